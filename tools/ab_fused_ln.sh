for i in 1 2 3; do
LNX_NO_FUSED_LN=1 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']; print('unfused', d['ms_per_step'], d['loss'], k['convmlp_fwd']['ms_per_step'], k['convmlp_bwd']['ms_per_step'])" || exit 1
python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']; print('fused  ', d['ms_per_step'], d['loss'], k['convmlp_fwd']['ms_per_step'], k['convmlp_bwd']['ms_per_step'])" || exit 1
done
