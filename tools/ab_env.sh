# A/B of one environment switch on one box: tools/ab_env.sh VAR [rounds]; prints ms/step with VAR=1 and without, alternating
V=$1; N=${2:-3}
for i in $(seq $N); do
env $V=1 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$V=1 ', d['ms_per_step'], d['loss'])" || exit 1
python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('default', d['ms_per_step'], d['loss'])" || exit 1
done
