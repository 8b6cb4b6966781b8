set -o pipefail
O=gpurun_out/r5d; mkdir -p $O
python tools/bench_meta.py --batch 256 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/bench_meta.py --batch 128 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "meta_head" > $O/t_meta_op.log 2>&1; echo "meta_op rc=$?" | tee -a $O/summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -s -k "forward_fp32 or backward_matches or fp8 or xl_b128 or sm_b24" > $O/t_model.log 2>&1; echo "model rc=$?" | tee -a $O/summary.txt
grep "^\[" $O/t_model.log | cut -c1-330
tail -3 $O/t_model.log
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --no-sched-calibration 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pin-run', d['loss'], d['ms_per_step'], d['config3_n1']['ms_per_step'])" | tee -a $O/summary.txt; done
python tools/bench_gemm_forms.py xl lg sm 2>&1 | grep -v amdgpu.ids > $O/bare_gemm.log; echo "forms rc=$?" | tee -a $O/summary.txt
cat $O/bare_gemm.log
python bench.py --arch xl --batch 128 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_xl_bf16.log 2>&1; echo "xl bf16 rc=$?" | tee -a $O/summary.txt
python bench.py --arch xl --batch 128 --dtype fp8 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_xl_fp8.log 2>&1; echo "xl fp8 rc=$?" | tee -a $O/summary.txt
LNX_FP8_DGRAD=0 python bench.py --arch xl --batch 128 --dtype fp8 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_xl_fp8_nodgrad.log 2>&1; echo "xl fp8 nodgrad rc=$?" | tee -a $O/summary.txt
for f in $O/bench_xl_*.log; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['loss'], 'roofline', r.get('achieved'), r.get('peak'), r.get('frac'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $O/summary.txt
