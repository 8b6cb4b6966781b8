set -o pipefail
O=gpurun_out/r5b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "meta_head_chain" > $O/t_meta_op.log 2>&1; echo "meta_op rc=$?" | tee -a $O/summary.txt
tail -15 $O/t_meta_op.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_model.py::test_xl_b128_batch_invariance --deselect tests/test_gpu_model.py::test_lg384_b64_batch_invariance > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt
tail -5 $O/t_all.log
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -q -s -k "xl_b128 or lg384_b64" > $O/t_model_new.log 2>&1; echo "model_new rc=$?" | tee -a $O/summary.txt
grep "^\[" $O/t_model_new.log | cut -c1-400
for i in 1 2; do
python bench.py --no-cpu-baseline > $O/bench_chain_$i.log 2>&1; echo "bench chain rc=$?" | tee -a $O/summary.txt
LNX_META_CHAIN=0 python bench.py --no-cpu-baseline > $O/bench_nochain_$i.log 2>&1; echo "bench nochain rc=$?" | tee -a $O/summary.txt
done
python tools/trace_fills.py --batch 128 > $O/fills.log 2>&1; echo "fills rc=$?" | tee -a $O/summary.txt
LNX_WGRAD_STREAM=0 bash tools/quick_trace.sh r5b_b256 --no-sched-calibration --batch 256 > $O/trace_b256.log 2>&1; echo "trace256 rc=$?" | tee -a $O/summary.txt
for f in $O/bench_*.log; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['loss'], d.get('config3_n1',{}).get('ms_per_step'), d['backward_schedule'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $O/summary.txt
