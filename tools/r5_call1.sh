set -o pipefail
O=gpurun_out/r5a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -s -k "xl_b128 or lg384_b64" > $O/t_model_new.log 2>&1; echo "model_new rc=$?" | tee -a $O/summary.txt
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_ops.py -x -q -k "xl_lg or discarded or deferred or at_xl_rows" > $O/t_ops_new.log 2>&1; echo "ops_new rc=$?" | tee -a $O/summary.txt
timeout -k 10 600 python -m pytest tests/test_gpu_train_loop.py -x -q -s -k bench_line > $O/t_bench_pin.log 2>&1; echo "bench_pin rc=$?" | tee -a $O/summary.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_model.py::test_xl_b128_batch_invariance --deselect tests/test_gpu_model.py::test_lg384_b64_batch_invariance > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt
tail -3 $O/t_all.log
LNX_WGRAD_STREAM=0 bash tools/quick_trace.sh r5a_b128 --batch 128 --no-sched-calibration > $O/trace_b128.log 2>&1; echo "trace128 rc=$?" | tee -a $O/summary.txt
LNX_WGRAD_STREAM=0 bash tools/quick_trace.sh r5a_b256 --no-sched-calibration > $O/trace_b256.log 2>&1; echo "trace256 rc=$?" | tee -a $O/summary.txt
cat $O/summary.txt
