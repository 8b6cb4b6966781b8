set -o pipefail
O=gpurun_out/r5c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "meta_head_chain" > $O/t_meta_op.log 2>&1; echo "meta_op rc=$?" | tee -a $O/summary.txt
tail -8 $O/t_meta_op.log
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --no-sched-calibration 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pin-run', d['loss'], d['ms_per_step'], d['config3_n1']['ms_per_step'])" | tee -a $O/summary.txt; done
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_train_loop.py::test_bench_line_loss_is_pinned_and_checked_against_the_oracle > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt
tail -5 $O/t_all.log
LNX_WGRAD_STREAM=0 bash tools/quick_trace.sh r5c_b256 --no-sched-calibration --batch 256 > $O/trace_b256.log 2>&1; echo "trace256 rc=$?" | tee -a $O/summary.txt
grep -i "meta_chain" gpurun_out/r5c_b256/trace/r_kernel_stats.csv | cut -c1-160
python bench.py --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt
python bench.py --force-dp --no-cpu-baseline > $O/bench_dp.log 2>&1; echo "bench dp rc=$?" | tee -a $O/summary.txt
tail -c 3000 $O/bench_dp.log
cat $O/summary.txt
