set -o pipefail
O=gpurun_out/r5f; mkdir -p $O
python tools/bench_meta.py --batch 256 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/bench_meta.py --batch 128 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/meta_trace -o r -- python3 $GRAFT_REPO_ROOT/tools/bench_meta.py --batch 256 > $GRAFT_REPO_ROOT/$O/meta_trace.log 2>&1); grep "meta_chain" $O/meta_trace/r_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
rm -f $O/meta_trace/r_kernel_trace.csv
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "meta_head" > $O/t_meta_op.log 2>&1; echo "meta_op rc=$?" | tee -a $O/summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train_loop.py -x -q -s -k "forward_fp32 or backward_matches or xl_b128 or bench_line or stream_logic or odd_batches" > $O/t_model.log 2>&1; echo "model rc=$?" | tee -a $O/summary.txt
grep "^\[" $O/t_model.log | cut -c1-330
tail -3 $O/t_model.log
cat $O/summary.txt
