set -o pipefail
O=gpurun_out/r5e; mkdir -p $O
python tools/bench_meta.py --batch 256 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "meta_head" > $O/t_meta_op.log 2>&1; echo "meta_op rc=$?" | tee -a $O/summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train_loop.py -x -q -s -k "forward_fp32 or fp8 or xl_b128 or bench_line or stream_logic" > $O/t_model.log 2>&1; echo "model rc=$?" | tee -a $O/summary.txt
grep "^\[" $O/t_model.log | cut -c1-330
tail -3 $O/t_model.log
echo "== default" > $O/forms.log; python tools/bench_gemm_forms.py xl sm 2>&1 | grep -v amdgpu.ids >> $O/forms.log
echo "== LNX_NT_V7=1" >> $O/forms.log; LNX_NT_V7=1 python tools/bench_gemm_forms.py xl sm 2>&1 | grep -v amdgpu.ids >> $O/forms.log
echo "== LNX_NT_V9=0 LNX_NT_V7=0 (one-shot kernels)" >> $O/forms.log; LNX_NT_V9=0 LNX_NT_V7=0 python tools/bench_gemm_forms.py xl sm 2>&1 | grep -v amdgpu.ids >> $O/forms.log
cat $O/forms.log
cat $O/summary.txt
