set -o pipefail
O=gpurun_out/r5h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "layernorm or convmlp" > $O/t_ln.log 2>&1; echo "ln rc=$?" | tee -a $O/summary.txt
tail -5 $O/t_ln.log
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -k "forward_fp32 or backward_matches or train_step or sm_b24 or stream_logic or recompute or weight_gradient_stream or grad_scaler or direct_grad or fused_block_layernorm or odd_batches" > $O/t_model.log 2>&1; echo "model rc=$?" | tee -a $O/summary.txt
tail -4 $O/t_model.log
for i in 1 2; do
python bench.py --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('defer  ', d['ms_per_step'], d['config3_n1']['ms_per_step'], d['loss'])" | tee -a $O/summary.txt
LNX_LN_DEFER=0 python bench.py --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('nodefer', d['ms_per_step'], d['config3_n1']['ms_per_step'], d['loss'])" | tee -a $O/summary.txt
done
cat $O/summary.txt
