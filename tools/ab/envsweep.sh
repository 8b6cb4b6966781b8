cd $GRAFT_REPO_ROOT
for e in "LNX_HEADS_GROUP=0" "LNX_FREQ_DEFER=0" "LNX_LN_DEFER=0" "LNX_WGRAD_STREAM=0" "LNX_META_CHAIN=0" "LNX_NO_SIDE_STREAM=1"; do
  echo "== $e"
  env $e timeout -k 10 400 python -m pytest tests/test_gpu_model.py -x -q -k "train_step_matches_reference or sm_b24_production or gradients_match" 2>&1 | tail -1
done
