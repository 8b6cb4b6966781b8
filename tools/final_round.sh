#!/bin/bash
# Run on the GPU box (ONE gpurun call, so that every committed number of a round comes from one box and one build):
#   kernel stats (isolated + overlapped), HBM / MFMA counter passes, every bench log, the 128-image kernel stats, the stand-in-collective log.
# usage: bash tools/final_round.sh <tag>     then, here:  python tools/summarise_profile.py <tag>  and copy gpurun_out/bench_round/<tag>_* ,
#        gpurun_out/<tag>_b128/trace/r_kernel_stats.csv (-> profiles/<tag>_b128_kernel_stats.csv) and gpurun_out/<tag>_cu_hog.log into profiles/
T=${1:-rXX}
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh > $R/gpurun_out/${T}_profile_round.log 2>&1 || { echo "profile_round FAILED"; tail -5 $R/gpurun_out/${T}_profile_round.log; exit 1; }
cd $R
bash tools/bench_round.sh $T || exit 1
python tools/bench_meta.py --batch 256 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_meta_chain.log; python tools/bench_meta.py --batch 128 2>&1 | grep -v amdgpu.ids >> gpurun_out/${T}_meta_chain.log
python tools/bench_gemm_forms.py xl lg sm 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_bare_gemm.log
LNX_WGRAD_STREAM=0 bash tools/quick_trace.sh ${T}_b128 --batch 128 --no-sched-calibration > gpurun_out/${T}_b128_trace.log 2>&1 || { echo "b128 trace FAILED"; exit 1; }
H=gpurun_out/${T}_cu_hog.log
: > $H
for B in 128 256; do
    LNX_TILE_SCHED=static timeout -k 10 300 python tools/bench_cu_hog.py --batch $B --wgs 0,32,64,96 >> $H 2>&1 || { echo "hog static $B FAILED"; tail -3 $H; exit 1; }
    timeout -k 10 300 python tools/bench_cu_hog.py --batch $B --wgs 0,32,64,96 >> $H 2>&1 || { echo "hog atomic $B FAILED"; tail -3 $H; exit 1; }
done
grep -v "amdgpu.ids" $H | tail -24
echo final_round done
