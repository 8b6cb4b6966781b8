#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats + three PMC passes (HBM read bytes, HBM write bytes, MFMA-pipe busy cycles)
# of the default bench, each in its own run (no --pmc together with other trace domains).
# Outputs under gpurun_out/prof_round/; tools/summarise_profile.py turns them into the files kept in profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_round
mkdir -p $O
# --batch 256: ONE leg (the default N = 1 run also times the 128-image leg config3_n1 since round 5, which would mix two batch sizes into the per-kernel averages);
# --no-sched-calibration: exactly 13 steps traced
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 --batch 256 --no-sched-calibration"
# the step as it is timed: weight-gradient products on their own stream beside the data-gradient chain (kernel durations overlap)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_overlap -o r -- python3 $R/bench.py $ARGS > $O/trace_overlap.log 2>&1 &&
# every kernel alone on the chip (what bench.py's roofline object times; the PMC passes serialise the kernels anyway)
export LNX_WGRAD_STREAM=0 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r -- python3 $R/bench.py $ARGS > $O/trace.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o r -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o r -- python3 $R/bench.py $ARGS > $O/write.log 2>&1 &&
# matrix-core utilisation and the clock the chip holds: busy cycles of the MFMA pipe against the busy cycles of
# the shader engines, and GRBM_GUI_ACTIVE (summed over the 8 XCDs) against the kernel's duration
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o r -- python3 $R/bench.py $ARGS > $O/mfma.log 2>&1
echo rc=$?
rm -f $O/mfma/r_kernel_trace.csv.keep; cp $O/mfma/r_kernel_trace.csv $O/mfma_kernel_trace.csv 2>/dev/null
rm -f $O/trace/r_kernel_trace.csv $O/trace_overlap/r_kernel_trace.csv  # large; the stats files are what is kept
ls -la $O/*
