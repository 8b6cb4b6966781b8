#!/bin/bash
# Run on the GPU box (gpurun): the bench configurations whose logs are kept under profiles/<tag>_bench_*.log.
# usage: bash tools/bench_round.sh <tag>      (writes gpurun_out/bench_round/<tag>_bench_*.log; copy them into profiles/)
T=${1:-rXX}
O=gpurun_out/bench_round
mkdir -p $O
run() { n=$1; shift; echo "== $n: bench.py $*"; timeout -k 10 400 python bench.py "$@" > $O/${T}_bench_$n.log 2>&1 || { echo "FAILED $n"; tail -5 $O/${T}_bench_$n.log; return 1; }; tail -1 $O/${T}_bench_$n.log | cut -c1-230; }
run b256 &&
run dropin --drop-in --no-cpu-baseline &&
run hostinput --host-input --no-cpu-baseline &&
run flatfile --flat-file --no-cpu-baseline &&
run eval --eval --no-cpu-baseline &&
run dp --force-dp --no-cpu-baseline &&
run recompute --recompute --no-cpu-baseline &&
run lg384_b64 --arch lg --img 384 --batch 64 --no-cpu-baseline &&
run xl_b128_bf16 --arch xl --batch 128 --no-cpu-baseline &&
run xl_b128_fp8 --arch xl --batch 128 --dtype fp8 --no-cpu-baseline
