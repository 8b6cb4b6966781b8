#!/bin/bash
# Library with the measurement kernels linked in: every product object + tools/experiments/*.  Load it with
# LNX_LIB_PATH=tools/liblnx_experiments.so.  Extra compiler flags (e.g. -DV8_STAMP, -DV5_ABLATE) are passed through.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
csrc="$here/../../linnaeus_amd/csrc"
make -C "$csrc" -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I$csrc $*"
out=${LNX_EXPERIMENTS_OUT:-$here/../liblnx_experiments.so}
tmp=$(mktemp -d)
$HIPCC $FLAGS -c "$here/gemm_nt_v8.hip" -o "$tmp/gemm_nt_v8.o"
$HIPCC $FLAGS -c "$here/gemm_nt_v5.hip" -o "$tmp/gemm_nt_v5.o"
$HIPCC $FLAGS -x hip -c "$here/hook.cpp" -o "$tmp/hook.o"
$HIPCC -shared -fPIC --offload-arch=gfx950 "$csrc"/*.o "$tmp"/*.o -o "$out"
rm -rf "$tmp"
echo "built $out"
