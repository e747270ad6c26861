#!/bin/bash
# tools/build_variant.sh NAME FILE.hip [extra hipcc flags]: libgq_hip.so with ONE translation unit rebuilt with extra flags,
# into guidedquant_amd/abl_NAME/libgq_hip.so (git-ignored; select with GQ_LIB_PATH) -- kernel experiments side by side in one GPU call
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../guidedquant_amd/csrc"
make -s >/dev/null
out=../abl_$name; mkdir -p $out
base=$(basename $src .hip)
extra=""; case $base in qtip|ap_plane|ap_stream) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $extra "$@" -c $base.hip -o $out/$base.o
objs=$(ls *.o | grep -v "^$base.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $out/$base.o -fopenmp -L/opt/rocm/lib/llvm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib -o $out/libgq_hip.so
echo built $out/libgq_hip.so
