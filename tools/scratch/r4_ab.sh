#!/bin/bash
# A/B on one box: stream-kernel build variants (guidedquant_amd/abl_*/libgq_hip.so) against the current library and the round-3 kernel
for r in 1 2; do for v in base "$@"; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  python tools/bench_ap.py --bits 2 --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-120
  python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-120
done; done
unset GQ_LIB_PATH
echo "== round-3 kernel"; GQ_ST=0 python tools/bench_ap.py --bits 2 --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-120
GQ_ST=0 python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-120
for v in "$@"; do export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; echo "== cmp $v"; python tools/st_cmp.py 2>&1 | grep -v amdgpu | cut -c1-90; done
