#!/bin/bash
out=gpurun_out/r5_call38.txt; mkdir -p gpurun_out; : > $out
{
echo "### 3 bits: shared-image plane kernel with 8 waves per block (base) vs 16 (t3)"
for r in 1 2; do for v in base t3; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  a=$(python tools/bench_ap.py --bits 3 --shapes wqkv --launch norm 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  b=$(python tools/bench_ap.py --bits 3 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  c=$(python tools/bench_ap.py --bits 3 --shapes w1w3 --launch plain 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')
  echo "$v: wqkv norm $a  w1w3 norm_pairs $b  w1w3 plain $c  $(python bench.py --bits 3 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-75)"
done; done
export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_t3/libgq_hip.so
echo "### tests on t3"; timeout 1500 python -m pytest tests/test_ap_fused_gpu.py tests/test_ap_plane_rows_gpu.py tests/test_decode_default_gpu.py -q -m gpu 2>&1 | tail -4
} >> $out 2>&1
