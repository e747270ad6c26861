#!/bin/bash
out=gpurun_out/r5_call20.txt; mkdir -p gpurun_out; : > $out
{
echo "### phase stamps compiled out (base) vs in (stamps): bench --quick 300 steps, qtip, per-launch"
for r in 1 2 3; do
echo "base   $(python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c40-75)"
echo "stamps $(GQ_LIB_PATH=$PWD/guidedquant_amd/abl_stamps/libgq_hip.so python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c40-75)"
done
echo "base   qtip $(python bench.py --backend qtip --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c40-75)"
echo "stamps qtip $(GQ_LIB_PATH=$PWD/guidedquant_amd/abl_stamps/libgq_hip.so python bench.py --backend qtip --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c40-75)"
for v in base stamps; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  python tools/bench_ap.py --bits 2 --shapes wqkv --launch qkv_rope 2>&1 | grep shape | cut -c1-150
  python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-150
  python tools/bench_ap.py --bits 2 --shapes wo w2 --launch resid 2>&1 | grep shape | cut -c1-150
  python tools/bench_ap.py --bits 4 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-150
done
unset GQ_LIB_PATH
echo "### quick test sweep"; timeout 1200 python -m pytest tests/test_ap_stream_gpu.py tests/test_ap_fused_gpu.py tests/test_qtip_gpu.py tests/test_decode_gpu.py -q -m gpu 2>&1 | tail -3
} >> $out 2>&1
