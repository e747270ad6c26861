#!/bin/bash
mkdir -p gpurun_out
for xf in 0 1 2 3; do
  echo "== GQ_ST_XFLAGS=$xf"
  GQ_ST_XFLAGS=$xf timeout 120 python tools/bench_ap.py --bits 2 --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-160
  GQ_ST_XFLAGS=$xf timeout 120 python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-160
done
GQ_ST_XFLAGS=2 PT_FUSED=1 timeout 120 python tools/phase_timing.py 2 w1w3 > gpurun_out/r4_phase_stream_x2.txt 2>&1
