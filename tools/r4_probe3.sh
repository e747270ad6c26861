#!/bin/bash
mkdir -p gpurun_out
python tools/st_cmp.py 2>&1 | grep -v amdgpu.ids | cut -c1-100
PT_FUSED=1 timeout 120 python tools/phase_timing.py 2 w1w3 wqkv > gpurun_out/r4_phase_stream.txt 2>&1
for st in 1 0; do
  echo "== GQ_ST=$st"
  GQ_ST=$st timeout 120 python tools/bench_ap.py --bits 2 --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-160
  GQ_ST=$st timeout 120 python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-160
done
timeout 300 python -m pytest tests/test_ap_fused_gpu.py -x -q -m gpu 2>&1 | tail -3
