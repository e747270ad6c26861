#!/bin/bash
# round 4, probe 2: the stream kernel (ap_stream.hip) -- parity tests, phase stamps, kernel times against the round-3 kernel
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ap_fused_gpu.py -x -q -m gpu > gpurun_out/r4_t_fused.txt 2>&1; echo "fused rc=$?" 
PT_FUSED=1 timeout 120 python tools/phase_timing.py 2 w1w3 wqkv > gpurun_out/r4_phase_stream.txt 2>&1
for st in 1 0; do
  echo "== GQ_ST=$st"
  GQ_ST=$st timeout 120 python tools/bench_ap.py --bits 2 --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-160
  GQ_ST=$st timeout 120 python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-160
done
tail -5 gpurun_out/r4_t_fused.txt
