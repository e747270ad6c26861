#!/bin/bash
# stream kernel at 3 and 4 bits (GQ_ST=2: every RMSNorm launch it serves) against the round-3 kernels
GQ_ST=2 timeout 600 python -m pytest tests/test_ap_fused_gpu.py -x -q -m gpu -k "rmsnorm" 2>&1 | tail -3
for b in 3 4; do for st in 2 0; do
  echo "== bits=$b GQ_ST=$st"
  GQ_ST=$st python tools/bench_ap.py --bits $b --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-130
  GQ_ST=$st python tools/bench_ap.py --bits $b --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-130
done; done
