#!/bin/bash
# stream kernel for the residual launches (wo, w2) against the local-image kernel; 70B shapes
for st in 3 1; do
  echo "== GQ_ST=$st"
  GQ_ST=$st python tools/bench_ap.py --bits 2 --shapes wo w2 --launch resid 2>&1 | grep shape | cut -c1-130
  GQ_ST=$st python tools/bench_ap.py --bits 2 --shapes 70b_wo 70b_w2 --launch resid 2>&1 | grep shape | cut -c1-130
done
for st in 1 0; do
  echo "== GQ_ST=$st"
  GQ_ST=$st python tools/bench_ap.py --bits 2 --shapes 70b_wqkv --launch norm 2>&1 | grep shape | cut -c1-130
  GQ_ST=$st python tools/bench_ap.py --bits 2 --shapes 70b_w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-130
done
