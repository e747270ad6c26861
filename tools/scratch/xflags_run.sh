#!/bin/bash
# where the time of the shared-image kernel goes: the w1w3 / wqkv launches with parts of the kernel switched off (GQ_PL_XFLAGS)
for f in ${XF:-0 1024 1 2 32 8 16}; do
  echo "== xflags $f"
  GQ_PL_XFLAGS=$f python tools/bench_ap.py --bits ${XB:-2} --shapes w1w3 --launch norm_pairs | cut -c1-120
  GQ_PL_XFLAGS=$f python tools/bench_ap.py --bits ${XB:-2} --shapes wqkv --launch norm | cut -c1-120
done
