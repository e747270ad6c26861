#!/bin/bash
out=gpurun_out/r5_call25.txt; mkdir -p gpurun_out; : > $out
{
echo "### 3 / 4 bits: plane-kernel instances with the launch properties compiled in (GQ_PL_SPEC=1, default) vs the general ones (0); wqkv now on the plane kernel"
for b in 3 4; do for sp in 0 1; do
  echo "== bits $b GQ_PL_SPEC=$sp"
  GQ_PL_SPEC=$sp python tools/bench_ap.py --bits $b --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-150
  GQ_PL_SPEC=$sp python tools/bench_ap.py --bits $b --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-150
  GQ_PL_SPEC=$sp python tools/bench_ap.py --bits $b --shapes wo w2 --launch resid 2>&1 | grep shape | cut -c1-150
done; done
for b in 3 4; do for sp in 0 1; do
echo "decode bits=$b GQ_PL_SPEC=$sp $(GQ_PL_SPEC=$sp python bench.py --bits $b --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-75)"
done; done
echo "decode bits=2 $(python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
echo "### full GPU suite"; timeout 2000 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
} >> $out 2>&1
