#!/bin/bash
out=gpurun_out/r5_call62.txt; mkdir -p gpurun_out; : > $out
{
echo "### compiler flags for ap_stream.hip / ap_plane.hip: shipped (-O3) vs -O2, no high-RP reschedule stage, sched strategies max-ilp / max-memory-clause"
for v in base o2 nohrp maxilp maxmem base; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  t=$(timeout 600 python -m pytest tests/test_ap_stream_gpu.py -q -m gpu -x 2>&1 | tail -1 | cut -c1-40)
  echo "$v: tests [$t] 2-bit $(python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-60) 3-bit $(python bench.py --bits 3 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-60) 4-bit $(python bench.py --bits 4 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-60)"
done
} >> $out 2>&1
