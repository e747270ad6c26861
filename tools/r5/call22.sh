#!/bin/bash
out=gpurun_out/r5_call22.txt; mkdir -p gpurun_out; : > $out
{
echo "### icache2"; ./tools/ubench/icache2
echo "### compiled-in epilogues / launch properties (epi) vs run-time flags (base)"
for r in 1 2 3; do
echo "base $(python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c40-75)"
echo "epi  $(GQ_LIB_PATH=$PWD/guidedquant_amd/abl_epi/libgq_hip.so python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c40-75)"
done
echo "epi, stream only (GQ_PL_SPEC=0) $(GQ_PL_SPEC=0 GQ_LIB_PATH=$PWD/guidedquant_amd/abl_epi/libgq_hip.so python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c40-75)"
echo "epi, plane only (GQ_ST_EPI=0)  $(GQ_ST_EPI=0 GQ_LIB_PATH=$PWD/guidedquant_amd/abl_epi/libgq_hip.so python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c40-75)"
for v in base epi; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  python tools/bench_ap.py --bits 2 --shapes wqkv --launch qkv_rope 2>&1 | grep shape | cut -c1-150
  python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-150
  python tools/bench_ap.py --bits 2 3 --shapes wo w2 --launch resid 2>&1 | grep shape | cut -c1-150
done
export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_epi/libgq_hip.so
echo "### tests on the epi library"; timeout 1500 python -m pytest tests -q -m gpu -x -k "stream or plane or fused or decode" 2>&1 | tail -3
} >> $out 2>&1
