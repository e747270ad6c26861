#!/bin/bash
out=gpurun_out/r5_call29.txt; mkdir -p gpurun_out; : > $out
{
echo "### sampler stage 2: counter / position requested with the candidates (new) vs where they are used (old)"
for r in 1 2 3; do
echo "old $(GQ_LIB_PATH=$PWD/guidedquant_amd/abl_oldsamp/libgq_hip.so python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
echo "new $(python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
done
echo "### the reference's operator (plain launch, no prologue) at 2 bits: default dispatch vs the stream kernel wherever it has a form (GQ_ST=3)"
for cfg in "" "GQ_ST=3"; do echo "== [$cfg]"; env $cfg python tools/bench_ap.py --bits 2 --shapes wqkv wo w1w3 w2 --launch plain 2>&1 | grep shape | cut -c1-150; done
echo "### sampler tests"; timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_hf_routes_gpu.py -q -m gpu 2>&1 | tail -3
} >> $out 2>&1
