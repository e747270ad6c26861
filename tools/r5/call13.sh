#!/bin/bash
out=gpurun_out/r5_call13.txt; mkdir -p gpurun_out; : > $out
{
echo "### tests"; timeout 1500 python -m pytest tests/test_ap_fused_gpu.py tests/test_ap_gemv_gpu.py tests/test_decode_gpu.py tests/test_decode_default_gpu.py tests/test_hf_routes_gpu.py -q -m gpu 2>&1 | tail -6
echo "### exact kernel with the RMSNorm prologue (3 / 4 bit wqkv; 2-bit exact mode)"
python tools/bench_ap.py --bits 3 4 --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-150
GQ_AP_EXACT=1 python tools/bench_ap.py --bits 2 --shapes wqkv w1w3 --launch norm 2>&1 | grep shape | cut -c1-150
echo "### bench quick, 300 steps: greedy sampler on / off"
for r in 1 2; do python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-120; GQ_SAMPLE_GREEDY=0 python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-120; done
echo "### bench quick, driver form (20 steps): fold on / off"
for r in 1 2 3; do python bench.py --quick --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-120; GQ_FOLD_EMBED=0 python bench.py --quick --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-120; done
echo "### 3 / 4 bit and exact decode"
python bench.py --quick --bits 3 --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c1-120
python bench.py --quick --bits 4 --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c1-120
python bench.py --quick --mode exact --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c1-120
} >> $out 2>&1
