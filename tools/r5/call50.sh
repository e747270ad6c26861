#!/bin/bash
out=gpurun_out/r5_call50.txt; mkdir -p gpurun_out; : > $out
{
echo "### attention: positions interleaved over the waves (new) vs 16 contiguous positions per wave (contig)"
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_decode_default_gpu.py -q -m gpu 2>&1 | tail -3
for S in 17 33 65 101 257; do echo "-- cache $S new"; ROPED=1 python tools/bench_attn.py $S 1 2>&1 | tail -4 | tr '\n' ' '; echo; echo "-- cache $S contig"; GQ_LIB_PATH=$PWD/guidedquant_amd/abl_contig/libgq_hip.so ROPED=1 python tools/bench_attn.py $S 1 2>&1 | tail -4 | tr '\n' ' '; echo; done
for r in 1 2 3; do
echo "contig $(GQ_LIB_PATH=$PWD/guidedquant_amd/abl_contig/libgq_hip.so python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
echo "new    $(python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
done
echo "long context new    $(python tools/bench_long_context.py 2>/dev/null | tail -1 | cut -c1-200)"
echo "long context contig $(GQ_LIB_PATH=$PWD/guidedquant_amd/abl_contig/libgq_hip.so python tools/bench_long_context.py 2>/dev/null | tail -1 | cut -c1-200)"
} >> $out 2>&1
