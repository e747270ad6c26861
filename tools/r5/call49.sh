#!/bin/bash
out=gpurun_out/r5_call49.txt; mkdir -p gpurun_out; : > $out
{
echo "### attention: merge over the stream groups that can hold a position (new) vs all 32 (oldattn)"
timeout 900 python -m pytest tests/test_decode_gpu.py -q -m gpu 2>&1 | tail -3
for S in 33 101; do echo "-- cache $S new"; ROPED=1 python tools/bench_attn.py $S 1 2>&1 | tail -4; echo "-- cache $S oldattn"; GQ_LIB_PATH=$PWD/guidedquant_amd/abl_oldattn/libgq_hip.so ROPED=1 python tools/bench_attn.py $S 1 2>&1 | tail -4; done
for r in 1 2 3; do
echo "oldattn $(GQ_LIB_PATH=$PWD/guidedquant_amd/abl_oldattn/libgq_hip.so python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
echo "new     $(python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
done
} >> $out 2>&1
