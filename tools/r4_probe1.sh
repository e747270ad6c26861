#!/bin/bash
# round 4, probe 1: phase stamps of the decode-graph launch forms + a baseline bench line on the same box
mkdir -p gpurun_out
PT_FUSED=1 python tools/phase_timing.py 2 w1w3 wqkv > gpurun_out/r4_phase_fused.txt 2>&1
python tools/phase_timing.py 2 w1w3 wqkv wo w2 > gpurun_out/r4_phase_plain.txt 2>&1
python bench.py --steps 200 --warmup 20 > gpurun_out/r4_bench0.json 2> gpurun_out/r4_bench0.err
tail -c 600 gpurun_out/r4_bench0.json
