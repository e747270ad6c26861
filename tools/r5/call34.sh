#!/bin/bash
# final records of the round: driver-form bench line (all sub-records), kernel trace + PMC passes of the same build
mkdir -p gpurun_out
bash tools/prof_bench.sh r05b > gpurun_out/r05b_prof.log 2>&1
cp gpurun_out/r05b_w1w3_traffic.json profiles/r05_w1w3_traffic.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05b_bench_default.log 2>&1
tail -1 gpurun_out/r05b_bench_default.log > gpurun_out/r05b_bench_default_line.json
python bench.py --quick --steps 400 --warmup 100 2>/dev/null | tail -1 | cut -c1-400 > gpurun_out/r05b_bench_400.txt
