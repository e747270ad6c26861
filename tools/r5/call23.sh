#!/bin/bash
out=gpurun_out/r5_call23.txt; mkdir -p gpurun_out; : > $out
{
echo "### token steps per graph replay (bench.py --quick --steps 300 --warmup 60)"
for r in 1 2; do for k in 1 2 4 10 20; do
echo "steps_per_replay=$k $(GQ_STEPS_PER_REPLAY=$k python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
done; done
echo "qtip k=1 $(python bench.py --backend qtip --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
echo "qtip k=4 $(GQ_STEPS_PER_REPLAY=4 python bench.py --backend qtip --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-75)"
echo "### full GPU suite"; timeout 2000 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
} >> $out 2>&1
