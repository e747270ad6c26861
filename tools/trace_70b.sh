#!/bin/bash
# kernel mix of the Llama-3.3-70B 2-bit decode step on one GPU (rocprofv3 kernel trace of bench.py --model ... --quick)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
S=/tmp/prof_70b; rm -rf $S; mkdir -p $S $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d $S/tr -o t -- python $R/bench.py --model meta-llama/Llama-3.3-70B-Instruct --steps 60 --warmup 20 --quick --no-cpu-baseline --no-other-configs > $S/log 2>&1
{ for f in $(find $S/tr -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-160 | head -14; done; tail -1 $S/log | cut -c1-300; } | tee $R/gpurun_out/r04_70b_decode_kernel_trace.txt
