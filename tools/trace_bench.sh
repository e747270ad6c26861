#!/bin/bash
tag=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
S=/tmp/prof_$tag; rm -rf $S; mkdir -p $S
cd /tmp
rocprofv3 --kernel-trace --stats -d $S/bench_trace -o t -- python $R/bench.py --steps 100 --warmup 100 --no-cpu-baseline "$@" > $S/bench_trace.log 2>&1
for f in $(find $S/bench_trace -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-150 | head -24; done > $R/gpurun_out/${tag}_trace.txt
tail -1 $S/bench_trace.log | cut -c1-200 >> $R/gpurun_out/${tag}_trace.txt
cat $R/gpurun_out/${tag}_trace.txt
