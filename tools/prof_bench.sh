#!/bin/bash
# round profile: kernel trace of bench.py + HBM traffic counters of the AP-GEMV kernels (separate passes, no traces)
tag=${1:-r01}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_bench_trace -o t -- python $R/bench.py --steps 100 --warmup 100 --no-cpu-baseline > $R/gpurun_out/${tag}_bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_fetch -o p -- python $R/tools/bench_ap.py --bits 2 --iters 20 > $R/gpurun_out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_write -o p -- python $R/tools/bench_ap.py --bits 2 --iters 20 > $R/gpurun_out/${tag}_write.log 2>&1
GQ_AP_EXACT=1 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_fetch_exact -o p -- python $R/tools/bench_ap.py --bits 2 --iters 20 > $R/gpurun_out/${tag}_fetch_exact.log 2>&1
tail -1 $R/gpurun_out/${tag}_bench_trace.log | cut -c1-200
