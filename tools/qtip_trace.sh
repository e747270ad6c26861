#!/bin/bash
# QTIP-backend decode: tokens/s + kernel mix (rocprofv3 kernel trace) on the power-of-two 7B-like model
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
S=/tmp/prof_qtip; rm -rf $S; mkdir -p $S $R/gpurun_out
cd /tmp
python $R/tools/qtip_decode_bench.py 8192 32 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d $S/tr -o t -- python $R/tools/qtip_decode_bench.py 8192 8 > $S/log 2>&1
for f in $(find $S/tr -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-150 | head -22; done | tee $R/gpurun_out/qtip_trace.txt
