#!/bin/bash
# QTIP-backend decode: tokens/s + kernel mix (rocprofv3 kernel trace); $1 = MLP width (11008 = Llama-2-7b, 8192 = power-of-two stand-in)
export TMPDIR=/tmp
W=${1:-11008}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
S=/tmp/prof_qtip; rm -rf $S; mkdir -p $S $R/gpurun_out
cd /tmp
python $R/tools/qtip_decode_bench.py $W 32 2>&1 | tail -1 | tee $R/gpurun_out/qtip_trace_$W.txt
rocprofv3 --kernel-trace --stats -d $S/tr -o t -- python $R/tools/qtip_decode_bench.py $W 8 > $S/log 2>&1
for f in $(find $S/tr -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-150 | head -16; done | tee -a $R/gpurun_out/qtip_trace_$W.txt
