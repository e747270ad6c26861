#!/bin/bash
# PMC passes (separate rocprofv3 runs, no trace domains) of the QTIP matvec / fused linear micro-benchmark
tag=${1:-r02}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
S=/tmp/prof_qtip_$tag; rm -rf $S; mkdir -p $S
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $S/q1 -o p -- python $R/tools/bench_qtip_mv.py > $S/q1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $S/q2 -o p -- python $R/tools/bench_qtip_mv.py > $S/q2.log 2>&1
for d in q1 q2; do for f in $(find $S/$d -name "*.db"); do echo "== pass $d (tools/bench_qtip_mv.py)"; python $R/tools/rocpd_summary.py $f | cut -c1-170 | grep -A12 "qtip_.*(n=" | head -120; done; done > $R/gpurun_out/${tag}_qtip_counters.txt
wc -l $R/gpurun_out/${tag}_qtip_counters.txt
