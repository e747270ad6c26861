#!/bin/bash
# usage: tools/prof_ap_pmc.sh <tag> [bench_ap args...]  -- three PMC passes (separate rocprofv3 runs, no trace domains), raw
# data under /tmp, the per-kernel counter means of the plane / exact kernels -> gpurun_out/<tag>_counters.txt
tag=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
S=/tmp/prof_ap_$tag; rm -rf $S; mkdir -p $S
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $S/p1 -o p -- python $R/tools/bench_ap.py "$@" > $S/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $S/p2 -o p -- python $R/tools/bench_ap.py "$@" > $S/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $S/p3 -o p -- python $R/tools/bench_ap.py "$@" > $S/p3.log 2>&1
for d in p1 p2 p3; do for f in $(find $S/$d -name "*.db"); do echo "== pass $d (tools/bench_ap.py $@)"; python $R/tools/rocpd_summary.py $f | cut -c1-150 | grep -A10 "ap_.*(n=" | head -60; done; done > $R/gpurun_out/${tag}_counters.txt
wc -l $R/gpurun_out/${tag}_counters.txt
