#!/bin/bash
# usage: tools/prof_ap.sh <tag> [bench_ap args...]   -> gpurun_out/<tag>_{trace,pmc*}
tag=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_trace -o t -- python $R/tools/bench_ap.py "$@" > $R/gpurun_out/${tag}_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $R/gpurun_out/${tag}_pmc1 -o p -- python $R/tools/bench_ap.py "$@" > $R/gpurun_out/${tag}_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/${tag}_pmc2 -o p -- python $R/tools/bench_ap.py "$@" > $R/gpurun_out/${tag}_pmc2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/${tag}_pmc3 -o p -- python $R/tools/bench_ap.py "$@" > $R/gpurun_out/${tag}_pmc3.log 2>&1
find $R/gpurun_out/${tag}_trace -name "*stats*" | head
