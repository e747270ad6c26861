#!/bin/bash
# usage: tools/prof_gemm_pmc.sh <tag> N K S bits -- PMC passes (separate rocprofv3 runs, no trace domains) over the prefill GEMM
tag=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
S=/tmp/prof_gemm_$tag; rm -rf $S; mkdir -p $S
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $S/p1 -o p -- python $R/tools/gemm_one.py "$@" > $S/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $S/p2 -o p -- python $R/tools/gemm_one.py "$@" > $S/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $S/p3 -o p -- python $R/tools/gemm_one.py "$@" > $S/p3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum -d $S/p4 -o p -- python $R/tools/gemm_one.py "$@" > $S/p4.log 2>&1
for d in p1 p2 p3 p4; do for f in $(find $S/$d -name "*.db"); do echo "== pass $d (tools/gemm_one.py $@)"; python $R/tools/rocpd_summary.py $f | cut -c1-150 | grep -A10 "ap_gemm.*(n=" | head -40; done; done > $R/gpurun_out/${tag}_counters.txt
tail -3 $S/p4.log >> $R/gpurun_out/${tag}_counters.txt
wc -l $R/gpurun_out/${tag}_counters.txt
