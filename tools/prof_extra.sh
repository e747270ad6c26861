#!/bin/bash
# extra PMC passes (separate rocprofv3 runs, no trace domains): the QTIP fused linear / matvec and the local-image AP kernel
tag=${1:-r01}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
S=/tmp/prof_extra_$tag; rm -rf $S; mkdir -p $S
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $S/q1 -o p -- python $R/tools/bench_qtip_linear.py > $S/q1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU -d $S/q2 -o p -- python $R/tools/bench_qtip_linear.py > $S/q2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $S/l1 -o p -- python $R/tools/bench_ap.py --bits 2 --shapes wo w2 --iters 40 > $S/l1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $S/l2 -o p -- python $R/tools/bench_ap.py --bits 2 --shapes wo w2 --iters 40 > $S/l2.log 2>&1
for d in q1 q2; do for f in $(find $S/$d -name "*.db"); do echo "== pass $d (tools/bench_qtip_linear.py)"; python $R/tools/rocpd_summary.py $f | cut -c1-150 | grep -A12 "qtip_.*(n=" | head -80; done; done > $R/gpurun_out/${tag}_qtip_linear_counters.txt
for d in l1 l2; do for f in $(find $S/$d -name "*.db"); do echo "== pass $d (tools/bench_ap.py --bits 2 --shapes wo w2)"; python $R/tools/rocpd_summary.py $f | cut -c1-150 | grep -A12 "ap_plane_local_kernel.*(n=" | head -40; done; done > $R/gpurun_out/${tag}_ap_local_counters.txt
wc -l $R/gpurun_out/${tag}_qtip_linear_counters.txt $R/gpurun_out/${tag}_ap_local_counters.txt
