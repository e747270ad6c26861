#!/bin/bash
# tools/prof_cmp.sh <variant...>: PMC passes of the w1w3 launch (2-bit, RMSNorm + pairs) per library variant -> gpurun_out/cmp_<v>.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQC?_[A-Z_]*(ICACHE|IFETCH|INST_LEVEL|DCACHE)[A-Z_]*" | sort -u > $R/gpurun_out/counters_icache.txt
for v in "$@"; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$R/guidedquant_amd/abl_$v/libgq_hip.so; fi
  S=/tmp/cmp_$v; rm -rf $S; mkdir -p $S
  args="--bits 2 --shapes ${CMP_SHAPE:-w1w3} --launch ${CMP_LAUNCH:-norm_pairs} --iters 50"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $S/p1 -o p -- python $R/tools/bench_ap.py $args > $S/p1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM -d $S/p2 -o p -- python $R/tools/bench_ap.py $args > $S/p2.log 2>&1
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $S/p3 -o p -- python $R/tools/bench_ap.py $args > $S/p3.log 2>&1
  for d in p1 p2 p3; do for f in $(find $S/$d -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-150 | grep -A12 "ap_.*(n=" | head -40; done; tail -3 $S/$d.log | grep -i -E "error|invalid" ; done > $R/gpurun_out/cmp_$v.txt
done
