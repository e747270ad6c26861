#!/bin/bash
# prefill GEMM tile shapes on one box (GQ_GEMM_SHAPE: 0 first kernel, 14 / 24 / 18 / 28 = RF, CF of the pipelined one); VARIANTS, BITS from the environment
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
for f in ${VARIANTS:-0 14 24 18}; do echo "## bench GQ_GEMM_SHAPE=$f bits ${BITS:-2}"; GQ_GEMM_SHAPE=$f timeout 900 python tools/bench_prefill.py ${BITS:-2} 2>&1 | cut -c1-200; done
} > gpurun_out/gemm_rf.txt 2>&1
grep -c . gpurun_out/gemm_rf.txt
