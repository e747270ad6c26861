#!/bin/bash
# split-K tile A/B on one box: GQ_GEMM_KSHAPE=14 | 18 | auto at S = 128 / 512 / 2048 (tools/bench_prefill.py), BITS from the environment
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
for f in 14 18 auto; do echo "## GQ_GEMM_KSHAPE=$f bits ${BITS:-4}"; if [ $f = auto ]; then unset GQ_GEMM_KSHAPE; else export GQ_GEMM_KSHAPE=$f; fi; timeout 900 python tools/bench_prefill.py ${BITS:-4} 2>&1 | cut -c1-200; done
} > gpurun_out/gemm_ksplit_ab.txt 2>&1
