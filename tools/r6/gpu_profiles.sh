#!/bin/bash
# round-6 profiles: tools/prof_bench.sh r06 (kernel trace of bench.py + FETCH / WRITE / SQ passes of the w1w3 2-bit launch), the SQ
# counters of the 4-bit decode-to-fp16 kernel on w1w3, and the kernel trace of the 4-bit decode
R=$(pwd)
tools/prof_bench.sh r06 2>&1 | tail -30
tools/prof_ap_pmc.sh r06_dq_4bit_w1w3 --bits 4 --shapes w1w3 --iters 40 --launch norm_pairs
export TMPDIR=/tmp
S=/tmp/prof_r06_4bit; rm -rf $S; mkdir -p $S; cd /tmp
rocprofv3 --kernel-trace --stats -d $S/t -o t -- python $R/bench.py --bits 4 --steps 100 --warmup 100 --quick > $S/t.log 2>&1
for f in $(find $S/t -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-160 | head -12; done > $R/gpurun_out/r06_4bit_decode_kernel_trace.txt
tail -1 $S/t.log | cut -c1-600 >> $R/gpurun_out/r06_4bit_decode_kernel_trace.txt
cat $R/gpurun_out/r06_4bit_decode_kernel_trace.txt | head -14
