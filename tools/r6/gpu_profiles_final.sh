#!/bin/bash
# final-build profiles: tools/prof_bench.sh r06 again (kernel trace of bench.py + FETCH / WRITE / SQ passes of the w1w3 2-bit launch) and the
# kernel trace of the EXACT-mode decode (bench.py --mode exact): the launches behind the 650 tokens/s of DESIGN.md section 3.7 (iv)
R=$(pwd)
tools/prof_bench.sh r06 2>&1 | tail -24
export TMPDIR=/tmp
S=/tmp/prof_r06_exact; rm -rf $S; mkdir -p $S; cd /tmp
rocprofv3 --kernel-trace --stats -d $S/t -o t -- python $R/bench.py --mode exact --steps 100 --warmup 100 --quick > $S/t.log 2>&1
for f in $(find $S/t -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-170 | head -12; done > $R/gpurun_out/r06_exact_decode_kernel_trace.txt
tail -1 $S/t.log | cut -c1-600 >> $R/gpurun_out/r06_exact_decode_kernel_trace.txt
cat $R/gpurun_out/r06_exact_decode_kernel_trace.txt | head -14
