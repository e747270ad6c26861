#!/bin/bash
# round profile: kernel trace of bench.py + HBM traffic / SQ counters of the dominant AP-GEMV kernel (w1w3) in
# SEPARATE rocprofv3 passes (no --pmc together with traces); writes only the text summaries under gpurun_out/<tag>_*.txt
tag=${1:-r01}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
S=/tmp/prof_$tag; rm -rf $S; mkdir -p $S
cd /tmp
rocprofv3 --kernel-trace --stats -d $S/bench_trace -o t -- python $R/bench.py --steps 100 --warmup 100 --no-cpu-baseline > $S/bench_trace.log 2>&1
for f in $(find $S/bench_trace -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-150 | head -24; done > $R/gpurun_out/${tag}_bench_kernel_trace_stats.txt
tail -1 $S/bench_trace.log | cut -c1-400 >> $R/gpurun_out/${tag}_bench_kernel_trace_stats.txt
rocprofv3 --pmc FETCH_SIZE -d $S/fetch -o p -- python $R/tools/bench_ap.py --bits 2 --shapes w1w3 --iters 40 > $S/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $S/write -o p -- python $R/tools/bench_ap.py --bits 2 --shapes w1w3 --iters 40 > $S/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $S/pmc1 -o p -- python $R/tools/bench_ap.py --bits 2 --shapes w1w3 --iters 40 > $S/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $S/pmc2 -o p -- python $R/tools/bench_ap.py --bits 2 --shapes w1w3 --iters 40 > $S/pmc2.log 2>&1
for d in fetch write pmc1 pmc2; do
  for f in $(find $S/$d -name "*.db"); do echo "== pass $d"; python $R/tools/rocpd_summary.py $f | cut -c1-150 | grep -B1 -A10 "ap_plane_kernel\|ap_gemv_quad_kernel" | head -40; done
done > $R/gpurun_out/${tag}_ap_gemv_w1w3_counters.txt
cat $R/gpurun_out/${tag}_bench_kernel_trace_stats.txt | head -14
