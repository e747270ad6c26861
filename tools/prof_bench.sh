#!/bin/bash
# round profile: kernel trace of bench.py + HBM traffic / SQ counters of the dominant AP-GEMV kernel -- the w1w3 launch exactly
# as the decode graph issues it (RMSNorm prologue + gate/up pair epilogue: tools/bench_ap.py --launch norm_pairs) -- in SEPARATE
# rocprofv3 passes (no --pmc together with traces); writes the text summaries and <tag>_w1w3_traffic.json under gpurun_out/
tag=${1:-r04}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
S=/tmp/prof_$tag; rm -rf $S; mkdir -p $S
cd /tmp
rocprofv3 --kernel-trace --stats -d $S/bench_trace -o t -- python $R/bench.py --steps 100 --warmup 100 --quick > $S/bench_trace.log 2>&1
for f in $(find $S/bench_trace -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-160 | head -30; done > $R/gpurun_out/${tag}_bench_kernel_trace_stats.txt
tail -1 $S/bench_trace.log | cut -c1-1200 >> $R/gpurun_out/${tag}_bench_kernel_trace_stats.txt
BA="python $R/tools/bench_ap.py --bits 2 --shapes w1w3 --iters 40 --launch norm_pairs"
rocprofv3 --kernel-trace --stats -d $S/ktrace -o p -- $BA > $S/ktrace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $S/fetch -o p -- $BA > $S/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $S/write -o p -- $BA > $S/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $S/pmc1 -o p -- $BA > $S/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $S/pmc2 -o p -- $BA > $S/pmc2.log 2>&1
{
  echo "== standalone kernel trace (tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs), bench line: $(tail -1 $S/ktrace.log)"
  for f in $(find $S/ktrace -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-160 | grep -A1 "^kernel\|ap_plane\|ap_stream" | head -8; done
  for d in fetch write pmc1 pmc2; do
    for f in $(find $S/$d -name "*.db"); do echo "== pass $d"; python $R/tools/rocpd_summary.py $f | cut -c1-150 | grep -B1 -A10 "ap_stream_kernel\|ap_plane_kernel\|ap_gemv_quad_kernel" | head -40; done
  done
} > $R/gpurun_out/${tag}_ap_gemv_w1w3_counters.txt
python - <<PY > $R/gpurun_out/${tag}_w1w3_traffic.json
import glob, json, sys
sys.path.insert(0, "$R/tools")
from rocpd_summary import counter_mean
kn = "ap_stream_kernel"
f = counter_mean(glob.glob("$S/fetch/**/*.db", recursive=True)[0], kn, "FETCH_SIZE")
if f is None:
    kn = "ap_plane_kernel"
    f = counter_mean(glob.glob("$S/fetch/**/*.db", recursive=True)[0], kn, "FETCH_SIZE")
w = counter_mean(glob.glob("$S/write/**/*.db", recursive=True)[0], kn, "WRITE_SIZE")
print(json.dumps({"kernel": f[2][:80], "launch": "norm_pairs", "bits": 2, "N": 28672, "K": 4096, "mode": "default",
                  "source": "profiles/${tag}_ap_gemv_w1w3_counters.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/prof_bench.sh)",
                  "FETCH_SIZE_KiB_per_launch": round(f[0], 1), "WRITE_SIZE_KiB_per_launch": round(w[0], 1), "dispatches_averaged": f[1],
                  "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream (MI355X_MICROARCH.md, HBM section): doubled",
                  "hbm_bytes_per_launch": int(round((2 * f[0] + w[0]) * 1024))}, indent=1))
PY
cat $R/gpurun_out/${tag}_bench_kernel_trace_stats.txt | head -16; cat $R/gpurun_out/${tag}_w1w3_traffic.json
