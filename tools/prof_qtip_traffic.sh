#!/bin/bash
# HBM traffic of the bare QTIP trellis matvec (gq_qtip_matvec 11008x4096, R = 2 -- the launch bench.py's qtip roofline times) from
# SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; no trace domains) -> gpurun_out/<tag>_qtip_matvec_traffic.json + .txt
tag=${1:-r03}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
S=/tmp/prof_qtip_tr_$tag; rm -rf $S; mkdir -p $S
cd /tmp
BA="python $R/tools/bench_qtip_mv.py"
export MV_ONLY=1 MV_SHAPE=11008x4096
rocprofv3 --kernel-trace --stats -d $S/ktrace -o p -- $BA > $S/ktrace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $S/fetch -o p -- $BA > $S/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $S/write -o p -- $BA > $S/write.log 2>&1
{
  echo "== kernel trace (MV_ONLY=1 MV_SHAPE=11008x4096 tools/bench_qtip_mv.py): $(grep -h matvec $S/ktrace.log | tail -1 | cut -c1-200)"
  for f in $(find $S/ktrace -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-160 | grep -A1 "^kernel\|qtip_matvec" | head -8; done
  for d in fetch write; do for f in $(find $S/$d -name "*.db"); do echo "== pass $d"; python $R/tools/rocpd_summary.py $f | cut -c1-150 | grep -A3 "qtip_matvec.*(n=" | head -12; done; done
} > $R/gpurun_out/${tag}_qtip_matvec_traffic.txt
python - <<PY > $R/gpurun_out/${tag}_qtip_matvec_traffic.json
import glob, json, sys
sys.path.insert(0, "$R/tools")
from rocpd_summary import counter_mean
f = counter_mean(glob.glob("$S/fetch/**/*.db", recursive=True)[0], "qtip_matvec_kernel", "FETCH_SIZE")
w = counter_mean(glob.glob("$S/write/**/*.db", recursive=True)[0], "qtip_matvec_kernel", "WRITE_SIZE")
print(json.dumps({"kernel": f[2][:80], "M": 11008, "K": 4096, "R": 2,
                  "source": "profiles/${tag}_qtip_matvec_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/prof_qtip_traffic.sh)",
                  "FETCH_SIZE_KiB_per_launch": round(f[0], 1), "WRITE_SIZE_KiB_per_launch": round(w[0], 1), "dispatches_averaged": f[1],
                  "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream (MI355X_MICROARCH.md, HBM section): doubled",
                  "hbm_bytes_per_launch": int(round((2 * f[0] + w[0]) * 1024))}, indent=1))
PY
cat $R/gpurun_out/${tag}_qtip_matvec_traffic.txt; cat $R/gpurun_out/${tag}_qtip_matvec_traffic.json
