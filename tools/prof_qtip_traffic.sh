#!/bin/bash
# HBM traffic of the bare QTIP trellis matvec (gq_qtip_matvec, R = 2; the three Llama-2-7b shapes bench.py's qtip rooflines time) from
# SEPARATE rocprofv3 --pmc passes per shape (FETCH_SIZE, WRITE_SIZE; no trace domains) -> gpurun_out/<tag>_qtip_matvec_traffic.json + .txt
tag=${1:-r04}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp
BA="python $R/tools/bench_qtip_mv.py"
export MV_ONLY=1
: > $R/gpurun_out/${tag}_qtip_matvec_traffic.txt
echo '{"shapes": [' > $R/gpurun_out/${tag}_qtip_matvec_traffic.json
first=1
for shape in 4096x4096 11008x4096 4096x11008; do
  export MV_SHAPE=$shape
  S=/tmp/prof_qtip_tr_${tag}_$shape; rm -rf $S; mkdir -p $S
  rocprofv3 --kernel-trace --stats -d $S/ktrace -o p -- $BA > $S/ktrace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $S/fetch -o p -- $BA > $S/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $S/write -o p -- $BA > $S/write.log 2>&1
  {
    echo "== kernel trace (MV_ONLY=1 MV_SHAPE=$shape tools/bench_qtip_mv.py): $(grep -h matvec $S/ktrace.log | tail -1 | cut -c1-200)"
    for f in $(find $S/ktrace -name "*.db"); do python $R/tools/rocpd_summary.py $f | cut -c1-160 | grep -A1 "^kernel\|qtip_matvec" | head -8; done
    for d in fetch write; do for f in $(find $S/$d -name "*.db"); do echo "== pass $d"; python $R/tools/rocpd_summary.py $f | cut -c1-150 | grep -A3 "qtip_matvec.*(n=" | head -12; done; done
  } >> $R/gpurun_out/${tag}_qtip_matvec_traffic.txt
  [ $first = 1 ] || echo ',' >> $R/gpurun_out/${tag}_qtip_matvec_traffic.json
  first=0
  python - <<PY >> $R/gpurun_out/${tag}_qtip_matvec_traffic.json
import glob, json, sys
sys.path.insert(0, "$R/tools")
from rocpd_summary import counter_mean
M, K = (int(v) for v in "$shape".split("x"))
f = counter_mean(glob.glob("$S/fetch/**/*.db", recursive=True)[0], "qtip_matvec_kernel", "FETCH_SIZE")
w = counter_mean(glob.glob("$S/write/**/*.db", recursive=True)[0], "qtip_matvec_kernel", "WRITE_SIZE")
print(json.dumps({"kernel": f[2][:80], "M": M, "K": K, "R": 2,
                  "source": "profiles/${tag}_qtip_matvec_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/prof_qtip_traffic.sh)",
                  "FETCH_SIZE_KiB_per_launch": round(f[0], 1), "WRITE_SIZE_KiB_per_launch": round(w[0], 1), "dispatches_averaged": f[1],
                  "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream (MI355X_MICROARCH.md, HBM section): doubled",
                  "hbm_bytes_per_launch": int(round((2 * f[0] + w[0]) * 1024))}, indent=1))
PY
done
echo ']}' >> $R/gpurun_out/${tag}_qtip_matvec_traffic.json
cat $R/gpurun_out/${tag}_qtip_matvec_traffic.txt | head -40; python -c "import json; print(json.load(open('$R/gpurun_out/${tag}_qtip_matvec_traffic.json')))" | cut -c1-600
