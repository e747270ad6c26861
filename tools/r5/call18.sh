#!/bin/bash
out=gpurun_out/r5_call18.txt; mkdir -p gpurun_out; : > $out
{
echo "### full GPU suite"; timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -8
echo "### wqkv norm 3 / 4 bit after the grid rule"; python tools/bench_ap.py --bits 3 4 --shapes wqkv --launch norm 2>&1 | grep shape | cut -c1-150
echo "### default bench (driver form)"; SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r5_bench_err.txt | tail -1 > gpurun_out/r05_bench_default_line.json; echo "bench wall seconds: $SECONDS"
python - <<'PY'
import json
l=json.load(open("gpurun_out/r05_bench_default_line.json"))
print({k:l[k] for k in ("value","ms_per_step")}, json.dumps(l["roofline"])[:900])
print("exact", l.get("exact_mode_tok_s")); oc=l.get("other_configs",{})
for k,v in oc.items(): print(k, json.dumps(v)[:300])
PY
echo "### profile"; bash tools/prof_bench.sh r05 2>&1 | tail -30
} >> $out 2>&1
