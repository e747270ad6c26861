#!/bin/bash
out=gpurun_out/r5_call12.txt; mkdir -p gpurun_out; : > $out
{
echo "### sampler tests"; python -m pytest tests/test_decode_gpu.py tests/test_hf_routes_gpu.py -q -m gpu 2>&1 | tail -4; echo "### greedy sampler on / off"; for r in 1 2; do python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-120; GQ_SAMPLE_GREEDY=0 python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-120; done; echo "### default bench (driver form)"; SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r5_bench_err.txt | tail -1 > gpurun_out/r05_bench_default_line.json; echo "bench wall seconds: $SECONDS"; tail -3 gpurun_out/r5_bench_err.txt
python - <<'PY'
import json
l=json.load(open("gpurun_out/r05_bench_default_line.json"))
print({k:l[k] for k in ("value","ms_per_step")}, l["roofline"])
print("exact", l.get("exact_mode_tok_s")); oc=l.get("other_configs",{})
for k,v in oc.items(): print(k, json.dumps(v)[:400])
for r in l["roofline_by_shape"]["rows"]: print(r)
PY
echo "### lm_head floor"; python tools/r5/lmhead_floor.py 2>&1 | tail -1
echo "### profile"; bash tools/prof_bench.sh r05 2>&1 | tail -30
} >> $out 2>&1
