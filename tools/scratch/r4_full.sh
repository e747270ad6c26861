#!/bin/bash
# full GPU check of the round: every -m gpu test, smoke, the default (driver-form) bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r4_gputests.txt 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r4_gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('exact_mode_tok_s'))
print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:160])
oc=d.get('other_configs',{})
for k,v in oc.items(): print(k, json.dumps(v)[:300])
PY
