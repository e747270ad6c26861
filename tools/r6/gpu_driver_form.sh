#!/bin/bash
# the literal driver command (no --quick), N times, each line kept; PYTHONFAULTHANDLER on
# usage: tools/r6/gpu_driver_form.sh TAG [N]
TAG=${1:-r06}
N=${2:-3}
mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
for i in $(seq 1 $N); do
  T0=$SECONDS
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_form_$i.json 2> gpurun_out/${TAG}_driver_form_$i.err
  echo "run $i rc $? wall $((SECONDS - T0)) s"; tail -c 600 gpurun_out/${TAG}_driver_form_$i.err | tail -3
  python3 - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_driver_form_$i.json").read().strip().splitlines()[-1])
    oc = d.get("other_configs", {})
    print("value", d["value"], "tok_s_400", d["config"].get("tok_s_400"), "roofline", d["roofline"].get("frac"), d["roofline"].get("avg_launch_us"),
          "cpu", (d.get("cpu_baseline") or {}).get("value"), "exact", d.get("exact_mode_tok_s"), "wall", d.get("bench_wall_s"))
    for k, v in oc.items():
        print("  ", k, {kk: vv for kk, vv in v.items() if kk in ("tok_s", "error", "skipped", "default_call_tok_s", "leg_wall_s", "hf_module_tree_static_cache_tok_s", "hf_module_tree_captured_step_tok_s")})
except Exception as e:
    print("unparsed:", e)
PY
done
