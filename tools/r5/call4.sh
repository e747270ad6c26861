#!/bin/bash
out=gpurun_out/r5_call4.txt; mkdir -p gpurun_out; : > $out
{
echo "### tests"; python -m pytest tests/test_hf_routes_gpu.py tests/test_handover_gpu.py tests/test_decode_gpu.py tests/test_qkv_rope_gpu.py -q -m gpu 2>&1 | tail -60
echo "### bench quick: embedding folded into the sampler / separate launch"
for r in 1 2; do
python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-120
GQ_FOLD_EMBED=0 python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-120
done
echo "### hf record"; python - <<'PY'
import json, torch, bench
print(json.dumps(bench.hf_generate_record(torch.device("cuda:0")), indent=1))
PY
} >> $out 2>&1
