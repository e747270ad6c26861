#!/bin/bash
out=gpurun_out/r5_call7.txt; mkdir -p gpurun_out; : > $out
{
echo "### handover debug"; timeout 300 python tools/r5/ho_debug.py 2>&1 | grep -v amdgpu | tail -30
echo "### hf record"; timeout 900 python - <<'PY'
import json, torch, bench
print(json.dumps(bench.hf_generate_record(torch.device("cuda:0")), indent=1))
PY
echo "### tp + pipeline tests"; timeout 1500 python -m pytest tests/test_tp_gpu.py tests/test_pipeline_nccl_gpu.py -q -m gpu 2>&1 | grep -v "^  File" | tail -40
} >> $out 2>&1
