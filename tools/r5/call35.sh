#!/bin/bash
out=gpurun_out/r5_call35.txt; mkdir -p gpurun_out; : > $out
{
timeout 1500 python -m pytest tests/test_hf_routes_gpu.py tests/test_decode_gpu.py tests/test_hf_path_gpu.py -q -m gpu 2>&1 | tail -4
python - <<'PY'
import json, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
rec = bench.hf_generate_record(torch.device("cuda:0"))
print(json.dumps({k: v for k, v in rec.items() if "tok_s" in k or "mem" in k}))
PY
} >> $out 2>&1
