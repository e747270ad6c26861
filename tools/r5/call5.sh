#!/bin/bash
out=gpurun_out/r5_call5.txt; mkdir -p gpurun_out; : > $out
{
echo "### tests"; timeout 900 python -m pytest tests/test_hf_routes_gpu.py tests/test_handover_gpu.py tests/test_decode_gpu.py tests/test_qkv_rope_gpu.py -q -m gpu 2>&1 | grep -v "^  File" | tail -150
echo "### pipeline tests"; timeout 1200 python -m pytest tests/test_pipeline_nccl_gpu.py -q -m gpu -x 2>&1 | tail -30
echo "### hf record"; timeout 900 python - <<'PY'
import json, torch, bench
print(json.dumps(bench.hf_generate_record(torch.device("cuda:0")), indent=1))
PY
} >> $out 2>&1
{
echo "### 3-bit stream kernel with 16 waves (GQ_ST=2: every RMSNorm launch it serves): w1w3 / wqkv norm, base = 8 waves; GQ_ST=1 = plane kernel"
for v in base w16r2 w16r3 w16r4; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"; GQ_ST=2 python tools/bench_ap.py --bits 3 --shapes wqkv w1w3 --launch norm 2>&1 | grep shape | cut -c1-150
done
unset GQ_LIB_PATH
echo "== plane kernel"; python tools/bench_ap.py --bits 3 --shapes wqkv w1w3 --launch norm 2>&1 | grep shape | cut -c1-150
} >> gpurun_out/r5_call5.txt 2>&1
