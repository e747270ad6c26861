#!/bin/bash
out=gpurun_out/r5_call6.txt; mkdir -p gpurun_out; : > $out
{
echo "### tests"; timeout 1500 python -m pytest tests/test_qtip_gpu.py tests/test_hf_routes_gpu.py tests/test_handover_gpu.py tests/test_tp_gpu.py -q -m gpu 2>&1 | grep -v "^  File" | tail -60
echo "### qtip decode: pre-transformed edges on / off"
for r in 1 2; do
python bench.py --backend qtip --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-140
GQ_QTIP_PRE=0 python bench.py --backend qtip --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-140
done
echo "### hf record"; timeout 900 python - <<'PY'
import json, torch, bench
print(json.dumps(bench.hf_generate_record(torch.device("cuda:0")), indent=1))
PY
echo "### 3-bit 16-wave stream variants"
for v in w16r2 w16r3 w16r4; do export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; echo "== $v"; GQ_ST=2 python tools/bench_ap.py --bits 3 --shapes wqkv w1w3 --launch norm 2>&1 | tail -3 | cut -c1-300; done
} >> $out 2>&1
