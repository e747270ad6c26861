#!/bin/bash
# round 5, GPU call 1: knock-outs of the stream kernel, the masked LDS read probe, local-kernel phase stamps, exact-kernel knobs
out=gpurun_out/r5_call1.txt; mkdir -p gpurun_out; : > $out
{
echo "### lds_masked"; tools/ubench/lds_masked
echo "### stream variants (us per launch; wqkv / w1w3, RMSNorm prologue)"
for r in 1 2; do for v in r4 base ko1 ko2 ko8 ko16 ko19; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  python tools/bench_ap.py --bits 2 --shapes wqkv w1w3 --launch norm 2>&1 | grep shape | cut -c1-150
done; done
unset GQ_LIB_PATH
echo "### base, the decode graph's launch forms"
python tools/bench_ap.py --bits 2 --shapes wqkv --launch qkv_rope 2>&1 | grep shape | cut -c1-150
python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-150
python tools/bench_ap.py --bits 2 --shapes wo w2 --launch resid 2>&1 | grep shape | cut -c1-150
echo "### phase stamps: stream kernel (base)"; PT_FUSED=1 python tools/phase_timing.py 2 wqkv w1w3 2>&1 | grep -v amdgpu
echo "### phase stamps: local kernel"; python tools/phase_timing.py 2 wo w2 2>&1 | grep -v amdgpu
echo "### tests: base"; python -m pytest tests/test_ap_stream_gpu.py tests/test_ap_fused_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "### tests: ko2"; GQ_LIB_PATH=$PWD/guidedquant_amd/abl_ko2/libgq_hip.so python -m pytest tests/test_ap_stream_gpu.py tests/test_ap_fused_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "### exact kernels: D x BPC (2-bit, plain launch)"
for d in 1 2; do for b in 2 3 4; do echo "== D=$d BPC=$b"; GQ_AP_EXACT=1 GQ_AP_D=$d GQ_AP_BPC=$b python tools/bench_ap.py --bits 2 3 --shapes wqkv wo w1w3 w2 2>&1 | grep shape | cut -c1-120; done; done
echo "### bench quick (base / r4)"
python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-200
GQ_LIB_PATH=$PWD/guidedquant_amd/abl_r4/libgq_hip.so python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-200
} >> $out 2>&1
