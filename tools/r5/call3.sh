#!/bin/bash
# round 5, GPU call 3: the statistics hand-over after the fixes -- parity tests, producer store modes, the decode graph
out=gpurun_out/r5_call3.txt; mkdir -p gpurun_out; : > $out
{
echo "### tests"; python -m pytest tests/test_handover_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "### producer store modes (us): resid / resid_ho per PL_SSQ_MODE (base = 2)"
for r in 1 2; do for v in base ssq0 ssq1 ssq3; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== $v"
  python tools/bench_ap.py --bits 2 --shapes wo w2 --launch resid_ho 2>&1 | grep shape | cut -c1-150
done; done
unset GQ_LIB_PATH
python tools/bench_ap.py --bits 2 --shapes wo w2 --launch resid 2>&1 | grep shape | cut -c1-150
echo "### consumer forms"
for r in 1 2; do
python tools/bench_ap.py --bits 2 --shapes wqkv --launch qkv_rope 2>&1 | grep shape | cut -c1-150
python tools/bench_ap.py --bits 2 --shapes wqkv --launch qkv_rope_ho 2>&1 | grep shape | cut -c1-150
python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs 2>&1 | grep shape | cut -c1-150
python tools/bench_ap.py --bits 2 --shapes w1w3 --launch norm_pairs_ho 2>&1 | grep shape | cut -c1-150
done
echo "### bench quick: hand-over on / off, per producer store mode"
for r in 1 2; do
python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-120
GQ_SSQ_HANDOVER=0 python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-120
GQ_LIB_PATH=$PWD/guidedquant_amd/abl_ssq1/libgq_hip.so python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-120
GQ_LIB_PATH=$PWD/guidedquant_amd/abl_ssq3/libgq_hip.so python bench.py --quick --steps 300 --warmup 50 2>/dev/null | tail -1 | cut -c1-120
done
echo "### phase stamps with the hand-over"; PT_FUSED=2 python tools/phase_timing.py 2 wqkv w1w3 2>&1 | grep -v amdgpu
echo "### the AP GPU suites"; python -m pytest tests/test_ap_stream_gpu.py tests/test_ap_fused_gpu.py tests/test_qkv_rope_gpu.py tests/test_decode_default_gpu.py tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -8
} >> $out 2>&1
