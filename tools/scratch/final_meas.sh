#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
echo "## bench_ap default"; timeout 400 python tools/bench_ap.py 2>&1 | tail -12
echo "## bench_ap exact"; GQ_AP_EXACT=1 timeout 400 python tools/bench_ap.py 2>&1 | tail -12
echo "## bench_ap plane forced"; GQ_PL_MIN_MWEIGHTS=0 timeout 400 python tools/bench_ap.py 2>&1 | tail -12
echo "## bench_ap 70b default"; timeout 400 python tools/bench_ap.py --bits 2 --shapes 70b_wqkv 70b_wo 70b_w1w3 70b_w2 2>&1 | tail -4
echo "## bench_kernels"; timeout 400 python tools/bench_kernels.py 2>&1 | tail -14
echo "## qtip linear"; timeout 300 python tools/bench_qtip_linear.py 2>&1 | tail -4
echo "## qtip decode (Llama-2-7b shape, MLP 11008)"; python tools/qtip_decode_bench.py 11008 32 2>&1 | tail -1
echo "## qtip decode (power-of-two MLP 8192)"; python tools/qtip_decode_bench.py 8192 32 2>&1 | tail -1
echo "## qtip decode through the separate ops"; GQ_NATIVE_QTIP=0 python tools/qtip_decode_bench.py 11008 32 2>&1 | tail -1
for b in 2 3 4; do echo "## bench bits $b"; python bench.py --bits $b 2>&1 | tail -1; done
echo "## bench exact"; python bench.py --mode exact --no-cpu-baseline 2>&1 | tail -1
echo "## bench 70B"; timeout 900 python bench.py --model meta-llama/Llama-3.3-70B-Instruct --no-cpu-baseline 2>&1 | tail -1
} > gpurun_out/final_meas.txt 2>&1
tail -5 gpurun_out/final_meas.txt | cut -c1-300
