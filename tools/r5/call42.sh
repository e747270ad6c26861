#!/bin/bash
out=gpurun_out/r5_call42.txt; mkdir -p gpurun_out; : > $out
q() { python bench.py "$@" 2>/dev/null | tail -1 | cut -c40-75; }
{
echo "### lm_head blocks per CU (GQ_DENSE_BPC, default 3): 2-bit decode"
for v in 2 3 4 5 6 8; do echo "GQ_DENSE_BPC=$v $(GQ_DENSE_BPC=$v q --quick --steps 300 --warmup 60)"; done
echo "### 4 bits: wo / w2 on the local-image kernel (GQ_PL_LOCAL_MAXBITS=4) vs default"
for cfg in "" "GQ_PL_LOCAL_MAXBITS=4"; do
  echo "[$cfg] $(env $cfg python tools/bench_ap.py --bits 4 --shapes wo w2 --launch resid 2>&1 | grep shape | sed 's/.*"shape": "\([a-z0-9]*\)".*"us": \([0-9.]*\).*/\1 \2/' | tr '\n' ' ') decode $(env $cfg python bench.py --bits 4 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-75)"
done
echo "### 4 bits: wo on the plane kernel instead of the exact kernel (GQ_PL_MIN_MWEIGHTS=16)"
echo "[GQ_PL_MIN_MWEIGHTS=16] $(GQ_PL_MIN_MWEIGHTS=16 python tools/bench_ap.py --bits 4 --shapes wo --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/') decode $(GQ_PL_MIN_MWEIGHTS=16 python bench.py --bits 4 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-75)"
echo "### 3 bits: wo on the exact kernel (GQ_PL_LOCAL=0 GQ_PL_MIN_MWEIGHTS=20) vs default local"
echo "[default] $(python tools/bench_ap.py --bits 3 --shapes wo --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/')"
echo "### Llama-3.3-70B 2-bit: stream kernel for every RMSNorm launch it serves (GQ_ST=2) vs default"
for cfg in "" "GQ_ST=2"; do echo "[$cfg] $(env $cfg python bench.py --model meta-llama/Llama-3.3-70B-Instruct --quick --steps 100 --warmup 20 2>/dev/null | tail -1 | cut -c40-75)"; done
echo "### exact mode: blocks per CU of the exact kernel (GQ_AP_BPC)"
for v in 1 2 3 4; do echo "GQ_AP_BPC=$v $(GQ_AP_BPC=$v q --mode exact --quick --steps 200 --warmup 40)"; done
echo "default $(q --mode exact --quick --steps 200 --warmup 40)"
echo "### QTIP: GQ_QTIP_SX"
for v in 0 1; do echo "GQ_QTIP_SX=$v $(GQ_QTIP_SX=$v q --backend qtip --quick --steps 200 --warmup 40)"; done
} >> $out 2>&1
