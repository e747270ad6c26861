#!/bin/bash
out=gpurun_out/r5_call46.txt; mkdir -p gpurun_out; : > $out
q() { python bench.py "$@" 2>/dev/null | tail -1 | cut -c40-75; }
{
echo "### 4 bits: wo through the plane path (now the local-image kernel) vs the exact kernel"
for cfg in "" "GQ_PL_MIN_MWEIGHTS=16"; do echo "[$cfg] wo $(env $cfg python tools/bench_ap.py --bits 4 --shapes wo --launch resid 2>&1 | grep shape | sed 's/.*"us": \([0-9.]*\).*/\1/') decode $(env $cfg python bench.py --bits 4 --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-75)"; done
echo "### QTIP decode forms (each switch off in turn)"
echo "default $(q --backend qtip --quick --steps 200 --warmup 40)"
for k in GQ_QTIP_MLP_MID GQ_QTIP_ATTN_FOLD GQ_QTIP_OUT_SEG GQ_QTIP_KSPLIT; do echo "$k=0 $(env $k=0 python bench.py --backend qtip --quick --steps 200 --warmup 40 2>/dev/null | tail -1 | cut -c40-75)"; done
echo "### Llama-3.3-70B 2-bit: K split of a row group (GQ_PL_LOG2CS) on the shared-image launches"
for cfg in "" "GQ_PL_LOG2CS=0" "GQ_PL_LOG2CS=1" "GQ_PL_LOG2CS=2"; do echo "[$cfg] $(env $cfg python bench.py --model meta-llama/Llama-3.3-70B-Instruct --quick --steps 100 --warmup 20 2>/dev/null | tail -1 | cut -c40-75)"; done
} >> $out 2>&1
