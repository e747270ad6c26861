#!/bin/bash
out=gpurun_out/r5_call63.txt; mkdir -p gpurun_out; : > $out
q() { python bench.py "$@" 2>/dev/null | tail -1 | cut -c40-60; }
{
echo "### HIP runtime environment knobs vs the 2-bit decode (bench.py --quick --steps 300 --warmup 60)"
echo "default: $(q --quick --steps 300 --warmup 60)"
for cfg in "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=256" "GPU_MAX_HW_QUEUES=1" "GPU_MAX_HW_QUEUES=8" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "ROC_SYSTEM_SCOPE_SIGNAL=0" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_SDMA=0"; do
  echo "$cfg: $(env $cfg python bench.py --quick --steps 300 --warmup 60 2>/dev/null | tail -1 | cut -c40-60)"
done
echo "default: $(q --quick --steps 300 --warmup 60)"
} >> $out 2>&1
