#!/bin/bash
out=gpurun_out/r5_call30.txt; mkdir -p gpurun_out; : > $out
export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_stamps/libgq_hip.so
{
for e in 0 2 0 2; do echo "######## GQ_ST_EPI=$e"; GQ_ST_EPI=$e PT_FUSED=1 python tools/phase_timing.py 2 w1w3 2>&1 | tail -40; done
} >> $out 2>&1
