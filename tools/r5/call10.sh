#!/bin/bash
out=gpurun_out/r5_call10.txt; mkdir -p gpurun_out; : > $out
{
echo "### the crashing test alone, launches serialised"
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -X faulthandler -m pytest tests/test_decode_default_gpu.py -x -q -m gpu -k "split_along_k" 2>&1 | tail -60
echo "### the rest of the suite behind it"; timeout 1800 python -m pytest tests/test_decode_default_gpu.py tests/test_decode_gpu.py tests/test_handover_gpu.py tests/test_hf_routes_gpu.py tests/test_lnq_gpu.py tests/test_lutgemm_gpu.py tests/test_pipeline_nccl_gpu.py tests/test_prefill_native_gpu.py tests/test_qkv_rope_gpu.py tests/test_qtip_gpu.py tests/test_tp_gpu.py tests/test_compile_contract_gpu.py -q -m gpu --deselect tests/test_decode_default_gpu.py::test_decode_with_the_down_projection_split_along_k_over_blocks 2>&1 | tail -15
} >> $out 2>&1
