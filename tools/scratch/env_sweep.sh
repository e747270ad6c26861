#!/bin/bash
# runtime environment knobs against the decode rate (bench.py --quick): one knob per run under its own timeout (ROC_SYSTEM_SCOPE_SIGNAL=0
# hung a run for the whole call once), baseline first and last
run() { echo "== $*: $(env "$@" timeout 120 python bench.py --quick --steps 400 --warmup 100 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run A=1
