#!/bin/bash
# plane-kernel configuration knobs (env) against the decode rate, every run under a timeout
run() { echo "== $*: $(env "$@" timeout 120 python bench.py --quick --steps 400 --warmup 100 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')"; }
run A=1
run GQ_PL_LOG2CS=0
run GQ_PL_LOG2CS=1
run GQ_PL_LOG2CS=2
run GQ_PL_S=1
run GQ_PL_HIMG=0
run GQ_PL_RAWX=0
run A=1
