#!/bin/bash
# A/B/C on one box: bench.py --quick of the current library against the variants named on the command line, alternating, 3 rounds,
# every run under its own timeout
for r in 1 2 3; do for v in base "$@"; do
  if [ "$v" = base ]; then unset GQ_LIB_PATH; else export GQ_LIB_PATH=$PWD/guidedquant_amd/abl_$v/libgq_hip.so; fi
  echo "== bench $v: $(timeout 120 python bench.py --quick --steps 400 --warmup 100 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')"
done; done
