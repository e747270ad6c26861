#!/bin/bash
# lm_head GEMV blocks per CU (GQ_DENSE_BPC) against the decode rate (bench.py --quick), two rounds
for r in 1 2; do for b in 4 2 3 6 8 16; do echo "== GQ_DENSE_BPC=$b: $(GQ_DENSE_BPC=$b python bench.py --quick --steps 400 --warmup 100 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*')"; done; done
