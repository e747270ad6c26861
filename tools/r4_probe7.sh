#!/bin/bash
for st in 3 0; do
  echo "== GQ_ST=$st"
  GQ_ST=$st python tools/bench_ap.py --bits 2 --shapes 70b_wo 70b_w2 --launch resid 2>&1 | grep shape | cut -c1-130
done
GQ_ST=3 GQ_PL_MIN_MWEIGHTS=0 python - <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import oracle
oracle.build()
from ap_helpers import _check_fast, _fast
from test_ap_fused_gpu import run_fused, _layer, _rows
_fast()
os.environ["GQ_ST"] = "3"
from guidedquant_amd import _lib; _lib.lib().gq_reset_env_cache()
N, K, bits = 8192, 28672, 2
rng, q, lut = _layer(N, K, bits, 11)
x = rng.normal(0, 1, K).astype(np.float16); x[rng.choice(K, 5, replace=False)] *= 300
res = rng.normal(0, 1, N).astype(np.float16)
got = run_fused(x, q, lut, bits, residual=res)
rows = _rows(rng, N)
# residual epilogue: compare y = got - res in fp16 semantics through the helper on the no-residual launch
got0 = run_fused(x, q, lut, bits)
_check_fast(got0, x, q, lut, bits, oracle, rows=rows)
exp = (res.astype(np.float16) + got0.astype(np.float16)).astype(np.float16)
print("resid identical:", np.array_equal(exp.view(np.uint16), got.view(np.uint16)))
print("one-launch K=28672 ok")
PY
