"""GPU probe: accuracy of the plane-MFMA (fast mode) GEMV as a function of the activation vector's dynamic range.
One (or several) hot channel(s) of magnitude R times the typical element; error against the fp64 oracle, in units of
sum|w||x| of the NON-hot elements and in units of the fp16 rounding of y."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ap_helpers import run_fused  # noqa: E402
from guidedquant_amd import _lib  # noqa: E402
from oracle import oracle  # noqa: E402

oracle.build()
N, K, bits = 256, 4096, 2
rng = np.random.default_rng(0)
codes = rng.integers(0, 4, (N, K), dtype=np.uint8)
q = oracle.ap_pack(codes, bits)
lut = np.sort(rng.normal(0, 0.02, (N, 4)).astype(np.float16), axis=1)
W = oracle.ap_dequant(q, lut, bits).astype(np.float64)
os.environ["GQ_PL_MIN_MWEIGHTS"] = "0"
for local in (1, 0):
    os.environ["GQ_PL_LOCAL"] = str(local)
    _lib.lib().gq_reset_env_cache()
    _lib.lib().gq_set_ap_mode(0)
    for nhot in (1, 6):
        for lr in range(0, 15, 2):
            R = 2.0**lr
            x = rng.normal(0, 1, K)
            hot = rng.choice(K, nhot, replace=False)
            x[hot] = R * np.sign(x[hot])
            x = x.astype(np.float16)
            got = run_fused(x, q, lut, bits).astype(np.float64)
            y64 = oracle.ap_gemv_f64(x, q, lut, bits)[0]
            xs = np.abs(x.astype(np.float64)).copy()
            xs[hot] = 0
            base = np.abs(W) @ xs
            err = np.abs(got - y64)
            r16 = 2.0**-11 * np.abs(y64)
            over = np.maximum(err - r16, 0)
            print(f"local={local} nhot={nhot} R=2^{lr:2d}  max err/sum|w||x|_nonhot = {(err / base).max():.2e}   max (err - fp16 rounding)/base = {(over / base).max():.2e}"
                  f"   rms err/rms y = {np.sqrt((err**2).mean() / (y64**2).mean()):.2e}", flush=True)
