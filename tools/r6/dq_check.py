"""round 6: the decode-to-fp16 MFMA kernel (ap_gemv.hip::ap_gemv_dq_kernel, GQ_DQ bit mask) against the default dispatch: accuracy on
sampled rows (tests/ap_helpers._check_fast: the fast-mode envelope against the oracle's fp64 product) and us per launch (bench.py's
bench_ap_shape: > 512 MB of weights rotating inside one captured graph), Llama-3-8B shapes, plain launch and the decode graph's form."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from ap_helpers import _check_fast, rmsnorm_ref, run_fused, silu_mul_ref  # noqa: E402
from guidedquant_amd import _lib, pack  # noqa: E402
from oracle import oracle  # noqa: E402

oracle.build()
L = _lib.lib()
L.gq_set_ap_mode(0)


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    L.gq_reset_env_cache()


# MODEL=70b: Llama-3.3-70B's launches (timing only makes sense with CHECK=0: the accuracy pass uses the 8B shapes)
SHAPES = {"8b": bench.SHAPES_8B, "70b": {"wqkv": (10240, 8192), "wo": (8192, 8192), "w1w3": (57344, 8192), "w2": (8192, 28672)},
          "7b": {"wqkv": (12288, 4096), "wo": (4096, 4096), "w1w3": (22016, 4096)}}[os.environ.get("MODEL", "8b")]
bits_list = [int(b) for b in os.environ.get("BITS", "2,3,4").split(",")]
if os.environ.get("CHECK", "1") != "0":
    for bits in bits_list:
        for N, K in ((6144, 4096), (4096, 14336), (1000, 4096), (28672, 4096)):
            setenv(GQ_DQ=7, GQ_DQ_MIN_MWEIGHTS=0)
            rng = np.random.default_rng(bits + N + K)
            q = pack.random_planes(N, K, bits, seed=N + K + bits)
            lut = np.sort(rng.normal(0, 0.02, (N, 1 << bits)).astype(np.float16), axis=1)
            x = rng.normal(0, 1, K)
            x[rng.choice(K, 4, replace=False)] *= 30.0
            x = x.astype(np.float16)
            rows = np.unique(np.concatenate([np.arange(0, 24), np.arange(N - 24, N), rng.integers(0, N, 48)]))
            got = run_fused(x, q, lut, bits)
            _check_fast(got, x, q, lut, bits, oracle, rows=rows)
            nw = (1 + 0.1 * rng.normal(0, 1, K)).astype(np.float16)
            got_n = run_fused(x, q, lut, bits, norm_weight=nw, eps=1e-5)
            _check_fast(got_n, rmsnorm_ref(x, nw, 1e-5), q, lut, bits, oracle, rows=rows)
            if N % 2 == 0:
                pr = run_fused(x, q, lut, bits, norm_weight=nw, eps=1e-5, flags=4, out_elems=N // 2)
                assert np.array_equal(pr.view(np.uint16), silu_mul_ref(got_n[0::2], got_n[1::2]).view(np.uint16))
            res = rng.normal(0, 1, N).astype(np.float16)
            got_r = run_fused(x, q, lut, bits, residual=res, flags=1)
            assert np.array_equal(got_r.view(np.uint16), (res.astype(np.float16) + got.astype(np.float16)).view(np.uint16))
            print("ok   bits", bits, N, K, flush=True)
for bits in bits_list:
    for nm, (N, K) in SHAPES.items():
        form = {"wqkv": "norm", "wo": "resid", "w1w3": "norm_pairs", "w2": "resid"}[nm]
        row = {}
        for tag, dq in (("default", 0), ("dq", 7), ("default2", 0), ("dq2", 7)):
            setenv(GQ_DQ=dq, GQ_DQ_MIN_MWEIGHTS=0)
            row[tag] = (bench.bench_ap_shape(nm, N, K, bits, iters=100)["us"], bench.bench_ap_shape(nm, N, K, bits, iters=100, fused=form)["us"])
        print(json.dumps({"bits": bits, "shape": nm, "plain/graph-form us": row}), flush=True)
