"""exact-order 2-bit kernels (GQ_AP_EXACT): us per launch, weights streaming from HBM vs resident in the memory-side cache, over the
configuration knobs of pick_quad_cfg (GQ_AP_D ring depth, GQ_AP_BPC blocks per CU, GQ_AP_T block size, GQ_AP_PT pair table)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from guidedquant_amd import _lib
L = _lib.lib(); L.gq_set_ap_mode(1)
def setenv(**kw):
    for k, v in kw.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    L.gq_reset_env_cache()
bits = int(os.environ.get("BITS", "2"))
forms = {"wqkv": "norm", "wo": "resid", "w1w3": "norm_pairs", "w2": "resid"}
SHAPES = {"8b": bench.SHAPES_8B, "70b": {"wqkv": (10240, 8192), "wo": (8192, 8192), "w1w3": (57344, 8192), "w2": (8192, 28672)}}[os.environ.get("MODEL", "8b")]
for nm in os.environ.get("SHAPES", "w1w3,w2,wqkv,wo").split(","):
    N, K = SHAPES[nm]
    for cfg in json.loads(os.environ.get("CFGS", "[{}]")):
        setenv(GQ_AP_D=None, GQ_AP_BPC=None, GQ_AP_T=None, GQ_AP_PT=None)
        setenv(**cfg)
        try:
            r = {"plain_hbm": bench.bench_ap_shape(nm, N, K, bits, iters=100)["us"], "plain_cached": bench.bench_ap_shape(nm, N, K, bits, iters=100, min_ws=1)["us"],
                 "form_hbm": bench.bench_ap_shape(nm, N, K, bits, iters=100, fused=forms[nm])["us"]}
        except Exception as e:
            r = {"error": str(e)[:80]}
        print(json.dumps({"shape": nm, "bits": bits, "cfg": cfg, **r}), flush=True)
