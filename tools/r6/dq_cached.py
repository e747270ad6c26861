"""dq kernel with the weights resident in the memory-side cache (2 buffers of one matrix rotating) vs streaming from HBM (> 512 MB rotating)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from guidedquant_amd import _lib
L = _lib.lib(); L.gq_set_ap_mode(0)
os.environ["GQ_DQ"] = "7"; os.environ["GQ_DQ_MIN_MWEIGHTS"] = "0"; L.gq_reset_env_cache()
for bits in (2, 3, 4):
    for nm in ("w1w3", "w2"):
        N, K = bench.SHAPES_8B[nm]
        hbm = bench.bench_ap_shape(nm, N, K, bits, iters=100)["us"]
        mall = bench.bench_ap_shape(nm, N, K, bits, iters=100, min_ws=1)["us"]
        print(json.dumps({"bits": bits, "shape": nm, "hbm_us": hbm, "cache_resident_us": mall}), flush=True)
os.environ["GQ_DQ"] = "0"; L.gq_reset_env_cache()
for bits in (2, 3, 4):
    N, K = bench.SHAPES_8B["w1w3"]
    print(json.dumps({"default kernels bits": bits, "hbm_us": bench.bench_ap_shape("w1w3", N, K, bits, iters=100)["us"],
                      "cache_resident_us": bench.bench_ap_shape("w1w3", N, K, bits, iters=100, min_ws=1)["us"]}), flush=True)
