python -m pytest tests/test_ap_gemv_gpu.py -m gpu -x -q 2>&1 | tail -2
for d in 1 2; do for bpc in 2 3 4; do echo "== exact D=$d BPC=$bpc"; GQ_AP_EXACT=1 GQ_AP_D=$d GQ_AP_BPC=$bpc python tools/bench_ap.py --bits 2 2>&1 | grep shape; done; done
echo "== exact bits 3 4"; GQ_AP_EXACT=1 python tools/bench_ap.py --bits 3 4 2>&1 | grep shape
