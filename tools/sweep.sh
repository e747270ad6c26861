python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== default"; python tools/bench_ap.py --bits 2 2>&1 | grep shape
for bpc in 1 2 3 4; do for d in 2 4; do echo "== BPC=$bpc D=$d"; GQ_AP_BPC=$bpc GQ_AP_D=$d python tools/bench_ap.py --bits 2 --shapes w1w3 w2 2>&1 | grep shape; done; done
echo "== bits 3,4 default"; python tools/bench_ap.py --bits 3 4 2>&1 | grep shape
