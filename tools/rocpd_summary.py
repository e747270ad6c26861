"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel duration stats and per-kernel mean PMC counters."""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    if cols:
        q = "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc"
        try:
            rows = list(cur.execute(q))
            tot = sum(r[5] for r in rows) or 1
            print(f"{'kernel':90s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
            for n, c, a, mn, mx, s in rows[:25]:
                print(f"{n[:90]:90s} {c:6d} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.1f}")
        except Exception as e:  # noqa
            print("kernels view:", e, cols)
    try:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        if ccols:
            rows = list(cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
            last = None
            for k, c, v, n in rows:
                if k != last:
                    print("\n" + k[:120], f"(n={n})")
                    last = k
                print(f"   {c:32s} {v:18.1f}")
    except Exception as e:  # noqa
        print("counters:", e)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print("==", p)
        main(p)
