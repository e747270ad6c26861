"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel duration stats -- one row per (kernel name, grid size), so that the
wqkv / w1w3 launches of one template (and wo / w2) get their own rows -- and per-kernel mean PMC counters."""
import sqlite3
import sys


def main(path, top=40):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    if cols:
        gcols = [c for c in ("grid_x", "grid_size_x", "grid_size", "workgroup_x", "workgroup_size_x") if c in cols]
        gsel = ", ".join(gcols) if gcols else "0"
        q = (f"select name, {gsel}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
             f"group by name{''.join(', ' + c for c in gcols)} order by sum(end-start) desc")
        try:
            rows = list(cur.execute(q))
            ng = max(1, len(gcols))
            tot = sum(r[-1] for r in rows) or 1
            print(f"{'kernel':84s} {'/'.join(gcols) or '-':>16s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
            for r in rows[:top]:
                n, g = r[0], "/".join(str(v) for v in r[1:1 + ng])
                c, a, mn, mx, s = r[1 + ng:]
                print(f"{n[:84]:84s} {g:>16s} {c:6d} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.1f}")
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


def counter_mean(path, kernel_substr, counter):
    """mean of one PMC counter over the dispatches of kernels whose name contains kernel_substr (None if absent)"""
    con = sqlite3.connect(path)
    try:
        rows = list(con.execute("select avg(value), count(*), min(kernel_name) from counters_collection where counter_name = ? and kernel_name like ?",
                                (counter, f"%{kernel_substr}%")))
    except Exception:
        return None
    return rows[0] if rows and rows[0][0] is not None else None


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print("==", p)
        main(p)
