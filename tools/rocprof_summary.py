"""Summarises a rocprofv3 rocpd sqlite database (kernel trace, optional PMC) into a text table.
Usage: python tools/rocprof_summary.py <results.db> [--pmc]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("dptx::", "")
    return re.sub(r"\(.*\)$", "", name)[:100]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    print(f"# kernel trace: {len(rows)} dispatches, {tot / 1e3:.3f} ms GPU time")
    print(f"{'kernel':100s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:100s} {n:6d} {us:12.1f} {us / n:10.2f} {100 * us / tot:6.2f}")
    if "--pmc" in sys.argv:
        try:
            q = cur.execute("select p.name, p.counter_name, p.counter_value from pmc_events p").fetchall()
        except sqlite3.OperationalError as ex:
            print("pmc query failed:", ex, [r[1] for r in cur.execute("pragma table_info(pmc_events)")])
            return
        pm = {}
        for name, cname, val in q:
            a = pm.setdefault((short(name), cname), [0, 0.0])
            a[0] += 1
            a[1] += val
        print(f"\n{'kernel':100s} {'counter':>12s} {'dispatches':>10s} {'sum':>16s} {'avg/dispatch':>16s}")
        for (k, c), (n, v) in sorted(pm.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:100s} {c:>12s} {n:10d} {v:16.1f} {v / n:16.1f}")


if __name__ == "__main__":
    main()
