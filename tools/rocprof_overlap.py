"""How much of a rocprofv3 kernel trace overlaps: wall time covered by >= 1 / >= 2 kernels, from the rocpd sqlite database.
Usage: python tools/rocprof_overlap.py <results.db>"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    rows = cur.execute("select start, end from kernels").fetchall()
    ev = []
    for s, e in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    busy1 = busy2 = 0
    depth, prev = 0, None
    for t, d in ev:
        if prev is not None and depth >= 1:
            busy1 += t - prev
            if depth >= 2:
                busy2 += t - prev
        depth += d
        prev = t
    tot = sum(e - s for s, e in rows)
    span = ev[-1][0] - ev[0][0] if ev else 0
    print(f"# overlap: {len(rows)} dispatches, sum of kernel durations {tot / 1e6:.3f} ms, time with >= 1 kernel running {busy1 / 1e6:.3f} ms, "
          f"with >= 2 running {busy2 / 1e6:.3f} ms ({100.0 * busy2 / max(busy1, 1):.1f} %), trace span {span / 1e6:.1f} ms")


if __name__ == "__main__":
    main()
