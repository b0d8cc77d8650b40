#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 results database (kernel name, launches, average / minimum / total time).

    rocprof_db_stats.py results.db [name-part [limit]]
    rocprof_db_stats.py results.db --seq N      the last N dispatches in order: start offset, duration, gap before (us)
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    like = sys.argv[2] if len(sys.argv) > 2 else ""
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    if like == "--seq":
        rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id "
                           f"order by d.start desc limit {limit}").fetchall()[::-1]
        t0, prev = rows[0][1], rows[0][1]
        print("start_us,dur_us,gap_us,kernel")
        for name, a, b in rows:
            print(f"{(a - t0) / 1e3:.1f},{(b - a) / 1e3:.1f},{(a - prev) / 1e3:.1f},{name[:70]}")
            prev = b
        return
    rows = cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), sum(d.end-d.start) "
                       f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 5 desc").fetchall()
    print("kernel,launches,avg_us,min_us,total_ms")
    for r in [r for r in rows if like in r[0]][:limit]:
        print(f"{r[0]},{r[1]},{r[2] / 1e3:.1f},{r[3] / 1e3:.1f},{r[4] / 1e6:.3f}")


if __name__ == "__main__":
    main()
