"""Dump the kernel timeline of a `rocprofv3 --kernel-trace` run (rocpd sqlite): start, duration, queue, name.

usage: python tools/prof_timeline.py <trace.db> [first_us last_us]
"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute(f"select start, end, name, {q or '0'} from kernels order by start"))
t0 = rows[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
print("# columns available:", cols)
for s, e, n, qq in rows:
    us = (s - t0) / 1e3
    if lo <= us <= hi:
        print(f"{us:12.1f} {(e - s) / 1e3:10.1f} q{qq} {n[:60]}")
