"""Summarise rocprofv3 (rocpd sqlite) outputs into a small text file for profiles/.

usage: python tools/prof_summary.py <stats.db> [<pmc.db> ...] > profiles/rNN_xxx.txt
Kernel table = `rocprofv3 --kernel-trace --stats` (top_kernels view); PMC tables = per-kernel sum and
per-launch mean of each collected counter (`--pmc X --kernel-trace`, one counter per pass).
"""
import sqlite3
import sys


def kernel_stats(path):
    cur = sqlite3.connect(path).cursor()
    print(f"# kernel stats: {path}")
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:72]:72s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}")
    print()


def pmc(path):
    cur = sqlite3.connect(path).cursor()
    print(f"# counters: {path}")
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name order by sum(value) desc")
    print(f"{'kernel':60s} {'counter':12s} {'launches':>8s} {'sum':>16s} {'mean/launch':>14s} {'avg_ns':>10s}")
    for k, c, n, s, a, d in rows:
        print(f"{k[:60]:60s} {c:12s} {n:8d} {s:16.1f} {a:14.2f} {d:10.0f}")
    print()


if __name__ == "__main__":
    kernel_stats(sys.argv[1])
    for p in sys.argv[2:]:
        pmc(p)
