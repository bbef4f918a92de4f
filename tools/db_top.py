"""Prints the top_kernels view of a rocprofv3 results database.  python tools/db_top.py results.db [rows]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
print(f"{'calls':>6} {'total ms':>10} {'avg us':>10} {'%':>6}  kernel")
for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (n,)):
    print(f"{calls:6d} {total / 1e3:10.2f} {avg:10.1f} {pct:6.2f}  {name[:110]}")
