"""Timeline of the last `n` kernel dispatches of a rocprofv3 rocpd database: start offset, duration, gap to the previous kernel's end.
    python tools/rocprof_timeline.py <results.db> [n]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(cur.execute("select name, start, end from kernels order by start"))[-n:]
t0, prev = rows[0][1], rows[0][1]
busy = 0
for name, st, en in rows:
    print("%10.1f us  +%8.1f gap  %9.1f us  %s" % ((st - t0) / 1e3, (st - prev) / 1e3, (en - st) / 1e3, name.split("(")[0][:90]))
    prev = en
    busy += en - st
print("span %.1f us, kernels busy %.1f us" % ((rows[-1][2] - t0) / 1e3, busy / 1e3))
