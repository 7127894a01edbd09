#!/usr/bin/env python
"""Per-dispatch durations of the kernels whose name contains a substring, from a rocprofv3 (rocpd sqlite) result.
usage: python tools/rocprof_durations.py DB substring"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print("tables:", tabs)
    sys.exit(0)
cols = [d[0] for d in cur.execute(f"select * from {view} limit 1").description]
ncol = "name" if "name" in cols else "kernel_name"
rows = list(cur.execute(f"select {ncol}, start, end from {view} where {ncol} like ? order by start", (f"%{sys.argv[2]}%",)))
d = [(e - s) / 1e3 for _, s, e in rows]
print(len(d), "dispatches; first 12 (us):", [round(x, 1) for x in d[:12]], "even-mean", round(sum(d[0::2]) / max(len(d[0::2]), 1), 1), "odd-mean", round(sum(d[1::2]) / max(len(d[1::2]), 1), 1))
