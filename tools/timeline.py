#!/usr/bin/env python
"""Prints the GPU timeline (start offset, duration, gap to the previous op) of N kernel dispatches and memory copies of a
rocprofv3 --kernel-trace --memory-copy-trace result (rocpd sqlite), ending SKIP ops before the last one.
usage: python tools/timeline.py DB [N] [SKIP]"""
import sqlite3
import sys

db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
rows = []
if "kernels" in tabs:
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows += [(s, e, nm.split("(")[0].replace("void ", "")) for s, e, nm in cur.execute("select start, end, name from kernels")]
for t in ("memory_copies", "memory_copy"):
    if t in tabs:
        rows += [(s, e, "memcpy " + str(nm)) for s, e, nm in cur.execute(f"select start, end, name from {t}")]
rows.sort()
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[-n - skip:len(rows) - skip]
t0 = rows[0][0]
prev_end = None
for s, e, nm in rows:
    gap = "" if prev_end is None else f"{(s - prev_end) / 1000:8.1f}"
    print(f"{(s - t0) / 1000:10.1f} us  dur {(e - s) / 1000:8.1f} us  gap {gap:>8}  {nm[:70]}")
    prev_end = e if prev_end is None else max(prev_end, e)
