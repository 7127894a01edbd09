#!/usr/bin/env python
"""Turns a rocprofv3 (ROCm 7.2, rocpd sqlite) result into the text summary committed under profiles/.

  kernel stats : rocprofv3 --kernel-trace --stats -d DIR -o NAME -- <cmd>     -> DIR/NAME_results.db
  PMC counters : rocprofv3 --pmc C1 C2 ... -d DIR -o NAME -- <cmd>            (separate run, no --stats)

usage: python tools/rocprof_summary.py DB [--pmc] > profiles/rNN_xxx.md
"""
import sqlite3
import sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    cols = [d[0] for d in cur.execute("select * from top_kernels limit 1").description]
    mm = "min_ns" in cols and "max_ns" in cols
    rows = list(cur.execute(f"select name, total_calls, total_duration, average, percentage{', min_ns, max_ns' if mm else ''} from top_kernels"))
    print("| kernel | calls | total (us) | avg (us) | % of GPU time |" + (" min (us) | max (us) |" if mm else ""))
    print("|---|---:|---:|---:|---:|" + ("---:|---:|" if mm else ""))
    for row in rows:
        name, calls, total, avg, pct = row[:5]
        short = name.split("(")[0].replace("void ", "")
        print(f"| `{short}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f} |" + (f" {row[5] / 1e3:.1f} | {row[6] / 1e3:.1f} |" if mm else ""))


def pmc(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        print("no counters_collection view; tables:", tabs)
        return
    cols = [d[0] for d in cur.execute(f"select * from {view} limit 1").description]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    q = f"select {kcol}, counter_name, count(*), sum(value), avg(value) from {view} group by {kcol}, counter_name order by {kcol}"
    print("| kernel | counter | dispatches | sum | avg per dispatch |")
    print("|---|---|---:|---:|---:|")
    for name, cname, n, tot, avg in cur.execute(q):
        short = name.split("(")[0].replace("void ", "")
        if not short.startswith("gsr_"):
            continue
        print(f"| `{short}` | {cname} | {n} | {tot:.6g} | {avg:.6g} |")


if __name__ == "__main__":
    (pmc if "--pmc" in sys.argv else kernel_stats)(sys.argv[1])
