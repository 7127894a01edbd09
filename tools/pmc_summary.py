#!/usr/bin/env python
"""Summarises the rocprofv3 --pmc passes written by tools/pmc_run.sh:
   * prints a markdown table of the counters per gsr_* kernel (averages per dispatch), and
   * writes profiles/pmc_latest.json = {workload: {stage: HBM bytes per launch}} which bench.py reports as
     roofline.traffic.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB and come from
separate passes; on gfx950 FETCH_SIZE reports HALF of the bytes of wide (16 B/lane) reads, so the read side is doubled
(our kernels read with 16-byte loads); WRITE_SIZE is used as reported (uncalibrated).

usage: python tools/pmc_summary.py gpurun_out/pmc config2 > profiles/rNN_pmc.md
"""
import collections
import json
import os
import sqlite3
import sys

# kernel-name prefix -> bench.py stage (template arguments vary: matched on the name in front of '<')
STAGE_PREFIX = {
    "gsr_preprocess_kernel": "preprocess", "gsr_scan_reduce_kernel": "count_scan", "gsr_scan_sums_kernel": "count_scan",
    "gsr_scan_apply_kernel": "count_scan", "gsr_tile_hist_kernel": "count_scan", "gsr_table_colscan_kernel": "count_scan",
    "gsr_tile_scan_kernel": "count_scan", "gsr_scatter_kernel": "scatter", "gsr_cursor_init_kernel": "scatter",
    "gsr_tile_sort_lds_kernel": "tile_sort", "gsr_tile_sort_near_kernel": "tile_sort", "gsr_tile_sort_global_kernel": "tile_sort",
    "gsr_blend_fwd_kernel": "blend_forward", "gsr_blend_bwd_kernel": "blend_backward", "gsr_gauss_bwd_kernel": "gauss_backward",
}


class _StageOf:
    def get(self, kernel, default=None):
        return STAGE_PREFIX.get(kernel.split("<")[0], default)


STAGE_OF = _StageOf()


def load(db):
    cur = sqlite3.connect(db).cursor()
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, cn, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        short = name.split("(")[0].replace("void ", "")
        if short.startswith("gsr_"):
            out[short][cn].append(val)
    return out


def launches_of(tot, stage):
    """dispatch count of the stage's most-launched kernel in the SQ_INSTS_VALU pass"""
    return max((v["SQ_INSTS_VALU"][1] for k, v in tot.items() if STAGE_OF.get(k) == stage and "SQ_INSTS_VALU" in v), default=0)


def main():
    d, workload = sys.argv[1], sys.argv[2]
    res = collections.defaultdict(dict)
    tot = collections.defaultdict(dict)   # per kernel and counter: sum over its dispatches, dispatch count
    for f in sorted(os.listdir(d)):
        if f.endswith("_results.db"):
            for k, v in load(os.path.join(d, f)).items():
                for c, vals in v.items():
                    res[k][c] = sum(vals) / len(vals)
                    tot[k][c] = (sum(vals), len(vals))
    counters = sorted({c for v in res.values() for c in v})
    print("| kernel | " + " | ".join(counters) + " |")
    print("|---|" + "---:|" * len(counters))
    for k in sorted(res):
        print(f"| `{k}` | " + " | ".join(f"{res[k].get(c, float('nan')):.4g}" for c in counters) + " |")
    # per-iteration traffic of a stage: sums over ALL dispatches of its kernels divided by the dispatch count of the
    # stage's most-launched kernel, so that a kernel launched once (a first, non-speculative call) counts once
    traffic = collections.defaultdict(float)
    launches = collections.defaultdict(lambda: {"FETCH_SIZE": 0, "WRITE_SIZE": 0})
    for k, v in tot.items():
        st = STAGE_OF.get(k)
        if st:
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                launches[st][c] = max(launches[st][c], v.get(c, (0, 0))[1])
    for k, v in tot.items():
        st = STAGE_OF.get(k)
        if st and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            traffic[st] += 2.0 * v["FETCH_SIZE"][0] / launches[st]["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"][0] / launches[st]["WRITE_SIZE"] * 1024
    print("\nHBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB -> B):")
    for st, b in traffic.items():
        print(f"  {st:16s} {b / 1e6:9.1f} MB")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_latest.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[workload] = {k: round(v) for k, v in traffic.items()}
    # wave-level VALU instructions per launch of each stage (bench.py: roofline.valu)
    insts = collections.defaultdict(float)
    for k, v in tot.items():
        st = STAGE_OF.get(k)
        if st and "SQ_INSTS_VALU" in v and launches[st]["FETCH_SIZE"]:
            insts[st] += v["SQ_INSTS_VALU"][0] / max(v["SQ_INSTS_VALU"][1], 1) * (v["SQ_INSTS_VALU"][1] / max(launches_of(tot, st), 1))
    cur[workload]["_insts_valu"] = {k: round(v) for k, v in insts.items()}
    # provenance: which kernels these counters were measured on (bench.py refuses to replay them onto other kernels)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import provenance
    cur[workload]["_provenance"] = provenance.stamp()
    json.dump(cur, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
