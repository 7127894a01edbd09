#!/bin/bash
# PMC counters of the decode kernels (tools/decode_probe.py), separate passes -> gpurun_out/decode_pmc/pmc.md
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/decode_pmc; mkdir -p "$OUT"
CMD="python tools/decode_probe.py"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT" -o p1 -- $CMD > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS -d "$OUT" -o p2 -- $CMD > "$OUT/p2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d "$OUT" -o p3 -- $CMD > "$OUT/p3.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT" -o p4 -- $CMD > "$OUT/p4.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT" -o p5 -- $CMD > "$OUT/p5.log" 2>&1
python - <<'PY'
import collections, os, sqlite3
d = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/decode_pmc")
res = collections.defaultdict(dict)
for f in sorted(os.listdir(d)):
    if f.endswith("_results.db"):
        cur = sqlite3.connect(os.path.join(d, f)).cursor()
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for name, cn, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            short = name.split("(")[0].replace("void ", "")
            if short.startswith("gsd_"):
                acc[short][cn].append(val)
        for k, v in acc.items():
            for c, vals in v.items():
                res[k][c] = sum(vals) / len(vals)
cs = sorted({c for v in res.values() for c in v})
lines = ["| kernel | " + " | ".join(cs) + " |", "|---|" + "---:|" * len(cs)]
for k in sorted(res):
    lines.append(f"| `{k}` | " + " | ".join(f"{res[k].get(c, float('nan')):.4g}" for c in cs) + " |")
open(os.path.join(d, "pmc.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf "$OUT"/*/ 2>/dev/null; find "$OUT" -name "*.db" -delete
