#!/bin/bash
# usage (GPU box): KERNELS=regex bash tools/gpu_pmc_probe.sh "CTR1 CTR2 ..." ["CTR..." ...] -- one rocprofv3 --pmc pass per argument over tools/decode_probe.py
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcprobe; rm -rf "$OUT"; mkdir -p "$OUT"
CMD=${CMD:-python tools/decode_probe.py}
if [ "${LIST:-0}" = 1 ]; then rocprofv3 --list-avail 2>/dev/null | grep -oE "Name:\s+[A-Za-z0-9_]+|^\s*[A-Z][A-Za-z0-9_]{4,}" | sort -u | tr '\n' ' ' | fold -w 200 > "$OUT/avail.txt"; fi
i=0
for grp in "$@"; do i=$((i+1)); rocprofv3 --pmc $grp -d "$OUT" -o p$i -- $CMD > "$OUT/p$i.log" 2>&1; done
python - <<'PY'
import collections, os, re, sqlite3
d = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmcprobe")
pat = re.compile(os.environ.get("KERNELS", "gsd_"))
res = collections.defaultdict(dict)
for root, _, files in os.walk(d):
    for f in files:
        if f.endswith("_results.db"):
            cur = sqlite3.connect(os.path.join(root, f)).cursor()
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for name, cn, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
                short = name.split("(")[0].replace("void ", "")
                if pat.search(short): acc[short][cn].append(val)
            for k, v in acc.items():
                for c, vals in v.items(): res[k][c] = sum(vals) / len(vals)
for k in sorted(res):
    print(k)
    for c in sorted(res[k]): print("    %-34s %.5g" % (c, res[k][c]))
PY
find "$OUT" -name "*.db" -delete; find "$OUT" -type d -empty -delete
