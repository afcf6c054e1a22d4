#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the bench attention launch under the two unit orders (tools/pmc_attn_order.py).  usage (GPU box): bash tools/pmc_attn_order.sh
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_attn_order"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d "$OUT/$c" -o $c --output-format csv -- python3 "$ROOT/tools/pmc_attn_order.py" > "$OUT/$c.log" 2>&1 || { echo "pass $c failed"; tail -5 "$OUT/$c.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
rows = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_fwd3" not in k: continue
        mm = re.search(r"(attn_fwd3_kernel<[^>]*>)", k)
        rows[(mm.group(1), r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for (k, c), v in sorted(rows.items()):
    v.sort()
    vals = [x[1] for x in v]
    h = len(vals) // 2
    scale = 2.0 if c == "FETCH_SIZE" else 1.0      # gfx950: FETCH_SIZE counts 64 B per 128-B request
    unit = 1e3 if max(vals) < 1e9 else 1e6          # KB or bytes
    a, p = sum(vals[:h]) / h * scale / unit, sum(vals[h:]) / (len(vals) - h) * scale / unit
    print(f"{k} {c}{' x2' if scale == 2 else ''}: XCD-affine order {a:8.1f} MB | position order {p:8.1f} MB per launch")
PY
