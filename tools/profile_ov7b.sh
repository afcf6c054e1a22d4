#!/bin/bash
# rocprofv3 --kernel-trace --stats over the OneVision-7B width (BASELINE.json configs[2]: 256-frame videos, D = 3584, M = 8), one
# stream x a row batch of 4 - the per-kernel summary behind the 7B numbers of DESIGN.md section 7.
# usage (GPU box): bash tools/profile_ov7b.sh   -> gpurun_out/ov7b/ov7b_kernel_stats.csv
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/ov7b"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export HIDDEN=3584 FRAMES=256 STEPS=3
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o ov7b -- python3 "$ROOT/tools/diag_batch_modes.py" 8 1x4 > "$OUT/run.log" 2> "$OUT/run.err" || { echo "rocprof failed"; tail -5 "$OUT/run.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
out = sys.argv[1]
f = sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*$", "", n).replace("void ", "")[:70]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(out, "ov7b_kernel_stats.csv"), "w") as g:
    g.write("# rocprofv3 --kernel-trace --stats -- python3 tools/diag_batch_modes.py 8 1x4 (HIDDEN=3584 FRAMES=256 STEPS=3): kernel, calls, total us, average us, percent\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
        g.write(f'"{short(r["Name"])}",{r["Calls"]},{float(r["TotalDurationNs"]) / 1e3:.1f},{float(r["AverageNs"]) / 1e3:.1f},{100 * float(r["TotalDurationNs"]) / tot:.2f}\n')
print(open(os.path.join(out, "ov7b_kernel_stats.csv")).read())
PY
