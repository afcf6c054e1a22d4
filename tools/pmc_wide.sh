#!/bin/bash
# PMC passes over the head_dim-448 forward (one shape, both kernel forms): where the cycles of a wave go.
# usage (GPU box): bash tools/pmc_wide.sh   -> gpurun_out/pmc_wide/*.csv + summary on stdout
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_wide"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SHAPES=12544x6272 MODES=${MODES:-1,2}
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace -d "$OUT/p$i" -o p$i --output-format csv -- python3 "$ROOT/tools/diag_wide_groups.py" > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_fwd_hd" not in k:
            continue
        name = "hd2" if "hd2" in k else "hd1"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in sorted(acc.items()):
    print(name, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
