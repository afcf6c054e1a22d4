#!/bin/bash
# PMC passes over the head_dim-448 backward kernels (the shape of tools/diag_bwd_hd_ablate.py): where the cycles of a wave go.
# usage (GPU box): bash tools/pmc_bwd_hd.sh   -> gpurun_out/pmc_bwd_hd/*.csv + summary on stdout
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_bwd_hd"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export BHD_V=pmc
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace -d "$OUT/p$i" -o p$i --output-format csv -- python3 "$ROOT/tools/diag_bwd_hd_ablate.py" one > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_bwd_hd" not in k:
            continue
        name = "MODE " + k.split("attn_bwd_hd_kernel<")[1].split(",")[1].strip()
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    print(name, {c: round(v) for c, v in m.items()})
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        print("   of wave cycles: active %.2f wait_any %.2f wait_inst %.2f | lds: wait_inst %.2f active %.2f idx_active/busy %.2f conflict/idx_active %.2f | mfma busy %.2f" % (
            m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_WAIT_INST_LDS", 0) / wc,
            m.get("SQ_ACTIVE_INST_LDS", 0) / wc, m.get("SQ_LDS_IDX_ACTIVE", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1),
            m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m.get("SQ_BUSY_CYCLES", 1), 1)))
PY
