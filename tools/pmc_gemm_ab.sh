#!/bin/bash
# PMC passes: 256-row GEMM kernels against the 128x256 two-per-CU kernel on the same shapes (tools/pmc_gemm_ab.py).
# usage (GPU box): bash tools/pmc_gemm_ab.sh   -> gpurun_out/pmc_gemm_ab/ + summary on stdout
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_gemm_ab"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace -d "$OUT/p$i" -o p$i --output-format csv -- python3 "$ROOT/tools/pmc_gemm_ab.py" > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
disp = collections.defaultdict(dict)
name = {}
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k:
            continue
        i = int(r["Dispatch_Id"])
        disp[i][r["Counter_Name"]] = float(r["Counter_Value"])
        mm = re.search(r"(gemm\w*<[^>]*>)", k)
        name[i] = (mm.group(1) if mm else k[:44], r.get("Grid_Size", "?"))
grp = collections.defaultdict(list)
for i, d in sorted(disp.items()):
    grp[name[i]].append(d)
for key, ds in grp.items():
    m = {c: sum(d.get(c, 0) for d in ds) / len(ds) for c in ds[0]}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8 or 1
    print(key, "n", len(ds), "cycles %.0f mfma_util %.3f" % (gui, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / gui),
          "| wave cycles: active %.2f wait_any %.2f wait_inst %.2f | L2 hit %.3f (req %.1f M) FETCHx2 %.1f MB | lds busy %.2f confl %.2f" % (
              m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc,
              m.get("TCC_HIT_sum", 0) / ((m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) or 1),
              (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) / 1e6, 2 * m.get("FETCH_SIZE", 0) / 1e3 if m.get("FETCH_SIZE", 0) < 1e9 else 2 * m.get("FETCH_SIZE", 0) / 1e6,
              m.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / gui, m.get("SQ_LDS_BANK_CONFLICT", 0) / (m.get("SQ_LDS_IDX_ACTIVE", 0) or 1)))
PY
