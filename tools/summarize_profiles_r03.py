"""Turns gpurun_out/prof_<tag>/ (tools/profile_round3.sh: every pass runs bench.py itself) into the summaries committed under
profiles/:
  <tag>_kernel_stats_1stream.csv / <tag>_kernel_stats.csv     name, calls, total us, average us, % (rocprofv3 --stats)
  <tag>_bench_under_rocprof*.json                              the bench line printed by the same command
  <tag>_mfma_utilisation.csv                                   matrix-pipe busy share per kernel and shape (PMC)
  <tag>_attn_fwd_hbm_traffic.json                              FETCH_SIZE x2 (gfx950) + WRITE_SIZE per launch over the bench's
                                                               real launch mix (PMC); feeds roofline.traffic
usage: python tools/summarize_profiles_r03.py <tag>"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*", "", name).replace("void ", "")


def stats(sub, out):
    files = sorted(glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]
    if not files:
        print("no kernel_stats for", sub)
        return
    rows = list(csv.DictReader(open(files[0])))
    with open(os.path.join(dst, out), "w") as f:
        f.write("name,calls,total_us,average_us,percent\n")
        for r in rows:
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs'])/1e3:.1f},{float(r['AverageNs'])/1e3:.2f},{r['Percentage']}\n")
    js = os.path.join(src, sub + ".json")
    if os.path.exists(js):
        line = [l for l in open(js) if l.startswith("{")]
        if line:
            open(os.path.join(dst, out.replace("kernel_stats", "bench_under_rocprof").replace(".csv", ".json")), "w").write(line[-1])


def per_dispatch(sub):
    """[{kernel, grid, counter: value...}] one record per dispatch, in dispatch order"""
    files = sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    rec = collections.OrderedDict()
    for fn in files:
        for r in csv.DictReader(open(fn)):
            d = rec.setdefault(int(r["Dispatch_Id"]), {})
            d["kernel"], d["grid"] = short(r["Kernel_Name"]), r.get("Grid_Size", "")
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [rec[k] for k in sorted(rec)]


stats("stats1", f"{tag}_kernel_stats_1stream.csv")
stats("stats2", f"{tag}_kernel_stats.csv")

mine = ("attn_", "gemm", "layernorm", "row_add", "frame_")
with open(os.path.join(dst, f"{tag}_mfma_utilisation.csv"), "w") as f:
    f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace on bench.py itself (tools/profile_round3.sh:\n")
    f.write("# one stream, a row batch of 2 videos per step), per-dispatch averages\n")
    f.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) / (GRBM_GUI_ACTIVE / 8 XCDs); profiled clocks are lower than un-profiled ones\n")
    f.write("# dispatches of one kernel are grouped by their MFMA-busy count (a function of the shape)\n")
    f.write("kernel,grid_threads,dispatches,mfma_busy_cycles_sum,kernel_cycles,mfma_util\n")
    groups = collections.defaultdict(list)
    for d in per_dispatch("mfma"):
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and d.get("GRBM_GUI_ACTIVE", 0) > 0 and d["kernel"].startswith(mine):
            groups[(d["kernel"], d["grid"], round(d["SQ_VALU_MFMA_BUSY_CYCLES"], -5))].append(d)
    for (k, grid, _), ds in sorted(groups.items()):
        busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in ds) / len(ds)
        cyc = sum(d["GRBM_GUI_ACTIVE"] for d in ds) / len(ds) / 8.0
        f.write(f"\"{k}\",{grid},{len(ds)},{busy:.0f},{cyc:.0f},{busy / 1024.0 / cyc:.3f}\n")

fetch, write = per_dispatch("fetch"), per_dispatch("write")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py itself (tools/profile_round3.sh: one "
               "stream, a row batch of 2 videos per step, 3 timed + 1 warm-up + instrumented steps), averaged over ALL dispatches of a "
               "kernel = its real launch mix in the bench (per step of 2 videos: 2 formation launches over 6272 keys + 1 evolution "
               "launch over 12544 keys for the plain variant, 2 formation launches for the frame-score variant); KB per dispatch; "
               "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 reports half of a 16-B/lane coalesced stream, MI355X_MICROARCH.md HBM), "
               "write bytes = WRITE_SIZE x 1024; FETCH_SIZE counts L2 misses, Infinity-Cache hits included; the fp32 partials of the "
               "cut units are part of the write traffic, the merge kernel is listed separately",
       "kernels": {}}
for k in sorted({d["kernel"] for d in fetch if d["kernel"].startswith("attn_")}):
    fv = [d["FETCH_SIZE"] for d in fetch if d["kernel"] == k and "FETCH_SIZE" in d]
    wv = [d["WRITE_SIZE"] for d in write if d["kernel"] == k and "WRITE_SIZE" in d]
    if not fv or not wv:
        continue
    out["kernels"][k] = {"dispatches": len(fv), "FETCH_SIZE_KB": round(sum(fv) / len(fv), 1), "WRITE_SIZE_KB": round(sum(wv) / len(wv), 1),
                         "read_bytes_corrected": int(2 * 1024 * sum(fv) / len(fv)), "write_bytes": int(1024 * sum(wv) / len(wv))}
dom = [k for k in out["kernels"] if k.startswith("attn_fwd3_kernel") and k.rstrip(">").endswith(", 0")]
if dom:
    v = out["kernels"][dom[0]]
    out["bench_avg_bytes_per_launch"] = v["read_bytes_corrected"] + v["write_bytes"]
    out["launch_mix"] = f"{dom[0]}: {v['dispatches']} dispatches of the bench run, 2 of 3 over 6272 keys, 1 of 3 over 12544 keys (per video pair)"
json.dump(out, open(os.path.join(dst, f"{tag}_attn_fwd_hbm_traffic.json"), "w"), indent=1)
print(sorted(x for x in os.listdir(dst) if x.startswith(tag)))
