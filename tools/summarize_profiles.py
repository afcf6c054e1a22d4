"""Turns gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the small summaries committed under profiles/:
  <tag>_kernel_stats_1video_in_flight.csv / <tag>_kernel_stats.csv   name, calls, total us, average us, % (rocprofv3 --stats)
  <tag>_bench_under_rocprof*.json                                     the bench line printed by the same command
  <tag>_mfma_utilisation.csv                                          matrix-pipe busy share per kernel (PMC)
  <tag>_attn_fwd_hbm_traffic.json                                     FETCH_SIZE x2 (gfx950) + WRITE_SIZE per launch (PMC)
usage: python tools/summarize_profiles.py <tag>"""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*", "", name).replace("void ", "")


def newest(files):
    """gpurun merges every call's files into gpurun_out/: keep the most recent run of a sub-directory only"""
    return sorted(files, key=os.path.getmtime)[-1:]


def stats(sub, out):
    files = newest(glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True))
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


def counters(sub):
    files = newest(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in files:
        for r in csv.DictReader(open(fn)):
            agg[(short(r["Kernel_Name"]), r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


stats("stats1", f"{tag}_kernel_stats_1video_in_flight.csv")
stats("stats2", f"{tag}_kernel_stats.csv")

def per_dispatch(sub):
    """[{kernel, grid, counter: value...}] one record per dispatch"""
    files = newest(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True))
    rec = collections.defaultdict(dict)
    for fn in files:
        for r in csv.DictReader(open(fn)):
            d = rec[(fn, r["Dispatch_Id"])]
            d["kernel"], d["grid"] = short(r["Kernel_Name"]), r.get("Grid_Size", "")
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(rec.values())


with open(os.path.join(dst, f"{tag}_mfma_utilisation.csv"), "w") as f:
    f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES (tools/profile_round.sh), per dispatch averages at the bench shapes\n")
    f.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) / (GRBM_GUI_ACTIVE / 8 XCDs); profiled clocks are lower than un-profiled ones\n")
    f.write("# dispatches of one kernel are grouped by their MFMA-busy count (it is a function of the shape), first dispatch of a group dropped (cold)\n")
    f.write("kernel,grid_threads,dispatches,mfma_busy_cycles_sum,kernel_cycles,mfma_util\n")
    for op in ("attn", "colsum", "gemm"):
        groups = collections.defaultdict(list)
        for d in per_dispatch(f"mfma_{op}"):
            if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and d.get("GRBM_GUI_ACTIVE", 0) > 0:
                groups[(d["kernel"], d["grid"], round(d["SQ_VALU_MFMA_BUSY_CYCLES"], -5))].append(d)
        for (k, grid, _), ds in sorted(groups.items()):
            ds = ds[1:] if len(ds) > 2 else ds
            busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in ds) / len(ds)
            cyc = sum(d["GRBM_GUI_ACTIVE"] for d in ds) / len(ds) / 8.0
            f.write(f"\"{k}\",{grid},{len(ds)},{busy:.0f},{cyc:.0f},{busy / 1024.0 / cyc:.3f}\n")

def ordered(sub, counter):
    """{(kernel, grid): [values in dispatch order]} - tools/bench_ops.py attn runs all its S=6272 launches, then all its
    S=12544 ones (the same number of each), and both shapes can share one grid: first half / second half."""
    files = newest(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True))
    agg = collections.defaultdict(list)
    for fn in files:
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == counter:
                agg[(short(r["Kernel_Name"]), r.get("Grid_Size", ""))].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in agg.items()}


fetch, write = ordered("fetch", "FETCH_SIZE"), ordered("write", "WRITE_SIZE")
shapes = {}
for (k, grid), f_ in sorted(fetch.items()):
    if "attn_fwd3_kernel" not in k:
        continue
    w_ = write.get((k, grid), [0.0] * len(f_))
    two = len(f_) % 2 == 0 and len(f_) >= 2 and abs(sum(f_[len(f_) // 2:]) / max(sum(f_[:len(f_) // 2]), 1.0) - 1.0) > 0.2
    parts = ((("S=6272", slice(0, len(f_) // 2)), ("S=12544", slice(len(f_) // 2, None))) if two else (("", slice(None)),))
    for label, sl in parts:
        fv, wv = f_[sl], (w_[sl] if len(w_) == len(f_) else w_)
        shapes[f"{k} grid {grid} {label}".strip()] = {
            "dispatches": len(fv), "FETCH_SIZE_KB": round(sum(fv) / len(fv), 1), "WRITE_SIZE_KB": round(sum(wv) / len(wv), 1),
            "read_bytes_corrected": int(2 * 1024 * sum(fv) / len(fv)), "write_bytes": int(1024 * sum(wv) / len(wv))}
if shapes:
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_round.sh) on tools/bench_ops.py attn "
                   "(R=12544, H=8, S=6272 and S=12544; both run the levelled stream-K schedule: 256 workgroups of 8 waves - the "
                   "fp32 partials of the cut units are part of the write traffic, the merge kernel is not counted), KB per "
                   "dispatch; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 reports half of a 16-B/lane coalesced stream, "
                   "MI355X_MICROARCH.md HBM), write bytes = WRITE_SIZE x 1024; FETCH_SIZE counts L2 misses, Infinity-Cache hits "
                   "included; algorithmic bytes: Q + K + V read once = 51.4 MB (S=6272) / 77.1 MB (S=12544), O written once = 25.7 MB",
           "kernels": shapes}
    tot = [(v["read_bytes_corrected"] + v["write_bytes"], v) for v in shapes.values()]
    if len(tot) >= 2:
        lo, hi = min(t[0] for t in tot), max(t[0] for t in tot)
        out["bench_avg_bytes_per_launch"] = int((4 * lo + hi) / 5)      # per video: 4 launches at S=6272, 1 at S=12544
    elif tot:
        out["bench_avg_bytes_per_launch"] = tot[0][0]
    json.dump(out, open(os.path.join(dst, f"{tag}_attn_fwd_hbm_traffic.json"), "w"), indent=1)
print(sorted(os.listdir(dst)))
