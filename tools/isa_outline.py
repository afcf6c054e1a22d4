"""Diagnostic (build container): compile one csrc/*.hip to gfx950 assembly and print an outline of one kernel's instruction
stream (M = MFMA, r = ds_read, w = ds_write, D = LDS-DMA / buffer load, G = global, v / s = other vector / scalar, waits and
barriers spelled out), plus its register / spill / scratch figures.
usage: python tools/isa_outline.py gemm128.hip 'gemm128_kernelI4BF16Li0E' [-DFLAG ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
flags = sys.argv[3:]
out = "/tmp/isa_" + os.path.splitext(os.path.basename(src))[0] + ".s"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", out,
       os.path.join(ROOT, "memory-augmented-vlm_amd", "csrc", src)] + flags
subprocess.run(cmd, check=True)
s = open(out).read()
for m in re.finditer(r"^(_Z\S*" + re.escape(pat) + r"\S*):[^\n]*\n(.*?)\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    seq = []
    for line in body.split("\n"):
        t = line.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith("."):
            if t.startswith(".LBB"):
                seq.append("\n" + t)
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"): k = "M"
        elif op.startswith("ds_read") or op.startswith("ds_load"): k = "r"
        elif op.startswith("ds_write") or op.startswith("ds_store"): k = "w"
        elif op.startswith("buffer_load"): k = "D"
        elif op.startswith("buffer_store") or op.startswith("global_"): k = "G"
        elif op.startswith("scratch_"): k = "!SCRATCH!"
        elif op == "s_barrier": k = "|BAR|"
        elif op == "s_waitcnt": k = "[" + t.split(None, 1)[1] + "]"
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): k = "<" + " ".join(t.split()[:2]) + ">"
        elif op.startswith("s_setprio"): k = "p"
        elif op.startswith("v_accvgpr"): k = "a"
        elif op.startswith("v_"): k = "v"
        elif op.startswith("s_"): k = "s"
        else: k = "?" + op
        seq.append(k)
    res, prev, cnt = [], None, 0
    for k in seq:
        if k == prev and len(k) == 1:
            cnt += 1
        else:
            if prev is not None:
                res.append(prev + (str(cnt) if cnt > 1 else ""))
            prev, cnt = k, 1
    res.append(prev + (str(cnt) if cnt > 1 else ""))
    print("==", name)
    print(" ".join(res))
    meta = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\n\s+- \.", s + "\n  - .", re.S)
    for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size"):
        mm = re.search(r"\.amdhsa_kernel " + re.escape(name) + r".*?\.end_amdhsa_kernel", s, re.S)
    md = re.search(r"- \.agpr_count:.*?\.name:\s+" + re.escape(name) + r".*?\.wavefront_size", s, re.S)
    if md:
        for key in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
            mm = re.search(r"\." + key + r":\s+(\d+)", md.group(0))
            if mm:
                print(f"   .{key} {mm.group(1)}", end="")
        print()
