"""In-tree build of libmavlm.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "gemm256.hip", "gemm256p.hip", "attention.hip", "attention3.hip", "attention_hd.hip", "attention_bwd.hip", "backward.hip", "variants.hip", "elementwise.hip", "mavlm_api.hip", "prof.hip"]
HEADERS = ["mavlm_common.h", "mavlm_kernels.h", os.path.join("..", "..", "include", "mavlm.h")]


def library_path() -> str:
    # MAVLM_LIB: diagnostics only (A/B builds of the same sources with different -D flags)
    return os.environ.get("MAVLM_LIB") or os.path.join(_HERE, "lib", "libmavlm.so")


def _stale(out: str) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    for f in SOURCES + HEADERS:
        p = os.path.join(_HERE, "csrc", f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_library(force: bool = False, verbose: bool = False) -> str:
    out = library_path()
    if not force and not _stale(out):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmavlm.so (ROCm toolchain required)")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", out + ".tmp"]
    cmd += [os.path.join(_HERE, "csrc", s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(out + ".tmp", out)
    return out
