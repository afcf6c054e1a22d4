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


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def _src_hash() -> str:
    """Content hash of everything that goes into the library (sources, headers, flags).  Staleness is decided on
    content, not on mtimes: a snapshot copy of the tree (the GPU box) does not preserve them."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        p = os.path.join(_HERE, "csrc", f)
        h.update(f.encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale(out: str) -> bool:
    if not os.path.exists(out):
        return True
    try:
        with open(out + ".srchash") as fh:
            return fh.read().strip() != _src_hash()
    except OSError:
        return True


def build_library(force: bool = False, verbose: bool = False) -> str:
    out = library_path()
    if not force and not _stale(out):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmavlm.so (ROCm toolchain required)")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # several ranks of one job may import the package at once: one builds, the others wait and then find it fresh
    import fcntl
    with open(os.path.join(os.path.dirname(out), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(out):
                return out
            tmp = f"{out}.tmp.{os.getpid()}"
            cmd = [hipcc] + FLAGS + ["-o", tmp] + [os.path.join(_HERE, "csrc", s) for s in SOURCES]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, out)
            with open(out + ".srchash", "w") as fh:
                fh.write(_src_hash())
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out
