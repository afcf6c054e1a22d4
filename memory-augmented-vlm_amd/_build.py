"""In-tree build of libmavlm.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "gemm256.hip", "gemm256p.hip", "gemm128.hip", "attention.hip", "attention3.hip", "attention_hd.hip", "attention_bwd.hip", "attention_bwd_hd.hip", "backward.hip", "variants.hip", "elementwise.hip", "mavlm_api.hip", "prof.hip"]
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


def _compile_one(hipcc, src, obj, verbose):
    """One translation unit -> object file, skipped when the object was built from the same bytes (source, headers,
    flags) - so editing one kernel file recompiles one file."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in [src] + [os.path.join(_HERE, "csrc", x) for x in HEADERS]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    stamp = obj + ".srchash"
    try:
        if os.path.exists(obj) and open(stamp).read().strip() == h.hexdigest():
            return None
    except OSError:
        pass
    cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + ["-c", "-o", obj, src]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        return "hipcc failed on " + src + ":\n" + r.stdout + r.stderr
    with open(stamp, "w") as fh:
        fh.write(h.hexdigest())
    return None


def build_library(force: bool = False, verbose: bool = False) -> str:
    out = library_path()
    if not force and not _stale(out):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmavlm.so (ROCm toolchain required)")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objdir = os.path.join(os.path.dirname(out), "obj")
    os.makedirs(objdir, exist_ok=True)
    # several ranks of one job may import the package at once: one builds, the others wait and then find it fresh
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    with open(os.path.join(os.path.dirname(out), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(out):
                return out
            srcs = [os.path.join(_HERE, "csrc", s) for s in SOURCES]
            objs = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in SOURCES]
            if force:
                for o in objs:
                    if os.path.exists(o + ".srchash"):
                        os.remove(o + ".srchash")
            workers = max(1, min(6, (os.cpu_count() or 2) - 1))
            with ThreadPoolExecutor(workers) as ex:
                errs = [e for e in ex.map(lambda so: _compile_one(hipcc, so[0], so[1], verbose), zip(srcs, objs)) if e]
            if errs:
                raise RuntimeError("\n".join(errs))
            tmp = f"{out}.tmp.{os.getpid()}"
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, out)
            with open(out + ".srchash", "w") as fh:
                fh.write(_src_hash())
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out
