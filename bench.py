#!/usr/bin/env python3
"""Headline benchmark: Memory-Fuser frames/sec (196 vis-tokens x 1024d, 64 memory tokens).

One "step" = one pass of the hot path over synthetic 64-frame videos resident in HBM
(BASELINE.json configs[1]: 2 recurrent chunks of 32 frames, 64 memory tokens, D=1024, H=8, bf16):
  prompt embedding lookup -> temporal PE add -> chunk 0 (formation) -> chunk 1 (evolution over the FIFO +
  formation) incl. frame scores -> Memory-Fuser MLP + token-type add + fine frames + concat into one token block.
Vision tower and LLM are outside the path.

    python bench.py --gpus N --steps K --warmup W [--mode replica|shard-video]

Multi-GPU: one process per GPU over RCCL (`nccl`).  Either the driver launches the ranks (`python -m
torch.distributed.run ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE are read from the environment) or -
when WORLD_SIZE is unset - this script starts its own N rank processes BEFORE anything touches a GPU (the parent
only waits; the reference launches with torchrun the same way, scripts/train/finetune_short.sh:49-53).  It fails
loudly (non-zero exit) when WORLD_SIZE != --gpus or fewer than N GPUs are visible; `--rehearse-gloo` is the one
explicit exception (N <= 6 ranks share the visible GPUs over gloo: a functional rehearsal, not a measurement).

  --mode replica      (headline) videos are the independent unit (the recurrence couples the chunks of one video), so
                      every rank runs its own videos per step (weak scaling) and the ranks all-gather their final
                      memory state [M,P,D] over RCCL, asynchronously, overlapped with the next video.
  --mode shard-video  BASELINE.json configs[3]: ONE long video, its memory ROWS sharded over the ranks
                      (distributed.RowShardedMemory), per-step all-gather of the new memory (strong scaling).

Timing (contract in the task description): W warm-up steps, then blocks of EXACTLY K steps, each bracketed by a
barrier + torch.cuda.synchronize() on both sides, MAX over ranks per block.  The block is repeated (>= 5 times and
until >= ~2 s of timed GPU work) and `value` / `ms_per_step` are those of the MEDIAN block; min / max and the shader
clock sampled during the blocks are reported beside it.  Rank 0 prints ONE JSON line with `roofline` (dominant kernel,
measured live with HIP events in a second, instrumented pass of the same K steps) and `cpu_baseline` (N=1 only).
"""
import argparse
import ctypes
import glob
import json
import os
import socket
import subprocess
import sys
import threading
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0

FRAMES, MEM_TOKENS, HIDDEN, HEADS, PATCHES, DEPTH = 64, 64, 1024, 8, 196, 2
QWEN2_VOCAB = 151936
MAX_RANKS_PER_CARD = 6      # the pool's process guard (rehearsals only)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=("replica", "shard-video"), default="replica")
    ap.add_argument("--repeats", type=int, default=5, help="minimum number of timed K-step blocks")
    ap.add_argument("--min-seconds", type=float, default=8.0,
                    help="keep repeating blocks until this much timed work (>= 8 s: a utilisation sampler sees the GPU busy)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="functional rehearsal: ranks may share GPUs and talk over gloo (never a measurement)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame-scores", action="store_true", help="skip the column-sum pass (not the headline)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather of the final memory state")
    ap.add_argument("--videos-in-flight", type=int, default=2,
                    help="HIP streams per GPU, each stepping its own video(s) (fills partial-wave tails)")
    ap.add_argument("--batch", type=int, default=2,
                    help="videos stepped TOGETHER per stream as a row batch (stacked memory rows in every weight-shared GEMM)")
    ap.add_argument("--shard-frames", type=int, default=1024, help="shard-video: frames of the one long video")
    ap.add_argument("--shard-hidden", type=int, default=3584, help="shard-video: hidden width (3584 = OV-7B)")
    ap.add_argument("--shard-mem-tokens", type=int, default=8)
    ap.add_argument("--m8-extra", action="store_true", help="also time the reference-default M=8 shape (extras)")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the single-video latency leg (profiling runs: keeps other launch shapes out of the kernel statistics)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------------------
# launcher: runs in a process that never initialises the GPU
# --------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`--gpus N` without a launcher: start N rank processes of this script (children, fresh interpreters), wait, and
    exit with the first non-zero code.  Nothing here touches the GPU (`torch.cuda.device_count()` does not)."""
    import torch
    ndev = torch.cuda.device_count()
    if args.rehearse_gloo:
        if ndev < 1:
            sys.exit("bench.py --rehearse-gloo: no GPU visible")
        if args.gpus > MAX_RANKS_PER_CARD * ndev:
            sys.exit(f"bench.py --rehearse-gloo: at most {MAX_RANKS_PER_CARD} ranks may share one GPU")
    elif ndev < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible - refusing to run fewer ranks than asked "
                 "(use --rehearse-gloo for a functional rehearsal on shared GPUs)")
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MAVLM_BENCH_LAUNCHER="self", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if args.rehearse_gloo:
            env["MAVLM_DIST_BACKEND"] = "gloo"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:          # one rank failed: the others would hang in a collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


# --------------------------------------------------------------------------------------------------------------
# helpers that run inside a rank
# --------------------------------------------------------------------------------------------------------------
def algorithmic_flops(M=MEM_TOKENS, frames=FRAMES, D=HIDDEN, L=DEPTH, cap=10):
    """SURVEY.md §8(d) formulas for one video (K/V of cached memories projected once)."""
    R, S = M * PATCHES, 32 * PATCHES
    chunks = frames // 32
    total = 0.0
    for t in range(chunks):
        total += L * (20.0 * R * D * D + 4.0 * S * D * D + 4.0 * R * S * D)
        if t > 0:
            n = min(t, cap)
            total += 4.0 * R * D * D + 4.0 * R * D * D + 4.0 * R * (n * R) * D
    total += 16.0 * (min(chunks, cap) * R) * D * D
    return total


def build_model(device, hidden=HIDDEN, mem_tokens=MEM_TOKENS, max_frames=600, seed=1234):
    import torch
    import memory_augmented_vlm_amd  # noqa: F401
    from memory_augmented_vlm_amd.model import llava_arch as arch

    class Base(torch.nn.Module):
        def __init__(self, config):
            super().__init__()
            self.embed_tokens = torch.nn.Embedding(QWEN2_VOCAB, config.hidden_size)

    class Model(arch.LlavaMetaModel, Base):
        pass

    hf = types.SimpleNamespace(hidden_size=hidden, num_memory_tokens=mem_tokens, memory_max_frames=max_frames)
    torch.manual_seed(seed)
    model = Model(hf).eval()
    model.image_newline = torch.nn.Parameter(torch.randn(hidden) * 0.02)
    with torch.no_grad():   # LayerNorm affine away from identity so the epilogue work is real
        for m in model.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(torch.rand_like(m.weight) * 0.2 - 0.1)
                m.bias.add_(torch.rand_like(m.bias) * 0.2 - 0.1)
    return model.to(device).to(torch.bfloat16), arch


class ClockSampler:
    """Shader clock of the card under load: the current level of pp_dpm_sclk, sampled while the timed blocks run."""

    def __init__(self, device_index):
        self.path = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
                if want in os.path.realpath(os.path.dirname(p)):
                    self.path = p
            if self.path is None:
                cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
                self.path = cands[0] if len(cands) == 1 else None
        except Exception:
            self.path = None
        self.samples, self._stop, self._t = [], threading.Event(), None

    def _read(self):
        try:
            for line in open(self.path):
                if line.rstrip().endswith("*"):
                    return int(line.split(":")[1].strip().split("M")[0])
        except Exception:
            return None

    def __enter__(self):
        if self.path:
            def loop():
                while not self._stop.is_set():
                    v = self._read()
                    if v:
                        self.samples.append(v)
                    time.sleep(0.02)
            self._t = threading.Thread(target=loop, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._t:
            self._t.join(1.0)

    def summary(self):
        if not self.samples:
            return None
        s = sorted(self.samples)
        return {"sclk_mhz_median": s[len(s) // 2], "sclk_mhz_min": s[0], "sclk_mhz_max": s[-1], "samples": len(s),
                "source": "pp_dpm_sclk (DPM level, reads up to ~10 % above the in-kernel clock: MI355X_MICROARCH.md DVFS item 6)"}


def timed_blocks(step, sync, args, world, device):
    """W warm-up steps, then >= args.repeats blocks of EXACTLY K steps (barrier + synchronize on both sides of each
    block, MAX over ranks).  Returns the sorted list of block times in seconds."""
    import torch
    import torch.distributed as dist

    def block():
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step()
    times = [block()]
    n = max(args.repeats, int(args.min_seconds / max(times[0], 1e-6)) + 1)
    n = min(n, 50)
    for _ in range(n - 1):
        times.append(block())
    return times


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands each job a
    share of its host: oversubscribing it with one thread per hardware thread makes the BLAS threads fight)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline():
    """the reference's algorithm on the host cores, C1 shape = ONE recurrent step through the memory
    transformer (formation, depth 2, no evolution at step 0), 8 frames x 196 tokens x 1024, 64 memory tokens, H = 8.
    What is timed is the oracle's torch restatement (oracle/torch_path.py: the same ATen call sequence the reference
    executes, pinned against the imported reference in tests/) - the reference's own Python never travels to this box.
    fp32 and bf16, all usable host cores and one thread, 2 warm-ups then the median of 5 (single-thread legs: fewer
    runs - a bounded sample of ~20-30 s in total, stated per leg)."""
    import numpy as np
    import torch
    from oracle import memory_path as O
    from oracle import torch_path as TP
    F = 8
    cfg = O.PathConfig(hidden=HIDDEN, heads=HEADS, mem_tokens=MEM_TOKENS, depth=DEPTH)
    w = O.make_weights(cfg, seed=77)
    seg = O.bf16_round(O.hash_normal_like((F, PATCHES, HIDDEN), 78))
    flop = DEPTH * (20.0 * cfg.mem_rows * HIDDEN ** 2 + 4.0 * F * PATCHES * HIDDEN ** 2 + 4.0 * cfg.mem_rows * F * PATCHES * HIDDEN)
    cores = usable_cpus()
    legs = {}
    t_start = time.perf_counter()

    def leg(name, dtype, threads, warm, reps):
        torch.set_num_threads(threads)
        run = TP.cpu_reference_step_timer(cfg, w, [seg], dtype)
        ts = []
        for i in range(warm + reps):
            t0 = time.perf_counter()
            out = run()
            dt = time.perf_counter() - t0
            if i >= warm:
                ts.append(dt)
        assert bool(torch.isfinite(out.float()).all())
        ts.sort()
        med = ts[len(ts) // 2]
        legs[name] = {"frames_per_s": round(F / med, 2), "s_per_step_median": round(med, 4), "s_min": round(ts[0], 4),
                      "s_max": round(ts[-1], 4), "gflops": round(flop / med / 1e9, 1), "threads": threads,
                      "warmups": warm, "runs": reps}

    leg("fp32_all_cores", torch.float32, cores, 2, 5)
    leg("bf16_all_cores", torch.bfloat16, cores, 2, 5)
    leg("fp32_1_thread", torch.float32, 1, 0, 1)          # ~7 s per run on a 5 GHz core: one cold run
    leg("bf16_1_thread", torch.bfloat16, 1, 1, 2)
    torch.set_num_threads(cores)
    best = max(("fp32_all_cores", "bf16_all_cores"), key=lambda k: legs[k]["frames_per_s"])
    return {"value": legs[best]["frames_per_s"], "unit": "frames/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model_string(), "best_leg": best, "legs": legs,
            "host_threads_visible": len(os.sched_getaffinity(0)),
            "protocol": "BASELINE.md §3: 2 warm-ups + median of 5 on all usable cores (affinity capped by the cgroup CPU "
                        "quota); single-thread legs bounded to 1 cold run (fp32) / 1 warm-up + 2 runs (bf16)",
            "sample": f"C1 = one recurrent step (formation, depth {DEPTH}): {F} frames x {PATCHES} tokens x {HIDDEN}, "
                      f"M={MEM_TOKENS}, H={HEADS}; {flop / 1e9:.1f} GFLOP per step; torch {torch.__version__} CPU ATen "
                      f"path of oracle/torch_path.py; whole baseline took {time.perf_counter() - t_start:.1f} s"}


def kernel_table(lib, capi, steps):
    nk = len(capi.KERNEL_KINDS)
    ms = (ctypes.c_double * nk)()
    ln = (ctypes.c_int64 * nk)()
    fl = (ctypes.c_double * nk)()
    by = (ctypes.c_double * nk)()
    capi.check(lib.mavlm_prof_read(ms, ln, fl, by, nk), "mavlm_prof_read")
    kernels = {}
    for i, name in enumerate(capi.KERNEL_KINDS):
        if ln[i]:
            kernels[name] = {"launches_per_step": ln[i] / steps, "avg_ms": round(ms[i] / ln[i], 5),
                             "ms_per_step": round(ms[i] / steps, 4),
                             "tflops": round(fl[i] / (ms[i] * 1e-3) / 1e12, 1) if fl[i] else None,
                             "alg_gbs": round(by[i] / (ms[i] * 1e-3) / 1e9, 1)}
    return kernels, ms, ln, fl, by


# --------------------------------------------------------------------------------------------------------------
# modes
# --------------------------------------------------------------------------------------------------------------
def run_replica(args, rank, world, local, device, dist_info):
    import torch
    import torch.distributed as dist
    from memory_augmented_vlm_amd import distributed as D
    from memory_augmented_vlm_amd import _capi as capi
    model, arch = build_model(device)
    rm = model.recurrent_memory_transformer
    rm.compute_frame_scores = not args.no_frame_scores
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    NS, NB = max(1, args.videos_in_flight), max(1, args.batch)      # streams, videos per row batch
    B = NS * NB                                                      # videos per step per GPU
    xs = [torch.randn((FRAMES, PATCHES, HIDDEN), generator=g).to(device).to(torch.bfloat16) for _ in range(B)]  # in HBM
    x = xs[0]
    idx_cpu = torch.arange(FRAMES)
    pool = arch.MemoryPathPool(model, NS, batch=NB)
    pool1 = arch.MemoryPathPool(model, 1, batch=NB) if NS > 1 else pool     # instrumented pass: ONE stream
    for pl in (pool, pool1):
        for slot in pl.slots:
            slot.recurrent_memory_transformer.compute_frame_scores = not args.no_frame_scores
        for bs in pl.bslots:
            bs.compute_frame_scores = not args.no_frame_scores
    mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=device)
    frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=device)
    gathered = torch.empty((world, MEM_TOKENS, PATCHES, HIDDEN), device=device, dtype=torch.bfloat16) if world > 1 else None
    # the gather runs asynchronously while the next video overwrites the FIFO ring -> gather from a private copy
    send = torch.empty((MEM_TOKENS, PATCHES, HIDDEN), device=device, dtype=torch.bfloat16) if world > 1 else None
    pending = [None]
    do_gather = world > 1 and not args.no_gather
    last = [None]

    def final_memory():
        """newest memory of this rank's first video (what the north-star's all-gather ships)"""
        if NB > 1:
            return pool.bslots[0].memory_cache(0)[-1]
        return rm.memory_cache[-1]

    def step(single=False, one_video=False):
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        if one_video:
            toks, _ = arch.video_memory_tokens(model, x, idx_cpu, mp, fp, model.image_newline)
        elif single:
            toks = pool1.run([(xi, idx_cpu) for xi in xs[:NB]], mp, fp, model.image_newline)[0]
        else:
            toks = pool.run([(xi, idx_cpu) for xi in xs], mp, fp, model.image_newline)[0]
        if do_gather and not single and not one_video:
            if pending[0] is not None:
                pending[0].wait()                       # previous gather (overlapped with this step) is done with `send`
            send.copy_(final_memory())                  # final memory state of this rank's (first) video
            _, pending[0] = D.all_gather_memory_state(send, out=gathered, async_op=True)
        last[0] = toks
        return toks

    def sync():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    with torch.no_grad(), ClockSampler(device.index) as clk:
        times = timed_blocks(step, sync, args, world, device)
    toks = last[0]
    rows_expected = 10 + 2 * MEM_TOKENS * PATCHES + 1 + 9 + 32 * PATCHES + 1
    assert toks.shape == (rows_expected, HIDDEN) and bool(torch.isfinite(toks.float()).all())
    if do_gather:       # every rank's final memory arrived: slot r of the gather equals what rank r sent
        mine = gathered[rank].float()
        assert bool(torch.isfinite(gathered.float()).all()) and torch.equal(mine, send.float())

    # ---- instrumented pass: the same K steps with a HIP-event pair around every kernel launch, ONE stream (one row batch of
    # NB videos in flight: kernel durations are not smeared by the other stream)
    lib = capi.lib()
    with torch.no_grad():
        lib.mavlm_prof_enable(1)
        for _ in range(args.steps):
            step(single=True)
        sync()
        kernels, ms, ln, fl, by = kernel_table(lib, capi, args.steps)
        lib.mavlm_prof_enable(0)
        # single-video latency (one video, one stream, no row batch): what a request waits for.  Its engine is created
        # here (the pools above run row batches), and the card has just idled through the read-back of the event table:
        # warm up for ~0.1 s, then the best of 3 blocks of 10
        single_ms = single_graph_ms = None
        if not args.no_latency:
            single_ms, single_graph_ms = single_video_latency(model, arch, x, idx_cpu, device)
    # the fused dense + residual + LayerNorm epilogue exchanges row statistics between workgroups with a bounded spin: a
    # timeout (never seen) would mean a wrong result - fail loudly rather than print a number
    ln_status = {}
    engines = [bs._engine for bs in pool.bslots + pool1.bslots if bs._engine is not None] + \
              [sl.recurrent_memory_transformer._engine for sl in pool.slots if sl.recurrent_memory_transformer._engine is not None]
    for e in engines:
        st = e.ln_exchange_status()
        if st is not None:
            ln_status["launches"] = ln_status.get("launches", 0) + st[0]
            ln_status["timeouts"] = ln_status.get("timeouts", 0) + st[1]
    assert ln_status.get("timeouts", 0) == 0, ln_status
    dom = "attention_fwd"
    di = capi.KERNEL_KINDS.index(dom)
    info = (ctypes.c_int32 * 4)()       # which instantiation the plan picks at the formation shape (S = one chunk's keys)
    capi.check(lib.mavlm_attention_plan(MEM_TOKENS * PATCHES, 32 * PATCHES, HEADS * NB, info), "mavlm_attention_plan")
    kname = f"attn_fwd3_kernel<BF16, {info[0]}, 0>"
    mi = capi.KERNEL_KINDS.index("attention_merge")
    fi = capi.KERNEL_KINDS.index("attention_fwd_frames")
    knote = ("HIP-event bracket around this kernel only (all its launches of a step: formation and evolution shapes; the last "
             f"formation layer of a step runs the frame-score variant attn_fwd3_kernel<BF16, {info[0]}, 1> on the SAME schedule - "
             f"{ms[fi] / max(ln[fi], 1) * 1e3:.1f} us per launch, kernels.attention_fwd_frames - which replaces the "
             f"column-sum pass); a launch serves the {NB} video(s) of a row batch ({HEADS * NB} (video, head) pairs); "
             + (f"schedule: levelled stream-K, {info[1]} workgroups of {info[0]} waves, {info[2]} level(s); its merge kernel "
                f"attn_combine_sk_kernel<BF16> is bracketed separately: {ms[mi] / max(ln[mi], 1) * 1e3:.1f} us per launch "
                "(kernels.attention_merge)" if info[1] else "plain grid"))
    achieved = fl[di] / (ms[di] * 1e-3) / 1e12
    # every attention-forward launch of a step, weighted by time (plain + frame-score variant + merges)
    att_ms = ms[di] + ms[fi] + ms[mi]
    att_weighted = (fl[di] + fl[fi]) / (att_ms * 1e-3) / 1e12 if att_ms > 0 else 0.0
    traffic, tnote = None, None
    for tp in ("r04_attn_fwd_hbm_traffic.json", "r03_attn_fwd_hbm_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", tp)
        if os.path.exists(tpath):      # PMC summary committed from a separate rocprofv3 --pmc run (tools/pmc_traffic.sh)
            tj = json.load(open(tpath))
            traffic = tj.get("bench_avg_bytes_per_launch")
            tnote = (f"NOT measured in this run: constant from profiles/{tp} (separate rocprofv3 --pmc passes over the same "
                     "command, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, averaged over this kernel's launches of a step = "
                     f"its real launch mix: {tj.get('launch_mix', 'see file')}); algorithmic bytes per launch in alg_bytes_per_launch")
            break
    roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_note": tnote,
                "kernel": kname, "kernel_note": knote,
                "avg_launch_ms": round(ms[di] / ln[di], 5), "launches_per_step": ln[di] / args.steps,
                "alg_flops_per_launch": fl[di] / ln[di], "alg_bytes_per_launch": by[di] / ln[di],
                "hbm_gbs_algorithmic": round(by[di] / (ms[di] * 1e-3) / 1e9, 1),
                "attention_fwd_all_launches_tflops": round(att_weighted, 1),
                "attention_fwd_all_launches_frac": round(att_weighted / MFMA_PEAK_TFLOPS, 4),
                "clock_note": "frac is against the nominal 2.5 PFLOP/s (2.4 GHz); sclk of THIS box while the blocks ran: "
                              "timing.clock; the box profiles/r04_* were taken on: profiles/README.md"}

    extras = {}
    if args.m8_extra or (world == 1 and os.environ.get("MAVLM_BENCH_M8", "1") != "0"):
        extras["m8_checkpoint_shape"] = run_m8_extra(args, device, arch)
    if world == 1 and not args.no_latency and os.environ.get("MAVLM_BENCH_OV7B", "1") != "0":
        try:
            extras["ov7b_width_configs2"] = run_ov7b_extra(args, device)
        except Exception as e:                # an extra never costs the line (e.g. a box short of memory)
            extras["ov7b_width_configs2"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank != 0:
        return None
    times_sorted = sorted(times)
    med = times_sorted[len(times_sorted) // 2]
    ms_per_step = med / args.steps * 1e3
    value = world * args.steps * B * FRAMES / med
    flops = algorithmic_flops()
    out = {
        "metric": "Memory-Fuser frames/sec (196 vis-tokens x 1024d, 64 mem tokens)",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "configs[1]: 64-frame video, 2 recurrent chunks of 32 frames, 64 memory tokens, "
                               "196 tokens/frame, D=1024, H=8, depth 2, frame scores on; "
                               f"a step = {B} independent videos per GPU: {NS} HIP stream(s) x a row batch of {NB} (the memory "
                               "rows of a batch are stacked into every weight-shared GEMM / LayerNorm launch)",
                   "frames": FRAMES, "mem_tokens": MEM_TOKENS, "hidden": HIDDEN, "parallelism": f"replica x{world}",
                   "videos_per_step_per_gpu": B, "streams": NS, "row_batch": NB, "allgather_final_memory": bool(do_gather),
                   "frame_scores": not args.no_frame_scores, "mode": "replica"},
        "timing": {"blocks": len(times), "steps_per_block": args.steps, "statistic": "median block",
                   "ms_per_step_min": round(times_sorted[0] / args.steps * 1e3, 4),
                   "ms_per_step_max": round(times_sorted[-1] / args.steps * 1e3, 4),
                   "value_best_block": round(world * args.steps * B * FRAMES / times_sorted[0], 1),
                   "timed_seconds_total": round(sum(times), 3), "clock": clk.summary()},
        "ranks": dist_info,
        "roofline": roofline,
        "alg_tflop_per_video": round(flops / 1e12, 3),
        "path_mfma_frac": round(B * flops / (med / args.steps) / 1e12 / MFMA_PEAK_TFLOPS, 4),
        "single_video_latency_ms": round(single_ms, 3) if single_ms is not None else None,
        "single_video_latency_graph_ms": round(single_graph_ms, 3) if single_graph_ms is not None else None,
        "kernel_timing_note": f"per-kernel numbers from an instrumented pass with ONE stream (a row batch of {NB} video(s)) in flight",
        "kernels": kernels,
        "fused_layernorm_exchange": ln_status or None,
    }
    out.update(extras)
    return out


def single_video_latency(model, arch, x, idx_cpu, device):
    """(eager ms, hipGraph-replay ms) for ONE video on one stream, no row batch - what a request behind the reference's own
    entry point (batch 1, llava_arch.py:436) waits for.  Eager = `video_memory_tokens` (~45 launches); graph = the same launch
    sequence captured once (`GraphedVideoMemory` on the model's own engine: what `enable_memory_graphs()` replays for a
    repeated shape; the replay includes the copies into / out of the graph's static buffers).  Warm-up ~0.1 s, best of 3
    blocks of 10 each."""
    import torch
    mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=device)
    frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=device)
    out = None

    def prompts():
        return (torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight),
                torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight))

    def eager():
        mp, fp = prompts()
        return arch.video_memory_tokens(model, x, idx_cpu, mp, fp, model.image_newline)[0]

    def best(fn):
        for _ in range(30):
            fn()
        blocks = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / 10 * 1e3)
        return min(blocks)

    e_ms = best(eager)
    ref = eager().clone()
    g = arch.GraphedVideoMemory(model, x.shape[0], idx_cpu, slot=model)
    dst = torch.empty_like(ref)

    def graphed():
        mp, fp = prompts()
        dst.copy_(g(x, mp, fp, model.image_newline))
        return dst
    g_ms = best(graphed)
    assert torch.equal(graphed(), ref)                 # replay = eager launches, bit for bit
    del g
    return e_ms, g_ms


def run_m8_extra(args, device, arch):
    """Extras: the reference-default (checkpoint-compatible) shape M = 8, D = 1024, 64 frames, same path."""
    import torch
    model, _ = build_model(device, mem_tokens=8, seed=4321)
    NS, NB = max(1, args.videos_in_flight), 8          # 8 videos per row batch: 8 x 1568 = the 12 544 rows of the headline shape
    B = NS * NB
    g = torch.Generator(device="cpu").manual_seed(55)
    xs = [torch.randn((FRAMES, PATCHES, HIDDEN), generator=g).to(device).to(torch.bfloat16) for _ in range(B)]
    idx_cpu = torch.arange(FRAMES)
    pool = arch.MemoryPathPool(model, NS, batch=NB)
    mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=device)
    frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=device)

    def step():
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        return pool.run([(xi, idx_cpu) for xi in xs], mp, fp, model.image_newline)[0]

    with torch.no_grad():
        for _ in range(5):
            step()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    fl = algorithmic_flops(M=8)
    lat = (None, None)
    if not args.no_latency:
        with torch.no_grad():
            lat = single_video_latency(model, arch, xs[0], idx_cpu, device)
    return {"single_video_latency_ms": round(lat[0], 3) if lat[0] else None,
            "single_video_latency_graph_ms": round(lat[1], 3) if lat[1] else None,
            "frames_per_s": round(args.steps * B * FRAMES / med, 1), "ms_per_step": round(med / args.steps * 1e3, 4),
            "videos_per_step": B, "streams": NS, "row_batch": NB, "mem_tokens": 8, "alg_tflop_per_video": round(fl / 1e12, 4),
            "path_mfma_frac": round(B * fl / (med / args.steps) / 1e12 / MFMA_PEAK_TFLOPS, 4), "this_rank_only": True}


def run_ov7b_extra(args, device):
    """Extras: BASELINE.json configs[2] - the LLaVA-OneVision-7B width (hidden 3584, heads of 448), 8 memory tokens, 256-frame
    videos (8 recurrent chunks): one video on one stream (what the reference's own entry point gets) and 2 streams x a row batch
    of 4.  Bounded: 2 warm-up + 3 timed passes each."""
    import torch
    hidden, frames, M = 3584, 256, 8
    model, arch = build_model(device, hidden=hidden, mem_tokens=M, seed=2468)
    g = torch.Generator(device="cpu").manual_seed(56)
    x0 = torch.randn((frames, PATCHES, hidden), generator=g).to(device).to(torch.bfloat16)
    xs = [x0] + [x0.roll(b, 0) for b in range(1, 8)]
    idx_cpu = torch.arange(frames)
    mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=device)
    frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=device)
    fl = algorithmic_flops(M=M, frames=frames, D=hidden)
    out = {"hidden": hidden, "frames": frames, "mem_tokens": M, "alg_tflop_per_video": round(fl / 1e12, 3), "this_rank_only": True}
    for name, (ns, nb) in (("single_video", (1, 1)), ("streams2_batch4", (2, 4))):
        pool = arch.MemoryPathPool(model, ns, batch=nb)
        vids = xs[:ns * nb]

        def step():
            mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
            fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
            return pool.run([(xi, idx_cpu) for xi in vids], mp, fp, model.image_newline)[0]
        with torch.no_grad():
            for _ in range(2):
                step()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[1]
        out[name] = {"ms_per_step": round(med * 1e3, 3), "videos_per_step": ns * nb,
                     "frames_per_s": round(ns * nb * frames / med, 1),
                     "path_mfma_frac": round(ns * nb * fl / med / 1e12 / MFMA_PEAK_TFLOPS, 4)}
        del pool
    return out


def run_shard_video(args, rank, world, local, device, dist_info):
    """BASELINE.json configs[3]: one T-frame video, memory rows sharded over the ranks (RowShardedMemory)."""
    import torch
    import torch.distributed as dist
    from memory_augmented_vlm_amd import distributed as D
    T, Dh, M = args.shard_frames, args.shard_hidden, args.shard_mem_tokens
    model, arch = build_model(device, hidden=Dh, mem_tokens=M, max_frames=max(600, T))
    rm = model.recurrent_memory_transformer
    sh = D.RowShardedMemory(rm)
    g = torch.Generator(device="cpu").manual_seed(7)      # the SAME video on every rank (it is one video)
    x = torch.empty((T, PATCHES, Dh), device=device, dtype=torch.bfloat16)
    for i in range(0, T, 64):
        x[i:i + 64] = torch.randn((min(64, T - i), PATCHES, Dh), generator=g).to(device).to(torch.bfloat16)
    idx = arch._device_indices(torch.arange(T), device)
    last = [None]

    def step():
        xp = model.positional_encoding(x, idx, indices_checked=True)
        sh.reset()
        for i in range(0, T, 32):
            # the next chunk's K/V projection overlaps the all-gather of this step's memory rows
            cache, _ = sh.step(xp[i:i + 32], prefetch=xp[i + 32:i + 64] if i + 32 < T else None)
        last[0] = cache
        return cache

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    with torch.no_grad(), ClockSampler(device.index) as clk:
        times = timed_blocks(step, sync, args, world, device)
    cache = last[0]
    assert len(cache) == min(10, -(-T // 32)) and bool(torch.isfinite(torch.stack(list(cache)).float()).all())
    if world > 1:       # every rank ends with the identical FIFO
        chk = torch.stack([c.float().sum() for c in cache])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi)
    if rank != 0:
        return None
    ts = sorted(times)
    med = ts[len(ts) // 2]
    fl = algorithmic_flops(M=M, frames=T, D=Dh) - 16.0 * (min(T // 32, 10) * M * PATCHES) * Dh * Dh   # no fuser in this loop
    return {
        "metric": "Memory-Fuser frames/sec, ONE long video with memory rows sharded over the GPUs (configs[3])",
        "value": round(args.steps * T / med, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(med / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"configs[3]: one {T}-frame video, {T // 32} recurrent chunks, M={M}, D={Dh}, H=8; rank g owns "
                               f"{M // world} memory token(s); per step: all-gather of the new memory rows + all-reduce of "
                               "the frame scores", "frames": T, "mem_tokens": M, "hidden": Dh,
                   "parallelism": f"row-shard x{world}", "mode": "shard-video"},
        "timing": {"blocks": len(times), "steps_per_block": args.steps, "statistic": "median block",
                   "ms_per_step_min": round(ts[0] / args.steps * 1e3, 4), "ms_per_step_max": round(ts[-1] / args.steps * 1e3, 4),
                   "timed_seconds_total": round(sum(times), 3), "clock": clk.summary()},
        "ranks": dist_info,
        "roofline": {"bound": "mfma", "achieved": round(fl / (med / args.steps) / 1e12 / world, 1), "peak": MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(fl / (med / args.steps) / 1e12 / world / MFMA_PEAK_TFLOPS, 4),
                     "traffic": None, "note": "whole sharded step per GPU (algorithmic flop of the video / ranks / time); the "
                                              "chunk K/V projection is computed redundantly on every rank and not counted twice"},
    }


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        launch_ranks(args, argv)            # never returns
    world_env = int(env_world or "1")
    if world_env != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env} - the launcher and the flag disagree")

    import torch
    import torch.distributed as dist
    from memory_augmented_vlm_amd import distributed as D
    from memory_augmented_vlm_amd import _capi as capi
    ndev = torch.cuda.device_count()
    rehearsal = args.rehearse_gloo or os.environ.get("MAVLM_DIST_BACKEND") == "gloo"
    if ndev < 1:
        sys.exit("bench.py: no GPU visible (the HIP path has no CPU fallback)")
    if args.gpus > ndev and not rehearsal:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible")
    rank, world, local = D.init_from_env("gloo" if rehearsal and world_env > 1 else None)
    if world != args.gpus:
        sys.exit(f"bench.py: process group has {world} rank(s), --gpus {args.gpus}")
    device = torch.device("cuda", local % ndev)
    torch.cuda.set_device(device)
    capi.lib()   # fail loudly if the HIP library is missing
    if os.environ.get("MAVLM_GEMM_TILE"):      # A/B hook (diagnostics): 128 / 256 / 257, default automatic
        capi.check(capi.lib().mavlm_set_gemm_tile(int(os.environ["MAVLM_GEMM_TILE"])), "set_gemm_tile")
    if os.environ.get("MAVLM_ATTN_IMPL"):
        capi.check(capi.lib().mavlm_set_attention_impl(int(os.environ["MAVLM_ATTN_IMPL"])), "set_attention_impl")

    dist_info = {"ranks_seen": 1, "world_size": world, "backend": None, "launcher": os.environ.get("MAVLM_BENCH_LAUNCHER", "external" if world > 1 else "none"),
                 "devices_visible": ndev, "rehearsal_shared_gpus": bool(rehearsal and world > 1)}
    if world > 1:
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)                                   # every rank really is there
        devs = [None] * world
        dist.all_gather_object(devs, f"{socket.gethostname()}:{torch.cuda.current_device()}")
        dist_info.update(ranks_seen=int(ones.item()), backend=dist.get_backend(), rank_devices=devs)
        if int(ones.item()) != args.gpus:
            sys.exit(f"bench.py: {int(ones.item())} rank(s) answered the all-reduce, --gpus {args.gpus}")
        if not rehearsal and len(set(devs)) != world:
            sys.exit(f"bench.py: ranks share a GPU ({devs}) - not a measurement")

    if args.mode == "shard-video":
        out = run_shard_video(args, rank, world, local, device, dist_info)
    else:
        out = run_replica(args, rank, world, local, device, dist_info)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.mode == "replica":
            out["cpu_baseline"] = cpu_baseline()
            out["speedup_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
