#!/usr/bin/env python3
"""Headline benchmark: Memory-Fuser frames/sec (196 vis-tokens x 1024d, 64 memory tokens).

One "step" = one pass of the hot path over one synthetic 64-frame video resident in HBM
(BASELINE.json configs[1]: 2 recurrent chunks of 32 frames, 64 memory tokens, D=1024, H=8, bf16):
  prompt embedding lookup -> temporal PE add -> chunk 0 (formation) -> chunk 1 (evolution over the FIFO +
  formation) incl. frame scores -> Memory-Fuser MLP + token-type add + fine frames + concat into one token block.
Vision tower and LLM are outside the path.

Multi-GPU (--gpus N, launched by torch.distributed.run): videos are the independent unit (the recurrence couples
the chunks of one video), so every rank runs its own video per step (weak scaling) and the ranks all-gather their
final memory state [M,P,D] over RCCL, asynchronously, overlapped with the next video.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel, measured
live with HIP events in a second, instrumented pass of the same K steps) and `cpu_baseline` (the numpy oracle
timed on the host cores, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0

FRAMES, MEM_TOKENS, HIDDEN, HEADS, PATCHES, DEPTH = 64, 64, 1024, 8, 196, 2
QWEN2_VOCAB = 151936


def algorithmic_flops():
    """SURVEY.md §8(d) formulas for one 64-frame video (K/V of cached memories projected once)."""
    R, S, D, L = MEM_TOKENS * PATCHES, 32 * PATCHES, HIDDEN, DEPTH
    formation = L * (20.0 * R * D * D + 4.0 * S * D * D + 4.0 * R * S * D)
    evolution = 4.0 * R * D * D + 4.0 * R * D * D + 4.0 * R * R * D          # n = 1 cached memory
    fuser = 16.0 * (2 * R) * D * D
    return 2 * formation + evolution + fuser


def build_model(device):
    import memory_augmented_vlm_amd  # noqa: F401
    from memory_augmented_vlm_amd.model import llava_arch as arch

    class Base(torch.nn.Module):
        def __init__(self, config):
            super().__init__()
            self.embed_tokens = torch.nn.Embedding(QWEN2_VOCAB, config.hidden_size)

    class Model(arch.LlavaMetaModel, Base):
        pass

    hf = types.SimpleNamespace(hidden_size=HIDDEN, num_memory_tokens=MEM_TOKENS)
    torch.manual_seed(1234)
    model = Model(hf).eval()
    model.image_newline = torch.nn.Parameter(torch.randn(HIDDEN) * 0.02)
    with torch.no_grad():   # LayerNorm affine away from identity so the epilogue work is real
        for m in model.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.add_(torch.rand_like(m.weight) * 0.2 - 0.1)
                m.bias.add_(torch.rand_like(m.bias) * 0.2 - 0.1)
    return model.to(device).to(torch.bfloat16), arch


def cpu_baseline():
    """The CPU oracle (a port of the reference algorithm, validated against the reference in tests/) on one
    64-frame video of the same shape, fp32, all host cores through numpy's BLAS."""
    import numpy as np
    from oracle import memory_path as O
    cfg = O.PathConfig(hidden=HIDDEN, heads=HEADS, mem_tokens=MEM_TOKENS, depth=DEPTH)
    w = O.make_weights(cfg, seed=77)
    x = O.bf16_round(O.hash_normal_like((FRAMES, PATCHES, HIDDEN), 78))
    emb = np.zeros((48900, HIDDEN), np.float32)
    idx = np.arange(FRAMES)
    t0 = time.perf_counter()
    toks = O.video_tokens(x, idx, cfg, w, emb, "fp32")
    dt = time.perf_counter() - t0
    assert np.isfinite(toks).all()
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = len(os.sched_getaffinity(0))
    return {"value": FRAMES / dt, "unit": "frames/s", "cores": int(threads), "kind": "port",
            "sample": f"1 video = {FRAMES} frames (2 chunks x 32, M={MEM_TOKENS}, D={HIDDEN}), fp32 numpy oracle, "
                      f"{dt:.1f} s, host has {len(os.sched_getaffinity(0))} usable cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame-scores", action="store_true", help="skip the column-sum pass (not the headline)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather of the final memory state")
    ap.add_argument("--videos-in-flight", type=int, default=2,
                    help="independent videos per step per GPU, each on its own HIP stream (fills partial-wave tails)")
    args = ap.parse_args()

    from memory_augmented_vlm_amd import distributed as D
    from memory_augmented_vlm_amd import _capi as capi
    rank, world, local = D.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device("cuda", local % max(1, torch.cuda.device_count()))   # (rehearsals may share one GPU)
    torch.cuda.set_device(device)
    capi.lib()   # fail loudly if the HIP library is missing
    if os.environ.get("MAVLM_GEMM_TILE"):      # A/B hook (diagnostics): 128 / 256 / 257, default automatic
        capi.check(capi.lib().mavlm_set_gemm_tile(int(os.environ["MAVLM_GEMM_TILE"])), "set_gemm_tile")
    if os.environ.get("MAVLM_ATTN_IMPL"):
        capi.check(capi.lib().mavlm_set_attention_impl(int(os.environ["MAVLM_ATTN_IMPL"])), "set_attention_impl")

    model, arch = build_model(device)
    rm = model.recurrent_memory_transformer
    rm.compute_frame_scores = not args.no_frame_scores
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    B = max(1, args.videos_in_flight)
    xs = [torch.randn((FRAMES, PATCHES, HIDDEN), generator=g).to(device).to(torch.bfloat16) for _ in range(B)]  # in HBM
    x = xs[0]
    idx_cpu = torch.arange(FRAMES)
    pool = arch.MemoryPathPool(model, B)
    for slot in pool.slots:
        slot.recurrent_memory_transformer.compute_frame_scores = not args.no_frame_scores
    mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=device)
    frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=device)
    gathered = torch.empty((world, MEM_TOKENS, PATCHES, HIDDEN), device=device, dtype=torch.bfloat16) if world > 1 else None
    # the gather runs asynchronously while the next video overwrites the FIFO ring -> gather from a private copy
    send = torch.empty((MEM_TOKENS, PATCHES, HIDDEN), device=device, dtype=torch.bfloat16) if world > 1 else None
    pending = [None]
    do_gather = world > 1 and not args.no_gather

    def step(single=False):
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        if single or B == 1:
            toks, _ = arch.video_memory_tokens(model, x, idx_cpu, mp, fp, model.image_newline)
        else:
            toks = pool.run([(xi, idx_cpu) for xi in xs], mp, fp, model.image_newline)[0]
        if do_gather:
            if pending[0] is not None:
                pending[0].wait()                       # previous gather (overlapped with this step) is done with `send`
            send.copy_(rm.memory_cache[-1])             # final memory state of this rank's (first) video
            _, pending[0] = D.all_gather_memory_state(send, out=gathered, async_op=True)
        return toks

    def sync():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    with torch.no_grad():
        for _ in range(args.warmup):
            toks = step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            toks = step()
        sync()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rows_expected = 10 + 2 * MEM_TOKENS * PATCHES + 1 + 9 + 32 * PATCHES + 1
    assert toks.shape == (rows_expected, HIDDEN) and bool(torch.isfinite(toks.float()).all())

    # ---- instrumented pass: the same K steps with a HIP-event pair around every kernel launch
    lib = capi.lib()
    nk = len(capi.KERNEL_KINDS)
    ms = (ctypes.c_double * nk)()
    ln = (ctypes.c_int64 * nk)()
    fl = (ctypes.c_double * nk)()
    by = (ctypes.c_double * nk)()
    with torch.no_grad():
        lib.mavlm_prof_enable(1)
        for _ in range(args.steps):
            step(single=True)      # ONE video in flight: kernel durations are not smeared by the other stream
        sync()
        capi.check(lib.mavlm_prof_read(ms, ln, fl, by, nk), "mavlm_prof_read")
        lib.mavlm_prof_enable(0)
    kernels = {}
    for i, name in enumerate(capi.KERNEL_KINDS):
        if ln[i]:
            avg_ms = ms[i] / ln[i]
            kernels[name] = {"launches_per_step": ln[i] / args.steps, "avg_ms": round(avg_ms, 5),
                             "ms_per_step": round(ms[i] / args.steps, 4),
                             "tflops": round(fl[i] / (ms[i] * 1e-3) / 1e12, 1) if fl[i] else None,
                             "alg_gbs": round(by[i] / (ms[i] * 1e-3) / 1e9, 1)}
    # Dominant kernel SYMBOL: attn_fwd3_kernel<BF16> (5 launches per video of one template; the GEMM time is spread
    # over four epilogue instantiations of two tile kernels, see `kernels`).
    dom = "attention_fwd"
    di = capi.KERNEL_KINDS.index(dom)
    achieved = fl[di] / (ms[di] * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_attn_fwd3_hbm_traffic.json")
    if os.path.exists(tpath):      # PMC summary committed from a separate rocprofv3 --pmc run (tests/pmc_traffic.sh)
        traffic = json.load(open(tpath)).get("bench_avg_bytes_per_launch")
    roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_note": "HBM bytes per launch, PMC FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, profiles/"
                                "r01_attn_fwd3_hbm_traffic.json; algorithmic bytes per launch in alg_bytes_per_launch",
                "kernel": "attn_fwd3_kernel<BF16>",
                "avg_launch_ms": round(ms[di] / ln[di], 5), "launches_per_step": ln[di] / args.steps,
                "alg_flops_per_launch": fl[di] / ln[di], "alg_bytes_per_launch": by[di] / ln[di],
                "hbm_gbs_algorithmic": round(by[di] / (ms[di] * 1e-3) / 1e9, 1)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps * B * FRAMES / elapsed
        flops = algorithmic_flops()
        out = {
            "metric": "Memory-Fuser frames/sec (196 vis-tokens x 1024d, 64 mem tokens)",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: 64-frame video, 2 recurrent chunks of 32 frames, 64 memory tokens, "
                                   "196 tokens/frame, D=1024, H=8, depth 2, frame scores on; "
                                   f"a step = {B} independent video(s) per GPU, each on its own HIP stream",
                       "frames": FRAMES, "mem_tokens": MEM_TOKENS, "hidden": HIDDEN, "parallelism": f"replica x{world}", "videos_per_step_per_gpu": B,
                       "allgather_final_memory": bool(do_gather), "frame_scores": not args.no_frame_scores},
            "roofline": roofline,
            "alg_tflop_per_video": round(flops / 1e12, 3),
            "path_mfma_frac": round(B * flops / (elapsed / args.steps) / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "kernel_timing_note": "per-kernel numbers from an instrumented pass with ONE video in flight",
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
