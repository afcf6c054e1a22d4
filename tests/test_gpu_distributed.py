"""-m gpu: the row-sharded single-video mode (SURVEY.md §8e option 2) - W ranks, each owning M/W memory tokens, against
the single-GPU engine.  The ranks share this box's one GPU and talk over gloo (the driver's 8-GPU node uses RCCL; the
collective calls are the same torch.distributed ones)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


FRAMES = {1024: [3, 2, 3, 1, 2], 3584: [2, 1, 1, 2]}      # chunks per width; FIFO cap 3 -> the ring wraps in both


def _worker(rank, world, port, out_dir, hidden=1024):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import memory_augmented_vlm_amd  # noqa: F401
    from memory_augmented_vlm_amd import distributed as D
    from oracle import memory_path as O
    from gpu_util import to_dev
    from test_gpu_path import make_projector
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.PathConfig(hidden=hidden, heads=8, mem_tokens=8, depth=2, cache_cap=3)
    w = O.make_weights(cfg, seed=71)
    proj = make_projector(cfg, w, "bf16", cache_cap=3)
    sharded = D.RowShardedMemory(proj)
    frames = FRAMES[hidden]
    segs = [to_dev(O.bf16_round(O.hash_normal_like((f, 196, hidden), 7100 + t))) for t, f in enumerate(frames)]
    oracle_ref = os.path.join(out_dir, "oracle_mem.npy")
    omem = np.load(oracle_ref) if os.path.exists(oracle_ref) else None
    with torch.no_grad():
        proj.memory_cache = []
        ref = []
        for s_ in segs:
            cache, scores = proj(s_)
            ref.append((cache[-1].clone(), scores[-1].clone(), len(cache)))
        # NON-contiguous chunks (strided views of a wider buffer): `step` makes them contiguous, and the copy made for the
        # prefetch must be the one the next step runs on - the C library matches the prefetched chunk by pointer (round 4:
        # the projection used to be discarded silently for such inputs)
        wide = [torch.cat([s_, torch.zeros_like(s_)], dim=2) for s_ in segs]
        nc = [w_[:, :, :hidden] for w_ in wide]
        assert not nc[0].is_contiguous()
        sharded.reset()
        n_pre = 0
        for t, s_ in enumerate(nc):
            # (odd steps hand the next chunk over: its K/V projection runs under the all-gather, mavlm_project_chunk)
            pre = nc[t + 1] if t % 2 == 1 and t + 1 < len(nc) else None
            n_pre += pre is not None
            cache, scores = sharded.step(s_, prefetch=pre)
            assert len(cache) == ref[t][2]
            err = float((cache[-1].float() - ref[t][0].float()).norm() / ref[t][0].float().norm())
            serr = float((scores.float() - ref[t][1].float()).norm() / ref[t][1].float().norm())
            oerr = -1.0
            if omem is not None:          # newest memory of step t against the CPU oracle (computed once, by the parent)
                o = torch.from_numpy(omem[t]).to(cache[-1].device)
                oerr = float((cache[-1].float().reshape(-1) - o.reshape(-1)).norm() / o.norm())
            np.save(os.path.join(out_dir, f"err_{rank}_{t}.npy"), np.array([err, serr, oerr]))
        from memory_augmented_vlm_amd import _capi as capi
        assert capi.lib().mavlm_prefetch_hits(sharded._engine.ctx) == n_pre > 0        # every prefetched projection was used
        # the Memory-Fuser MLP over the FIFO, row-sharded (each rank fuses its tokens, one all-gather): against the module
        # applied to the whole FIFO on this rank (row-independent GEMMs: the same kernel per row block -> same bits)
        from memory_augmented_vlm_amd.model import llava_arch as arch
        torch.manual_seed(7)
        fuser = arch.MemoryFuserMLP(hidden).to("cuda").to(torch.bfloat16)
        e0 = torch.randn(hidden, device="cuda").bfloat16() * 0.02
        got = sharded.fused_memory(fuser, e0)
        whole = torch.cat(list(cache), 0)
        b2 = fuser[2].bias.float() + e0.float()
        from memory_augmented_vlm_amd import _ops as ops
        want = ops.linear(ops.linear(whole.reshape(-1, hidden), fuser[0].weight, fuser[0].bias.float(), capi.EPI_GELU),
                          fuser[2].weight, b2, capi.EPI_BIAS).view(-1, 196, hidden)
        assert got.shape == want.shape
        assert float((got.float() - want.float()).norm() / want.float().norm()) < 2e-3
        # every rank holds the same full FIFO
        mine = torch.stack(list(cache)).float().cpu()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        assert all(torch.equal(g, mine) for g in gathered)
    dist.barrier()
    dist.destroy_process_group()


_ORACLE = {}


def _oracle_chain(hidden, out_dir):
    """The same chunks through the CPU oracle (emulation mode), once per width, in the parent: newest memory per step and
    the calibrated chain tolerance (oracle with float64 accumulation against itself, test_gpu_path.chain_tol)."""
    if hidden not in _ORACLE:
        from oracle import memory_path as O
        from test_gpu_path import chain_tol, run_oracle_steps
        cfg = O.PathConfig(hidden=hidden, heads=8, mem_tokens=8, depth=2, cache_cap=3)
        w = O.make_weights(cfg, seed=71)
        segs = [O.bf16_round(O.hash_normal_like((f, 196, hidden), 7100 + t)) for t, f in enumerate(FRAMES[hidden])]
        ref = run_oracle_steps(cfg, w, "bf16", segs, np.float32)
        alt = run_oracle_steps(cfg, w, "bf16", segs, np.float64)
        _ORACLE[hidden] = (np.stack([r[0][-1] for r in ref]),
                           [chain_tol(O.rel_l2(a[0][-1], r[0][-1])) for a, r in zip(alt, ref)])
    np.save(os.path.join(out_dir, "oracle_mem.npy"), _ORACLE[hidden][0])
    return _ORACLE[hidden][1]


@pytest.mark.parametrize("world,hidden,with_oracle", [(2, 1024, True), (4, 1024, True), (2, 3584, True)])
def test_row_sharded_video_matches_single_gpu(world, hidden, with_oracle, tmp_path):
    """FIFO cap 3 over 4-5 chunks (wraps): the sharded recurrence tracks the single-GPU engine within the 16-bit chain
    noise (different kernel plans at 1/W of the rows -> different fp32 summation order; no systematic drift), and
    - independent of the engine - the CPU oracle within the calibrated chain tolerance (D = 1024).  hidden = 3584 is the
    LLaVA-OneVision-7B width of BASELINE.json configs[3] (wide-head kernels), against the oracle as well (round 3).  The
    ranks run the FUSED engine on their row shard (mavlm_config.q_tokens), one in-place all-gather per step."""
    tols = _oracle_chain(hidden, str(tmp_path)) if with_oracle else None
    port = 29600 + world + (hidden // 1024)
    mp.spawn(_worker, args=(world, port, str(tmp_path), hidden), nprocs=world, join=True)
    for rank in range(world):
        for t in range(len(FRAMES[hidden])):
            err, serr, oerr = np.load(tmp_path / f"err_{rank}_{t}.npy")
            assert err < 6e-3 and serr < 5e-3, (rank, t, err, serr)
            if with_oracle:
                assert 0 <= oerr < tols[t], (rank, t, oerr, tols[t])


@pytest.mark.parametrize("mode", ["replica", "shard-video"])
def test_bench_two_rank_rehearsal(mode, tmp_path):
    """`bench.py --gpus 2` end to end, both modes: self-launch of the two ranks, process group, the timed loop with its
    collectives (all-gather of the final memory state / per-step all-gather + all-reduce of the row shards), rank 0's JSON
    line with ranks_seen == 2.  The ranks share this box's GPU and talk over gloo (`--rehearse-gloo`; the line says so)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-gloo", "--steps", "2", "--warmup", "1",
           "--repeats", "1", "--min-seconds", "0", "--no-cpu-baseline", "--mode", mode]
    if mode == "shard-video":
        cmd += ["--shard-frames", "96", "--shard-hidden", "1024", "--shard-mem-tokens", "8"]
    env = dict(os.environ, MAVLM_BENCH_M8="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"]["ranks_seen"] == 2 and out["ranks"]["rehearsal_shared_gpus"] is True
    assert out["ranks"]["backend"] == "gloo" and out["value"] > 0 and out["config"]["mode"] == mode
    if mode == "replica":
        assert out["config"]["allgather_final_memory"] is True and out["scaling"] == "weak"
    else:
        assert out["scaling"] == "strong"


_NCCL_SMOKE = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MAVLM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MAVLM_ROOT"], "tests"))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import distributed as D
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
# the in-place form of RowShardedMemory.step: the send buffer is this rank's slice of the receive buffer, async
slot = torch.randn(8, 196, 1024, device="cuda").bfloat16()
keep = slot.clone()
work = dist.all_gather_into_tensor(slot.view(-1), slot[0:8].reshape(-1), async_op=True)
sc = torch.arange(5, device="cuda", dtype=torch.float32)
w2 = dist.all_reduce(sc, async_op=True)
work.wait(); w2.wait(); torch.cuda.synchronize()
assert torch.equal(slot, keep) and torch.equal(sc.cpu(), torch.arange(5, dtype=torch.float32))
g, _ = D.all_gather_memory_state(slot)
assert g.shape == (1, 8, 196, 1024) and torch.equal(g[0], keep)
# a 1-rank row shard of a real projector over RCCL's group: the sharded step degenerates to the single-GPU step
from oracle import memory_path as O
from test_gpu_path import make_projector
from gpu_util import to_dev
cfg = O.PathConfig(hidden=256, heads=2, mem_tokens=4, depth=2)
proj = make_projector(cfg, O.make_weights(cfg, seed=72))
sh = D.RowShardedMemory(proj)
seg = to_dev(O.bf16_round(O.hash_normal_like((2, 196, 256), 7200)))
with torch.no_grad():
    sh.reset(); c1, s1 = sh.step(seg)
    proj.memory_cache = []; c2, s2 = proj(seg)
assert torch.equal(c1[-1], c2[-1])
dist.barrier(); dist.destroy_process_group()
print("NCCL_SMOKE_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""


def test_rccl_one_rank_smoke(tmp_path):
    """No multi-GPU node is available to the builder: every multi-rank test above talks gloo on one shared card.  This at
    least LOADS RCCL (`backend="nccl"` is RCCL on ROCm) in a one-rank group on this box and runs the collective forms the
    multi-GPU paths use - `all_gather_into_tensor` IN PLACE (send = this rank's slice of the receive buffer, async), an async
    all-reduce, the final-memory all-gather helper - plus a 1-rank RowShardedMemory step.  Not a scaling measurement."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MAVLM_ROOT=root)
    r = subprocess.run([sys.executable, "-c", _NCCL_SMOKE], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and "NCCL_SMOKE_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
