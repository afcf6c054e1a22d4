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


def _worker(rank, world, port, out_dir):
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
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2, cache_cap=3)
    w = O.make_weights(cfg, seed=71)
    proj = make_projector(cfg, w, "bf16", cache_cap=3)
    sharded = D.RowShardedMemory(proj)
    frames = [3, 2, 3, 1, 2]
    segs = [to_dev(O.bf16_round(O.hash_normal_like((f, 196, 1024), 7100 + t))) for t, f in enumerate(frames)]
    with torch.no_grad():
        proj.memory_cache = []
        ref = []
        for s_ in segs:
            cache, scores = proj(s_)
            ref.append((cache[-1].clone(), scores[-1].clone(), len(cache)))
        for t, s_ in enumerate(segs):
            cache, scores = sharded.step(s_)
            assert len(cache) == ref[t][2]
            err = float((cache[-1].float() - ref[t][0].float()).norm() / ref[t][0].float().norm())
            serr = float((scores.float() - ref[t][1].float()).norm() / ref[t][1].float().norm())
            np.save(os.path.join(out_dir, f"err_{rank}_{t}.npy"), np.array([err, serr]))
        # every rank holds the same full FIFO
        mine = torch.stack(list(cache)).float().cpu()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        assert all(torch.equal(g, mine) for g in gathered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_video_matches_single_gpu(world, tmp_path):
    """5 chunks, FIFO cap 3 (wraps): the sharded recurrence tracks the single-GPU engine within the 16-bit chain
    noise (different kernel plans at 1/W of the rows -> different fp32 summation order; no systematic drift)."""
    port = 29600 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        for t in range(5):
            err, serr = np.load(tmp_path / f"err_{rank}_{t}.npy")
            assert err < 6e-3 and serr < 5e-3, (rank, t, err, serr)
