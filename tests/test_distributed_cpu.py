"""World-size-2 gloo test of the multi-GPU host logic (video sharding + all-gather of the final memory state)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from memory_augmented_vlm_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_videos, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    mine = list(D.shard_range(n_videos, r, w))
    # stand-in for the final memory state of the last video this rank processed: [M,P,D] filled with its video id
    state = torch.full((2, 3, 4), float(mine[-1] if mine else -1))
    out, work = D.all_gather_memory_state(state, async_op=True)
    work.wait()
    flag = torch.zeros(1)
    if r == 0:
        flag.fill_(1.0)
    dist.broadcast(flag, src=0)          # the reference's only hot-path collective (llava_arch.py:378-386)
    q.put((rank, mine, out[:, 0, 0, 0].tolist(), float(flag)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_videos", [5, 2])
def test_shard_and_allgather_world2(n_videos):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_videos, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    all_ids = res[0][1] + res[1][1]
    assert all_ids == list(range(n_videos))                      # disjoint, complete, ordered
    expect = [float(res[0][1][-1]), float(res[1][1][-1])]
    assert res[0][2] == expect and res[1][2] == expect           # both ranks see both states
    assert res[0][3] == res[1][3] == 1.0


def test_shard_range_properties():
    for n in range(0, 20):
        for w in (1, 2, 3, 8):
            parts = [list(D.shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_single_process_gather_is_identity():
    s = torch.arange(24.).reshape(2, 3, 4)
    out, work = D.all_gather_memory_state(s)
    assert work is None and torch.equal(out[0], s)


def test_row_sharded_memory_argument_checks():
    """RowShardedMemory (SURVEY.md §8e option 2): host-side contract without a GPU - the memory tokens must divide over
    the ranks; a single process owns all rows."""
    import types
    from memory_augmented_vlm_amd import distributed as D
    proj = types.SimpleNamespace(num_memory_tokens=8, patch_size=196)
    s = D.RowShardedMemory(proj)
    assert (s.world, s.rank, s.tokens, s.token0) == (1, 0, 8, 0) and s.cache == []


def _shard_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import types
    D.init_from_env("gloo")
    s = D.RowShardedMemory(types.SimpleNamespace(num_memory_tokens=8, patch_size=196))
    try:
        D.RowShardedMemory(types.SimpleNamespace(num_memory_tokens=7, patch_size=196))
        bad = False
    except ValueError:
        bad = True
    # the in-place all-gather the sharded step issues: every rank contributes its token rows of a [M, P, D] slot
    slot = torch.zeros(8, 2, 3)
    slot[s.token0:s.token0 + s.tokens] = rank + 1.0
    own = slot[s.token0:s.token0 + s.tokens].clone()
    dist.all_gather_into_tensor(slot.view(-1), own.reshape(-1))
    q.put((rank, s.tokens, s.token0, bad, slot[:, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_memory_world2_partition_and_gather():
    """world 2 over gloo: rank g owns memory tokens [4g, 4g+4); the per-step exchange (all-gather of the owned rows into the
    FIFO slot) leaves every rank with the full entry; a token count that does not divide is refused."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[:4] for g in got] == [(0, 4, 0, True), (1, 4, 4, True)]
    assert got[0][4] == got[1][4] == [1.0] * 4 + [2.0] * 4


def _dropout_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import types
    from memory_augmented_vlm_amd.model import llava_arch as arch
    D.init_from_env("gloo")
    torch.manual_seed(1000 + 17 * rank)                    # different RNG streams per rank, as in a real job
    me = types.SimpleNamespace(device=torch.device("cpu"))
    draws = [arch.LlavaMetaForCausalLM.get_synced_dropout_decision(me, 0.5) for _ in range(24)]
    q.put((rank, draws))
    dist.barrier()
    dist.destroy_process_group()


def test_synced_dropout_decision_world2():
    """llava_arch.py:378-386: rank 0 draws Bernoulli(p), a 1-element broadcast makes every rank take the same branch
    (frames dropped or kept) although their RNG streams differ."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dropout_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1] and all(isinstance(d, bool) for d in res[0])
    assert 2 <= sum(res[0]) <= 22                          # a fair coin, not a constant
