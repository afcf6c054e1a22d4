"""CPU: the plans that are part of a result's rounding - key splits of small grids, the levelled stream-K schedule, the
rule for computing the frame scores inside the forward - are pure host functions of the shape in the C library, and the
oracle's emulation re-states them (oracle/memory_path.py).  The parity gates rely on the two agreeing for EVERY shape:
swept here without a GPU (no kernel is launched)."""
import ctypes
import itertools

import pytest

import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi
from oracle import memory_path as O

ROWS = [1, 100, 128, 784, 1568, 3136, 4100, 8192, 8320, 12544, 25088]
KEYS = [1, 64, 196, 588, 1000, 1024, 4032, 4096, 6272, 12544, 62720, 250880]
HEADS = [1, 2, 3, 8, 64]


@pytest.fixture(scope="module")
def lib():
    return capi.lib()


def test_attention_plans_match_the_oracle(lib):
    info = (ctypes.c_int32 * 4)()
    for R, S, H in itertools.product(ROWS, KEYS, HEADS):
        capi.check(lib.mavlm_attention_plan(R, S, H, info), "plan")
        G, QB, full, levels = O.streamk_plan(R, S, H)
        ns, _ = O.split_plan(R, S, H)
        assert info[1] == G, (R, S, H)
        if G:
            assert info[0] == QB // 32 and info[2] == len(levels) and info[3] == 1, (R, S, H)
        else:
            assert info[3] == ns, (R, S, H)
        # a workspace is needed exactly when a schedule cuts keys
        need = lib.mavlm_attention_ws_floats(R, S, H)
        assert (need > 0) == bool(G or ns > 1), (R, S, H)
        if G:
            assert need == G * len(levels) * (QB * 128 + QB), (R, S, H)


def test_streamk_unit_order_is_a_bijection_and_matches_the_oracle(lib):
    """Round 4: the levelled stream-K schedule deals its units XCD-major (every XCD walks a contiguous range of the head-major
    unit order).  Positions -> units must be a bijection (every unit exactly once, whole or cut), the C map and the oracle's
    must agree, and at the bench launch (2 videos x 8 heads x 49 blocks of 256 queries) an XCD owns exactly two heads."""
    for R, S, H in [(12544, 6272, 16), (12544, 6272, 8), (12544, 12544, 16), (12544, 125440, 8), (1568, 6272, 64), (8320, 4096, 8),
                    (25088, 6272, 8), (100000, 4096, 3), (1100, 6400, 64)]:
        G, QB, full, levels = O.streamk_plan(R, S, H)
        if not G:
            continue
        units = -(-R // QB) * H
        seen = []
        for si in range(full):
            for v in range(G):
                u = lib.mavlm_attention_plan_unit(R, S, H, -1, si, v)
                assert u == O.streamk_unit_of_round(G, full, levels, si, v), (R, S, H, si, v)
                seen.append(u)
        for lv, (k, base, n) in enumerate(levels):
            for ul in range(n):
                u = lib.mavlm_attention_plan_unit(R, S, H, lv, 0, ul)
                assert u == O.streamk_unit_of_level(G, full, levels, lv, ul), (R, S, H, lv, ul)
                seen.append(u)
        assert sorted(seen) == list(range(units)), (R, S, H)
        # the workgroups of one XCD (virtual ids [x G/8, (x+1) G/8)) walk a contiguous unit range
        W = G // 8
        for x in range(8):
            mine = [O.streamk_unit_of_round(G, full, levels, si, v) for si in range(full) for v in range(x * W, (x + 1) * W)]
            for lv, (k, base, n) in enumerate(levels):
                mine += [O.streamk_unit_of_level(G, full, levels, lv, ul) for ul in range(n) if (ul << k) // W == x]
            assert sorted(mine) == list(range(min(mine), max(mine) + 1)), (R, S, H, x)
    G, QB, full, levels = O.streamk_plan(12544, 6272, 16)
    assert (G, QB, full) == (256, 256, 3)
    assert [O.streamk_unit_of_round(G, full, levels, 0, 32 * x) for x in range(8)] == [98 * x for x in range(8)]


def test_wide_head_plans_match_the_oracle(lib):
    """head_dim 448: the levelled stream-K plan of the 32-query-wave kernel (more 128-query units than 256 workgroups; H up to
    the heads of a row batch) and the key splits of the small single-video grids; head_dim 256 (16-query kernel): splits only."""
    info = (ctypes.c_int32 * 4)()
    for R, S, H in itertools.product(ROWS, KEYS, [1, 2, 8, 32, 64]):
        G, full, levels = O.streamk_plan_wide(R, S, H)
        ns, _ = O.split_plan_wide(R, S, H)
        capi.check(lib.mavlm_attention_hd_plan_info(R, S, H, 448, info), "plan info")
        assert info[0] == G, (R, S, H)
        need = lib.mavlm_attention_hd_ws_floats(R, S, H, 448)
        if G:
            assert info[1] == full and info[2] == len(levels) and info[3] == 1, (R, S, H)
            assert need == G * len(levels) * (128 * 448 + 128), (R, S, H)
            cuts = O.streamk_split_tiles_wide(R, S, H)
            assert len(cuts) == sum(n for _, _, n in levels), (R, S, H)
            nt = -(-S // 32)
            for rng in cuts.values():                         # a cut unit's ranges tile its keys exactly, in order
                assert rng[0][0] == 0 and rng[-1][1] == nt and all(a[1] == b[0] for a, b in zip(rng, rng[1:])), (R, S, H)
        else:
            assert info[3] == ns, (R, S, H)
            assert (need > 0) == (ns > 1), (R, S, H)
            if ns > 1:
                assert need == ns * (R * H * 448 + H * R), (R, S, H)
        capi.check(lib.mavlm_attention_hd_plan_info(R, S, H, 256, info), "plan info")
        assert info[0] == 0 and info[3] == ns, (R, S, H)
        assert lib.mavlm_attention_hd_ws_floats(R, S, H, 256) == (ns * (R * H * 256 + H * R) if ns > 1 else 0), (R, S, H)


def test_frame_score_rule_matches_the_oracle(lib):
    for R, H, P in itertools.product(ROWS, HEADS, [4, 60, 64, 100, 196, 198]):
        for F in (1, 2, 3, 32, 40, 64, 65):
            S = F * P
            assert bool(lib.mavlm_frame_scores_fused(R, S, H, P)) == O.frame_scores_fused(R, S, H, P), (R, S, H, P)
            assert (lib.mavlm_attention_frames_ws_floats(R, S, H, P) > 0) == (P % 4 == 0 and P >= 64 and F <= 64), (R, S, H, P)
        assert not lib.mavlm_frame_scores_fused(R, 3 * P + 4, H, P)          # keys that are not whole frames
    try:
        capi.check(lib.mavlm_set_frame_score_mode(0), "mode")
        assert not lib.mavlm_frame_scores_fused(12544, 6272, 8, 196)
    finally:
        lib.mavlm_set_frame_score_mode(1)
    assert lib.mavlm_frame_scores_fused(12544, 6272, 8, 196) and not lib.mavlm_frame_scores_fused(1568, 6272, 8, 196)


def test_column_sum_plan_is_consistent(lib):
    info = (ctypes.c_int32 * 2)()
    for R, S, H in itertools.product(ROWS, KEYS, [1, 3, 8]):
        capi.check(lib.mavlm_attention_colsum_plan(R, S, H, info), "colsum plan")
        wgs, planes = info[0], info[1]
        nkb, ntq = -(-S // 128), -(-R // 64)
        total = nkb * H * ntq
        assert 1 <= wgs <= min(512, total) and planes >= 1, (R, S, H)
        assert lib.mavlm_attention_colsum_floats(R, S, H) >= planes * H * S, (R, S, H)
        # the planes bound: a unit's ntq tiles are spread over at most ceil((ntq - 1) / floor(total / wgs)) + 1 workgroups
        q = total // wgs
        assert planes <= (ntq - 1 + q - 1) // q + 1, (R, S, H)
