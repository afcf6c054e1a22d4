"""Chunk boundaries of the recurrent memory loop.

Mirrors ``uniform_segment_variant`` of the reference (llava/model/memory_module/segment.py:169-192): fixed
chunks of ``d`` frames, the remainder (if any) as a final shorter chunk.  Host-side integer math only."""


def uniform_segment_variant(features, d=32):
    """``features``: anything with ``.shape[0]`` = T (the reference passes ``image.mean(dim=1)`` only for its
    length, llava_arch.py:528) or an int.  Returns ``[0, d, 2d, ..., T]``."""
    T = int(features) if isinstance(features, int) else int(features.shape[0])
    full = T // d
    bounds = [i * d for i in range(full + 1)]
    if full * d < T:
        bounds.append(T)
    return bounds


# ------------------------------------------------------------------------------------------------------------------
# Inactive variants: scene-aware segmentation / sampling, segment.py:3-53,252-337 of the
# reference (imported at llava_arch.py:34, the call at :516 is commented out).  The per-frame pooling and the adjacent
# cosine similarities run as HIP kernels; the boundary / budget logic is host-side integer work on <= a few hundred
# values, kept separate (`*_from_similarity`) so that it is testable without a GPU.
# ------------------------------------------------------------------------------------------------------------------
import torch

from ... import _capi as capi
from ... import _ops as ops


def cal_depth_score(sim_scores):
    """depth[i] = (highest value reached climbing left from i while values do not fall) + (same to the right)
    - 2 sim[i]   (segment.py:3-25).  `sim_scores`: 1-D CPU tensor; returns a tensor of the same dtype."""
    s = sim_scores.detach().cpu()
    vals = s.tolist()
    n = len(vals)
    out = torch.zeros(n, dtype=s.dtype)
    for i in range(n):
        left = vals[i]
        j = i - 1
        while j >= 0 and vals[j] >= left:
            left = vals[j]
            j -= 1
        right = vals[i]
        j = i + 1
        while j < n and vals[j] >= right:
            right = vals[j]
            j += 1
        out[i] = torch.tensor(left, dtype=s.dtype) + torch.tensor(right, dtype=s.dtype) - 2 * s[i]
    return out


def adjacent_similarity(features, eps=1e-2):
    """cosine similarity of consecutive frame vectors [T,D] -> CPU float32 [T-1] (HIP kernel; segment.py:33)."""
    if not features.is_cuda:
        raise capi.MavlmError("segment: frame features are not on a GPU (no CPU fallback)")
    return ops.adjacent_cosine(features.float().contiguous(), eps).cpu()


def segment_from_similarity(sim_scores, num_frames, alpha=0.5, k=None):
    """Host half of `segment` (segment.py:34-53): first similarity replaced by the second, depth scores, boundaries =
    depth > mean + alpha*std (or the top-k), the sequence end appended unless the last boundary is T-1."""
    sim = sim_scores.clone()
    sim[0] = sim[1]
    depth = cal_depth_score(sim)
    if k is not None:
        b = torch.topk(depth, k).indices.sort()[0]
    else:
        std, mean = torch.std_mean(depth)
        b = (depth > mean + alpha * std).nonzero().squeeze(-1)
    b = b.tolist()
    if not b or b[-1] != num_frames - 1:
        b.append(num_frames)
    return sorted(set(b)), depth


def segment(features, alpha=0.5, k=None):
    """features [T,D] on the GPU -> (boundaries, depth_scores)  (segment.py:28-53)."""
    if features.shape[0] == 1:
        return [0], torch.zeros(1)
    return segment_from_similarity(adjacent_similarity(features), features.shape[0], alpha, k)


def uniform_segment(features, d=32):
    """Chunks of d frames with the remainder FIRST (segment.py:130-166)."""
    T = int(features) if isinstance(features, int) else int(features.shape[0])
    if T <= d:
        return [0, T]
    first = T % d
    return ([0] if first == 0 else [0, first]) + list(range(first + d, T + 1, d))


def scenes_priority_from_boundaries(boundaries, depth_scores, num_frames, sample_num=32):
    """Host half of `sample_scenes_priority` (segment.py:270-337): exactly `sample_num` distinct frame indices, spread
    over the scenes in proportion to their length, or - with more scenes than samples - the centres of the scenes that
    start at the most surprising boundaries; shortfalls are filled from the unused frames with torch.randperm."""
    T = num_frames
    bounds = sorted(set(list(boundaries) + [0, T]))
    scenes = len(bounds) - 1
    picked = []
    if scenes <= sample_num:
        lengths = [bounds[i + 1] - bounds[i] for i in range(scenes)]
        spare, total = sample_num - scenes, sum(lengths)
        budget = [1 + int(spare * n / total) for n in lengths]
        while sum(budget) < sample_num:
            budget[sum(budget) % scenes] += 1
        while sum(budget) > sample_num:
            budget[budget.index(max(budget))] -= 1
        for i in range(scenes):
            lo, hi, want = bounds[i], bounds[i + 1], budget[i]
            if hi - lo <= want:
                picked.extend(range(lo, hi))
            else:
                picked.extend(torch.linspace(lo, hi - 1, steps=want).round().long().tolist())
    else:
        scores = [0] + [depth_scores[b - 1].item() for b in bounds[1:-1]]
        order = sorted(range(scenes), key=lambda i: -scores[i])[:sample_num]       # stable: ties keep scene order
        picked = [(bounds[i] + bounds[i + 1]) // 2 for i in order]
    picked = sorted(set(picked))
    if len(picked) < sample_num:
        pool = sorted(set(range(T)) - set(picked))
        need = sample_num - len(picked)
        if len(pool) >= need:
            picked.extend(pool[i] for i in torch.randperm(len(pool))[:need].tolist())
        else:
            picked.extend(pool)
    return sorted(picked)[:sample_num]


def sample_scenes_priority(features, sample_num=32, alpha=0.3, k=None):
    """features [T,P,D] on the GPU -> `sample_num` frame indices (segment.py:252-337)."""
    if features.dim() != 3:
        raise capi.MavlmError("sample_scenes_priority expects [frames, patches, dim]")
    if not features.is_cuda:
        raise capi.MavlmError("sample_scenes_priority: features are not on a GPU (no CPU fallback)")
    T = features.shape[0]
    _, means = ops.frame_mean(features.contiguous(), want_f32=True)
    bounds, depth = segment(means, alpha=alpha, k=k)
    return scenes_priority_from_boundaries(bounds, depth, T, sample_num)


def adjusted_from_similarity(sim_scores, num_frames, alpha=0.5, k=None, min_distance=32, max_distance=64):
    """Host half of `adjusted_segment` (segment.py:56-128): depth-score boundaries (at most 15 by threshold, else the
    top 15), then gaps shorter than `min_distance` are merged and gaps longer than `max_distance` get evenly spaced
    extra boundaries; the sequence end is appended or merged into the last boundary."""
    T = num_frames
    depth = cal_depth_score(sim_scores)
    if k is not None:
        b = torch.topk(depth, k).indices.sort()[0]
    else:
        std, mean = torch.std_mean(depth)
        b = (depth > mean + alpha * std).nonzero().squeeze(-1)
        if len(b) > 15:
            b = torch.topk(depth, 15).indices.sort()[0]
    b = b.tolist()
    if not b or b[-1] != T:
        b.append(T)
    if b[0] != 0:
        b.insert(0, 0)
    b = sorted(set(b))
    out = [b[0]]
    for cand in b[1:-1]:
        gap = cand - out[-1]
        if gap < min_distance:
            continue
        if gap > max_distance:
            extra, start = int(gap / max_distance), out[-1]
            for i in range(1, extra + 1):
                nb = start + round(gap * i / (extra + 1))
                if out[-1] < nb < cand:
                    out.append(nb)
        out.append(cand)
    gap = T - out[-1]
    if gap >= min_distance or out[-1] == 0:
        out.append(T)
    else:
        out[-1] = T
    return out


def adjusted_segment(features, alpha=0.5, k=None, min_distance=32, max_distance=64):
    """features [T,D] on the GPU -> boundary list (segment.py:56-128; torch's default cosine eps 1e-8, no first-score
    patch - as the reference)."""
    if features.shape[0] == 1:
        return [0]
    return adjusted_from_similarity(adjacent_similarity(features, eps=1e-8), features.shape[0], alpha, k, min_distance,
                                    max_distance)
