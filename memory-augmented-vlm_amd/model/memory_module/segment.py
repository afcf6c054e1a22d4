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
