"""MI355X-native recurrent memory-token + Memory-Fuser path (gfx950 HIP kernels behind a C ABI).

Layout mirrors the reference's ``llava.model`` surface for this path only:

    model/memory_module/MemoryController.py   Config, TransformerProjector (+ parameter containers)
    model/memory_module/position_encoding.py  TemporalPositionalEncoding
    model/memory_module/segment.py            uniform_segment_variant
    model/llava_arch.py                       MemoryPathMixin: module construction + the video memory driver
    csrc/ , lib/libmavlm.so                   HIP kernels + C ABI (include/mavlm.h)
"""
from . import _capi  # noqa: F401
from ._build import build_library, library_path  # noqa: F401

__all__ = ["build_library", "library_path"]
