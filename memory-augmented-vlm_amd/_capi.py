"""ctypes binding of include/mavlm.h.  No torch types cross this boundary: device pointers are ints
(``tensor.data_ptr()``), the stream is ``torch.cuda.current_stream().cuda_stream``.

There is no CPU fallback: if the library is missing every call raises (build it with
``python -c "import __graft_entry__ as g; g.build()"``)."""
import ctypes as C
import os

from ._build import library_path

MAX_DEPTH = 8
BF16, F16 = 0, 1
EPI_BIAS, EPI_RELU, EPI_GELU, EPI_RES_F32, EPI_F32 = 0, 1, 2, 3, 4
E_ARG, E_SHAPE, E_STATE = -1, -2, -3

vp = C.c_void_p
i32 = C.c_int32


class Config(C.Structure):
    _fields_ = [("hidden", i32), ("heads", i32), ("patches", i32), ("mem_tokens", i32), ("depth", i32),
                ("inter", i32), ("cache_cap", i32), ("max_chunk_frames", i32), ("dtype", i32), ("eps", C.c_float),
                ("batch", i32), ("q_token0", i32), ("q_tokens", i32), ("fused_ln", i32)]


LN_AUTO, LN_NEVER = 0, 1      # mavlm_config.fused_ln
LN_MAX_STREAMS = 8            # MAVLM_LN_MAX_STREAMS: streams that may run the fused Residual kernel concurrently (include/mavlm.h)


class AttnWeights(C.Structure):
    _fields_ = [("wq", vp), ("bq", vp), ("wo", vp), ("bo", vp), ("ln_g", vp), ("ln_b", vp)]


class Weights(C.Structure):
    _fields_ = [("mem0", vp), ("w_kv_seg", vp), ("b_kv_seg", vp),
                ("layer_attn", AttnWeights * MAX_DEPTH),
                ("w_up", vp * MAX_DEPTH), ("b_up", vp * MAX_DEPTH),
                ("w_down", vp * MAX_DEPTH), ("b_down", vp * MAX_DEPTH),
                ("ln2_g", vp * MAX_DEPTH), ("ln2_b", vp * MAX_DEPTH),
                ("evo", AttnWeights), ("w_kv_evo", vp), ("b_kv_evo", vp),
                ("w_f1", vp), ("b_f1", vp), ("w_f2", vp), ("b_f2_type0", vp), ("type1", vp)]


class Buffers(C.Structure):
    _fields_ = [("mem_ring", vp), ("evo_kv_ring", vp), ("workspace", vp), ("workspace_bytes", C.c_size_t)]


# name -> (restype, argtypes); must list every symbol include/mavlm.h declares
SIGNATURES = {
    "mavlm_abi_version": (C.c_int, []),
    "mavlm_create": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
    "mavlm_destroy": (None, [vp]),
    "mavlm_workspace_bytes": (C.c_size_t, [C.POINTER(Config)]),
    "mavlm_workspace_layout": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_size_t), i32]),
    "mavlm_bind_weights": (C.c_int, [vp, C.POINTER(Weights)]),
    "mavlm_bind_buffers": (C.c_int, [vp, C.POINTER(Buffers)]),
    "mavlm_ln_status_async": (C.c_int, [vp, vp, i32, vp]),
    "mavlm_reset": (C.c_int, [vp]),
    "mavlm_cache_len": (C.c_int, [vp]),
    "mavlm_newest_slot": (C.c_int, [vp]),
    "mavlm_steps": (C.c_int, [vp]),
    "mavlm_pe_add": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mavlm_step": (C.c_int, [vp, vp, i32, vp, i32, vp]),
    "mavlm_step_batch": (C.c_int, [vp, C.POINTER(vp), i32, vp, i32, vp]),
    "mavlm_batch": (C.c_int, [vp]),
    "mavlm_project_chunk": (C.c_int, [vp, vp, i32, vp]),
    "mavlm_project_chunk_ahead": (C.c_int, [vp, vp, i32, vp]),
    "mavlm_prefetch_hits": (C.c_int, [vp]),
    "mavlm_fuse_emit_batch": (C.c_int, [vp, C.POINTER(vp), vp, i32, vp, i32, vp, i32, vp, i32, vp, C.c_int64,
                                        C.POINTER(C.c_int64), vp]),
    "mavlm_fuse_emit": (C.c_int, [vp, vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, C.c_int64,
                                  C.POINTER(C.c_int64), vp]),
    "mavlm_linear": (C.c_int, [vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mavlm_attention": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, C.c_float, i32, vp]),
    "mavlm_attention_hd": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, C.c_float, i32, vp]),
    "mavlm_linear_ln_ws_bytes": (C.c_int64, [i32, i32, i32]),
    "mavlm_linear_ln": (C.c_int, [vp, i32, vp, i32, vp, vp, i32, vp, vp, C.c_float, vp, i32, vp, i32, i32, i32, vp, C.c_int64,
                                  i32, vp]),
    "mavlm_set_fused_layernorm": (C.c_int, [i32]),
    "mavlm_workspace_ln_ctl_offset": (C.c_int64, [C.POINTER(Config)]),
    "mavlm_ln_ctl_offset": (C.c_int64, [vp]),
    "mavlm_linear_ws_floats": (C.c_int64, [i32, i32, i32, i32, i32]),
    "mavlm_linear_ws": (C.c_int, [vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, C.c_int64, i32, vp]),
    "mavlm_attention_ws_floats": (C.c_int64, [i32, i32, i32]),
    "mavlm_attention_ws": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, C.c_float, vp, C.c_int64, i32, vp]),
    "mavlm_attention_hd_ws_floats": (C.c_int64, [i32, i32, i32, i32]),
    "mavlm_attention_hd_plan_info": (C.c_int, [i32, i32, i32, i32, vp]),
    "mavlm_attention_hd_ws": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, C.c_float, vp, C.c_int64,
                                        i32, vp]),
    "mavlm_attention_colsum_hd": (C.c_int, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, C.c_float, i32, vp]),
    "mavlm_attention_colsum": (C.c_int, [vp, i32, vp, i32, vp, vp, C.c_int64, i32, i32, i32, C.c_float, i32, vp]),
    "mavlm_layernorm": (C.c_int, [vp, vp, i32, vp, vp, vp, i32, i32, C.c_float, i32, vp]),
    "mavlm_row_add": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mavlm_pool_bilinear": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mavlm_attention_bwd": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, vp, i32,
                                      i32, i32, i32, C.c_float, i32, vp]),
    "mavlm_attention_bwd_hd": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, vp, i32,
                                         i32, i32, i32, i32, C.c_float, i32, vp]),
    "mavlm_linear_splitk": (C.c_int, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp]),
    "mavlm_layernorm_bwd_ws_floats": (C.c_int64, [i32]),
    "mavlm_layernorm_bwd": (C.c_int, [vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, C.c_float, i32, vp]),
    "mavlm_transpose": (C.c_int, [vp, i32, i32, i32, vp, i32, vp]),
    "mavlm_rowsum": (C.c_int, [vp, i32, i32, i32, vp, i32, vp]),
    "mavlm_act": (C.c_int, [i32, vp, vp, vp, C.c_int64, i32, vp]),
    "mavlm_attention_probs": (C.c_int, [vp, i32, vp, vp, i32, i32, i32, i32, C.c_float, i32, vp]),
    "mavlm_attention_dscores": (C.c_int, [vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, C.c_float, i32, vp]),
    "mavlm_rowdot_heads": (C.c_int, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "mavlm_frame_mean": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "mavlm_adjacent_cosine": (C.c_int, [vp, vp, i32, i32, C.c_float, vp]),
    "mavlm_gru_sequence": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mavlm_set_attention_bwd_fused": (C.c_int, [i32]),
    "mavlm_set_gemm_tile": (C.c_int, [i32]),
    "mavlm_set_gemm_rows": (C.c_int, [i32]),
    "mavlm_set_gemm_order": (C.c_int, [i32]),
    "mavlm_set_attention_impl": (C.c_int, [i32]),
    "mavlm_attention_plan": (C.c_int, [i32, i32, i32, C.POINTER(i32)]),
    "mavlm_attention_plan_unit": (C.c_int, [i32, i32, i32, i32, i32, i32]),
    "mavlm_set_attention_unit_order": (C.c_int, [i32]),
    "mavlm_set_attention_streamk_min_tiles": (C.c_int, [i32]),
    "mavlm_set_attention_streamk_waves": (C.c_int, [i32]),
    "mavlm_set_attention_colsum_wgs": (C.c_int, [i32]),
    "mavlm_set_attention_wide_groups": (C.c_int, [i32]),
    "mavlm_set_frame_score_mode": (C.c_int, [i32]),
    "mavlm_set_splitk_layernorm": (C.c_int, [i32]),
    "mavlm_set_gemm_short_splits": (C.c_int, [i32]),
    "mavlm_frame_scores_fused": (C.c_int, [i32, i32, i32, i32]),
    "mavlm_attention_frames_ws_floats": (C.c_int64, [i32, i32, i32, i32]),
    "mavlm_attention_frames": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, C.c_float, i32, vp, C.c_int64,
                                          vp, i32, vp]),
    "mavlm_attention_colsum_floats": (C.c_int64, [i32, i32, i32]),
    "mavlm_attention_colsum_plan": (C.c_int, [i32, i32, i32, C.POINTER(i32)]),
    "mavlm_prof_enable": (C.c_int, [i32]),
    "mavlm_prof_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), i32]),
}

KERNEL_KINDS = ("gemm", "attention_fwd", "attention_colsum", "layernorm", "row_add", "misc", "attention_bwd",
                "gemm_splitk", "transpose", "attention_merge", "attention_fwd_frames", "gemm_layernorm")

_lib = None


class MavlmError(RuntimeError):
    pass


def lib():
    """Load libmavlm.so (once).  Raises loudly if it has not been built - there is no fallback path."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise MavlmError(f"{path} is missing: the HIP library has not been built "
                             "(run __graft_entry__.build()); there is no CPU fallback for this path")
        # Load PyTorch-ROCm's HIP runtime first: torch bundles its own libamdhip64, and the streams / device
        # pointers we are handed belong to THAT runtime.  Loading libmavlm.so before torch would bind it to the
        # system copy (two runtimes in one process -> hipErrorNoDevice on the first launch).
        import torch  # noqa: F401
        l = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.mavlm_abi_version() != 3:
            raise MavlmError("libmavlm.so ABI version mismatch - rebuild")
        _lib = l
    return _lib


_ERR = {E_ARG: "bad argument (null pointer / size / alignment)", E_SHAPE: "shape not supported by the gfx950 kernels",
        E_STATE: "weights/buffers not bound or empty cache"}


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc < 0:
        raise MavlmError(f"{what}: {_ERR.get(rc, rc)}")
    raise MavlmError(f"{what}: hipError_t {rc}")
