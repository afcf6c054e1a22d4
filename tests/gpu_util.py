"""Helpers shared by the -m gpu tests: numpy <-> device tensors on the 16-bit grids, oracle weights -> module."""
import numpy as np
import torch

from oracle import memory_path as O

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def to_dev(a: np.ndarray, mode="bf16"):
    """float32 numpy (values already on the grid, or to be rounded) -> 16-bit device tensor."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda").to(DT[mode])


def to_np(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def f32_dev(a: np.ndarray):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda")


def load_oracle_weights(module_root, w: dict, prefix_map=None):
    """Copy oracle weights (reference state-dict names) into a torch module tree (strict on names)."""
    sd = module_root.state_dict()
    used = 0
    for k, v in w.items():
        if k in sd:
            assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
            sd[k].copy_(torch.from_numpy(np.ascontiguousarray(v)))
            used += 1
    return used
