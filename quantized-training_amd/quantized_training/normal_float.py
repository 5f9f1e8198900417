"""NormalFloat code books (NF4 and relatives): 2^k levels at evenly spaced quantiles of N(0, 1), normalised
to [-1, 1] (upstream src/quantized_training/normal_float.py:4-62).  Host-side: the code book only feeds the
65 536-entry value map; the kernels then treat `nfK` like any other table dtype."""
import numpy as np
import torch

__all__ = ["create_normal_map", "quantize_to_nf"]


def _linspace_f32(start, end, steps):
    """torch.linspace in float32: first half from the start, second half from the end."""
    step = (np.float32(end) - np.float32(start)) / np.float32(steps - 1)
    i = np.arange(steps, dtype=np.float32)
    lo = np.float32(start) + step * i
    hi = np.float32(end) - step * (np.float32(steps - 1) - i)
    return np.where(np.arange(steps) < steps // 2, lo, hi).astype(np.float32)


def create_normal_map(offset=0.9677083, use_extra_value=True, k=4):
    """Sorted float32 tensor of the 2^k levels (asymmetric: one more positive level when use_extra_value)."""
    try:
        from scipy.stats import norm
    except ImportError as ie:  # pragma: no cover
        raise ImportError("Scipy is required for `create_normal_map`.") from ie
    half = 2 ** (k - 1)
    neg = -norm.ppf(_linspace_f32(offset, 0.5, half)[:-1])
    if use_extra_value:
        pos = norm.ppf(_linspace_f32(offset, 0.5, half + 1)[:-1])
        zeros = [0.0]
    else:
        pos = norm.ppf(_linspace_f32(offset, 0.5, half)[:-1])
        zeros = [0.0, 0.0]
    levels = np.sort(np.concatenate([pos, zeros, neg]).astype(np.float32))
    levels = levels / levels.max()
    assert levels.size == 2 ** k
    return torch.from_numpy(levels.astype(np.float32))


def quantize_to_nf(input: torch.Tensor, k: int = 4, use_extra_value=True, int_bits=None):
    """(indices, values): index of the nearest level for every element and the level table in the
    input's dtype; with `int_bits` the levels are scaled to integers of that width first."""
    values = create_normal_map(k=k, use_extra_value=use_extra_value)
    if int_bits is not None:
        values = torch.round(values * (2 ** (int_bits - 1) - 1))
    values = values.to(device=input.device, dtype=input.dtype)
    x = torch.clamp(input, min=values.amin(), max=values.amax())
    indices = torch.argmin(torch.abs(values - x.unsqueeze(-1)), dim=-1)
    return indices, values
