"""Block (tile) helpers for the block-scaled formats (upstream src/quantized_training/mx_utils.py:19-134)."""
import torch

__all__ = ["_shared_exponents", "_reshape_to_blocks", "_undo_reshape_to_blocks", "FP32_MIN_NORMAL"]

FP32_EXPONENT_BIAS = 127
FP32_MIN_NORMAL = 2 ** (-FP32_EXPONENT_BIAS + 1)


def _shared_exponents(A, method="max", axes=None, ebits=0):
    """floor(log2(max |A|)) over `axes` (keepdim), in A's dtype (upstream mx_utils.py:19-55)."""
    if method == "max":
        if axes is None:
            shared = torch.max(torch.abs(A))
        else:
            shared = A
            for ax in axes:
                shared, _ = torch.max(torch.abs(shared), dim=ax, keepdim=True)
    elif method == "none":
        shared = torch.abs(A)
    else:
        raise Exception("Unrecognized shared exponent selection method %s" % (method))
    shared = torch.floor(torch.log2(shared + FP32_MIN_NORMAL * (shared == 0).type(shared.dtype)))
    if ebits > 0:
        emax = 2 ** (ebits - 1) - 1
        shared[shared > emax] = float("NaN")
        shared[shared < -emax] = -emax
    return shared


def _reshape_to_blocks(A, axes, block_size):
    """[..., n, ...] -> [..., n_blocks, block, ...] along every axis in `axes`, zero-padded to a multiple of
    block_size (upstream mx_utils.py:58-118).  Returns (A, shifted axes, unpadded shape, padded shape)."""
    if axes is None:
        raise Exception("axes required in order to determine which dimension toapply block size to")
    if block_size == 0:
        raise Exception("block_size == 0 in _reshape_to_blocks")
    axes = sorted((x + A.dim() if x < 0 else x) for x in axes)
    assert all(x >= 0 for x in axes)
    for i in range(len(axes)):
        axes[i] += i
        A = torch.unsqueeze(A, dim=axes[i] + 1)
    orig_shape = A.size()
    pad = [0, 0] * len(orig_shape)
    need = False
    for ax in axes:
        rem = orig_shape[ax] % block_size
        if rem:
            pad[2 * ax] = block_size - rem
            need = True
    if need:
        A = torch.nn.functional.pad(A, list(reversed(pad)), mode="constant")
    padded_shape = A.size()
    shape = list(padded_shape)
    for ax in axes:
        if shape[ax] >= block_size:
            assert shape[ax] % block_size == 0
            shape[ax + 1] = block_size
            shape[ax] = shape[ax] // block_size
        else:
            shape[ax + 1] = shape[ax]
            shape[ax] = 1
    return A.view(shape), axes, orig_shape, padded_shape


def _undo_reshape_to_blocks(A, padded_shape, orig_shape, axes):
    A = A.view(padded_shape)
    if list(padded_shape) != list(orig_shape):
        A = A[tuple(slice(0, x) for x in orig_shape)]
    for ax in reversed(axes):
        A = torch.squeeze(A, dim=ax + 1)
    return A
