"""Launch fusions of the TRAINING step (BASELINE configs[4]: int8 activations + weights with delayed scaling, E5M2 gradients through the
backward hooks -- quantize.py:116-179, fake_quantize.py:217-246, run_glue_no_trainer.py:647-667 upstream).

The reference's hooks call fake-quantizers back to back on the same tensor.  A gradient that leaves a LayerNorm goes through the
residual add's backward-pre quantizer, its two backward quantizers and the dense layer's backward-pre quantizer; a LayerNorm's output
goes through the input quantizers of query / key / value.  Every call is a launch (8 us observed, >= 4.5 us whatever it does inside a
replayed hipGraph), the step has ~280 of them, plus 73 column-sum launches for the bias gradients.

A CHAIN is a static list of fake-quantizer calls that the module structure guarantees to follow each other on known tensors: when the
chain's HEAD is called, qt_fake_quant_chain_bf16 evaluates every member in one launch -- each with its own scale and amax slot, bit
for bit what its own launch would compute -- and leaves each member's result as a one-shot hand-over.  The hooks still run as the
reference's do; a member's call checks that it received exactly the tensor the chain predicted (storage, version, shape) and hands the
result out, counted once.  Should it receive anything else, it re-zeroes its amax slot (the chain's speculative contribution) and
computes as usual.  Chains are planned from the modules' structure (`plan`), never from a trace.

Nothing here changes a value: the launches are the same functions of the same inputs; the bias gradient's fp32 column sums are added
in another (fixed) order than qt_colsum_bf16's."""
import ctypes
import os

import torch

from . import _native

__all__ = ["plan", "unplan", "enabled", "STATS"]


class _Counters:
    chains = 0            # chain launches
    members = 0           # fake-quantizer calls served by a chain launch (heads included)
    colsums = 0           # bias gradients handed over
    misses = 0            # members that received another tensor than predicted

    @classmethod
    def reset(cls):
        cls.chains = cls.members = cls.colsums = cls.misses = 0


STATS = _Counters


def enabled():
    return os.environ.get("QT_TRAIN_CHAINS", "1") != "0"


class Chain:
    """members: [(fake-quantizer, src)] -- src -1: the head's input, else the index of the member whose result it reads.
    colsum: (member index, Linear) -- that member's result is the Linear's grad_output: its column sums are the bias gradient."""

    def __init__(self, members, colsum=None, name=""):
        self.members = members
        self.colsum = colsum
        self.name = name


_COLSUM = {}              # (data_ptr, version, shape) of a grad_output -> its column sums (one-shot, taken by the Linear's backward)


def take_colsum(g):
    hit = _COLSUM.pop((g.data_ptr(), g._version, tuple(g.shape)), None)
    if hit is not None:
        STATS.colsums += 1
    return hit


def _member_ok(fq, device):
    from .fake_quantize import FusedAmaxObsFakeQuantize
    from .quantizer.quantizer import QScheme
    return (isinstance(fq, FusedAmaxObsFakeQuantize) and fq._quantize and not fq.is_per_channel and not fq.record_histogram
            and fq.outlier_threshold is None and fq.qscheme in (None, QScheme.PER_TENSOR_SYMMETRIC)
            and not getattr(fq, "_emit_fp8", None)      # (hooks on a member still see its call, input and result: forward() runs for every member)
            and (not fq._observe or (fq.amax_history.numel() > 0 and fq.amax_history.dim() == 1 and fq.amax_history.device == device))
            and fq.scale.numel() == 1 and fq.scale.device == device and fq.scale.dtype == torch.float32)


class _ChainFn(torch.autograd.Function):
    """The chain's results with the straight-through gradient of every member (fake_quantize.py:250-252 upstream): the gradient of x
    is the sum of the gradients of the results that were used, added in member order."""

    @staticmethod
    def forward(ctx, x, run):
        outs = run(x)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        total = None
        for g in grads:
            if g is not None:
                total = g if total is None else total + g
        return total, None


_MAX = {}
_SCRATCH = {}


def _format_max(fq):
    """Largest finite magnitude of the fake-quantizer's value map (bounds the fixed-point range of the column sums)."""
    key = str(fq.dtype)
    hit = _MAX.get(key)
    if hit is None:
        import numpy as np
        m = _native.build_map_u16(fq.dtype).astype(np.uint32) << 16
        v = np.abs(m.view(np.float32))
        hit = _MAX[key] = float(v[np.isfinite(v)].max())
    return hit


def _chain_scratch(nbytes, device):
    """Zeroed accumulators + tickets of the column sums, per (device, stream, capture) like every scratch launches share (fused.splitk_scratch);
    every launch leaves them zero."""
    from .fused import _capture_id
    st = torch.cuda.current_stream(device).cuda_stream
    key = (device.index if device.index is not None else torch.cuda.current_device(), st, _capture_id(ctypes.c_void_p(st)))
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is None and len(_SCRATCH) > 8:
            for k in [k for k in _SCRATCH if k[2] not in (0, key[2])]:
                del _SCRATCH[k]
        buf = _SCRATCH[key] = torch.zeros(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
    return buf


def run_chain(head, chain, X):
    """Evaluates every member of `chain` on X in one launch; returns the head's result or None (the head then runs alone)."""
    from .fake_quantize import _launch_format, _stream_ptr, launch_scale_update, _Stats, _take_preupdate
    if not (enabled() and X.is_cuda and X.dtype == torch.bfloat16 and X.is_contiguous() and X.dim() >= 2 and X.numel() > 0
            and X.data_ptr() % 16 == 0):
        return None
    cols = X.shape[-1]
    rows = X.numel() // cols
    if cols % 8:
        return None
    dev = X.device
    members = chain.members
    if members[0][0] is not head or not all(_member_ok(fq, dev) for fq, _ in members):
        return None
    for fq, _ in members:
        fq._move_to(dev)
    fmt0 = _launch_format(head._qt_format, head.qmap)
    if any(_launch_format(fq._qt_format, fq.qmap).key() != fmt0.key() or str(fq.dtype) != str(head.dtype) for fq, _ in members[1:]):
        return None
    if fmt0.kind == _native.QT_FMT_LUT and not (fmt0.p1 & 1):
        return None
    if fmt0.kind not in (_native.QT_FMT_LUT, _native.QT_FMT_FP_SAT, _native.QT_FMT_INT):
        return None
    need_grad = torch.is_grad_enabled() and X.requires_grad
    colsum = chain.colsum if os.environ.get("QT_TRAIN_COLSUM", "1") != "0" else None
    if colsum is not None and (need_grad or colsum[1].bias is None or not colsum[1].bias.requires_grad or colsum[1].out_features != cols):
        colsum = None
    st = _stream_ptr(X)
    L = _native.lib()

    def launch(x):
        outs = [torch.empty_like(x) for _ in members]
        stages = (_native.QtChainStage * len(members))()
        for i, (fq, src) in enumerate(members):
            if fq._observe:
                launch_scale_update(fq.amax_history, fq.scale, fq.quant_max, fq.force_scale_power_of_two, st)
            stages[i].scale_f32_dev = fq.scale.data_ptr()
            stages[i].amax_bits_dev = fq.amax_history.data_ptr() if fq._observe else None
            stages[i].out_dev = outs[i].data_ptr()
            stages[i].src = src
        gb = ws = None
        if colsum is not None:
            ws = _chain_scratch(L.qt_fake_quant_chain_ws_bytes(rows, cols), dev)
            gb = torch.empty(cols, dtype=torch.bfloat16, device=dev)
        rc = L.qt_fake_quant_chain_bf16(x.data_ptr(), rows, cols, stages, len(members), ctypes.byref(fmt0),
                                        head.qmap.data_ptr() if fmt0.kind == _native.QT_FMT_LUT else None,
                                        colsum[0] if colsum is not None else -1, _format_max(head), gb.data_ptr() if gb is not None else None,
                                        ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, st)
        _native.check(rc, "qt_fake_quant_chain_bf16")
        if gb is not None:
            g = outs[colsum[0]]
            if len(_COLSUM) > 64:
                _COLSUM.clear()
            _COLSUM[(g.data_ptr(), g._version, tuple(g.shape))] = gb
        return outs

    outs = _ChainFn.apply(X, launch) if need_grad else launch(X)
    STATS.chains += 1
    STATS.members += 1
    _Stats.add(X.numel())                                      # the head's own call
    for i, (fq, src) in enumerate(members):
        if i == 0:
            continue
        want = X if src < 0 else outs[src]
        fq.__dict__["_qt_chain_result"] = (want.data_ptr(), want._version, tuple(want.shape), outs[i], want)
    return outs[0]


def take_member_result(fq, X):
    """Called at the top of a fake-quantizer's forward: the result a chain launch left for THIS call, or None.  A call that received
    another tensor than the chain predicted drops the speculative amax and proceeds on its own."""
    pend = fq.__dict__.get("_qt_chain_result")
    if pend is None:
        return None
    fq.__dict__["_qt_chain_result"] = None
    ptr, version, shape, out, _keep = pend
    from .fake_quantize import _Stats, _take_preupdate
    if X.data_ptr() == ptr and X._version == version and tuple(X.shape) == shape and X.is_contiguous():
        STATS.members += 1
        _Stats.add(X.numel())
        if fq._observe:
            _take_preupdate(fq.amax_history)                   # the batched scale update (or the chain's own) served this call
        return out
    STATS.misses += 1
    if fq._observe and fq.amax_history.numel() > 0:
        # the chain has already rolled this quantizer's history for the call and added an amax that belongs to no call: start the slot again
        from .fake_quantize import _PREUPDATED
        fq.amax_history[0].zero_()
        _PREUPDATED.add(fq.amax_history.data_ptr())            # ... and do not roll a second time
    return None


def _fq(holder, key):
    return holder[key] if holder is not None and key in holder else None


def ensure_planned(model):
    """plan(model) whenever the set of fake-quantizers has changed since the last plan (they are created lazily by the first step)."""
    from .fake_quantize import FusedAmaxObsFakeQuantize
    count = sum(1 for m in model.modules() if isinstance(m, FusedAmaxObsFakeQuantize))
    if model.__dict__.get("_qt_train_plan") != (count, enabled()):
        model.__dict__["_qt_train_plan"] = (count, enabled())
        return plan(model)
    return None


def unplan(model):
    from .fake_quantize import FusedAmaxObsFakeQuantize
    for m in model.modules():
        if isinstance(m, FusedAmaxObsFakeQuantize):
            m.__dict__.pop("_qt_chain", None)
            m.__dict__.pop("_qt_chain_result", None)


def plan(model):
    """Attaches chains to the fake-quantizers of `model` from its module structure; returns how many.  Call it once the lazily
    created fake-quantizers exist (after a first training step); idempotent.
      * output blocks `LayerNorm(residual(dropout(dense(h)), x))` (modules/quantizable/attention.py::_bert_output_forward; upstream
        modeling_bert.py:174-214) with inactive dropout, backward: residual.error_pre_process[0] -> residual.error_post_process[0], [1]
        -> dense.error_pre_process[0] (+ the dense layer's bias gradient);
      * every other QAT Linear with a backward-pre quantizer: that call + the bias gradient;
      * attention blocks whose query / key / value read one tensor, forward: the three input quantizers."""
    from .modules.qat.linear import Linear as QATLinear
    unplan(model)
    if not enabled():
        return 0
    n = 0
    chained = set()
    for mod in model.modules():
        dense, res, ln = getattr(mod, "dense", None), getattr(mod, "residual", None), getattr(mod, "LayerNorm", None)
        if isinstance(dense, QATLinear) and res is not None and ln is not None and getattr(type(mod), "_qt_twin", False):
            drop = getattr(mod, "dropout", None)
            if drop is not None and getattr(drop, "p", 0.0) != 0.0:
                continue
            pre = _fq(getattr(res, "error_pre_process", None), "0")
            p0, p1 = _fq(getattr(res, "error_post_process", None), "0"), _fq(getattr(res, "error_post_process", None), "1")
            dpre = _fq(getattr(dense, "error_pre_process", None), "0")
            if pre is None or len(getattr(res, "error_pre_process", {})) != 1:
                continue
            members = [(pre, -1)]
            colsum = None
            if p0 is not None and p1 is not None and len(res.error_post_process) == 2:
                members += [(p0, 0), (p1, 0)]
                if dpre is not None and len(dense.error_pre_process) == 1:
                    members.append((dpre, 1))
                    colsum = (3, dense)
            if len(members) > 1:
                pre.__dict__["_qt_chain"] = Chain(members, colsum, name=type(mod).__name__)
                chained.update(id(f) for f, _ in members)
                n += 1
    for mod in model.modules():
        q, k, v = getattr(mod, "query", None), getattr(mod, "key", None), getattr(mod, "value", None)
        if isinstance(q, QATLinear) and isinstance(k, QATLinear) and isinstance(v, QATLinear) and hasattr(mod, "qk_matmul"):
            fqs = [_fq(getattr(l, "activation_pre_process", None), "0") for l in (q, k, v)]
            if all(f is not None and id(f) not in chained for f in fqs) and all(len(l.activation_pre_process) == 1 for l in (q, k, v)):
                fqs[0].__dict__["_qt_chain"] = Chain([(fqs[0], -1), (fqs[1], -1), (fqs[2], -1)], name="qkv inputs")
                chained.update(id(f) for f in fqs)
                n += 1
    for mod in model.modules():
        if isinstance(mod, QATLinear):
            dpre = _fq(getattr(mod, "error_pre_process", None), "0")
            if dpre is not None and id(dpre) not in chained and len(mod.error_pre_process) == 1 and mod.bias is not None:
                dpre.__dict__["_qt_chain"] = Chain([(dpre, -1)], (0, mod), name="grad_output + bias gradient")
                chained.add(id(dpre))
                n += 1
    return n
