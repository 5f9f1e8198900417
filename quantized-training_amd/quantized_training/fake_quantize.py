"""Fake-quantize module and autograd function on the MI355X HIP engine.

Same surface as the reference's ``fake_quantize.py`` (upstream
src/quantized_training/fake_quantize.py): ``get_quantization_map`` (:31-95),
``FusedAmaxObsFakeQuantFunction`` (:197-252), ``FusedAmaxObsFakeQuantize`` (:255-435,
identical buffer names / persistence so reference checkpoints load) and
``_DerivedObserverOrFakeQuantize`` (:438-474).

What is different underneath:
  * the value map is built by ``qt_build_map`` (C++, csrc/qt_host.cpp) once per (dtype, device)
    and shared by every instance instead of 128 KiB per instance;
  * one call = ``qt_scale_update`` (delayed-scaling state machine, on device) followed by ONE fused
    HIP pass that observes amax and writes the fake-quantized tensor -- no host synchronisation
    (the reference reads ``observer_enabled[0]`` / ``fake_quant_enabled[0]`` on the host twice per call);
  * device tensors ALWAYS go through libqt_hip.so; a missing library raises (``_native.lib``).
    CPU tensors (host-side plumbing, e.g. the MobileBERT-tiny CPU config) use the same formulas
    written with torch ops.
"""
import ctypes
import logging
import os
from typing import Optional

import torch
from torch.ao.quantization import FakeQuantizeBase

from . import _native
from .quantizer.quantizer import QScheme

__all__ = [
    "FusedAmaxObsFakeQuantize",
    "_DerivedObserverOrFakeQuantize",
    "get_quantization_map",
]

logger = logging.getLogger(__name__)

class _Stats:
    """Counts fake-quant applications the way the metric 'quantized elements/s' does
    (sum of numel(input) over fake-quant calls)."""
    elements = 0
    calls = 0

    @classmethod
    def reset(cls):
        cls.elements = 0
        cls.calls = 0

    @classmethod
    def add(cls, numel):
        cls.elements += int(numel)
        cls.calls += 1


STATS = _Stats

import re as _re

_RE_NF = _re.compile(r"nf(\d+)(?:_(\d+))?")

_MAP_CACHE = {}      # (dtype, device) -> bf16 tensor [65536]
_FORMAT_CACHE = {}   # dtype -> _native.QtFormat


def _format_for(dtype):
    f = _FORMAT_CACHE.get(dtype)
    if f is None:
        if isinstance(dtype, str) and _RE_NF.fullmatch(dtype):
            f = _native.QtFormat(_native.QT_FMT_LUT, 0, 0, 0.0, 0.0)       # code-book dtype: table only
        else:
            f = _native.format_for(dtype)
        _FORMAT_CACHE[dtype] = f
    return f


def _table_for(dtype, device):
    """The [65536] bf16 value map as one tensor (NormalFloat's (indices, values) pair is flattened)."""
    m = get_quantization_map(dtype, device)
    if isinstance(m, tuple):
        key = (dtype, "flat", torch.device(device) if device is not None else torch.device("cpu"))
        flat = _MAP_CACHE.get(key)
        if flat is None:
            idx, vals = m
            flat = vals[idx].contiguous()
            flat._qt_dtype = dtype
            _MAP_CACHE[key] = flat
        return flat
    return m


def get_quantization_map(dtype, device=None):
    """65 536-entry bf16 -> bf16 value map of ``dtype`` (upstream fake_quantize.py:31-95).

    ``table[bits(v)]`` is the nearest representable value of ``dtype`` to the bf16 value ``v``.
    Raises ``ValueError`` for an unknown dtype.  The returned tensor is shared: do not modify it.
    """
    dev = torch.device(device) if device is not None else torch.device("cpu")
    nf = _RE_NF.fullmatch(dtype) if isinstance(dtype, str) else None
    if nf:      # NormalFloat: (indices, values), as upstream returns for this family (fake_quantize.py:90-93)
        key = (dtype, dev)
        hit = _MAP_CACHE.get(key)
        if hit is None:
            from .normal_float import quantize_to_nf
            patterns = torch.arange(2 ** 16, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
            idx, vals = quantize_to_nf(patterns, int(nf.group(1)), int_bits=int(nf.group(2)) if nf.group(2) else None)
            hit = (idx.to(dev), vals.to(dev))
            _MAP_CACHE[key] = hit
        return hit
    key = (dtype, dev)
    hit = _MAP_CACHE.get(key)
    if hit is None:
        host = _MAP_CACHE.get((dtype, torch.device("cpu")))
        if host is None:
            table = _native.build_map_u16(dtype)          # raises ValueError on unknown dtype
            host = torch.from_numpy(table.view("int16")).view(torch.bfloat16)
            host._qt_dtype = dtype
            _MAP_CACHE[(dtype, torch.device("cpu"))] = host
        hit = host if dev.type == "cpu" else _device_map(host, dev)
        hit._qt_dtype = dtype    # lets the block-scaled GEMMs recognise the element format of values rounded with this map
        _MAP_CACHE[key] = hit
    return hit


QT_MAP_ENTRIES = _native.QT_MAP_ENTRIES
_ROWS_TAIL = 4096            # bf16 slots behind the map: 512 rows x 16 bytes


def _device_map(host, dev):
    """The map on a device, with the ROW FORM of the map (qt_build_rowparams) in the same allocation right behind its 65 536 entries
    where the row form covers the value range tensors live in: the elementwise kernels then need no 128 KiB table in LDS.  The
    returned tensor is the [65536] view; `_qt_rows` holds the qt_format.p1 bits that tell the kernels the tail is there."""
    import numpy as np
    m = host.view(torch.int16).numpy().view(np.uint16)
    rp = _native.build_rowparams(m)
    flagged = np.ctypeslib.as_array(rp.flagged)
    lo, hi = 127 - 30, 127 + 15      # flagged rows out there are rare enough for the kernel's per-element detour through the map
    usable = (rp.zero_sign in (0, 1) and not flagged[lo:hi].any() and not (rp.signed_rows and flagged[256 + lo:256 + hi].any()))
    if not usable:
        return host.to(dev)
    both = torch.empty(QT_MAP_ENTRIES + _ROWS_TAIL, dtype=torch.bfloat16)
    both[:QT_MAP_ENTRIES] = host
    rows = np.ctypeslib.as_array(rp.row).reshape(2048).astype(np.uint32)
    both[QT_MAP_ENTRIES:].view(torch.int16).copy_(torch.from_numpy(rows.view(np.int16).copy()))
    out = both.to(dev)[:QT_MAP_ENTRIES]
    out._qt_rows = 1 | (2 if rp.signed_rows else 0) | (4 if rp.sign_mask else 0) | (8 if rp.zero_sign == 1 else 0)
    rule = 0x8000 if (rp.zero_sign == 1 and rp.sign_mask) else 0x0000      # what the row form makes of the input -0.0
    out._qt_neg_zero = None
    if int(m[0x8000]) != rule:
        out._qt_rows |= 16
        out._qt_neg_zero = float(np.array([int(m[0x8000]) << 16], dtype=np.uint32).view(np.float32)[0])
    return out


def _launch_format(fmt, lut):
    """`fmt` with the row-form bits when `lut` is a map allocated by _device_map (and still that very allocation)."""
    bits = getattr(lut, "_qt_rows", 0) if lut is not None else 0
    if not bits or fmt.kind != _native.QT_FMT_LUT or fmt.p1:
        return fmt
    if lut.storage_offset() != 0 or lut.untyped_storage().nbytes() < 2 * (QT_MAP_ENTRIES + _ROWS_TAIL):
        return fmt
    return _native.QtFormat(fmt.kind, fmt.p0, bits, fmt.flo, lut._qt_neg_zero if bits & 16 else fmt.fhi)


# ----------------------------------------------------------------------------------------------
# device plumbing
# ----------------------------------------------------------------------------------------------
def handover_valid(t):
    """Producer kernels hand results to their consumers through Python attributes on the tensor (`_qt_fq_done_by`,
    `_qt_fp8`, ...), stamped with the tensor's version counter (`_qt_ver`): an in-place modification in between (a user
    forward hook, `add_`) makes them stale, and the consumer then does its own pass."""
    return getattr(t, "_qt_ver", None) == t._version


_LAZY = {}          # data_ptr -> weakref of a tensor whose values were not written (its FP8 codes were): views of it lose the attribute


def note_lazy(t):
    """Registers t (model_fusions._mark_lazy, rope_fq) so that a VIEW of it -- a reshape between the producer and the consuming hook
    drops Python attributes -- is still recognised by materialize_lazy.  The entry dies with the tensor."""
    import weakref
    ptr = t.data_ptr()
    _LAZY[ptr] = weakref.ref(t)
    weakref.finalize(t, lambda p=ptr: _LAZY.pop(p, None) if (_LAZY.get(p) is not None and _LAZY[p]() is None) else None)


def materialize_lazy(t):
    """A producer that knew its consumer multiplies FP8 codes wrote ONLY the codes of fq(t) (t._qt_lazy; model_fusions.rope_fq).  The
    fake-quantized values are exactly what the codes decode to, so whoever asks for them after all gets them here."""
    if t.__dict__.get("_qt_lazy", False):
        t.copy_(t._qt_fp8.to(t.dtype))
        t._qt_lazy = False
        t._qt_ver = t._version
        return
    if _LAZY:
        ref = _LAZY.get(t.data_ptr())
        base = ref() if ref is not None else None
        if (base is not None and base is not t and base.__dict__.get("_qt_lazy", False) and base.device == t.device and base.dtype == t.dtype
                and base.numel() == t.numel() and t.is_contiguous()):
            # a view of a lazy tensor (same storage, same extent): decode through the owner
            stamped = getattr(t, "_qt_ver", None) == t._version
            materialize_lazy(base)
            if stamped:
                t._qt_ver = t._version


def _stream_ptr(t):
    """Current stream of the tensor's device for the native call that follows; that call runs with the tensor's device
    current (`_native.note_device`), so `model.to("cuda:1")` or a `dispatch_model` placement needs no `set_device`."""
    _native.note_device(t.device.index)
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _is_device(t):
    return t.device.type == "cuda"


def _as_flag(v):
    """observer_enabled / fake_quant_enabled arrive as python bools from the module (no sync) or as
    the uint8[1] buffers when the function is called the upstream way (one host read)."""
    if isinstance(v, torch.Tensor):
        return bool(v.reshape(-1)[0].item())
    return bool(v)


# ---- batched delayed-scaling update (harness.GraphedTrainStep) --------------------------------------------------------
_PREUPDATED = set()       # data_ptr of every amax history whose NEXT call's scale update has already been done


def _take_preupdate(amax_history) -> bool:
    p = amax_history.data_ptr()
    if p in _PREUPDATED:
        _PREUPDATED.discard(p)
        return True
    return False


def launch_scale_update(amax_history, scale, quant_max, pow2, stream_ptr):
    """The delayed-scaling update in front of one observed call -- unless BatchedScaleUpdate already did it."""
    if _take_preupdate(amax_history):
        return
    _native.check(_native.lib().qt_scale_update(amax_history.data_ptr(), int(amax_history.shape[0]), int(scale.numel()),
                                                scale.data_ptr(), float(quant_max), int(bool(pow2)), stream_ptr),
                  "qt_scale_update")


class BatchedScaleUpdate:
    """The delayed-scaling update (amax = max(history), roll, scale = amax / quant_max) of many fake-quantizers as ONE
    launch (qt_scale_update_multi), done before the step that will call each of them; their next call then skips its own
    update.  Only fake-quantizers that observe, already hold a history and live on `device` take part."""

    def __init__(self, fake_quantizers, device):
        fqs = [f for f in fake_quantizers
               if getattr(f, "_observe", False) and f.amax_history.numel() > 0 and f.amax_history.device == device
               and f.amax_history.dtype == torch.float32 and f.scale.dtype == torch.float32]
        self.fqs = fqs
        i64 = dict(dtype=torch.int64, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.hist = torch.tensor([f.amax_history.data_ptr() for f in fqs], **i64)
        self.scale = torch.tensor([f.scale.data_ptr() for f in fqs], **i64)
        self.L = torch.tensor([int(f.amax_history.shape[0]) for f in fqs], **i32)
        self.C = torch.tensor([int(f.scale.numel()) for f in fqs], **i32)
        self.qmax = torch.tensor([float(f.quant_max) for f in fqs], dtype=torch.float32, device=device)
        self.pow2 = torch.tensor([int(bool(f.force_scale_power_of_two)) for f in fqs], **i32)
        self.device = device

    def launch(self):
        if not self.fqs:
            return
        _native.check(_native.lib().qt_scale_update_multi(self.hist.data_ptr(), self.L.data_ptr(), self.C.data_ptr(),
                                                          self.scale.data_ptr(), self.qmax.data_ptr(), self.pow2.data_ptr(),
                                                          len(self.fqs), _stream_ptr(self.hist)), "qt_scale_update_multi")
        for f in self.fqs:
            _PREUPDATED.add(f.amax_history.data_ptr())

    def forget(self):
        for f in self.fqs:
            _PREUPDATED.discard(f.amax_history.data_ptr())


class _PrecomputedFakeQuant(torch.autograd.Function):
    """fq(X) that was already computed (BatchedWeightFakeQuant): hands `out` on with the straight-through gradient of
    FusedAmaxObsFakeQuantFunction (fake_quantize.py:250-252 upstream)."""

    @staticmethod
    def forward(ctx, X, out):
        return out.view_as(out)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None


class BatchedWeightFakeQuant:
    """Every per-tensor weight fake-quantizer of a step as ONE launch per format (qt_fake_quant_multi_bf16) in front of the step
    (harness.GraphedTrainStep, after BatchedScaleUpdate): a weight does not change between the start of a step and the optimizer update
    at its end, so `weight_fake_quant(W)` of every QAT Linear (modules/qat/linear.py:40-41) can run first -- each with its own scale
    and amax slot, i.e. the per-tensor state machine is the reference's.  The call the Linear then issues finds its result
    (`_qt_pre`: valid for the very next call, and only for that very weight at that very version), counts its elements and returns
    it with the straight-through gradient.  `pairs`: (fake-quantizer, weight Parameter)."""

    def __init__(self, pairs, device):
        groups = {}
        for fq, W in pairs:
            if not (isinstance(fq, FusedAmaxObsFakeQuantize) and fq._quantize and not fq.is_per_channel and not fq.record_histogram
                    and fq.outlier_threshold is None and fq.qscheme in (None, QScheme.PER_TENSOR_SYMMETRIC)):
                continue
            if not (W.device == device and W.dtype == torch.bfloat16 and W.is_contiguous() and W.numel() % 8 == 0 and W.numel() > 0
                    and W.data_ptr() % 16 == 0 and fq.scale.numel() == 1 and fq.scale.dtype == torch.float32 and fq.scale.device == device):
                continue
            if fq._observe and (fq.amax_history.numel() == 0 or fq.amax_history.dim() != 1 or fq.amax_history.device != device):
                continue
            fq._move_to(device)
            fmt = _launch_format(fq._qt_format, fq.qmap)
            if fmt.kind == _native.QT_FMT_LUT and not (fmt.p1 & 1):
                continue
            if fmt.kind not in (_native.QT_FMT_LUT, _native.QT_FMT_FP_SAT, _native.QT_FMT_INT):
                continue
            lut = fq.qmap if fmt.kind == _native.QT_FMT_LUT else None
            groups.setdefault((fmt.key(), lut.data_ptr() if lut is not None else 0), (fmt, lut, []))[2].append((fq, W))
        self.groups = []
        for fmt, lut, members in groups.values():
            total = sum(W.numel() for _, W in members)
            out = torch.empty(total, dtype=torch.bfloat16, device=device)
            rows, outs, tiles, off = [], [], 0, 0
            for fq, W in members:
                y = out[off:off + W.numel()].view(W.shape)
                off += W.numel()
                nvec = W.numel() // 8
                rows.append([W.data_ptr(), y.data_ptr(), fq.scale.data_ptr(), fq.amax_history.data_ptr() if fq._observe else 0, nvec, tiles])
                tiles += (nvec + 1023) // 1024
                outs.append(y)
            items = torch.tensor(rows, dtype=torch.int64, device=device)
            self.groups.append((fmt, lut, members, outs, items, tiles))
        self.device = device

    def __len__(self):
        return sum(len(g[2]) for g in self.groups)

    def launch(self):
        for fmt, lut, members, outs, items, tiles in self.groups:
            _native.check(_native.lib().qt_fake_quant_multi_bf16(items.data_ptr(), len(members), tiles, ctypes.byref(fmt),
                                                                 lut.data_ptr() if lut is not None else None, _stream_ptr(items)),
                          "qt_fake_quant_multi_bf16")
            for (fq, W), y in zip(members, outs):
                fq.__dict__["_qt_pre"] = (W.data_ptr(), W._version, y)

    def forget(self):
        for _, _, members, _, _, _ in self.groups:
            for fq, _ in members:
                fq.__dict__["_qt_pre"] = None


def _rows_view(t):
    """(view, transposed) when `t` is a non-contiguous bf16 device tensor whose rows are contiguous:
    either its last dim has stride 1, or its last two dims are a transposed pair (K^T)."""
    if t.is_contiguous() or t.dtype != torch.bfloat16 or t.dim() < 2 or t.dim() > 4 or t.numel() == 0:
        return None
    v, transposed = t, False
    if v.stride(-1) != 1:
        if v.stride(-2) != 1:
            return None
        v, transposed = v.transpose(-1, -2), True
    if v.shape[-1] % 8 != 0 or any(st % 8 != 0 for st in v.stride()[:-1]) or v.data_ptr() % 16 != 0:
        return None
    return v, transposed


def _forward_rows(input, rows, observe, qmap, amax_history, scale, quant_max, pow2, fmt):
    v, transposed = rows
    L = _native.lib()
    st = _stream_ptr(v)
    fmt = _launch_format(fmt if fmt is not None else _native.QtFormat(_native.QT_FMT_LUT, 0, 0, 0.0, 0.0), qmap)
    if observe:
        launch_scale_update(amax_history, scale, quant_max, pow2, st)
    y = torch.empty(v.shape, dtype=v.dtype, device=v.device)
    shp = [1] * (4 - v.dim()) + list(v.shape)
    strd = [0] * (4 - v.dim()) + list(v.stride())
    _native.check(L.qt_fake_quant_rows_bf16(v.data_ptr(), y.data_ptr(), shp[0], shp[1], shp[2], shp[3],
                                            strd[0], strd[1], strd[2], ctypes.byref(fmt),
                                            qmap.data_ptr() if qmap is not None else None, scale.data_ptr(),
                                            amax_history.data_ptr() if observe else None, st), "qt_fake_quant_rows_bf16")
    return y.transpose(-1, -2) if transposed else y


def _channel_view(shape, ch_axis):
    ax = ch_axis + len(shape) if ch_axis < 0 else ch_axis
    outer = 1
    for d in shape[:ax]:
        outer *= d
    inner = 1
    for d in shape[ax + 1:]:
        inner *= d
    return outer, shape[ax], inner


def _hip_fake_quant(x, y, fmt, lut, scale, amax_hist, per_channel, ch_axis):
    """One fused HIP pass.  y may be None (observe only); amax_hist may be None (observer off)."""
    L = _native.lib()
    n = x.numel()
    if n == 0:
        return
    bf16 = x.dtype == torch.bfloat16
    xp, yp = x.data_ptr(), (y.data_ptr() if y is not None else None)
    lp = lut.data_ptr() if lut is not None else None
    sp = scale.data_ptr() if scale is not None else None
    ap = amax_hist.data_ptr() if amax_hist is not None else None     # slot 0 of the history
    st = _stream_ptr(x)
    if per_channel:
        outer, C, inner = _channel_view(tuple(x.shape), ch_axis)
        fn = L.qt_fake_quant_pc_bf16 if bf16 else L.qt_fake_quant_pc_f32
        _native.check(fn(xp, yp, outer, C, inner, ctypes.byref(_launch_format(fmt, lut)), lp, sp, ap, st), "qt_fake_quant_pc")
    else:
        fn = L.qt_fake_quant_bf16 if bf16 else L.qt_fake_quant_f32
        _native.check(fn(xp, yp, n, ctypes.byref(_launch_format(fmt, lut)), lp, sp, ap, st), "qt_fake_quant")


def hip_vmap(x, qmap, fmt=None):
    """quantized_ops::vmap on a device tensor."""
    L = _native.lib()
    x = x.contiguous()
    y = torch.empty_like(x)
    if x.numel() == 0:
        return y
    if fmt is None:
        fmt = _native.QtFormat(_native.QT_FMT_LUT, 0, 0, 0.0, 0.0)
    lp = qmap.data_ptr() if qmap is not None else None
    if x.dtype == torch.bfloat16:
        fn = L.qt_vmap_bf16
    elif x.dtype == torch.float32:
        fn = L.qt_vmap_f32
    elif x.dtype == torch.float16:
        fn = L.qt_vmap_f16
    else:
        return hip_vmap(x.float(), qmap, fmt).to(x.dtype)
    _native.check(fn(x.data_ptr(), y.data_ptr(), x.numel(), ctypes.byref(fmt), lp, _stream_ptr(x)), "qt_vmap")
    return y


# ----------------------------------------------------------------------------------------------
# CPU-tensor formulas (host plumbing; the one device use is _cpu_vmap's gather for non-bf16 index tables, decomposed.py)
# ----------------------------------------------------------------------------------------------
def _cpu_vmap(x, qmap):
    if x.dtype == torch.bfloat16:
        idx = x.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
    else:
        raw = x.to(torch.float32).contiguous().view(torch.int32)
        idx = ((raw >> 16) & 0xFFFF) | ((raw & 0xFFFF) != 0).to(torch.int32)
    return qmap[idx.reshape(-1).long()].to(x.dtype).reshape(x.shape)


def _state_update_cpu(amax_cur, amax_history, scale, quant_max, pow2):
    amax = torch.amax(amax_history, dim=0)
    if amax_history.shape[0] > 1:
        amax_history.copy_(torch.roll(amax_history, -1, 0))
    amax_history[0] = amax_cur
    sf = amax / quant_max
    sf = torch.where(amax > 0.0, sf, scale)
    sf = torch.where(torch.isfinite(amax), sf, scale)
    if pow2:
        sf = torch.pow(2, torch.ceil(torch.log2(sf)))
    scale.copy_(sf)


class FusedAmaxObsFakeQuantFunction(torch.autograd.Function):
    """Observe amax with delayed scaling, then fake-quantize (upstream fake_quantize.py:197-252).

    Argument list is the reference's plus one trailing optional ``qt_format`` (closed-form
    descriptor of ``qmap``; ``None`` = use the table).  Backward is the straight-through estimator.
    """

    @staticmethod
    def forward(ctx, input, observer_enabled, fake_quant_enabled, qmap, amax_history, scale,
                amax_history_len, quant_max, ch_axis=None, per_row_fake_quant=False,
                force_scale_power_of_two=False, qt_format=None, emit_fp8=None):
        observe = _as_flag(observer_enabled)
        quantize = _as_flag(fake_quant_enabled)
        if not observe and not quantize:
            return input

        if observe and amax_history.numel() == 0:                      # upstream :225-228
            if per_row_fake_quant:
                ax = ch_axis + input.ndim if ch_axis < 0 else ch_axis
                size = tuple(d if i == ax else 1 for i, d in enumerate(input.shape))
            else:
                size = ()
            amax_history.resize_((amax_history_len,) + size).fill_(0.0)
            scale.resize_(size).fill_(1.0)

        if not _is_device(input):
            return _forward_cpu(input, observe, quantize, qmap, amax_history, scale, quant_max, ch_axis,
                                per_row_fake_quant, force_scale_power_of_two)

        # Permuted attention views (q / k^T / v): the pass itself writes the canonical contiguous layout
        # (what the reference's vmap returns, decomposed.py:155) instead of a .contiguous() copy first.
        rows = None if (per_row_fake_quant or emit_fp8 is not None) else _rows_view(input)
        if rows is not None and quantize:
            return _forward_rows(input, rows, observe, qmap, amax_history, scale, quant_max,
                                 force_scale_power_of_two, qt_format)
        x = input.contiguous()
        if x.dtype not in (torch.bfloat16, torch.float32):
            return _forward_other_dtype(x, observe, quantize, qmap, amax_history, scale, quant_max, ch_axis,
                                        per_row_fake_quant, force_scale_power_of_two, qt_format)
        L = _native.lib()
        fmt = qt_format if qt_format is not None else _native.QtFormat(_native.QT_FMT_LUT, 0, 0, 0.0, 0.0)
        if observe:
            launch_scale_update(amax_history, scale, quant_max, force_scale_power_of_two, _stream_ptr(x))
        if (emit_fp8 is not None and quantize and not per_row_fake_quant and x.dtype == torch.bfloat16
                and x.numel() % 16 == 0 and x.numel() > 0):
            # one pass: bf16 fake-quantized tensor (unless emit_fp8 == "only") + the same values as FP8 bytes
            only = emit_fp8 == "only"
            y = None if only else torch.empty_like(x)
            y8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
            _native.check(L.qt_fake_quant_bf16_fp8(
                x.data_ptr(), y.data_ptr() if y is not None else None, y8.data_ptr(), x.numel(), ctypes.byref(fmt),
                scale.data_ptr(), amax_history.data_ptr() if observe else None, _stream_ptr(x)), "qt_fake_quant_bf16_fp8")
            y8 = y8.view(torch.float8_e5m2 if fmt.p0 == 2 else torch.float8_e4m3fn)
            if only:
                return y8
            ctx.mark_non_differentiable(y8)
            return y, y8            # the module attaches y8 to y for the QAT Linear that owns the hook (fused.py)
        y = torch.empty_like(x) if quantize else None
        per_channel = bool(per_row_fake_quant)
        if per_channel and scale.numel() != _channel_view(tuple(x.shape), ch_axis)[1]:
            if observe or scale.numel() != 1:
                raise ValueError(f"per-channel scale has {scale.numel()} entries, tensor has "
                                 f"{_channel_view(tuple(x.shape), ch_axis)[1]} channels on axis {ch_axis}")
            per_channel = False                                         # single broadcast scale
        _hip_fake_quant(x, y, fmt, qmap, scale, amax_history if observe else None, per_channel, ch_axis)
        return y if quantize else input

    @staticmethod
    def backward(ctx, grad_output, *unused):
        return (grad_output,) + (None,) * 12


class MXFakeQuantFunction(torch.autograd.Function):
    """Microscaling fake-quant: one scale per block of `block_size` elements along `axes`, from the block's
    amax (or its shared exponent), then `qmap[x / s] * s` (upstream fake_quantize.py:98-133).  The fused HIP
    pass (qt_fake_quant_mx_*) covers blocks along the last axis of a contiguous bf16 / fp32 device tensor;
    other layouts use the same formulas as torch ops with the table lookup on the HIP vmap kernel."""

    @staticmethod
    def forward(ctx, input, fake_quant_enabled, scale, qmap, axes, block_size, quant_max,
                force_scale_power_of_two=False, scale_qmap=None, qt_format=None):
        if not _as_flag(fake_quant_enabled):
            return input
        from .decomposed import expand, quantize_mx
        out = _hip_mx_or_none(input, scale, qmap, axes, block_size, quant_max, force_scale_power_of_two,
                              scale_qmap, qt_format)
        if out is not None:
            return out
        sf, q = quantize_mx(input, qmap, axes, block_size, quant_max, force_scale_power_of_two, scale_qmap=scale_qmap)
        scale.resize_(sf.shape).copy_(sf)
        return q * expand(sf, q.shape, block_size)

    @staticmethod
    def backward(ctx, grad_output):
        return (grad_output,) + (None,) * 9


def _hip_mx_or_none(input, scale, qmap, axes, block_size, quant_max, pow2, scale_qmap, fmt):
    """Fused device pass for the common layout; None when it does not apply."""
    if not _is_device(input) or input.dtype not in (torch.bfloat16, torch.float32) or input.dim() < 1:
        return None
    ax = axes if isinstance(axes, int) else (axes[0] if len(axes) == 1 else None)
    if ax is None or (ax % input.dim()) != input.dim() - 1:
        return None
    if not isinstance(block_size, int) or block_size not in (8, 16, 32, 64, 128) or pow2:
        return None
    cols = input.shape[-1]
    if cols % block_size != 0 or cols == 0 or input.numel() == 0:
        return None
    L = _native.lib()
    x = input.contiguous()
    y = torch.empty_like(x)
    nblk = cols // block_size
    sf = torch.empty(x.shape[:-1] + (nblk,), dtype=x.dtype, device=x.device)
    fmt = fmt if fmt is not None else _native.QtFormat(_native.QT_FMT_LUT, 0, 0, 0.0, 0.0)
    fn = L.qt_fake_quant_mx_bf16 if x.dtype == torch.bfloat16 else L.qt_fake_quant_mx_f32
    _native.check(fn(x.data_ptr(), y.data_ptr(), sf.data_ptr(), x.numel() // cols, cols, block_size,
                     ctypes.byref(fmt), qmap.data_ptr() if qmap is not None else None, float(quant_max),
                     scale_qmap.data_ptr() if scale_qmap is not None else None, _stream_ptr(x)), "qt_fake_quant_mx")
    scale.resize_(sf.shape).copy_(sf)
    return y


class GroupWiseAffineFakeQuantFunction(torch.autograd.Function):
    """Asymmetric block-wise integer fake-quant with per-block scale and zero point
    (upstream fake_quantize.py:136-194); plain torch ops in the tensor's dtype."""

    @staticmethod
    def forward(ctx, input, fake_quant_enabled, scale, zero_point, axes, block_size, quant_min, quant_max,
                scale_qmap=None):
        if not _as_flag(fake_quant_enabled):
            return input
        from .decomposed import expand, _const_like
        from .mx_utils import _reshape_to_blocks
        assert block_size > 0
        axes = [axes] if type(axes) == int else list(axes)
        axes = [x + input.ndim if x < 0 else x for x in axes]
        blocks, baxes, _, _ = _reshape_to_blocks(input, axes, block_size)
        block_axes = [x + 1 for x in baxes]
        lo = torch.amin(blocks, dim=block_axes)
        hi = torch.amax(blocks, dim=block_axes)
        sf = (hi - lo) / _const_like(quant_max - quant_min, hi)
        sf = torch.where(sf > 0.0, sf, _const_like(1.0, sf))
        zp = -lo / sf + quant_min
        if scale_qmap is not None:
            sf = torch.ops.quantized_ops.vmap(sf, scale_qmap)
            zp = torch.ops.quantized_ops.vmap(zp, scale_qmap)
        scale.resize_(sf.shape).copy_(sf)
        zero_point.resize_(zp.shape).copy_(zp)
        sfe = expand(sf, input.shape, block_size)
        zpe = expand(zp, input.shape, block_size)
        q = torch.clamp(torch.round(input / sfe + zpe), quant_min, quant_max)
        return (q - zpe) * sfe

    @staticmethod
    def backward(ctx, grad_output):
        return (grad_output,) + (None,) * 8


def _forward_cpu(input, observe, quantize, qmap, amax_history, scale, quant_max, ch_axis, per_channel, pow2):
    if observe:
        if per_channel:
            ax = ch_axis + input.ndim if ch_axis < 0 else ch_axis
            dims = tuple(i for i in range(input.ndim) if i != ax)
            amax_cur = torch.amax(torch.abs(input), dim=dims, keepdim=True)
        else:
            amax_cur = torch.amax(torch.abs(input))
        _state_update_cpu(amax_cur, amax_history, scale, quant_max, pow2)
    if quantize:
        s = scale.to(input.dtype)
        input = _cpu_vmap(input / s, qmap) * s
    return input


def _forward_other_dtype(x, observe, quantize, qmap, amax_history, scale, quant_max, ch_axis, per_channel, pow2, fmt):
    """fp16 (or other) device tensors: the arithmetic around the table stays in the tensor's dtype
    exactly as upstream; the table lookup itself is the HIP vmap kernel."""
    if observe:
        if per_channel:
            ax = ch_axis + x.ndim if ch_axis < 0 else ch_axis
            dims = tuple(i for i in range(x.ndim) if i != ax)
            amax_cur = torch.amax(torch.abs(x), dim=dims, keepdim=True)
        else:
            amax_cur = torch.amax(torch.abs(x))
        _state_update_cpu(amax_cur.float(), amax_history, scale, quant_max, pow2)
    if quantize:
        s = scale.to(x.dtype)
        x = hip_vmap(x / s, qmap, fmt) * s
    return x


class FusedAmaxObsFakeQuantize(FakeQuantizeBase):
    r"""Simulates quantize + dequantize of a tensor to ``dtype`` with amax-history (delayed) scaling.

    Constructor arguments, buffers (``fake_quant_enabled``, ``observer_enabled``, ``amax_history``,
    ``scale``, ``zero_point`` persistent; ``qmap``, ``scale_qmap``, ``histogram`` not) and
    ``extra_repr`` follow upstream fake_quantize.py:255-341.
    """

    qmap: torch.Tensor
    scale_qmap: Optional[torch.Tensor]
    amax_history: torch.Tensor
    scale: torch.Tensor
    zero_point: torch.Tensor

    def __init__(self, dtype: str, qscheme=None, quant_min: Optional[float] = None,
                 quant_max: Optional[float] = None, amax_history_len: int = None,
                 ch_axis: Optional[int] = None, block_size: Optional[int] = None,
                 record_histogram: bool = False, scale_dtype: Optional[str] = None,
                 force_scale_power_of_two: bool = False, outlier_threshold: Optional[float] = None,
                 **kwargs) -> None:
        super().__init__()
        if isinstance(qscheme, str):
            qscheme = QScheme(qscheme)
        self.dtype = dtype
        self.qscheme = qscheme
        self.quant_min = quant_min
        self.quant_max = quant_max
        self.amax_history_len = amax_history_len
        self.ch_axis = ch_axis
        self.block_size = block_size
        self.scale_dtype = scale_dtype
        self.force_scale_power_of_two = force_scale_power_of_two
        self.outlier_threshold = outlier_threshold
        device = kwargs.get("device", None)
        if device is None and isinstance(kwargs.get("factory_kwargs"), dict):
            device = kwargs["factory_kwargs"].get("device", None)
        self._qt_format = _format_for(dtype)                                  # raises ValueError on a bad dtype
        self.register_buffer("qmap", _table_for(dtype, device), persistent=False)
        scale_map = _table_for(scale_dtype, device) if scale_dtype is not None else None
        self.register_buffer("scale_qmap", scale_map, persistent=False)
        fk = {"device": device, "dtype": torch.float}
        self.register_buffer("amax_history", torch.tensor([], **fk))
        self.register_buffer("scale", torch.tensor([1.0], **fk))
        self.register_buffer("zero_point", torch.tensor([1.0], **fk))
        self.is_per_channel = self.qscheme == QScheme.PER_CHANNEL_SYMMETRIC
        self.record_histogram = record_histogram
        self.register_buffer("histogram", torch.zeros(254, **fk), persistent=False)
        # host mirrors of the uint8[1] enable buffers: forward() never reads the device copies
        self._observe = False
        self._quantize = True
        self._emit_fp8 = None          # set by fused.py when the consumer is an FP8-capable GEMM
        self.enable_observer(self.qscheme is not None)

    # -- enable flags: keep the buffers (state_dict compatibility) and the host mirrors in step --
    def enable_fake_quant(self, enabled: bool = True) -> None:
        self.fake_quant_enabled[0] = 1 if enabled else 0
        self._quantize = bool(enabled)

    def disable_fake_quant(self):
        self.enable_fake_quant(False)

    def enable_observer(self, enabled: bool = True) -> None:
        self.observer_enabled[0] = 1 if enabled else 0
        self._observe = bool(enabled)

    def disable_observer(self):
        self.enable_observer(False)

    def fp8_exact(self):
        """True when every fake-quantized value is an OCP FP8 value AND the scale is identically 1
        (specs without `qs`), i.e. fq(x) can be handed to an FP8 GEMM without changing any product."""
        return (self._qt_format.kind == _native.QT_FMT_FP_SAT and self.qscheme is None
                and self.outlier_threshold is None and not self._observe and self._quantize
                and getattr(self, "_scale_is_one", True))

    def stateless_map(self):
        """True when this fake-quantizer is a pure function of its input: the value map of `dtype` at scale 1 (specs without
        `qs`), quantize on, observer off -- what a GEMM may apply to its weight operand on the fly (fused.fqt_linear_or_none)."""
        return (self._quantize and not self._observe and self.qscheme is None and not self.is_per_channel
                and not self.record_histogram and self.outlier_threshold is None and getattr(self, "_scale_is_one", True))

    def map_producer_format(self, device):
        """(format with the row-form bits, device map) when a producing kernel may apply this fake-quantizer on the consumer's behalf in
        its ROW FORM (csrc/qt_model_ops.hip, *_map_* entry points): a stateless table format whose device map carries the row words
        (_device_map); else None."""
        if not (self.stateless_map() and self._qt_format.kind == _native.QT_FMT_LUT):
            return None
        self._move_to(device)
        fmt = _launch_format(self._qt_format, self.qmap)
        return (fmt, self.qmap) if (fmt.p1 & 1) else None

    def sync_flags_from_buffers(self):
        """Re-read the enable buffers (one host sync); call after writing them directly."""
        self._observe = bool(self.observer_enabled[0].item())
        self._quantize = bool(self.fake_quant_enabled[0].item())

    @torch.jit.export
    def calculate_qparams(self):
        if self.qscheme == QScheme.GROUP_WISE_AFFINE:
            return self.scale, self.zero_point
        return self.scale

    @torch.jit.export
    def extra_repr(self):
        return (
            "fake_quant_enabled={}, observer_enabled={}, dtype={}, amax_history_len={}, "
            "quant_max={}, qscheme={}, ch_axis={}, block_size={}, force_scale_power_of_two={}, "
            "scale={}".format(
                self.fake_quant_enabled, self.observer_enabled, self.dtype, self.amax_history_len,
                self.quant_max, self.qscheme, self.ch_axis, self.block_size,
                self.force_scale_power_of_two, self.scale,
            )
        )

    def _move_to(self, device):
        if self.scale.device != device or self.amax_history.device != device:
            self.to(device)
        # `model.to(device)` leaves a plain copy of the host map in the buffer: take the cached device map instead, which carries the
        # row form behind its entries (_device_map) and the `_qt_dtype` tag the block-scaled GEMMs look for
        if self.qmap.device != device or (device.type == "cuda" and getattr(self.qmap, "_qt_dtype", None) is None):
            self.qmap = _table_for(self.dtype, device)
            if self.scale_qmap is not None:
                self.scale_qmap = _table_for(self.scale_dtype, device)

    def expect_prequantized(self, tensor, x8, replacement=None):
        """A producing kernel computed fq(tensor) for this fake-quantizer's NEXT call; see forward().  Without
        `replacement` the quantized values were written into `tensor` itself (FP8 code in x8); with it `tensor` holds
        the unquantized values (it has other readers, e.g. a residual connection) and `replacement` = fq(tensor) as
        bf16.  The tensor is kept referenced until that call so that its storage cannot be recycled in between."""
        self.__dict__["_qt_expected"] = (tensor.data_ptr(), tensor.numel(), tensor._version, x8, replacement, tensor)

    def producer_fusable(self) -> bool:
        """True when this fake-quantizer is a pure stateless function a producing kernel may apply on its behalf
        (model_fusions.py): quantize on, observer off, per-tensor, unit scale, closed-form E4M3 / E5M2."""
        return (self._quantize and not self._observe and self.qscheme is None and not self.is_per_channel
                and not self.record_histogram and self.outlier_threshold is None and self.fp8_exact())

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        self.__dict__["_qt_calls"] = self.__dict__.get("_qt_calls", 0) + 1      # harness.GraphedTrainStep reads this
        if "_qt_deferred" in self.__dict__:
            # training step: this backward quantizer's call is evaluated by the fan-in launch of the node its result goes to (train_fusions.py)
            from . import train_fusions
            out = train_fusions.take_deferred(self, X)
            if out is not None:
                return out
        if "_qt_chain_result" in self.__dict__ or "_qt_chain" in self.__dict__:
            # training-step chains (train_fusions.py): a chain launch already evaluated this call, or this call heads a chain
            from . import train_fusions
            out = train_fusions.take_member_result(self, X)
            if out is not None:
                return out
            chain = self.__dict__.get("_qt_chain")
            if chain is not None:
                out = train_fusions.run_chain(self, chain, X)
                if out is not None:
                    return out
        done_by = getattr(X, "_qt_fq_done_by", None)
        if done_by is self and handover_valid(X):
            # the kernel that produced X already applied this fake-quantizer (and attached X._qt_fp8): the call the
            # reference issues here is satisfied by that fused computation, counted once
            _Stats.add(X.numel())
            if not self.__dict__.get("_qt_lazy_ok"):          # (set by model_fusions.codes_only_ok: the consumer decodes on demand)
                materialize_lazy(X)
            return X
        also = getattr(X, "_qt_also_done", None)
        if also is not None and handover_valid(X):
            # the producing kernel evaluated this fake-quantizer on its result as well (a norm with several consuming Linears,
            # model_fusions._norm_with_consumers): X holds the fake-quantized values (one format for all the consumers), these are this
            # call's own codes
            for fq, x8 in also:
                if fq is self:
                    _Stats.add(X.numel())
                    if X.__dict__.get("_qt_lazy", False) and not self.__dict__.get("_qt_lazy_ok"):
                        materialize_lazy(X)
                    out = X.view(X.shape)
                    out._qt_fp8 = x8
                    out._qt_ver = out._version
                    out._qt_origin = (X.data_ptr(), X._version, tuple(X.shape))
                    if X.__dict__.get("_qt_lazy", False):
                        out._qt_lazy = True                   # the producer wrote the codes only; this consumer decodes on demand
                    return out
        materialize_lazy(X)          # every path below reads X's values: a producer may have written its FP8 codes only
        if (done_by is not None and self._emit_fp8 and isinstance(done_by, FusedAmaxObsFakeQuantize) and handover_valid(X)
                and getattr(X, "_qt_fp8", None) is not None and X.is_cuda and X.dtype == torch.bfloat16 and X.is_contiguous()
                and not (torch.is_grad_enabled() and X.requires_grad) and self.producer_fusable() and done_by.producer_fusable()
                and done_by._qt_format.key() == self._qt_format.key()):
            # X was produced by an identical stateless fake-quantizer (a sibling's: q beside k, v; gate beside up), so it lies on
            # this format's grid, where the fake-quantizer is the identity: the result of this call is X, value for value.  The call
            # is still evaluated -- the pass below computes the FP8 code of every element of X -- but the bf16 tensor it would
            # write is a copy of X and is not written: the result is a view of X carrying the codes just computed.
            self._move_to(X.device)
            _Stats.add(X.numel())
            x8 = FusedAmaxObsFakeQuantFunction.apply(X, False, True, self.qmap, self.amax_history, self.scale, self.amax_history_len,
                                                      self.quant_max, None, False, False, self._qt_format, "only")
            out = X.view(X.shape)
            out._qt_fp8 = x8
            out._qt_ver = out._version
            out._qt_origin = (X.data_ptr(), X._version, tuple(X.shape))
            return out
        expect = self.__dict__.get("_qt_expected")
        if expect is not None:
            # same hand-over when the producer's tensor reaches the hook as a view (a reshape in between drops Python
            # attributes): the producer left the storage it wrote, valid for this -- the very next -- call only
            self.__dict__["_qt_expected"] = None
            ptr, numel, version, x8, replacement, _keep = expect
            if X.data_ptr() == ptr and X.numel() == numel and X._version == version and X.is_contiguous():
                _Stats.add(numel)
                if replacement is not None:
                    out = replacement.view(X.shape)
                    out._qt_fp8 = x8.view(X.shape)
                    out._qt_ver = out._version
                    out._qt_origin = (ptr, version, tuple(X.shape))       # fq(.) of X, for sibling GEMMs
                    if replacement.__dict__.get("_qt_lazy", False):       # the producer wrote the codes only
                        out._qt_lazy = True
                        if not self.__dict__.get("_qt_lazy_ok"):
                            materialize_lazy(out)
                    return out
                X._qt_fp8 = x8
                X._qt_ver = X._version
                return X
        pre = self.__dict__.get("_qt_pre")
        if pre is not None:
            # BatchedWeightFakeQuant computed this call in front of the step (same scale, same amax slot): valid for this -- the very
            # next -- call, on the very tensor it read
            self.__dict__["_qt_pre"] = None
            ptr, version, out = pre
            if X.data_ptr() == ptr and X._version == version and X.shape == out.shape and X.is_contiguous():
                _Stats.add(X.numel())
                if self._observe:
                    _take_preupdate(self.amax_history)                        # the batched scale update served this call
                return _PrecomputedFakeQuant.apply(X, out)
        self._move_to(X.device)

        if self.record_histogram:                                            # upstream :348-350
            exp = torch.floor(torch.log2(torch.abs(X.detach().float())))
            self.histogram += torch.histc(exp, 254, min=-126, max=127)

        if self.outlier_threshold is not None:                              # upstream :353-359
            orig_X = X.clone()
            mask = torch.abs(X) < self.outlier_threshold
            X = torch.where(mask, X, torch.zeros_like(X))
            outlier_pct = mask.bitwise_not().sum().item() / X.numel()
            self.max_outlier_pct = max(outlier_pct, getattr(self, "max_outlier_pct", 0.0))

        if self.qscheme == QScheme.MICROSCALING:
            X = MXFakeQuantFunction.apply(X, self._quantize, self.scale, self.qmap, self.ch_axis, self.block_size,
                                          self.quant_max, self.force_scale_power_of_two, self.scale_qmap,
                                          self._qt_format)
            if self._quantize:
                _Stats.add(X.numel())
        elif self.qscheme == QScheme.GROUP_WISE_AFFINE:
            X = GroupWiseAffineFakeQuantFunction.apply(X, self._quantize, self.scale, self.zero_point, self.ch_axis,
                                                       self.block_size, self.quant_min, self.quant_max, self.scale_qmap)
        if self.qscheme in (QScheme.MICROSCALING, QScheme.GROUP_WISE_AFFINE):
            if self.outlier_threshold is not None:
                X = torch.where(mask, X, orig_X)
            return X

        if not self._observe and not self._quantize:
            if self.outlier_threshold is not None:                          # the restore runs whatever is enabled (upstream :401-402)
                X = torch.where(mask, X, orig_X)
            return X
        _Stats.add(X.numel())
        orig_in = X
        X = FusedAmaxObsFakeQuantFunction.apply(
            X, self._observe, self._quantize, self.qmap, self.amax_history, self.scale,
            self.amax_history_len, self.quant_max, self.ch_axis, self.is_per_channel,
            self.force_scale_power_of_two, self._qt_format,
            self._emit_fp8 if (self._emit_fp8 and self.fp8_exact()) else None,
        )
        if isinstance(X, tuple):
            src = orig_in
            X, x8 = X
            X._qt_fp8 = x8
            X._qt_ver = X._version
            X._qt_origin = (src.data_ptr(), src._version, tuple(src.shape))    # which tensor this is fq(.) of (sibling GEMMs)
        elif X is not orig_in and self.qscheme is None and not self._observe:
            X._qt_origin = (orig_in.data_ptr(), orig_in._version, tuple(orig_in.shape))
            X._qt_ver = X._version

        if self.outlier_threshold is not None:                              # upstream :401-402
            X = torch.where(mask, X, orig_X)
        return X

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict,
                              missing_keys, unexpected_keys, error_msgs):
        # scale / amax_history start empty and are sized by the first observed call; let a
        # checkpoint of any size load into them (upstream :406-435)
        for name in ("scale", "amax_history"):
            key = prefix + name
            if key in state_dict:
                getattr(self, name).resize_(state_dict[key].shape)
            elif strict:
                missing_keys.append(key)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict,
                                      missing_keys, unexpected_keys, error_msgs)
        self.sync_flags_from_buffers()
        self._scale_is_one = bool((self.scale == 1.0).all().item())


class _DerivedObserverOrFakeQuantize(FakeQuantizeBase):
    """Fake-quantize whose scale is derived from other observers (e.g. bias = s_x * s_w),
    upstream fake_quantize.py:438-474."""

    def __init__(self, dtype, obs_or_fqs, derive_qparams_fn):
        super().__init__()
        self.obs_or_fqs = obs_or_fqs
        self.derive_qparams_fn = derive_qparams_fn
        self._qt_format = _format_for(dtype)
        self.register_buffer("qmap", get_quantization_map(dtype), persistent=False)
        self.observer_enabled[0] = 0
        self.dtype = dtype
        self.qscheme = obs_or_fqs[1].qscheme

    def forward(self, x):
        if self.qmap.device != x.device:
            self.to(x.device)
            self.qmap = get_quantization_map(self.dtype, x.device)
        scale = self.calculate_qparams()
        if not isinstance(scale, torch.Tensor):
            scale = scale[0]
        return FusedAmaxObsFakeQuantFunction.apply(
            x, False, bool(self.fake_quant_enabled[0].item()), self.qmap, None,
            scale.to(torch.float32).contiguous(), None, None, None, False, False, self._qt_format)

    def calculate_qparams(self):
        return self.derive_qparams_fn(self.obs_or_fqs)
