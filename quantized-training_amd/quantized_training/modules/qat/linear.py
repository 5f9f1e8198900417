"""Linear layer whose weight is fake-quantized on every forward
(upstream src/quantized_training/modules/qat/linear.py:15-81)."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.parametrize import (
    is_parametrized,
    transfer_parametrizations_and_params,
    type_before_parametrizations,
)

__all__ = ["Linear"]


GEMM_ROUTES = {}          # "train:<kind> MxNxK" -> "in_tree_bf16_gemm" | "library_bf16_gemm" (fused.routes_report)


def train_gemm_enabled():
    import os
    return os.environ.get("QT_TRAIN_GEMM", "1") != "0"


def train_gemm_or_none(a, b, bias, trans_a, trans_b, kind):
    """C = op(a) . op(b) (+ bias) through qt_train_gemm_bf16 (csrc/qt_train_gemm.hip), or None when the kernel does not take the problem
    (the caller then runs torch's GEMM).  a: [M, K] or, trans_a, [K, M]; b: [N, K] (a Linear weight) or, trans_b, [K, N]."""
    import ctypes
    from ... import _native
    if not (train_gemm_enabled() and a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2
            and a.stride(1) == 1 and b.stride(1) == 1):
        return None
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N = b.shape[1] if trans_b else b.shape[0]
    if (b.shape[0] if trans_b else b.shape[1]) != K:
        return None
    skinny = not trans_a and not trans_b and (N < 8 or N % 8 != 0) and M * N <= 4096          # a classifier head: one wave per output, deterministic
    ok = (a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
          and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.numel() == N
                                and bias.data_ptr() % (2 if skinny else 8) == 0))
          and ((K >= 8 and K % 8 == 0) if skinny else (K >= 256 and K % 64 == 0 and M % 8 == 0 and N % 8 == 0 and M >= 8 and N >= 8)))
    key = f"train:{kind} {M}x{N}x{K}"
    if not ok:
        GEMM_ROUTES.setdefault(key, "library_bf16_gemm")
        return None
    c = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    prob = (_native.QtGemmProblem * 1)()
    prob[0].a, prob[0].b, prob[0].c = a.data_ptr(), b.data_ptr(), c.data_ptr()
    prob[0].bias = bias.data_ptr() if bias is not None else None
    _native.note_device(a.device.index)
    rc = _native.lib().qt_train_gemm_bf16(prob, 1, int(trans_a), int(trans_b), M, N, K, a.stride(0), b.stride(0), N,
                                          ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
    if rc in (_native.QT_ERR_BAD_ARG, _native.QT_ERR_UNALIGNED):
        GEMM_ROUTES.setdefault(key, "library_bf16_gemm")
        return None
    _native.check(rc, "qt_train_gemm_bf16")
    GEMM_ROUTES.setdefault(key, "in_tree_bf16_gemm")
    return c


def train_gemm_group(As, Bs, trans_a, trans_b, kind, biases=None, outs=None):
    """Up to four products of ONE shape as one launch of qt_train_gemm_bf16 (query / key / value: the three forward products, the three
    input gradients, the three weight gradients); returns the list of results (`outs` when given), or None when the kernel does not take
    the problems."""
    import ctypes
    from ... import _native
    n = len(As)
    if not (train_gemm_enabled() and 1 <= n <= 4 and len(Bs) == n):
        return None
    if biases is not None and any(b is not None and not (b.dtype == torch.bfloat16 and b.is_contiguous() and b.data_ptr() % 8 == 0) for b in biases):
        return None
    a0, b0 = As[0], Bs[0]
    for a, b in zip(As, Bs):
        if not (a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2 and b.dim() == 2 and a.is_contiguous()
                and b.is_contiguous() and a.shape == a0.shape and b.shape == b0.shape and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0):
            return None
    M, K = (a0.shape[1], a0.shape[0]) if trans_a else (a0.shape[0], a0.shape[1])
    N = b0.shape[1] if trans_b else b0.shape[0]
    if (b0.shape[0] if trans_b else b0.shape[1]) != K or not (K >= 256 and K % 64 == 0 and M % 8 == 0 and N % 8 == 0 and M >= 8 and N >= 8):
        return None
    if biases is not None and any(b is not None and b.numel() != N for b in biases):
        return None
    Cs = outs if outs is not None else [torch.empty((M, N), dtype=torch.bfloat16, device=a0.device) for _ in range(n)]
    prob = (_native.QtGemmProblem * n)()
    for i in range(n):
        prob[i].a, prob[i].b, prob[i].c = As[i].data_ptr(), Bs[i].data_ptr(), Cs[i].data_ptr()
        prob[i].bias = biases[i].data_ptr() if biases is not None and biases[i] is not None else None
    _native.note_device(a0.device.index)
    rc = _native.lib().qt_train_gemm_bf16(prob, n, int(trans_a), int(trans_b), M, N, K, a0.stride(0), b0.stride(0), N,
                                          ctypes.c_void_p(torch.cuda.current_stream(a0.device).cuda_stream))
    if rc in (_native.QT_ERR_BAD_ARG, _native.QT_ERR_UNALIGNED):
        return None
    _native.check(rc, "qt_train_gemm_bf16")
    GEMM_ROUTES.setdefault(f"train:{kind} {n}x({M}x{N}x{K})", "in_tree_bf16_gemm, one launch")
    return Cs


# ---- the forward products of query / key / value as one launch ---------------------------------------------------------------------------
# Hugging Face's self-attention block calls the three projections one after the other on the same hidden states and only reshapes their
# results before it hands them to the attention function.  A projection that train_fusions marked as such a member (the fused training
# attention engaged behind it on an earlier step) returns its output tensor UNWRITTEN -- only while that block's forward is running
# (train_fusions.attention_block_active) -- and leaves the product here; the third member's forward -- or, whatever came in between, the
# attention function's entry, the end of the block's forward, the next other Linear, any backward -- launches what is pending: three
# problems of one shape as ONE launch of qt_train_gemm_bf16 (17 us against 3 x 8-11), otherwise one by one.
_FWD_PENDING = []         # (x2, wq, bias, y2)


def flush_forward():
    if not _FWD_PENDING:
        return
    todo = list(_FWD_PENDING)
    _FWD_PENDING.clear()
    from ... import train_fusions
    if len(todo) == 3 and train_gemm_group([t[0] for t in todo], [t[1] for t in todo], False, False, "forward q/k/v", [t[2] for t in todo],
                                           [t[3] for t in todo]) is not None:
        train_fusions.STATS.qkv_forward_groups += 1
        return
    for x2, w, b, y2 in todo:
        if train_gemm_group([x2], [w], False, False, "forward (deferred)", [b], [y2]) is None:
            y2.copy_(F.linear(x2, w, b))


def _defer_forward(x2, w, b):
    """The output tensor of a q / k / v member, its product left pending; None when the grouped kernel would not take it."""
    M, K = x2.shape
    N = w.shape[0]
    if not (train_gemm_enabled() and x2.is_contiguous() and w.is_contiguous() and w.shape[1] == K and K >= 256 and K % 64 == 0 and M % 8 == 0
            and N % 8 == 0 and M >= 8 and N >= 8 and x2.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and (b is None or (b.dtype == torch.bfloat16 and b.is_contiguous() and b.numel() == N and b.data_ptr() % 8 == 0))):
        return None
    if _FWD_PENDING and (_FWD_PENDING[0][0].shape != x2.shape or _FWD_PENDING[0][1].shape != w.shape):
        flush_forward()
    y2 = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
    _FWD_PENDING.append((x2, w, b, y2))
    if len(_FWD_PENDING) == 3:
        flush_forward()
    return y2


def train_gemm_backward(gys, ws, xs, kind):
    """Both backward products of up to four Linears of ONE shape -- gx_i = gy_i . Wq_i and gW_i = gy_i^T . x_i -- as ONE launch
    (qt_train_gemm_backward_bf16: the weight gradients' tiles and the input gradients' tiles share the chip); returns (gxs, gws) or None
    when the kernel does not take the problems (the caller then issues the products one by one).  Bit for bit the single launches' results."""
    import ctypes
    from ... import _native
    n = len(gys)
    if not (train_gemm_enabled() and 1 <= n <= 4 and len(ws) == n and len(xs) == n):
        return None
    from ... import train_fusions
    if not train_fusions._on("pairgemm"):
        return None
    g0, w0, x0 = gys[0], ws[0], xs[0]
    for g, w, x in zip(gys, ws, xs):
        for t, t0 in ((g, g0), (w, w0), (x, x0)):
            if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.is_contiguous() and t.shape == t0.shape and t.data_ptr() % 16 == 0):
                return None
    T, O = g0.shape
    I = w0.shape[1]
    if w0.shape[0] != O or tuple(x0.shape) != (T, I) or not (T >= 256 and T % 64 == 0 and O >= 256 and O % 64 == 0 and I >= 8 and I % 8 == 0):
        return None
    gxs = [torch.empty((T, I), dtype=torch.bfloat16, device=g0.device) for _ in range(n)]
    gws = [torch.empty((O, I), dtype=torch.bfloat16, device=g0.device) for _ in range(n)]
    items = (_native.QtLinearBackward * n)()
    for i in range(n):
        items[i].gy, items[i].wq, items[i].x, items[i].gx, items[i].gw = (gys[i].data_ptr(), ws[i].data_ptr(), xs[i].data_ptr(), gxs[i].data_ptr(),
                                                                          gws[i].data_ptr())
    _native.note_device(g0.device.index)
    rc = _native.lib().qt_train_gemm_backward_bf16(items, n, T, O, I, O, I, I, I, I, ctypes.c_void_p(torch.cuda.current_stream(g0.device).cuda_stream))
    if rc in (_native.QT_ERR_BAD_ARG, _native.QT_ERR_UNALIGNED):
        return None
    _native.check(rc, "qt_train_gemm_backward_bf16")
    GEMM_ROUTES.setdefault(f"train:dgrad + wgrad {kind}{n}x({T}x{O}x{I})" if n > 1 else f"train:dgrad + wgrad {kind}{T}x{O}x{I}", "in_tree_bf16_gemm, one launch")
    return gxs, gws


class _LinearColsumBias(torch.autograd.Function):
    """F.linear whose backward computes the bias gradient -- grad_output.sum(0), on the gradient the backward-pre hook already
    fake-quantized (quantize.py:116-179) -- with qt_colsum_bf16 (fp32 sums in a fixed order, one rounding) instead of torch's generic
    reduction: 12.6 -> ~5 us per Linear inside the replayed training step.  Input and weight gradients are the two GEMMs autograd's own
    linear backward runs (grad_output . W and grad_output^T . x)."""

    @staticmethod
    def forward(ctx, x, w, b, defer=False):
        ctx.save_for_backward(x, w)
        x2 = x.reshape(-1, x.shape[-1])
        if defer:
            y = _defer_forward(x2, w, b)
            if y is not None:
                return y.view(*x.shape[:-1], w.shape[0])
        flush_forward()
        y = train_gemm_or_none(x2, w, b, False, False, "forward") if x2.is_contiguous() and w.is_contiguous() else None
        if y is not None:
            return y.view(*x.shape[:-1], w.shape[0])
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        import ctypes
        from ... import _native
        flush_forward()
        x, w = ctx.saved_tensors
        gy2 = gy.reshape(-1, gy.shape[-1])
        gx = gw = gb = None
        x2 = x.reshape(-1, x.shape[-1])
        gyc = gy2 if gy2.is_contiguous() else None            # (a permuted view: torch's GEMMs take it as it is)
        from ... import train_fusions
        done = train_fusions.take_linear_grads(gy, x, w)      # query / key / value: the attention backward launched all six products as two
        if done is not None:
            gx, gw = done
            if not ctx.needs_input_grad[0]:
                gx = None
            if not ctx.needs_input_grad[1]:
                gw = None
        if done is None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and gyc is not None and w.is_contiguous() and x2.is_contiguous():
            both = train_gemm_backward([gyc], [w], [x2], "")      # the two products in one launch
            if both is not None:
                done = (both[0][0].view(x.shape), both[1][0])
                gx, gw = done
        if done is None and ctx.needs_input_grad[0]:
            # gx = gy . Wq: the weight is read with its rows as the contraction index (trans_b)
            gx = train_gemm_or_none(gyc, w, None, False, True, "dgrad") if gyc is not None and w.is_contiguous() else None
            gx = gx.view(x.shape) if gx is not None else gy2.mm(w).view(x.shape)
        if done is None and ctx.needs_input_grad[1]:
            # gW = gy^T . x: both operands are read with the token index as the contraction index (trans_a, trans_b)
            gw = train_gemm_or_none(gyc, x2, None, True, True, "wgrad") if gyc is not None and x2.is_contiguous() else None
            if gw is None:
                gw = gy2.t().mm(x2)
        if ctx.needs_input_grad[2]:
            from ... import train_fusions
            gb = train_fusions.take_colsum(gy)          # the launch that fake-quantized this gradient summed its columns on the way
            if gb is not None:
                return gx, gw, gb, None
            train_fusions.STATS.colsum_fallbacks += 1
            g = gy2 if gy2.is_contiguous() else gy2.contiguous()
            if g.dtype != torch.bfloat16 or g.data_ptr() % 16 or g.shape[1] % 8 or g.shape[0] == 0:
                gb = gy2.sum(0)                         # a view at an odd storage offset, another dtype: what F.linear's backward does
            else:
                gb = torch.empty(g.shape[1], dtype=g.dtype, device=g.device)
                _native.note_device(g.device.index)
                rc = _native.lib().qt_colsum_bf16(g.data_ptr(), gb.data_ptr(), g.shape[0], g.shape[1],
                                                  ctypes.c_void_p(torch.cuda.current_stream(g.device).cuda_stream))
                if rc in (_native.QT_ERR_BAD_ARG, _native.QT_ERR_UNALIGNED):
                    gb = gy2.sum(0)
                else:
                    _native.check(rc, "qt_colsum_bf16")
        return gx, gw, gb, None


class Linear(nn.Linear):
    """``F.linear(x, weight_fake_quant(W), b)``; shares ``weight`` / ``bias`` Parameters with the
    float module it was made from."""

    _FLOAT_MODULE = nn.Linear

    def __init__(self, in_features, out_features, bias=True, qconfig=None, device=None, dtype=None) -> None:
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__(in_features, out_features, bias, **factory_kwargs)
        assert qconfig, "qconfig must be provided for QAT module"
        self.qconfig = qconfig
        # the fake-quantizer's buffers are real even when the layer itself is built on `meta`
        fq_device = None if device is not None and str(device) == "meta" else device
        self.weight_fake_quant = qconfig.weight(factory_kwargs={"device": fq_device, "dtype": dtype})

    def forward(self, input):
        from ...fused import cached_weight, fused_linear_or_none
        out = fused_linear_or_none(self, input)
        if out is not None:
            return out
        # (opt-in) in eval with a frozen / stateless weight fake-quantizer the quantized weight is kept, see fused.py
        wq = cached_weight(self, "dense", lambda: self.weight_fake_quant(self.weight))
        b = self.bias
        if (torch.is_grad_enabled() and input.is_cuda and input.dtype == torch.bfloat16 and wq.dtype == torch.bfloat16
                and (b is None or b.dtype == torch.bfloat16) and (input.requires_grad or wq.requires_grad)):
            # training on the device: the three products on the in-tree GEMM (csrc/qt_train_gemm.hip), the bias gradient through
            # qt_colsum_bf16 or the chain launch that fake-quantized grad_output
            # what the backward will multiply with, for train_fusions.group_qkv_backward -- weak references: the autograd node keeps both
            # alive until its backward has run, and nothing here may extend the life of a step's tensors (or of its autograd graph)
            import weakref
            self.__dict__["_qt_train_xw"] = (weakref.ref(input), weakref.ref(wq))
            defer = False
            owner = self.__dict__.get("_qt_qkv_member")
            if owner is not None and not self._forward_hooks:
                from ... import train_fusions
                defer = train_fusions.attention_block_active(owner) and train_fusions._on("qkvfwd")
            return _LinearColsumBias.apply(input, wq, b, defer)
        return F.linear(input, wq, b)

    @classmethod
    def from_float(cls, mod):
        """The QAT twin of a float ``nn.Linear`` that carries a ``qconfig``; weight and bias stay the SAME Parameter
        objects (or parametrizations) as the float module's."""
        kind = type_before_parametrizations(mod)
        assert kind == cls._FLOAT_MODULE, f" qat.{cls.__name__}.from_float only works for {cls._FLOAT_MODULE.__name__}"
        assert getattr(mod, "qconfig", None), "Input float module must have a valid qconfig"
        twin = cls(mod.in_features, mod.out_features, bias=mod.bias is not None, qconfig=mod.qconfig,
                   device="meta")                      # parameters are adopted from `mod` right below
        for name in ("weight", "bias"):
            if is_parametrized(mod, name):
                transfer_parametrizations_and_params(mod, twin, name)
            else:
                setattr(twin, name, getattr(mod, name))
        return twin

    def to_float(self):
        """A plain ``nn.Linear`` holding detached copies of the current parameters."""
        plain = torch.nn.Linear(self.in_features, self.out_features, bias=self.bias is not None, device="meta")
        plain.weight = torch.nn.Parameter(self.weight.detach())
        plain.bias = None if self.bias is None else torch.nn.Parameter(self.bias.detach())
        return plain.train(self.training)
