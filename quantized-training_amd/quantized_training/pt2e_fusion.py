"""Fusion pass for PREPARED PT2E graphs (upstream examples/language_modeling/wikitext.py:60-136).

`prepare_pt2e` leaves a flat aten graph in which every fake-quantizer is a `call_module` node (upstream quantize_pt2e.py:262-273):

    fq_a(x) ; fq_w(W) ; aten.linear          -- Linears, q / k / v reading one fq_a
    silu(linear_g) * linear_u ; fq ; linear  -- the MLP
    to(fp32) pow mean add rsqrt mul to(bf16) mul(weight) ; fq      -- LlamaRMSNorm, decomposed
    add(fq_r(residual), linear) ; fq          -- the residual stream (the annotator quantizes an add's earlier-defined operand)
    add(fq_i(q * cos), rotate_half(q) * sin) ; fq_q ; matmul ; * scale ; + mask ; softmax ; fq_p ; matmul(., fq_v(v))

and running it node by node launches one kernel per node.  `fuse_prepared_graph` rewrites those chains into `call_module` nodes whose
modules drive the same HIP kernels the eager route (`quantize()`, model_fusions.py) uses: qt_linear_fq8_bf16 / qt_mlp_fq8_bf16 /
qt_linear_fqt_bf16 for the Linears (weight fake-quant in the GEMM's operand path, siblings in one launch), the RMSNorm kernel with the
residual add in front and the consumers' fake-quantizers behind it, the rotary kernel, and the FP8 attention core.  Every fused module
has the node sequence it replaces as its fallback (CPU tensors, gradients, other dtypes, formats the kernels do not take), computed
with the same torch calls in the same order -- on the CPU a fused graph is bit-identical to the graph it was made from
(tests/test_pt2e_cpu.py).  Fake-quantizers that a producing kernel evaluates stay in the graph as nodes and hand the result through
(FusedAmaxObsFakeQuantize.forward, `_qt_fq_done_by`); those that vanish into a kernel (weights, the attention core's four, the rotary's
inner two) are counted where they are computed, so `fake_quantize.STATS` reads the same for both graphs.

The pass never changes what is computed, only how many launches it takes; a chain it does not recognise is left alone.
"""
import copy
import ctypes
import operator
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.fx import GraphModule, Node

from . import _native
from .fake_quantize import STATS as _FQ_STATS, FusedAmaxObsFakeQuantize, _stream_ptr
from .modules.qat.linear import Linear as QATLinear

__all__ = ["fuse_prepared_graph", "unfuse_prepared_graph", "PreparedCausalLMLoss", "ShapeMemo", "PreparedLinear", "PreparedMLP", "PreparedRMSNorm", "PreparedAttention"]

aten = torch.ops.aten


# ---- modules the rewritten graph calls ------------------------------------------------------------------------------------------
class PreparedLinear(QATLinear):
    """`aten.linear(x, fq_w(W), b)` of a prepared graph as a QAT Linear (modules/qat/linear.py): same Parameters, the graph's weight
    fake-quantizer, and -- for the kernels that look at the consumer's input fake-quantizer -- the graph's activation fake-quantizer
    under `activation_pre_process["0"]` (not hooked: it stays a node in front of this one)."""

    def __init__(self, weight, bias, weight_fq, act_fq):
        nn.Module.__init__(self)
        self.in_features, self.out_features = int(weight.shape[1]), int(weight.shape[0])
        self.weight = weight
        if bias is None:
            self.register_parameter("bias", None)
        else:
            self.bias = bias
        self.weight_fake_quant = weight_fq
        self.activation_pre_process = nn.ModuleDict({} if act_fq is None else {"0": act_fq})
        self.__dict__["_qt_prepared"] = True


def _producer_ok(fq):
    return isinstance(fq, FusedAmaxObsFakeQuantize) and fq.producer_fusable() and os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0"


def _table_ok(fq):
    """A stateless table-format fake-quantizer (posit, fpN, ... without `qs`) a producing kernel may apply in its row form."""
    return (isinstance(fq, FusedAmaxObsFakeQuantize) and fq.stateless_map() and fq._qt_format.kind == _native.QT_FMT_LUT
            and os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0" and os.environ.get("QT_FUSED_PRODUCER_MAP", "1") != "0")


class PreparedMLP(nn.Module):
    """`silu(gate(x)) * up(x)` with the down projection's input fake-quantizer behind it (a node of its own: it hands through)."""

    def __init__(self, gate, up, down):
        super().__init__()
        self.gate_proj, self.up_proj = gate, up
        self.__dict__["down_proj"] = down                     # a reference, not a child: it is a node (and a child) of the graph
        self.__dict__["_qt_prepared"] = True

    def forward(self, x):
        from . import model_fusions as mf
        y = mf._fused_mlp_or_none(self, x) if self.__dict__["down_proj"] is not None else None
        if y is not None:
            return y
        gate, up = self.gate_proj(x), self.up_proj(x)
        if mf._eligible(gate, up) and gate.shape == up.shape and gate.shape[-1] % 8 == 0 and gate.numel() > 0:
            down = self.__dict__["down_proj"]
            fq = mf.consumer_fq(down) if down is not None else None
            if fq is not None and os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0":
                return mf.silu_mul_fq(gate, up, fq)
            mfq = mf.consumer_fq_map(down) if down is not None and os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0" else None
            if mfq is not None:
                y = mf.silu_mul_map(gate, up, mfq)
                if y is not None:
                    return y
            return mf.silu_mul(gate, up)
        return F.silu(gate) * up


class PreparedRMSNorm(nn.Module):
    """LlamaRMSNorm as exported (to fp32, pow, mean, + eps, rsqrt, mul, to bf16, weight *), optionally with the residual add in front
    (`forward(a, b)` -> (sum, y)); `consumer`: the fake-quantizer node behind the result, `sum_fq`: the one behind the sum."""

    def __init__(self, weight, eps, consumer=None, sum_fq=None):
        super().__init__()
        self.weight = weight
        self.eps = float(eps)
        self.__dict__["consumer"] = consumer
        self.__dict__["sum_fq"] = sum_fq

    def forward(self, a, b=None):
        from . import model_fusions as mf
        w = self.weight
        cols = a.shape[-1]
        ok = (mf._eligible(a, w) and cols % 8 == 0 and cols <= 16384 and a.numel() > 0 and w.is_contiguous() and w.shape == (cols,)
              and (b is None or (mf._eligible(b) and b.shape == a.shape)))
        if ok:
            fq = self.__dict__["consumer"]
            if _table_ok(fq):                                 # table formats: the row form (and the same for the residual stream's)
                sfq = self.__dict__["sum_fq"]
                sfq = sfq if (b is not None and _table_ok(sfq) and sfq.dtype == fq.dtype) else None
                got = mf.rmsnorm_map(a, b, w, self.eps, [fq], sum_fq=sfq)
                if got is not None:
                    return got[1] if b is None else got
            fq = fq if _producer_ok(fq) else None
            if b is None:
                return mf.rmsnorm_fq(a, w, self.eps, fq) if fq is not None else mf.rmsnorm(a, w, self.eps)
            sfq = self.__dict__["sum_fq"]
            sfq = sfq if _producer_ok(sfq) else None
            return _add_rmsnorm(a, b, w, self.eps, fq, sfq)
        s = a if b is None else a + b
        x32 = s.to(torch.float32)
        var = x32.pow(2).mean(-1, keepdim=True)
        y = w * (x32 * torch.rsqrt(var + self.eps)).to(s.dtype)
        return y if b is None else (s, y)


def _add_rmsnorm(a, b, weight, eps, fq, sum_fq):
    """(sum, y) in one launch (qt_add_rmsnorm_sumfq_bf16): sum = bf16(a + b), written as sum_fq(sum) when that fake-quantizer sits behind
    it; y = fq(RMSNorm(sum)) with its FP8 codes.  Both results are marked for the fake-quantizer nodes that follow."""
    from .model_fusions import _fp8_view
    cols = a.shape[-1]
    a2, b2 = a.contiguous(), b.contiguous()
    total = torch.empty_like(a2)
    y = torch.empty_like(a2)
    y8 = torch.empty(a2.shape, dtype=torch.uint8, device=a2.device) if fq is not None else None
    _native.check(_native.lib().qt_add_rmsnorm_sumfq_bf16(
        a2.data_ptr(), b2.data_ptr(), weight.data_ptr(), total.data_ptr(), y.data_ptr(), y8.data_ptr() if y8 is not None else None,
        a2.numel() // cols, cols, float(eps), ctypes.byref(fq._qt_format) if fq is not None else None,
        ctypes.byref(sum_fq._qt_format) if sum_fq is not None else None, _stream_ptr(a2)), "qt_add_rmsnorm_sumfq_bf16")
    if fq is not None:
        y._qt_fp8 = _fp8_view(y8, fq)
        y._qt_fq_done_by = fq
        y._qt_ver = y._version
    if sum_fq is not None:
        total._qt_fq_done_by = sum_fq
        total._qt_ver = total._version
    return total, y


class PreparedAttention(nn.Module):
    """Rotary embedding + the attention core of an exported LlamaAttention (eager attention, no KV cache, as many key heads as query
    heads): forward(q, k, v, cos, sin, mask) with q / k / v the [B, H, S, D] views of the projections' outputs -> [B, S, H * D]."""

    def __init__(self, fq_q, fq_k, fq_p, fq_v, inner_q, inner_k, scaling, out_proj):
        super().__init__()
        self.scaling = float(scaling)
        d = self.__dict__
        d["fqs"] = (fq_q, fq_k, fq_p, fq_v)
        d["inner"] = (inner_q, inner_k)
        d["_qt_out_proj"] = out_proj                          # consumer of the result: its input fake-quantizer rides on the kernel's epilogue

    def _rotate(self, x, cos_u, sin_u, inner):
        a = x * cos_u
        if inner is not None:
            a = inner(a)
        h = x.shape[-1] // 2
        return a + torch.cat((-x[..., h:], x[..., :h]), dim=-1) * sin_u

    def forward(self, q, k, v, cos, sin, mask):
        out = self._fused(q, k, v, cos, sin, mask)
        if out is None:
            out = self._fused_table(q, k, v, cos, sin, mask)
        if out is not None:
            return out
        fq_q, fq_k, fq_p, fq_v = self.__dict__["fqs"]
        inner_q, inner_k = self.__dict__["inner"]
        cos_u, sin_u = cos.unsqueeze(1), sin.unsqueeze(1)
        qr = fq_q(self._rotate(q, cos_u, sin_u, inner_q))
        kr = fq_k(self._rotate(k, cos_u, sin_u, inner_k).transpose(2, 3))
        s = torch.matmul(qr, kr) * self.scaling
        if mask is not None:
            s = s + mask
        p = F.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(fq_p(p), fq_v(v))
        o = o.transpose(1, 2).contiguous()
        return o.reshape(o.shape[0], o.shape[1], -1)

    def _fused_table(self, q, k, v, cos, sin, mask):
        """Stateless TABLE formats on all four matmul inputs: the rotary kernel applies fq_q / fq_k (and the inner pair) in their row
        form, fq_v runs as its own strided pass, and the core is qt_attention_fq_bf16 (bf16 matrix instructions, the probabilities'
        fake-quantizer inside, §4.4b of DESIGN.md)."""
        from . import fused, model_fusions as mf
        fq_q, fq_k, fq_p, fq_v = self.__dict__["fqs"]
        inner_q, inner_k = self.__dict__["inner"]
        if os.environ.get("QT_FUSED_ATTENTION", "auto") == "0":
            return None
        if not (mf._eligible(q, k, v, cos, sin) and q.dim() == 4 and q.shape == k.shape == v.shape):
            return None
        B, H, S, D = q.shape
        if D not in (64, 128) or S % 4 != 0 or B * H > 65535:
            return None
        if not all(_table_ok(f) for f in (fq_q, fq_k, fq_p, fq_v)) or len({f.dtype for f in (fq_q, fq_k, fq_p, fq_v)}) != 1:
            return None
        if any(f is not None and not (_table_ok(f) and f.dtype == fq_q.dtype) for f in (inner_q, inner_k)):
            return None
        if mf._row_stride(q) is None or mf._row_stride(k) is None:
            return None
        if cos.dim() != 3 or cos.shape[0] not in (1, B) or cos.shape[-1] != D or cos.shape[-2] != S or sin.shape != cos.shape:
            return None
        mk = fused._mask_strides(mask, B, H, S, S, q.device, 4)
        if mk is False:
            return None
        m, msb, msh, msq = mk
        pf = fq_p.map_producer_format(q.device)
        if pf is None:
            return None
        if cos.shape[0] != B:
            cos, sin = cos.expand(B, -1, -1), sin.expand(B, -1, -1)
        got = mf.rope_map(q, k, cos.contiguous(), sin.contiguous(), fq_q, fq_k, inner_q is not None, inner_k is not None)
        if got is None:
            return None
        qq, kq = got
        for f, t in ((fq_q, q), (fq_k, k), (inner_q, q), (inner_k, k)):      # evaluated inside the rotary launch
            if f is not None:
                f.__dict__["_qt_calls"] = f.__dict__.get("_qt_calls", 0) + 1
                _FQ_STATS.add(t.numel())
        fmt, qmap = pf
        out = torch.empty((B, S, H, D), dtype=torch.bfloat16, device=q.device)
        _FQ_STATS.add(B * H * S * S)                          # fq_p, inside the kernel
        fq_p.__dict__["_qt_calls"] = fq_p.__dict__.get("_qt_calls", 0) + 1
        proj = self.__dict__["_qt_out_proj"]
        fq_o = mf.consumer_fq_map(proj) if proj is not None else None
        if not (fq_o is not None and fq_o.dtype == fq_p.dtype and _table_ok(fq_o)):
            fq_o = None                                       # else: its node runs its own pass
        # (with fq_o: the output projection's input fake-quantizer, same format, on the kernel's epilogue; its node hands the result through)
        if fused.attention_rows_or_none(_native.lib(), _stream_ptr(q), qq, kq, v, fq_v, m, mask, (msb, msh, msq), out, (B, H, S, S, D), self.scaling, fmt,
                                        qmap.data_ptr(), fq_o is not None):
            pass                                              # (the value pass of the launch pair was fq_v's call: counted there)
        else:
            vq = fq_v(v).contiguous()
            fused.launch_attention_fq(_native.lib(), _stream_ptr(q), qq, kq, vq, m, mask, (msb, msh, msq), out, (B, H, S, S, D), self.scaling, fmt,
                                      qmap.data_ptr(), None, None, fq_o is not None)
        if fq_o is not None:
            fq_o.expect_prequantized(out, None)
        return out.reshape(B, S, H * D)

    def _fused(self, q, k, v, cos, sin, mask):
        from . import fused, model_fusions as mf
        fqs = self.__dict__["fqs"]
        inner_q, inner_k = self.__dict__["inner"]
        if os.environ.get("QT_FP8_ATTENTION", "1") == "0" or os.environ.get("QT_FP8_ATTENTION_KERNEL", "1") == "0":
            return None
        if not (mf._eligible(q, k, v, cos, sin) and q.dim() == 4 and q.shape == k.shape == v.shape):
            return None
        B, H, S, D = q.shape
        if D not in (64, 128) or S % 128 != 0 or S > 1024 or B * H > 65535:
            return None
        if not all(_producer_ok(f) for f in fqs) or len({f._qt_format.key() for f in fqs}) != 1:
            return None
        for f in (inner_q, inner_k):
            if f is not None and not _producer_ok(f):
                return None
        rq, rk = mf._row_stride(q), mf._row_stride(k)
        if rq is None or rk is None or v.stride(-1) != 1 or any(s % 8 for s in v.stride()[:3]) or v.data_ptr() % 16:
            return None
        if cos.shape[-1] != D or cos.shape[-2] != S or sin.shape != cos.shape:
            return None
        if cos.dim() != 3 or cos.shape[0] not in (1, B):
            return None
        if cos.shape[0] != B:
            cos, sin = cos.expand(B, -1, -1), sin.expand(B, -1, -1)
        cos, sin = cos.contiguous(), sin.contiguous()
        fq_q, fq_k, fq_p, fq_v = fqs
        dev = q.device
        q8 = torch.empty((B, H, S, D), dtype=torch.uint8, device=dev)
        k8 = torch.empty((B, H, S, D), dtype=torch.uint8, device=dev)
        vt8 = torch.empty((B, H, D, S), dtype=torch.uint8, device=dev)
        q_out = torch.empty((B, H, S, D), dtype=q.dtype, device=dev)            # codes only: decoded on demand (materialize_lazy)
        k_out = torch.empty((B, H, S, D), dtype=q.dtype, device=dev)
        fmt = lambda f: ctypes.byref(f._qt_format) if f is not None else None   # noqa: E731
        _native.check(_native.lib().qt_rope_fq_inner_value(
            q.data_ptr(), k.data_ptr(), cos.data_ptr(), sin.data_ptr(), None, None, q8.data_ptr(), k8.data_ptr(), B, S, H, H, D, rq, rk,
            fmt(fq_q), fmt(fq_k), fmt(inner_q), fmt(inner_k), v.data_ptr(), vt8.data_ptr(), v.stride(0), v.stride(1), v.stride(2), fmt(fq_v),
            _stream_ptr(q)), "qt_rope_fq_inner_value")
        for f, t in ((inner_q, q), (inner_k, k)):              # the two inner calls, evaluated inside that launch
            if f is not None:
                f.__dict__["_qt_calls"] = f.__dict__.get("_qt_calls", 0) + 1
                _FQ_STATS.add(t.numel())
        for t, t8, f in ((q_out, q8, fq_q), (k_out, k8, fq_k)):
            t._qt_lazy = True
            t._qt_fq_done_by = f
            t._qt_fp8 = mf._fp8_view(t8, f)
            t._qt_ver = t._version
        self.__dict__["_qt_vt8"] = (fused.value_key(v), fq_v, vt8)
        out = fused._attention_fp8_or_none(self, q_out, k_out, v, mask, self.scaling, fqs)
        if out is None:                                        # the kernel declined after all (mask layout): finish on the node sequence
            self.__dict__.pop("_qt_vt8", None)
            from .fake_quantize import materialize_lazy
            materialize_lazy(q_out)
            materialize_lazy(k_out)
            s = torch.matmul(fq_q(q_out), fq_k(k_out.transpose(2, 3))) * self.scaling
            if mask is not None:
                s = s + mask
            p = F.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
            o = torch.matmul(fq_p(p), fq_v(v)).transpose(1, 2).contiguous()
            return o.reshape(B, S, -1)
        return out.reshape(B, S, H * D)


class PreparedCausalLMLoss(nn.Module):
    """transformers' ForCausalLMLoss as exported: logits.float(), labels padded with -100 and shifted by one, mean cross entropy with
    ignore_index -100.  On the device: ONE pass over the bf16 logits (qt_causal_lm_loss_bf16), no fp32 copy of them."""

    def forward(self, logits, labels):
        if (logits.is_cuda and not torch.is_grad_enabled() and logits.dim() == 3 and logits.dtype == torch.bfloat16 and logits.stride(2) == 1
                and logits.stride(0) == logits.shape[1] * logits.stride(1) and logits.stride(1) % 8 == 0 and logits.data_ptr() % 16 == 0
                and labels.dtype == torch.long and labels.is_contiguous() and labels.shape == logits.shape[:2]):
            B, S, V = logits.shape
            scratch = torch.empty(B * S + 1, dtype=torch.float32, device=logits.device)
            _native.check(_native.lib().qt_causal_lm_loss_bf16(logits.data_ptr(), labels.data_ptr(), B, S, V, logits.stride(1), -100,
                                                               scratch.data_ptr(), scratch.data_ptr() + 4 * B * S, _stream_ptr(logits)),
                          "qt_causal_lm_loss_bf16")
            return scratch[B * S]
        x = logits.to(torch.float32)
        t = F.pad(labels, [0, 1], "constant", -100.0)[:, 1:]
        return F.cross_entropy(x.view(-1, x.shape[-1]), t.reshape(-1).to(x.device))


class ShapeMemo(nn.Module):
    """The part of an exported forward that depends on the input SHAPES only (HF builds the causal mask and the rotary tables from
    `arange(seq_len)` on every call): computed once per shape, kept.  Nothing a fake-quantizer or a Parameter's value enters is
    ever in here -- weights are fake-quantized on every forward, as the reference does.

    Kept results are never freed while the module lives: a hipGraph captured earlier (harness.GraphedWindow) replays reads of their
    addresses, so an eviction would hand that memory back to the allocator under a live graph.  Past `kMax` shapes a new shape is
    computed and returned without being kept.  The key also holds the device and the version counters of the constants the
    sub-graph reads, so `gm.to(other_device)` or an in-place update of one of them recomputes instead of returning stale tables."""

    kMax = 64

    def __init__(self, sub):
        super().__init__()
        self.sub = sub
        self.__dict__["kept"] = {}

    def _state_key(self):
        return tuple((str(b.device), b._version, b.data_ptr()) for b in self.sub.buffers())

    def forward(self, *sizes):
        key = (tuple(int(v) for v in sizes), self._state_key())
        kept = self.__dict__["kept"]
        out = kept.get(key)
        if out is None:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                return self.sub(*sizes)                       # a shape first seen inside a capture: computed in the graph, not kept
            out = self.sub(*sizes)
            if len(kept) < self.kMax:
                kept[key] = out
        return out


# ---- graph helpers ----------------------------------------------------------------------------------------------------------------
def _is(node, target):
    return isinstance(node, Node) and node.op == "call_function" and node.target == target


def _users(node):
    """Users that compute something (`_assert_tensor_metadata` nodes only check the exporter's metadata)."""
    return [u for u in node.users if not _is(u, aten._assert_tensor_metadata.default)]


def _asserts(node):
    return [u for u in node.users if _is(u, aten._assert_tensor_metadata.default)]


def _attr(gm, node):
    """The Parameter / buffer / constant behind a get_attr node, or None."""
    if not (isinstance(node, Node) and node.op == "get_attr"):
        return None
    obj = gm
    for part in str(node.target).split("."):
        if not hasattr(obj, part):
            return None
        obj = getattr(obj, part)
    return obj


def _scalar(gm, arg):
    """A python number, or the value of a one-element constant tensor node."""
    if isinstance(arg, (int, float)):
        return float(arg)
    t = _attr(gm, arg)
    if isinstance(t, torch.Tensor) and t.numel() == 1:
        return float(t.detach().reshape(-1)[0].item())
    return None


class _Pass:
    def __init__(self, gm):
        self.gm = gm
        self.g = gm.graph
        self.n = 0
        self.shim_of = {}                 # call_module node -> PreparedLinear
        self.counts = {"linear": 0, "sibling_groups": 0, "mlp": 0, "rmsnorm": 0, "add_rmsnorm": 0, "attention": 0}

    def fq_of(self, node):
        if isinstance(node, Node) and node.op == "call_module":
            mod = self.gm.get_submodule(str(node.target))
            if isinstance(mod, FusedAmaxObsFakeQuantize):
                return mod
        return None

    def add_module(self, prefix, mod):
        name = f"_qt_{prefix}_{self.n}"
        self.n += 1
        self.gm.add_submodule(name, mod)
        return name

    def erase(self, nodes):
        """Erase `nodes` (users first); metadata assertions hanging off them go too.  Nodes that still have other users stay."""
        pending = list(dict.fromkeys(n for n in nodes if isinstance(n, Node)))
        progress = True
        while pending and progress:
            progress = False
            for n in list(pending):
                for a in _asserts(n):
                    self.g.erase_node(a)
                if not n.users:
                    self.g.erase_node(n)
                    pending.remove(n)
                    progress = True

    # -- Linears ---------------------------------------------------------------------------------------------------------------
    def linears(self):
        by_input = {}
        for node in list(self.g.nodes):
            if not _is(node, aten.linear.default) or node.kwargs:
                continue
            x, wq = node.args[0], node.args[1]
            b = node.args[2] if len(node.args) > 2 else None
            w_fq = self.fq_of(wq)
            if w_fq is None or len(wq.users) != 1 or len(wq.args) != 1:
                continue
            W = _attr(self.gm, wq.args[0])
            if not isinstance(W, nn.Parameter) or W.dim() != 2:
                continue
            bias = None
            if b is not None:
                bias = _attr(self.gm, b)
                if not isinstance(bias, nn.Parameter):
                    continue                                   # a fake-quantized bias: left as it is
            a_fq = self.fq_of(x)
            shim = PreparedLinear(W, bias, w_fq, a_fq)
            if (a_fq is not None and a_fq.fp8_exact() and w_fq.fp8_exact() and a_fq.producer_fusable()):
                a_fq._emit_fp8 = "both"                       # the GEMMs multiply codes (fused.mark_fp8_producer)
            name = self.add_module("linear", shim)
            with self.g.inserting_before(node):
                new = self.g.call_module(name, (x,))
            new.meta = dict(node.meta)
            node.replace_all_uses_with(new)
            self.erase([node, wq, wq.args[0], b])
            self.shim_of[new] = shim
            by_input.setdefault(x, []).append(new)
            self.counts["linear"] += 1
        return by_input

    def mlps(self, by_input):
        for node in list(self.g.nodes):
            if not _is(node, aten.mul.Tensor) or len(node.args) != 2:
                continue
            act, up = node.args
            if not _is(act, aten.silu.default) or len(_users(act)) != 1 or up not in self.shim_of:
                continue
            gate = act.args[0]
            if gate not in self.shim_of or len(_users(gate)) != 1 or len(_users(up)) != 1 or gate.args[0] is not up.args[0]:
                continue
            x = gate.args[0]
            down = None
            users = _users(node)
            if len(users) == 1 and self.fq_of(users[0]) is not None:
                for u in _users(users[0]):
                    if u in self.shim_of:
                        down = self.shim_of[u]
                        break
            g_shim, u_shim = self.shim_of[gate], self.shim_of[up]
            if g_shim.weight.shape != u_shim.weight.shape:
                continue
            mlp = PreparedMLP(g_shim, u_shim, down)
            name = self.add_module("mlp", mlp)
            with self.g.inserting_before(node):
                new = self.g.call_module(name, (x,))
            new.meta = dict(node.meta)
            node.replace_all_uses_with(new)
            self.erase([node, act, gate, up])
            for n in (gate, up):
                by_input[x].remove(n)
                self.shim_of.pop(n)
            self.counts["mlp"] += 1

    def siblings(self, by_input):
        from .fused import SiblingGroup
        for x, nodes in by_input.items():
            shims = [self.shim_of[n] for n in nodes if n in self.shim_of]
            if 2 <= len(shims) <= 4 and len({s.weight.shape[1] for s in shims}) == 1 and self.fq_of(x) is not None:
                group = SiblingGroup(shims)
                for s in shims:
                    s.__dict__["_qt_sibling_group"] = group
                self.counts["sibling_groups"] += 1

    # -- RMSNorm ----------------------------------------------------------------------------------------------------------------
    def norms(self):
        for r in list(self.g.nodes):
            if not _is(r, aten.rsqrt.default):
                continue
            add_eps = r.args[0]
            if not _is(add_eps, aten.add.Tensor):
                continue
            mean, eps = add_eps.args[0], _scalar(self.gm, add_eps.args[1])
            if eps is None or not _is(mean, aten.mean.dim) or list(mean.args[1]) != [-1] or not (len(mean.args) > 2 and mean.args[2]):
                continue
            pw = mean.args[0]
            if not _is(pw, aten.pow.Tensor_Scalar) or pw.args[1] != 2:
                continue
            x32 = pw.args[0]
            if not _is(x32, aten.to.dtype) or x32.args[1] != torch.float32:
                continue
            x = x32.args[0]
            ru = _users(r)
            if len(ru) != 1 or not _is(ru[0], aten.mul.Tensor) or set(ru[0].args) != {x32, r} or len(_users(add_eps)) != 1 \
                    or len(_users(mean)) != 1 or len(_users(pw)) != 1 or set(_users(x32)) != {pw, ru[0]}:
                continue
            mul_x = ru[0]
            mu = _users(mul_x)
            if len(mu) != 1 or not _is(mu[0], aten.to.dtype) or mu[0].args[1] != torch.bfloat16:
                continue
            to_bf = mu[0]
            tu = _users(to_bf)
            if len(tu) != 1 or not _is(tu[0], aten.mul.Tensor):
                continue
            out = tu[0]
            W = _attr(self.gm, out.args[0])
            if out.args[1] is not to_bf or not isinstance(W, nn.Parameter) or W.dim() != 1:
                continue
            # the fake-quantizer behind the result (through an alias node)
            tail = out
            ou = _users(out)
            if len(ou) == 1 and _is(ou[0], aten.alias.default):
                tail = ou[0]
                ou = _users(tail)
            consumer = self.fq_of(ou[0]) if len(ou) == 1 else None
            # the residual add in front
            sum_fq, with_add, sum_user = None, False, None
            if _is(x, aten.add.Tensor) and len(x.args) == 2 and all(isinstance(a, Node) for a in x.args) and not x.kwargs:
                others = [u for u in _users(x) if u is not x32]
                sa, sb = (a.meta.get("val") for a in x.args)
                same = sa is None or sb is None or (tuple(sa.shape) == tuple(sb.shape) and sa.dtype == sb.dtype)
                if same and len(others) <= 1 and (not others or self.fq_of(others[0]) is not None):
                    with_add = True
                    if others:
                        sum_user = others[0]
                        sum_fq = self.fq_of(sum_user)
            mod = PreparedRMSNorm(W, eps, consumer, sum_fq)
            name = self.add_module("norm", mod)
            with self.g.inserting_before(x.next if with_add else x32):   # the sum's fake-quantizer node may sit in front of the norm's first node
                if with_add:
                    pair = self.g.call_module(name, (x.args[0], x.args[1]))
                    total = self.g.call_function(operator.getitem, (pair, 0))
                    new = self.g.call_function(operator.getitem, (pair, 1))
                    total.meta = dict(x.meta)
                else:
                    new = self.g.call_module(name, (x,))
            new.meta = dict(out.meta)
            out.replace_all_uses_with(new)
            if with_add:
                if sum_user is not None:
                    sum_user.replace_input_with(x, total)
                self.counts["add_rmsnorm"] += 1
            else:
                self.counts["rmsnorm"] += 1
            self.erase([out, to_bf, mul_x, r, add_eps, add_eps.args[1], mean, pw, x32, out.args[0]] + ([x] if with_add else []))
            if with_add and not total.users:
                self.g.erase_node(total)

    # -- rotary + attention core ------------------------------------------------------------------------------------------------------
    def _rotary(self, node):
        """node = add(fq_i(mul(x, unsqueeze(cos, 1))) | mul(x, ..), mul(cat([neg(slice(x, 3, h, end)), slice(x, 3, 0, h)], -1), unsqueeze(sin, 1)))
        -> (x, cos, sin, inner fake-quantizer or None, nodes to erase) or None."""
        if not _is(node, aten.add.Tensor) or len(node.args) != 2:
            return None
        first, second = node.args
        dead = [node]
        inner = self.fq_of(first)
        if inner is not None:
            if len(_users(first)) != 1:
                return None
            dead.append(first)
            first = first.args[0]
        if not _is(first, aten.mul.Tensor) or not _is(second, aten.mul.Tensor) or len(_users(first)) != 1 or len(_users(second)) != 1:
            return None
        x, cos_u = first.args
        cat, sin_u = second.args
        if not _is(cos_u, aten.unsqueeze.default) or cos_u.args[1] != 1 or not _is(sin_u, aten.unsqueeze.default) or sin_u.args[1] != 1:
            return None
        if not _is(cat, aten.cat.default) or len(cat.args[0]) != 2 or (cat.args[1] if len(cat.args) > 1 else 0) != -1 or len(_users(cat)) != 1:
            return None
        neg, lo = cat.args[0]
        if not _is(neg, aten.neg.default) or not _is(lo, aten.slice.Tensor) or not _is(neg.args[0], aten.slice.Tensor):
            return None
        hi = neg.args[0]
        if lo.args[0] is not x or hi.args[0] is not x or lo.args[1] != 3 or hi.args[1] != 3 or lo.args[2] != 0 or lo.args[3] != hi.args[2]:
            return None
        val = x.meta.get("val")
        if val is not None and int(val.shape[-1]) != 2 * int(lo.args[3]):
            return None
        if len(hi.args) > 3 and hi.args[3] < (1 << 62):
            return None
        if len(_users(lo)) != 1 or len(_users(hi)) != 1 or len(_users(neg)) != 1:
            return None
        dead += [first, second, cat, neg, lo, hi, cos_u, sin_u]
        return x, cos_u.args[0], sin_u.args[0], inner, dead

    def attentions(self):
        for sm in list(self.g.nodes):
            if not _is(sm, aten.softmax.int) or sm.args[1] != -1 or (sm.args[2] if len(sm.args) > 2 else None) != torch.float32:
                continue
            sc = sm.args[0]
            mask = None
            dead = [sm]
            if _is(sc, aten.add.Tensor) and len(_users(sc)) == 1:
                scaled, mask = sc.args
                dead.append(sc)
            else:
                scaled = sc
            if not _is(scaled, aten.mul.Tensor) or len(_users(scaled)) != 1:
                continue
            mm, scaling = scaled.args[0], _scalar(self.gm, scaled.args[1])
            if scaling is None or not _is(mm, aten.matmul.default) or len(_users(mm)) != 1:
                continue
            nq, nk = mm.args
            fq_q, fq_k = self.fq_of(nq), self.fq_of(nk)
            if fq_q is None or fq_k is None or len(_users(nq)) != 1 or len(_users(nk)) != 1:
                continue
            kt = nk.args[0]
            if not _is(kt, aten.transpose.int) or sorted(kt.args[1:]) != [2, 3] or len(_users(kt)) != 1:
                continue
            rq, rk = self._rotary(nq.args[0]), self._rotary(kt.args[0])
            if rq is None or rk is None or rq[1] is not rk[1] or rq[2] is not rk[2]:
                continue
            # forward: to(bf16), dropout(p, False), fq_p, matmul(., fq_v(v)), transpose(1, 2), contiguous, reshape
            cur = sm
            chain = []
            u = _users(cur)
            if len(u) == 1 and _is(u[0], aten.to.dtype) and u[0].args[1] == torch.bfloat16:
                cur = u[0]
                chain.append(cur)
                u = _users(cur)
            if len(u) == 1 and _is(u[0], aten.dropout.default) and (u[0].args[2] is False or u[0].args[1] == 0.0):
                cur = u[0]
                chain.append(cur)
                u = _users(cur)
            if len(u) != 1 or self.fq_of(u[0]) is None:
                continue
            np_ = u[0]
            fq_p = self.fq_of(np_)
            u = _users(np_)
            if len(u) != 1 or not _is(u[0], aten.matmul.default) or u[0].args[0] is not np_:
                continue
            mm2 = u[0]
            nv = mm2.args[1]
            fq_v = self.fq_of(nv)
            if fq_v is None or len(_users(nv)) != 1:
                continue
            u = _users(mm2)
            if len(u) != 1 or not _is(u[0], aten.transpose.int) or sorted(u[0].args[1:]) != [1, 2]:
                continue
            tr = u[0]
            u = _users(tr)
            if len(u) != 1 or not _is(u[0], aten.contiguous.default):
                continue
            cont = u[0]
            u = _users(cont)
            if len(u) != 1 or not (_is(u[0], aten.reshape.default) or _is(u[0], aten.view.default)):
                continue
            rs = u[0]
            shape = list(rs.args[1])
            if len(shape) != 3 or shape[2] != -1:
                continue
            out_proj = None
            ru = _users(rs)
            if len(ru) == 1 and self.fq_of(ru[0]) is not None:
                for c in _users(ru[0]):
                    if c in self.shim_of:
                        out_proj = self.shim_of[c]
                        break
            mod = PreparedAttention(fq_q, fq_k, fq_p, fq_v, rq[3], rk[3], scaling, out_proj)
            name = self.add_module("attention", mod)
            with self.g.inserting_before(rs):
                new = self.g.call_module(name, (rq[0], rk[0], nv.args[0], rq[1], rq[2], mask))
            new.meta = dict(rs.meta)
            rs.replace_all_uses_with(new)
            self.erase([rs, cont, tr, mm2, nv, np_] + chain[::-1] + dead + [scaled, scaled.args[1], mm, nq, nk, kt] + rq[4] + rk[4])
            self.counts["attention"] += 1


def _loss(p):
    """cross_entropy_loss(view(to_fp32(logits), [-1, V]), to(view(slice(pad(labels, [0, 1], constant, -100), 1, 1, end), [-1])))."""
    for ce in list(p.g.nodes):
        if not _is(ce, aten.cross_entropy_loss.default) or len(ce.args) != 2 or ce.kwargs:
            continue
        lv, tv = ce.args
        if not _is(lv, aten.view.default) or not _is(lv.args[0], aten.to.dtype) or lv.args[0].args[1] != torch.float32:
            continue
        logits = lv.args[0].args[0]
        if list(lv.args[1])[0] != -1 or len(lv.args[1]) != 2 or len(_users(lv)) != 1 or len(_users(lv.args[0])) != 1:
            continue
        t = tv
        chain = []
        if _is(t, aten.to.dtype_layout) or _is(t, aten.to.dtype) or _is(t, aten.to.device):
            chain.append(t)
            t = t.args[0]
        if not _is(t, aten.view.default) or list(t.args[1]) != [-1]:
            continue
        chain.append(t)
        sl = t.args[0]
        if not _is(sl, aten.slice.Tensor) or sl.args[1:3] != (1, 1) or (len(sl.args) > 3 and sl.args[3] < (1 << 62)):
            continue
        pad = sl.args[0]
        if not _is(pad, aten.pad.default) or list(pad.args[1]) != [0, 1] or (len(pad.args) > 3 and float(pad.args[3]) != -100.0):
            continue
        labels = pad.args[0]
        if any(len(_users(n)) != 1 for n in chain + [sl, pad]):
            continue
        name = p.add_module("loss", PreparedCausalLMLoss())
        with p.g.inserting_before(ce):
            new = p.g.call_module(name, (logits, labels))
        new.meta = dict(ce.meta)
        ce.replace_all_uses_with(new)
        p.erase([ce, lv, lv.args[0]] + chain + [sl, pad])
        p.counts["loss"] = p.counts.get("loss", 0) + 1


_IMPURE = {aten.dropout.default, aten.native_dropout.default, aten.rand.default, aten.randn.default, aten.rand_like.default,
           aten.randn_like.default, aten.bernoulli.default}


def _hoist_shape_only(p):
    """Move every node whose value depends on the input shapes alone (no placeholder values, no Parameter, no module call) into one
    ShapeMemo module.  The frontier -- such nodes with a user outside -- becomes that module's outputs."""
    gm, g = p.gm, p.g
    hoist = {}                                              # node -> True (tensor / value) for hoistable nodes, in graph order
    sizes = []

    def ok_arg(a):
        if isinstance(a, Node):
            return a in hoist
        if isinstance(a, (list, tuple)):
            return all(ok_arg(x) for x in a)
        if isinstance(a, dict):
            return all(ok_arg(x) for x in a.values())
        return True

    for n in g.nodes:
        if n.op == "call_function" and n.target == aten.sym_size.int and isinstance(n.args[0], Node) and n.args[0].op == "placeholder":
            hoist[n] = True
            sizes.append(n)
        elif n.op == "get_attr":
            obj = _attr(gm, n)
            if isinstance(obj, torch.Tensor) and not isinstance(obj, nn.Parameter) and not obj.requires_grad:
                # buffers and lifted constants (inv_freq, eps, the mask's fill value).  The exporter registers all of them as
                # persistent buffers of the graph module, so a `load_state_dict` may rewrite them in place: ShapeMemo keys its
                # results on their version counters (and device), and the root attributes stay where they are
                hoist[n] = True
        elif n.op == "call_function" and n.target not in _IMPURE and "rand" not in str(n.target) and ok_arg(n.args) and ok_arg(n.kwargs):
            hoist[n] = True                                 # constructors without inputs (arange(1)) included
    # get_attr leaves that nothing hoisted reads stay where they are
    inside = [n for n in hoist if n.op != "get_attr" and n not in sizes]
    if not inside:
        return
    frontier = [n for n in inside if any(u not in hoist for u in n.users) and not _is(n, aten._assert_tensor_metadata.default)]
    # values that are python numbers (sym ints) are cheap and may feed view shapes outside: leave integer arithmetic in the main graph too
    frontier = [n for n in frontier if isinstance(n.meta.get("val"), torch.Tensor)]
    if not frontier:
        return
    # the nodes the frontier needs
    need, stack = set(), list(frontier)
    while stack:
        n = stack.pop()
        if n in need or n in sizes:
            continue
        need.add(n)
        stack.extend(a for a in n.all_input_nodes if a in hoist)
    if sum(1 for n in need if n.op == "call_function") < 4:
        return
    target_device = next((q.device for q in gm.parameters() if q.device.type == "cuda"), None)
    sub = torch.fx.Graph()
    env = {}
    for sz in sizes:
        env[sz] = sub.placeholder(sz.name)
    holder = nn.Module()
    for n in g.nodes:
        if n not in need:
            continue
        if n.op == "get_attr":
            attr = f"c{len(env)}"
            setattr(holder, attr, _attr(gm, n)) if isinstance(_attr(gm, n), nn.Parameter) else holder.register_buffer(attr, _attr(gm, n), persistent=False)
            env[n] = sub.get_attr(attr)
        else:
            env[n] = sub.node_copy(n, lambda a: env[a])
            if target_device is not None and "device" in env[n].kwargs:
                # the exporter records the example inputs' device in constructor nodes (arange, full, ...); upstream's driver pins them
                # to the model's device after prepare_pt2e (wikitext.py:98-101) by walking model.graph -- which cannot see these nodes
                # any more once they live in the memo's sub-graph
                env[n].kwargs = dict(env[n].kwargs, device=target_device)
    sub.output(tuple(env[n] for n in frontier))
    memo = ShapeMemo(GraphModule(holder, sub))
    name = p.add_module("shapes", memo)
    first_user = None
    for n in g.nodes:
        if n not in hoist and any(a in frontier for a in n.all_input_nodes):
            first_user = n
            break
    last_size = sizes[-1] if sizes else None
    anchor = first_user
    with g.inserting_before(anchor):
        call = g.call_module(name, tuple(sizes))
        outs = [g.call_function(operator.getitem, (call, i)) for i in range(len(frontier))]
    for n, o in zip(frontier, outs):
        o.meta = dict(n.meta)
        for u in list(n.users):
            if u not in hoist:
                u.replace_input_with(n, o)
    # the sizes must be defined in front of the call
    for sz in sizes:
        if not _before(g, sz, call):
            call.prepend(sz)
    p.erase([n for n in reversed(list(g.nodes)) if n in need and n not in sizes])
    p.counts["shape_only_nodes"] = sum(1 for n in need if n.op == "call_function")


def _before(g, a, b):
    for n in g.nodes:
        if n is a:
            return True
        if n is b:
            return False
    return False


def _copy_graph(graph):
    g = torch.fx.Graph()
    out = g.graph_copy(graph, {})
    g.output(out)
    g._codegen = copy.deepcopy(graph._codegen)                # the exporter's calling convention (keyword arguments, output pytree)
    return g


# The fused helper modules re-register Parameters and fake-quantizer modules the prepared graph already owns (shared objects, so
# calibration state and `.to()` carry over).  Their `_qt_*` names must not leak into checkpoints: a state_dict of a fused graph has
# exactly the keys of the plain prepared graph (upstream's format), and loading one -- strict -- into either form works.
def _drop_helper_keys(module, state_dict, prefix, local_metadata):
    for k in [k for k in state_dict if k[len(prefix):].startswith("_qt_")]:
        del state_dict[k]
    return state_dict


def _forgive_helper_keys(module, incompatible):
    incompatible.missing_keys[:] = [k for k in incompatible.missing_keys if not k.startswith("_qt_")]


def _install_state_dict_hooks(model):
    if "_qt_state_hooks" in model.__dict__:
        return
    model.__dict__["_qt_state_hooks"] = (model._register_state_dict_hook(_drop_helper_keys),
                                         model.register_load_state_dict_post_hook(_forgive_helper_keys))


def _remove_state_dict_hooks(model):
    for h in model.__dict__.pop("_qt_state_hooks", ()):
        h.remove()


def unfuse_prepared_graph(model: GraphModule):
    """Put back the graph `fuse_prepared_graph` started from (fake-quantizer modules and Parameters are shared, so calibration state
    carries over) and drop the fused modules.  No-op on a graph that was not fused."""
    plain = model.__dict__.pop("_qt_unfused_graph", None)
    if plain is None:
        return False
    model.graph = plain
    model.recompile()
    for name in [n for n, _ in model.named_children() if n.startswith("_qt_")]:
        delattr(model, name)
    _remove_state_dict_hooks(model)
    return True


def fuse_prepared_graph(model: GraphModule):
    """Rewrite the chains listed in the module docstring of a prepared graph in place; returns the number of rewrites by kind.
    Idempotent; safe on any graph (unrecognised chains are left alone).  `QT_PT2E_FUSE=0` turns it into a no-op."""
    if os.environ.get("QT_PT2E_FUSE", "1") == "0" or "_qt_unfused_graph" in model.__dict__:
        return {}
    model.__dict__["_qt_unfused_graph"] = _copy_graph(model.graph)
    p = _Pass(model)
    by_input = p.linears()
    p.mlps(by_input)
    p.siblings(by_input)
    p.norms()
    p.attentions()
    _loss(p)
    _hoist_shape_only(p)
    model.graph.lint()
    model.recompile()
    _install_state_dict_hooks(model)
    return p.counts
