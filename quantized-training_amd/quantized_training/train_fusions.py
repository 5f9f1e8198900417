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
in another (fixed) order than qt_colsum_bf16's.

PRODUCER kernels (second half of the file) go one step further: LayerNorm, GELU, softmax and the whole attention core of a training
step as autograd Functions on kernels that compute torch's arithmetic with torch's rounding points and evaluate the fake-quantizer
calls around them in their own launch (_LayerNormTrainFn, _GeluTrainFn, _SoftmaxTrainFn, _AttentionTrainFn); the gradients that meet
at a LayerNorm's output are summed by that LayerNorm's backward (take_deferred, _fanin); nn.Embedding's weight gradient is torch's
bit for bit (_EmbeddingTrainFn).  One debug mask, QT_TRAIN_DEBUG (an integer, default 0 = every fusion on; DEBUG_BITS below), switches
fusions OFF for A/B runs and tests -- rounds 4-5 had grown seven separate switches."""
import ctypes
import os
import weakref

import torch

from . import _native

__all__ = ["plan", "unplan", "enabled", "STATS"]


class _Counters:
    chains = 0            # chain launches
    members = 0           # fake-quantizer calls served by a chain launch (heads included)
    colsums = 0           # bias gradients handed over
    colsum_fallbacks = 0  # bias gradients a Linear summed itself (qt_colsum_bf16: no launch had left them)
    misses = 0            # members that received another tensor than predicted
    attention = 0         # attention-core launches (forward or backward), each standing for 4 / 2 fake-quantizer calls
    fanins = 0            # gradient fan-in launches (qt_grad_fanin_bf16)
    deferred = 0          # backward fake-quantizer calls evaluated inside a fan-in launch
    embeddings = 0        # embedding weight gradients by qt_embedding_backward_bf16
    addlns = 0            # residual adds formed inside a LayerNorm launch
    qkv_groups = 0        # query / key / value backward products launched as one dgrad + one wgrad launch
    qkv_forward_groups = 0   # query / key / value forward products launched together
    missed = []           # ... their names and what differed (the first few)

    @classmethod
    def reset(cls):
        cls.chains = cls.members = cls.colsums = cls.colsum_fallbacks = cls.misses = cls.attention = cls.fanins = cls.deferred = cls.embeddings = cls.addlns = cls.qkv_groups = cls.qkv_forward_groups = 0
        cls.missed = []


STATS = _Counters


# QT_TRAIN_DEBUG: bits that switch a fusion OFF (values never change: every fusion computes what the launches it replaces compute).
#   1 chains     no chained fake-quantizer launches at all (implies everything below: they are built on chains)
#   2 colsum     bias gradients by qt_colsum_bf16 instead of riding on the chain launches
#   4 producers  torch's own LayerNorm / GELU / softmax kernels (implies attention, fanin, embedding, addln)
#   8 attention  the attention core as its sub-modules (library GEMMs, qt_softmax_*)
#  16 fanin      one fake-quantizer launch and one add per gradient arriving at a LayerNorm output (implies addln)
#  32 embedding  torch's embedding_dense_backward
#  64 addln      the residual add in front of a LayerNorm as its own launch
# 128 qkvgemm    the input and weight gradients of query / key / value as six launches instead of two
# 256 optimizer  torch's own clip_grad_norm_ + optimizer.step() launches (optim.py)
# 512 pairgemm   a Linear's input and weight gradient as two launches instead of one (q / k / v: two instead of one)
# 1024 qkvfwd    the forward products of query / key / value as three launches instead of one
DEBUG_BITS = {"chains": 1, "colsum": 2, "producers": 4, "attention": 8, "fanin": 16, "embedding": 32, "addln": 64, "qkvgemm": 128, "optimizer": 256,
              "pairgemm": 512, "qkvfwd": 1024}


def _on(name):
    try:
        mask = int(os.environ.get("QT_TRAIN_DEBUG", "0") or "0", 0)
    except ValueError:
        raise ValueError(f"QT_TRAIN_DEBUG={os.environ.get('QT_TRAIN_DEBUG')!r}: an integer mask of {DEBUG_BITS}") from None
    return not (mask & DEBUG_BITS[name])


def enabled():
    return _on("chains")


class Chain:
    """members: [(fake-quantizer, src)] -- src -1: the head's input, else the index of the member whose result it reads.
    colsum: (member index, Linear) -- that member's result is the Linear's grad_output: its column sums are the bias gradient."""

    def __init__(self, members, colsum=None, name=""):
        self.members = members
        self.colsum = colsum
        self.name = name


_COLSUM = {}              # (data_ptr, version, shape) of a grad_output -> (that tensor, its column sums): one-shot, taken by the Linear's backward


def put_colsum(g, gb):
    """Leaves the column sums of `g` for the Linear whose backward receives `g`.  The entry holds `g` itself: while it exists the address
    cannot be handed to another tensor, so a key can only ever match the tensor it was made for (an entry nobody takes -- a chain miss, a
    Linear that did not go through _LinearColsumBias -- used to outlive its tensor and could match a later gradient at the same address).
    More than 64 entries: the OLDEST go (never all of them); a Linear whose entry went computes its own sums."""
    while len(_COLSUM) >= 64:
        del _COLSUM[next(iter(_COLSUM))]
    _COLSUM[(g.data_ptr(), g._version, tuple(g.shape))] = (g, gb)


_LINEAR_GRADS = {}        # (data_ptr, version, shape) of a grad_output -> (that tensor, identity of x and Wq, grad_input, grad_weight): one-shot, like _COLSUM


_ATTN_ACTIVE = []         # the self-attention modules whose forward is running (innermost last)


def _attn_enter(module, args, kwargs=None):
    _ATTN_ACTIVE.append(module)


def _attn_leave(module, args, output):
    from .modules.qat.linear import flush_forward
    if _ATTN_ACTIVE and _ATTN_ACTIVE[-1] is module:
        _ATTN_ACTIVE.pop()
    flush_forward()                    # (nothing a projection left pending outlives the block that called it)


def attention_block_active(owner_ref):
    """True while the forward of the self-attention block `owner_ref` (a weak reference) is running."""
    owner = owner_ref() if owner_ref is not None else None
    return owner is not None and bool(_ATTN_ACTIVE) and _ATTN_ACTIVE[-1] is owner


def _mark_qkv_members(attn, lins):
    """The three projections in front of an attention core that runs as _AttentionTrainFn: from the next step on their forward products
    go out as one launch (modules/qat/linear.py, flush_forward) -- but only INSIDE the forward of this very block (hooks on the block say
    when): Hugging Face's self-attention forward only reshapes the three results before it hands them to the attention function, whereas a
    projection called from anywhere else must return a written tensor.  Only QAT Linears of one shape without forward hooks of their own
    (a hook would see the output before the launch that writes it)."""
    import weakref
    from .modules.qat.linear import Linear as QATLinear
    if not all(type(l) is QATLinear and not l._forward_hooks and l.weight.shape == lins[0].weight.shape and (l.bias is None) == (lins[0].bias is None)
               for l in lins):
        return
    if not attn.__dict__.get("_qt_qkv_hooks", False):
        attn.register_forward_pre_hook(_attn_enter)
        attn.register_forward_hook(_attn_leave, always_call=True)
        attn.__dict__["_qt_qkv_hooks"] = True
    ref = weakref.ref(attn)
    for l in lins:
        l.__dict__["_qt_qkv_member"] = ref


def group_qkv_backward(lins, gys):
    """The attention backward has just produced the (fake-quantized) grad_outputs of the query / key / value projections: their three
    input gradients gy . Wq and three weight gradients gy^T . x go out as TWO launches of qt_train_gemm_bf16 (three problems each: 17 / 19 us
    against 3 x 8 / 3 x 11) instead of six when the three Linear nodes run; each node then finds its pair (take_linear_grads), checked
    against the very x and Wq it saved.  Same tiles, same order of additions as the single launches: bit-identical."""
    from .modules.qat.linear import train_gemm_backward, train_gemm_group
    if not _on("qkvgemm") or len(lins) != 3 or len(gys) != 3:
        return False
    xs, ws, g2 = [], [], []
    for lin, gy in zip(lins, gys):
        xw = lin.__dict__.get("_qt_train_xw") if lin is not None else None
        if xw is None or gy is None:
            return False
        x, w = xw[0](), xw[1]()
        if x is None or w is None:
            return False
        if not (x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and w.is_contiguous() and gy.is_contiguous()
                and gy.shape[:-1] == x.shape[:-1] and gy.shape[-1] == w.shape[0] and w.shape[1] == x.shape[-1]):
            return False
        xs.append(x.reshape(-1, x.shape[-1])); ws.append(w); g2.append(gy.reshape(-1, gy.shape[-1]))
    both = train_gemm_backward(g2, ws, xs, "q/k/v ")              # all six products in ONE launch
    if both is not None:
        gxs, gws = both
    else:
        gxs = train_gemm_group(g2, ws, False, True, "dgrad q/k/v")
        if gxs is None:
            return False
        gws = train_gemm_group(g2, xs, True, True, "wgrad q/k/v")
        if gws is None:
            return False
    for lin, gy, x2, w, gx, gw in zip(lins, gys, xs, ws, gxs, gws):
        x = lin.__dict__["_qt_train_xw"][0]()
        while len(_LINEAR_GRADS) >= 16:
            del _LINEAR_GRADS[next(iter(_LINEAR_GRADS))]
        # (x and Wq are remembered by address and version, not by reference: Wq carries the step's autograd graph, and a reference that
        # outlives a stream capture's end broke the graph's instantiation)
        _LINEAR_GRADS[(gy.data_ptr(), gy._version, tuple(gy.shape))] = (gy, (x.data_ptr(), x._version, tuple(x.shape), w.data_ptr(), w._version),
                                                                          gx.view(x.shape), gw)
    STATS.qkv_groups += 1
    return True


def take_linear_grads(gy, x, w):
    hit = _LINEAR_GRADS.pop((gy.data_ptr(), gy._version, tuple(gy.shape)), None)
    if hit is None:
        return None
    _, ident, gx, gw = hit
    if ident != (x.data_ptr(), x._version, tuple(x.shape), w.data_ptr(), w._version):
        return None                    # (another forward of the same Linear in between: the node multiplies what IT saved)
    return gx, gw


def take_colsum(g):
    hit = _COLSUM.pop((g.data_ptr(), g._version, tuple(g.shape)), None)
    if hit is None:
        return None
    STATS.colsums += 1
    return hit[1]


def _member_ok(fq, device):
    from .fake_quantize import FusedAmaxObsFakeQuantize
    from .quantizer.quantizer import QScheme
    return (isinstance(fq, FusedAmaxObsFakeQuantize) and fq._quantize and not fq.is_per_channel and not fq.record_histogram
            and fq.outlier_threshold is None and fq.qscheme in (None, QScheme.PER_TENSOR_SYMMETRIC)
            and not getattr(fq, "_emit_fp8", None)      # (hooks on a member still see its call, input and result: forward() runs for every member)
            and (not fq._observe or (fq.amax_history.numel() > 0 and fq.amax_history.dim() == 1 and fq.amax_history.device == device))
            and fq.scale.numel() == 1 and fq.scale.device == device and fq.scale.dtype == torch.float32)


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
    colsum = chain.colsum if _on("colsum") else None
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
            put_colsum(g, gb)
        return outs

    with torch.no_grad():
        outs = launch(X)
    STATS.chains += 1
    STATS.members += 1
    _Stats.add(X.numel())                                      # the head's own call
    for i, (fq, src) in enumerate(members):
        if i == 0:
            continue
        want = X if src < 0 else outs[src]
        fq.__dict__["_qt_chain_result"] = (want.data_ptr(), want._version, tuple(want.shape), outs[i], want, False)
    if need_grad:
        # every member keeps an autograd node of its own (the straight-through gradient, fake_quantize.py:250-252), as in the unchained
        # path: the engine then adds the members' gradients into x in the very order it would have, bit for bit
        from .fake_quantize import _PrecomputedFakeQuant
        return _PrecomputedFakeQuant.apply(X, outs[0])
    return outs[0]


def take_member_result(fq, X):
    """Called at the top of a fake-quantizer's forward: the result a chain launch left for THIS call, or None.  A call that received
    another tensor than the chain predicted drops the speculative amax and proceeds on its own."""
    pend = fq.__dict__.get("_qt_chain_result")
    if pend is None:
        return None
    fq.__dict__["_qt_chain_result"] = None
    ptr, version, shape, out, _keep, connected = pend[:6]
    arm = pend[6] if len(pend) > 6 else None
    from .fake_quantize import _Stats, _take_preupdate, _PrecomputedFakeQuant
    if X.data_ptr() == ptr and X._version == version and tuple(X.shape) == shape and (X.is_contiguous() or X.stride() == _keep.stride()):
        STATS.members += 1
        _Stats.add(X.numel())
        if fq._observe:
            _take_preupdate(fq.amax_history)                   # the batched scale update (or the chain's own) served this call
        if arm is not None:
            # this call's result is an output of the producing kernel's own autograd node (`connected`): what comes back for it goes to
            # that node alone, so the consuming Linear's backward quantizer may leave its call to the node's fan-in launch (take_deferred)
            arm[0].__dict__["_qt_deferred"] = (tuple(X.shape), arm[1])
        if not connected and torch.is_grad_enabled() and X.requires_grad:
            return _PrecomputedFakeQuant.apply(X, out)         # the straight-through gradient of this call (fake_quantize.py:250-252)
        return out
    STATS.misses += 1
    if len(STATS.missed) < 8:
        STATS.missed.append((getattr(fq, "name", "?"), X.data_ptr() == ptr, X._version == version, tuple(X.shape), shape, X.is_contiguous()))
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


_PLAN_WARNED = False


def unplan(model):
    from .fake_quantize import FusedAmaxObsFakeQuantize
    for m in model.modules():
        if isinstance(m, FusedAmaxObsFakeQuantize):
            m.__dict__.pop("_qt_chain", None)
            m.__dict__.pop("_qt_chain_result", None)
        elif isinstance(m, torch.nn.LayerNorm):
            m.__dict__.pop("_qt_grad_head", None)


def plan(model):
    """Attaches chains to the fake-quantizers of `model` from its module structure; returns how many.  Call it once the lazily
    created fake-quantizers exist (after a first training step); idempotent.
      * output blocks `LayerNorm(residual(dropout(dense(h)), x))` (modules/quantizable/attention.py::_bert_output_forward; upstream
        modeling_bert.py:174-214), backward: residual.error_pre_process[0] -> residual.error_post_process[0], [1] and -- with inactive
        dropout, whose backward otherwise sits in between -- -> dense.error_pre_process[0] (+ the dense layer's bias gradient);
      * every other QAT Linear with a backward-pre quantizer: that call + the bias gradient;
      * attention blocks whose query / key / value read one tensor, forward: the three input quantizers."""
    from .modules.qat.linear import Linear as QATLinear
    unplan(model)
    if not enabled():
        return 0
    # the module layout the chains are read from: output blocks are this package's twins (modules/quantizable/attention.py) with
    # .dense / .residual / .LayerNorm, attention blocks carry .query / .key / .value beside .qk_matmul / .av_matmul.  A tree that has the
    # twins but not those names (another transformers layout, a model edited after quantize()) gets NO chains and one warning -- never
    # chains planned from half of a block.
    problems = []
    for name, mod in model.named_modules():
        if getattr(type(mod), "_qt_twin", False) and hasattr(mod, "residual"):
            missing = [a for a in ("dense", "LayerNorm") if not isinstance(getattr(mod, a, None), torch.nn.Module)]
            if missing:
                problems.append(f"{name} ({type(mod).__name__}) has no sub-module {missing}")
        if hasattr(mod, "qk_matmul") != hasattr(mod, "av_matmul") or (
                hasattr(mod, "qk_matmul") and hasattr(mod, "query") and not all(hasattr(mod, a) for a in ("key", "value"))):
            problems.append(f"{name} ({type(mod).__name__}) is not a query / key / value + qk_matmul / av_matmul attention block")
    if problems:
        global _PLAN_WARNED
        if not _PLAN_WARNED:
            _PLAN_WARNED = True
            import logging
            logging.getLogger(__name__).warning("quantized_training: training-step launch fusions are OFF for this model, its module layout "
                                                "is not the one they are planned from: %s", "; ".join(problems[:4]))
        return 0
    n = 0
    chained = set()
    for mod in model.modules():
        dense, res, ln = getattr(mod, "dense", None), getattr(mod, "residual", None), getattr(mod, "LayerNorm", None)
        if isinstance(dense, QATLinear) and res is not None and ln is not None and getattr(type(mod), "_qt_twin", False):
            drop = getattr(mod, "dropout", None)
            dropping = drop is not None and getattr(drop, "p", 0.0) != 0.0      # the dropout's backward then sits between the add and the dense layer
            pre = _fq(getattr(res, "error_pre_process", None), "0")
            p0, p1 = _fq(getattr(res, "error_post_process", None), "0"), _fq(getattr(res, "error_post_process", None), "1")
            dpre = _fq(getattr(dense, "error_pre_process", None), "0")
            if pre is None or len(getattr(res, "error_pre_process", {})) != 1:
                continue
            members = [(pre, -1)]
            colsum = None
            if p0 is not None and p1 is not None and len(res.error_post_process) == 2:
                members += [(p0, 0), (p1, 0)]
                if not dropping and dpre is not None and len(dense.error_pre_process) == 1:
                    members.append((dpre, 1))
                    colsum = (3, dense)
            if len(members) > 1:
                pre.__dict__["_qt_chain"] = Chain(members, colsum, name=type(mod).__name__)
                chained.update(id(f) for f, _ in members)
                ln.__dict__["_qt_grad_head"] = pre             # the gradient of this LayerNorm's input is exactly that chain's input
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


# ---- producer kernels of a training step that evaluate the chain behind them in their own launch ---------------------------------
def producers_enabled():
    return enabled() and _on("producers")


def _members_format(members, dev):
    """The one launch format of all members (same dtype, same map), or None."""
    from .fake_quantize import _launch_format
    if not members or not all(_member_ok(fq, dev) for fq, _ in members):
        return None
    head = members[0][0]
    for fq, _ in members:
        fq._move_to(dev)
    fmt0 = _launch_format(head._qt_format, head.qmap)
    if any(_launch_format(fq._qt_format, fq.qmap).key() != fmt0.key() or str(fq.dtype) != str(head.dtype) for fq, _ in members[1:]):
        return None
    if fmt0.kind == _native.QT_FMT_LUT and not (fmt0.p1 & 1):
        return None
    if fmt0.kind not in (_native.QT_FMT_LUT, _native.QT_FMT_FP_SAT, _native.QT_FMT_INT):
        return None
    return fmt0


def _stages(members, like, st):
    """(ctypes stage array, result tensors): every member's delayed-scaling update is issued (unless the step's batched update did it)."""
    from .fake_quantize import launch_scale_update
    outs = [torch.empty_like(like) for _ in members]
    stages = (_native.QtChainStage * len(members))()
    for i, (fq, src) in enumerate(members):
        if fq._observe:
            launch_scale_update(fq.amax_history, fq.scale, fq.quant_max, fq.force_scale_power_of_two, st)
        stages[i].scale_f32_dev = fq.scale.data_ptr()
        stages[i].amax_bits_dev = fq.amax_history.data_ptr() if fq._observe else None
        stages[i].out_dev = outs[i].data_ptr()
        stages[i].src = src
    return stages, outs


def _hand_over(members, produced, outs, connected=False, arm=None):
    """Leaves every member's result for its call: member i will be called on `produced` (src -1) or on member src's result.
    connected: the results are outputs of the producer's autograd node (no straight-through node is added when they are handed out);
    arm[i]: the backward fake-quantizer to arm for a deferred call once member i took its result (take_deferred)."""
    for i, (fq, src) in enumerate(members):
        want = produced if src < 0 else outs[src]
        fq.__dict__["_qt_chain_result"] = (want.data_ptr(), want._version, tuple(want.shape), outs[i], want, connected, arm[i] if arm else None)
    STATS.chains += 1


class _Token:
    __slots__ = ("__weakref__",)


_PENDING = {}             # data_ptr of a placeholder -> (the gradient a deferred backward fake-quantizer call received, the quantizer, the
#                           placeholder -- held, so its address is not reused while the entry exists --, weak reference to the graph's token)
_HOLDS = weakref.WeakValueDictionary()      # data_ptr -> every placeholder still alive, to tell one that lost its entry from a real gradient


def fanin_enabled():
    return producers_enabled() and _on("fanin")


def take_deferred(fq, X):
    """Called at the top of a fake-quantizer's forward when a producer launch armed it (`_qt_deferred`, one shot): this is the backward
    quantizer of a Linear whose input is an output of a LayerNorm's autograd node, called by the Linear's backward hook on grad_input
    (quantize.py:147-148 upstream).  The call's bookkeeping happens here -- scale update, counters -- and its result is a placeholder
    that only that node's backward will see: its fan-in launch (qt_grad_fanin_bf16) evaluates the call on the way into the sum the
    engine would have formed.  Anything unexpected: None, and the call proceeds as usual."""
    armed = fq.__dict__.pop("_qt_deferred", None)
    if armed is None or armed[1]() is None or torch.is_grad_enabled() or not fanin_enabled():
        return None                    # (armed[1]: weak reference to the producing node's token -- a graph that was dropped arms nothing)
    shape = armed[0]
    if not (X.is_cuda and X.dtype == torch.bfloat16 and X.is_contiguous() and tuple(X.shape) == shape and X.numel() % 8 == 0 and X.data_ptr() % 16 == 0
            and not _hooked(fq) and _member_ok(fq, X.device) and _members_format([(fq, -1)], X.device) is not None):
        return None
    from .fake_quantize import _Stats, _stream_ptr, launch_scale_update
    if fq._observe:
        launch_scale_update(fq.amax_history, fq.scale, fq.quant_max, fq.force_scale_power_of_two, _stream_ptr(X))
    STATS.members += 1
    STATS.deferred += 1
    _Stats.add(X.numel())
    hold = torch.empty_like(X)
    if len(_PENDING) > 64:
        # only entries whose autograd graph is gone (their token died: nothing can deliver those placeholders any more); a placeholder of
        # a backward that is still running is never dropped -- it is uninitialised memory that only its entry can turn into a gradient
        for k in [k for k, v in _PENDING.items() if v[3]() is None]:
            del _PENDING[k]
    _PENDING[hold.data_ptr()] = (X, fq, hold, armed[1])
    _HOLDS[hold.data_ptr()] = hold
    return hold


def _fanin_items(arrivals, dev):
    """(items, fmt, lut): the arrivals as (tensor, fake-quantizer or None) -- a deferred call's placeholder becomes the gradient the call
    received and its quantizer --, the one launch format of the quantized ones (None: nothing to quantize) and its map pointer."""
    from .fake_quantize import _hip_fake_quant, _launch_format
    items = []
    for g in arrivals:
        pend = _PENDING.pop(g.data_ptr(), None)
        if pend is not None and pend[2].numel() == g.numel():
            items.append((pend[0], pend[1], pend[2]))
        else:
            if pend is None and _HOLDS.get(g.data_ptr()) is g:
                raise RuntimeError("train_fusions: a deferred fake-quantizer call's placeholder arrived without its pending entry "
                                   "(it holds no gradient); set QT_TRAIN_DEBUG=16 (no fan-in launches) and report this")
            items.append((g.contiguous(), None, None))
    quant = [(fq, -1) for _, fq, _ in items if fq is not None]
    fmt = _members_format(quant, dev) if quant else None
    if quant and fmt is None:
        # (deferred calls of different formats: each as its own launch after all; their updates and counts are done)
        for i, (raw, fq, hold) in enumerate(items):
            if fq is not None:
                f1 = _launch_format(fq._qt_format, fq.qmap)
                _hip_fake_quant(raw, hold, f1, fq.qmap, fq.scale, fq.amax_history if fq._observe else None, False, None)
                items[i] = (hold, None, None)
        quant = []
    return [(raw, fq) for raw, fq, _ in items], fmt, (_lut_ptr(quant[0][0], fmt) if quant else None)


def _fanin_array(part):
    arr = (_native.QtFaninItem * len(part))()
    for i, (raw, fq) in enumerate(part):
        arr[i].x_dev = raw.data_ptr()
        arr[i].fq = 1 if fq is not None else 0
        arr[i].scale_f32_dev = fq.scale.data_ptr() if fq is not None else None
        arr[i].amax_bits_dev = fq.amax_history.data_ptr() if fq is not None and fq._observe else None
        arr[i].out_dev = None
    return arr


def _fanin(first, arrivals, prepared=None):
    """first + arrivals in order, as the engine adds them (bf16 adds); an arrival that is a deferred call's placeholder is evaluated on
    the way (its fake-quantizer on the gradient the call received).  One launch (per four arrivals)."""
    from .fake_quantize import _stream_ptr, _hip_fake_quant, _launch_format
    items, fmt, lut = prepared if prepared is not None else _fanin_items(arrivals, first.device)
    first = first.contiguous()
    if not (first.dtype == torch.bfloat16 and first.numel() % 8 == 0 and first.data_ptr() % 16 == 0
            and all(t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() == first.numel() and t.data_ptr() % 16 == 0 for t, _ in items)):
        # something the launch would refuse (an offset view, another dtype): the launches it stands for, one by one
        for raw, fq in items:
            if fq is not None:
                y = torch.empty_like(raw)
                _hip_fake_quant(raw.contiguous(), y, _launch_format(fq._qt_format, fq.qmap), fq.qmap, fq.scale, fq.amax_history if fq._observe else None, False, None)
                raw = y
            first = first + raw.view_as(first)
        return first
    if fmt is None:
        fmt = _native.QtFormat(_native.QT_FMT_FP_SAT, 2, -14, 0.0, 57344.0)       # (no quantized item: the format is not read)
    out = torch.empty_like(first)
    L = _native.lib()
    st = _stream_ptr(first)
    pos = 0
    while pos < len(items):
        part = items[pos:pos + 4]
        _native.check(L.qt_grad_fanin_bf16(first.data_ptr(), _fanin_array(part), len(part), out.data_ptr(), first.numel(), ctypes.byref(fmt), lut, st),
                      "qt_grad_fanin_bf16")
        STATS.fanins += 1
        first = out
        pos += 4
    return out


def _lut_ptr(head, fmt):
    return head.qmap.data_ptr() if fmt.kind == _native.QT_FMT_LUT else None


def _grad_chain(head):
    """(members, colsum) of the chain that starts at the backward quantizer `head`, validated for `head`'s device; (None, None) else."""
    chain = head.__dict__.get("_qt_chain") if head is not None else None
    if chain is None or chain.members[0][0] is not head:
        return None, None
    colsum = chain.colsum if _on("colsum") else None
    if colsum is not None and (colsum[1].bias is None or not colsum[1].bias.requires_grad):
        colsum = None
    return chain.members, colsum


def _float_scratch(kind, nbytes, dev):
    from .fused import splitk_scratch
    return splitk_scratch(kind, nbytes, 0, dev)[0]


class _LayerNormTrainFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension of a bf16 device tensor inside a training step: qt_layernorm_train_bf16 evaluates the input
    quantizers of the Linears that read the result in its launch; the backward (qt_layernorm_train_backward_bf16) evaluates the
    gradient chain of the residual add in front of it, and the dense layer's bias gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, consumers, grad_head, posts=None, addends=None):
        from .fake_quantize import _stream_ptr
        cols = x.shape[-1]
        rows = x.numel() // cols
        st = _stream_ptr(x)
        members = [(fq, -1) for fq in consumers]
        fmt = _members_format(members, x.device)
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        stages, outs = _stages(members, x, st)
        # addends (h, r): x is the residual add's result whose values nobody has computed yet (functional_modules._LazyAdd) -- this launch
        # forms bf16(h + r), stores it into x's memory (the backward reads it there) and normalises it
        src, res = (addends[0], addends[1]) if addends is not None else (x, None)
        _native.check(_native.lib().qt_layernorm_train_bf16(src.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                            rstd.data_ptr(), rows, cols, float(eps), stages, len(members), ctypes.byref(fmt),
                                                            _lut_ptr(members[0][0], fmt), res.data_ptr() if res is not None else None,
                                                            x.data_ptr() if res is not None else None, st), "qt_layernorm_train_bf16")
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.grad_head = grad_head
        ctx.fan = posts is not None
        if posts is None:
            _hand_over(members, y, outs)
            return y
        ctx.token = _Token()
        ref = weakref.ref(ctx.token)
        # the consumers' results are outputs of THIS node: what comes back for each arrives here separately, and the sum the engine
        # would have formed one launch per arrival is formed in one (`_fanin`), the consumers' deferred backward quantizers on the way
        _hand_over(members, y, outs, connected=True, arm=[(p, ref) if p is not None else None for p in posts])
        return (y,) + tuple(outs)

    @staticmethod
    def backward(ctx, dy, *gouts):
        from .fake_quantize import _stream_ptr
        x, weight, bias, mean, rstd = ctx.saved_tensors
        cols = x.shape[-1]
        rows = x.numel() // cols
        pad = (None, None) if ctx.fan else ()
        arrivals = [g for g in reversed(gouts) if g is not None]      # the engine's order: the consumers last to first, after dy
        members, colsum = _grad_chain(ctx.grad_head)
        fmt = _members_format(members, x.device) if members is not None else None
        ride = None
        if arrivals:
            if dy is None:
                dy = torch.zeros_like(x)
            prepared = _fanin_items(arrivals, x.device)
            # the sum is formed by the backward kernel itself while it loads dy -- when that kernel runs (a gradient chain behind this
            # LayerNorm), the arrivals' quantizers share its format and there are at most three; else by a launch of its own
            if (fmt is not None and len(prepared[0]) <= 3 and dy.dtype == torch.bfloat16 and dy.is_contiguous() and dy.data_ptr() % 16 == 0
                    and (prepared[1] is None or prepared[1].key() == fmt.key()) and all(t.data_ptr() % 16 == 0 for t, _ in prepared[0])):
                ride = prepared[0]
            else:
                dy = _fanin(dy, arrivals, prepared)
        dy = dy.contiguous()
        if fmt is None or dy.dtype != torch.bfloat16 or dy.data_ptr() % 16:
            gx, gw, gb = torch.ops.aten.native_layer_norm_backward(dy, x, [cols], mean.view(*x.shape[:-1], 1), rstd.view(*x.shape[:-1], 1), weight, bias,
                                                                   [True, True, True])
            return (gx, gw, gb, None, None, None) + pad
        if colsum is not None and colsum[1].out_features != cols:
            colsum = None
        L = _native.lib()
        st = _stream_ptr(x)
        dx = torch.empty_like(x)
        gw = torch.empty(cols, dtype=torch.bfloat16, device=x.device)
        gb = torch.empty(cols, dtype=torch.bfloat16, device=x.device)
        gbias = torch.empty(cols, dtype=torch.bfloat16, device=x.device) if colsum is not None else None
        stages, outs = _stages(members, x, st)
        pbytes = L.qt_layernorm_train_backward_groups(rows) * 3 * cols * 4
        part = _float_scratch("lnbwd", pbytes, x.device)
        _native.check(L.qt_layernorm_train_backward_bf16(dy.data_ptr(), x.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                                         rows, cols, stages, len(members), ctypes.byref(fmt), _lut_ptr(members[0][0], fmt),
                                                         colsum[0] if colsum is not None else -1, part.data_ptr(), part.numel() * 4, gw.data_ptr(),
                                                         gb.data_ptr(), gbias.data_ptr() if gbias is not None else None,
                                                         _fanin_array(ride) if ride else None, len(ride) if ride else 0, st),
                      "qt_layernorm_train_backward_bf16")
        if ride:
            STATS.fanins += 1
        _hand_over(members, dx, outs)
        if gbias is not None:
            g = outs[colsum[0]]
            put_colsum(g, gbias)
        return (dx, gw, gb, None, None, None) + pad


def _layernorm_plan(norm, x):
    """(consumers' input quantizers, their Linears' deferrable backward quantizers) when `norm(x)` of a training step can go through
    _LayerNormTrainFn, or None: bf16 device tensors under grad mode, rows of at most 1024 columns, and at least one consuming Linear whose
    input quantizer can ride on the launch (`_qt_consumers`, set by model_fusions.apply_bert_fusions)."""
    if not (producers_enabled() and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous()
            and type(norm) is torch.nn.LayerNorm and norm.elementwise_affine and norm.bias is not None and len(norm.normalized_shape) == 1
            and norm.weight.dtype == torch.bfloat16 and x.shape[-1] == norm.normalized_shape[0] and x.shape[-1] % 8 == 0 and x.shape[-1] <= 1024
            and x.data_ptr() % 16 == 0 and not norm._forward_hooks and not norm._forward_pre_hooks and not norm._backward_hooks):
        return None
    after = norm.__dict__.get("_qt_dropout_after")
    if after is not None and after.training and after.p > 0.0:
        return None                                            # (the consumers read dropout(norm(x)), not this launch's result)
    consumers, posts = [], []
    for lin in norm.__dict__.get("_qt_consumers") or []:
        for f in (getattr(lin, "error_post_process", None) or {}).values():
            f.__dict__.pop("_qt_deferred", None)               # (nothing armed by an earlier forward survives this one)
        holder = getattr(lin, "activation_pre_process", None)
        fq = _fq(holder, "0")
        if fq is None or len(holder) != 1:
            return None
        consumers.append(fq)
        # the consumer's backward quantizer on its grad_input (quantize.py:147-148, `--quantize_backprop ...,residual`), if it is the only
        # thing hooked onto the Linear's backward: its call may be deferred to this node's fan-in launch
        ph = getattr(lin, "error_post_process", None)
        post = _fq(ph, "0") if ph is not None and len(ph) == 1 and len(lin._backward_hooks) == 1 else None
        posts.append(post if post is not None and _member_ok(post, x.device) and not _hooked(post) else None)
    if not consumers or len(consumers) > 4 or _members_format([(f, -1) for f in consumers], x.device) is None:
        return None
    return consumers, posts


def layernorm_or_none(norm, x):
    """`norm(x)` of a training step through _LayerNormTrainFn, or None (the caller runs the module)."""
    plan_ = _layernorm_plan(norm, x)
    if plan_ is None:
        return None
    consumers, posts = plan_
    if fanin_enabled() and x.requires_grad:
        return _LayerNormTrainFn.apply(x, norm.weight, norm.bias, norm.eps, consumers, norm.__dict__.get("_qt_grad_head"), posts, None)[0]
    return _LayerNormTrainFn.apply(x, norm.weight, norm.bias, norm.eps, consumers, norm.__dict__.get("_qt_grad_head"))


def add_layernorm_or_none(block, h, r):
    """`block.LayerNorm(block.residual(h, r))` of a training step with the add inside the LayerNorm launch, or None.  The residual module
    is still CALLED -- its backward hooks (the gradient chain of `plan`) hang on that call -- but told to leave the sum's values to the
    launch (functional_modules._LazyAdd); nothing may hook its forward or read the sum in between."""
    from .modules.quantizable.functional_modules import AddFunctional
    norm, res = getattr(block, "LayerNorm", None), getattr(block, "residual", None)
    if not (fanin_enabled() and _on("addln") and type(res) is AddFunctional and norm is not None
            and h.shape == r.shape and h.dtype == r.dtype and r.is_cuda and r.is_contiguous() and r.data_ptr() % 16 == 0 and h.requires_grad
            and not res._forward_hooks and not res._forward_pre_hooks and getattr(res, "activation_pre_process", None) is None):
        return None
    plan_ = _layernorm_plan(norm, h)
    if plan_ is None:
        return None
    consumers, posts = plan_
    res.__dict__["_qt_lazy_add"] = True
    try:
        s_ = res(h, r)
    finally:
        res.__dict__.pop("_qt_lazy_add", None)
    STATS.addlns += 1
    return _LayerNormTrainFn.apply(s_, norm.weight, norm.bias, norm.eps, consumers, norm.__dict__.get("_qt_grad_head"), posts, (h, r))[0]


class _GeluTrainFn(torch.autograd.Function):
    """BertIntermediate's erf GELU inside a training step (qt_gelu_chain_bf16 / qt_gelu_backward_chain_bf16): the output dense's input
    quantizer rides on the forward, the intermediate dense's backward-pre quantizer and its bias gradient on the backward."""

    @staticmethod
    def forward(ctx, h, consumer, grad_head):
        from .fake_quantize import _stream_ptr
        cols = h.shape[-1]
        rows = h.numel() // cols
        st = _stream_ptr(h)
        members = [(consumer, -1)]
        fmt = _members_format(members, h.device)
        y = torch.empty_like(h)
        stages, outs = _stages(members, h, st)
        _native.check(_native.lib().qt_gelu_chain_bf16(h.data_ptr(), y.data_ptr(), rows, cols, stages, 1, ctypes.byref(fmt), _lut_ptr(consumer, fmt), st),
                      "qt_gelu_chain_bf16")
        _hand_over(members, y, outs)
        ctx.save_for_backward(h)
        ctx.grad_head = grad_head
        return y

    @staticmethod
    def backward(ctx, dy):
        from .fake_quantize import _stream_ptr
        (h,) = ctx.saved_tensors
        cols = h.shape[-1]
        rows = h.numel() // cols
        dy = dy.contiguous()
        members, colsum = _grad_chain(ctx.grad_head)
        fmt = _members_format(members, h.device) if members is not None else None
        if fmt is None or dy.dtype != torch.bfloat16 or dy.data_ptr() % 16:
            return torch.ops.aten.gelu_backward(dy, h, approximate="none"), None, None
        if colsum is not None and colsum[1].out_features != cols:
            colsum = None
        L = _native.lib()
        st = _stream_ptr(h)
        dx = torch.empty_like(h)
        stages, outs = _stages(members, h, st)
        gbias = ws = None
        if colsum is not None:
            ws = _chain_scratch(L.qt_fake_quant_chain_ws_bytes(rows, cols), h.device)
            gbias = torch.empty(cols, dtype=torch.bfloat16, device=h.device)
        _native.check(L.qt_gelu_backward_chain_bf16(dy.data_ptr(), h.data_ptr(), dx.data_ptr(), rows, cols, stages, len(members), ctypes.byref(fmt),
                                                    _lut_ptr(members[0][0], fmt), colsum[0] if colsum is not None else -1, _format_max(members[0][0]),
                                                    gbias.data_ptr() if gbias is not None else None, ws.data_ptr() if ws is not None else None,
                                                    ws.numel() if ws is not None else 0, st), "qt_gelu_backward_chain_bf16")
        _hand_over(members, dx, outs)
        if gbias is not None:
            g = outs[colsum[0]]
            put_colsum(g, gbias)
        return dx, None, None


def gelu_or_none(intermediate, h):
    """`intermediate_act_fn(h)` of a BertIntermediate inside a training step, or None."""
    consumer = intermediate.__dict__.get("_qt_consumer")
    holder = getattr(consumer, "activation_pre_process", None) if consumer is not None else None
    fq = _fq(holder, "0")
    if not (producers_enabled() and torch.is_grad_enabled() and h.is_cuda and h.dtype == torch.bfloat16 and h.is_contiguous() and h.dim() >= 2
            and h.shape[-1] % 8 == 0 and h.data_ptr() % 16 == 0 and fq is not None and len(holder) == 1
            and _members_format([(fq, -1)], h.device) is not None):
        return None
    dense = getattr(intermediate, "dense", None)
    head = _fq(getattr(dense, "error_pre_process", None), "0") if dense is not None else None
    return _GeluTrainFn.apply(h, fq, head)


class _SoftmaxTrainFn(torch.autograd.Function):
    """attn_scaling -> (+ mask) -> softmax of the quantizable attention blocks inside a training step (modules/quantizable/attention.py;
    upstream modeling_bert.py:142-158): qt_softmax_fq_probs_bf16 writes the probabilities and evaluates av_matmul's input quantizer on
    them; the backward (qt_softmax_backward_chain_bf16) evaluates qk_matmul's backward-pre quantizer on the gradient of the scores."""

    @staticmethod
    def forward(ctx, scores, mask, strides, scaling, fq_p, grad_head):
        from .fake_quantize import _stream_ptr, launch_scale_update
        B, H, Q, C = scores.shape
        st = _stream_ptr(scores)
        fmt = _members_format([(fq_p, -1)], scores.device)
        probs = torch.empty_like(scores)
        out = torch.empty_like(scores)
        if fq_p._observe:
            launch_scale_update(fq_p.amax_history, fq_p.scale, fq_p.quant_max, fq_p.force_scale_power_of_two, st)
        msb, msh, msq = strides
        _native.check(_native.lib().qt_softmax_fq_probs_bf16(scores.data_ptr(), mask.data_ptr() if mask is not None else None, out.data_ptr(),
                                                             probs.data_ptr(), B, H, Q, C, msb, msh, msq, float(scaling), ctypes.byref(fmt),
                                                             fq_p.qmap.data_ptr(), fq_p.scale.data_ptr(),
                                                             fq_p.amax_history.data_ptr() if fq_p._observe else None, st), "qt_softmax_fq_probs_bf16")
        _hand_over([(fq_p, -1)], probs, [out])
        ctx.save_for_backward(probs)
        ctx.scaling = float(scaling)
        ctx.grad_head = grad_head
        return probs

    @staticmethod
    def backward(ctx, dp):
        from .fake_quantize import _stream_ptr
        (probs,) = ctx.saved_tensors
        dp = dp.contiguous()
        members, _ = _grad_chain(ctx.grad_head)
        if members is None and ctx.grad_head is not None:
            members = [(ctx.grad_head, -1)]
        fmt = _members_format(members, probs.device) if members is not None else None
        if fmt is None or len(members) > 2 or dp.dtype != torch.bfloat16 or dp.data_ptr() % 16:
            ds = torch.ops.aten._softmax_backward_data(dp, probs, -1, probs.dtype)
            return ds * ctx.scaling, None, None, None, None, None
        st = _stream_ptr(probs)
        C = probs.shape[-1]
        rows = probs.numel() // C
        ds = torch.empty_like(probs)
        stages, outs = _stages(members, probs, st)
        _native.check(_native.lib().qt_softmax_backward_chain_bf16(dp.data_ptr(), probs.data_ptr(), ds.data_ptr(), rows, C, ctx.scaling, stages,
                                                                   len(members), ctypes.byref(fmt), _lut_ptr(members[0][0], fmt), st),
                      "qt_softmax_backward_chain_bf16")
        _hand_over(members, ds, outs)
        return ds, None, None, None, None, None


def softmax_or_none(attn, scores, attention_mask, scaling, dropout):
    """probs = softmax(attn_scaling(scores, scaling) + mask) of a quantizable attention block inside a training step, or None: nothing
    hooks the scaling or the softmax (`--quantize_forward gemm`), no active dropout, av_matmul's input quantizers exist."""
    if not (producers_enabled() and torch.is_grad_enabled() and scores.requires_grad and scores.is_cuda and scores.dtype == torch.bfloat16
            and scores.dim() == 4 and scores.is_contiguous() and scores.shape[-1] % 8 == 0 and scores.shape[-1] <= 2048 and scores.data_ptr() % 16 == 0):
        return None
    if dropout and attn.training:
        return None
    for name in ("attn_scaling", "softmax"):
        mod = getattr(attn, name, None)
        if mod is None or mod._forward_hooks or mod._forward_pre_hooks or mod._backward_hooks or mod._backward_pre_hooks \
                or getattr(mod, "activation_pre_process", None) is not None:
            return None
    if type(attn.softmax) is not torch.nn.Softmax:             # (the fp32 softmax of the LLaMA twin rounds at other points)
        return None
    holder = getattr(attn.av_matmul, "activation_pre_process", None)
    fq_p = _fq(holder, "0")
    if fq_p is None or _members_format([(fq_p, -1)], scores.device) is None:
        return None
    B, H, Q, C = scores.shape
    mask = None
    strides = (0, 0, 0)
    if attention_mask is not None:
        m = attention_mask[..., :C]
        if m.dtype != torch.bfloat16 or m.dim() != 4 or m.stride(-1) != 1 or m.device != scores.device or m.requires_grad:
            return None
        if m.shape[0] not in (1, B) or m.shape[1] not in (1, H) or m.shape[2] not in (1, Q):
            return None
        strides = (m.stride(0) if m.shape[0] == B and B > 1 else 0, m.stride(1) if m.shape[1] == H and H > 1 else 0,
                   m.stride(2) if m.shape[2] == Q and Q > 1 else 0)
        if (strides[0] | strides[1] | strides[2]) % 8 != 0 or m.data_ptr() % 16 != 0:
            return None
        mask = m
    head = _fq(getattr(attn.qk_matmul, "error_pre_process", None), "0")
    return _SoftmaxTrainFn.apply(scores, mask, strides, scaling, fq_p, head)


def _hooked(fq):
    return bool(fq._forward_hooks or fq._forward_pre_hooks)


def _served_call(fq, x, y, numel=None):
    """One hook call that a fused launch has already served (scale update issued, amax accumulated, result `y` written): counted as the
    hook counts it.  A fake-quantizer somebody hooked is called as the module it is, on the tensor the reference's hook would hand it,
    and finds `y` (take_member_result) -- its hooks see the call, its input and its result."""
    from .fake_quantize import _Stats
    if y is not None and _hooked(fq):
        fq.__dict__["_qt_chain_result"] = (x.data_ptr(), x._version, tuple(x.shape), y, x, False)
        with torch.no_grad():
            got = fq(x)
        if got.data_ptr() != y.data_ptr():
            raise RuntimeError("a fake-quantizer call served by the attention launch did not take its result")
        return
    STATS.members += 1
    _Stats.add(x.numel() if numel is None else numel)
    fq.__dict__["_qt_calls"] = fq.__dict__.get("_qt_calls", 0) + 1      # (harness.GraphedTrainStep batches the scale updates of the quantizers a step calls)


class _AttentionTrainFn(torch.autograd.Function):
    """The attention core between the query / key / value projections and the output projection inside a training step, one launch each
    way (qt_attention_train_bf16 / qt_attention_train_backward_bf16; modules/quantizable/attention.py, upstream modeling_bert.py:118-158):
    the four input quantizers of qk_matmul / av_matmul, both products, scaling + mask + softmax and -- when it shares their format -- the
    output projection's input quantizer forward; the two backward-pre quantizers, the four products and the softmax backward backward.
    The result and the gradients are laid out [B, S, H, D], so the permute copies of the unfused path do not exist."""

    @staticmethod
    def forward(ctx, q, k, v, mask, mstrides, scaling, fqs, fq_o, efqs, lins, drop_p=0.0):
        from .fake_quantize import _stream_ptr, _Stats, launch_scale_update
        B, H, S, D = q.shape
        dev = q.device
        st = _stream_ptr(q)
        members = [(f, -1) for f in fqs] + ([(fq_o, -1)] if fq_o is not None else [])
        fmt = _members_format(members, dev)
        qq, kq, vq = (torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=dev) for _ in range(3))
        probs = torch.empty((B, H, S, S), dtype=q.dtype, device=dev)
        pq = torch.empty_like(probs)
        out = torch.empty((B, S, H * D), dtype=q.dtype, device=dev)
        oq = torch.empty_like(out) if fq_o is not None else None
        outs = [qq, kq, vq, pq, oq]
        stages = (_native.QtChainStage * 5)()
        for i, (fq, _) in enumerate(members):
            if fq._observe:
                launch_scale_update(fq.amax_history, fq.scale, fq.quant_max, fq.force_scale_power_of_two, st)
            stages[i].scale_f32_dev = fq.scale.data_ptr()
            stages[i].amax_bits_dev = fq.amax_history.data_ptr() if fq._observe else None
            stages[i].out_dev = outs[i].data_ptr()
            stages[i].src = -1
        msb, msh, msq = mstrides
        # attention-probability dropout (between the softmax and av_matmul): the keep mask is drawn by torch's generator (one launch;
        # seeds and stream capture behave as for nn.Dropout), the kernels apply torch's dropout arithmetic with it, forward and backward
        keep = torch.empty((B, H, S, S), dtype=torch.uint8, device=dev).bernoulli_(1.0 - drop_p) if drop_p > 0.0 else None
        ctx.drop_scale = 1.0 / (1.0 - drop_p) if drop_p > 0.0 else 1.0
        _native.check(_native.lib().qt_attention_train_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), q.stride(2), q.stride(1),
                                                            mask.data_ptr() if mask is not None else None, msb, msh, msq, stages, probs.data_ptr(),
                                                            out.data_ptr(), keep.data_ptr() if keep is not None else None, ctx.drop_scale, B, H, S, D,
                                                            float(scaling), ctypes.byref(fmt), _lut_ptr(fqs[0], fmt), st),
                      "qt_attention_train_bf16")
        STATS.attention += 1
        # the four hook calls this launch stands for, in the hooks' order (qk_matmul: q, k^T; av_matmul: P, v): counted as the hooks count
        # them -- and a fake-quantizer somebody hooked is CALLED, on the view the reference's hook would hand it, and finds its result
        for fq, x, y in ((fqs[0], q, qq), (fqs[1], k.transpose(2, 3), kq.transpose(2, 3)), (fqs[3], probs, pq), (fqs[2], v, vq)):
            _served_call(fq, x, y)
        if fq_o is not None:
            _hand_over([(fq_o, -1)], out, [oq])                # (called by the output projection's own hook, on `out` viewed [B, S, H * D])
        ctx.save_for_backward(*((qq, kq, vq, probs, pq) + ((keep,) if keep is not None else ())))
        ctx.scaling = float(scaling)
        ctx.efqs = efqs
        ctx.lins = lins
        return out.view(B, S, H, D)

    @staticmethod
    def backward(ctx, dout):
        from .fake_quantize import _stream_ptr, _Stats, launch_scale_update
        qq, kq, vq, probs, pq = ctx.saved_tensors[:5]
        keep = ctx.saved_tensors[5] if len(ctx.saved_tensors) > 5 else None
        B, H, S, D = qq.shape
        dev = qq.device
        ev, eq = ctx.efqs
        members = [(ev, -1), (eq, -1)]
        fmt = _members_format(members, dev)
        dout = dout.contiguous()
        if fmt is None or dout.dtype != torch.bfloat16 or dout.data_ptr() % 16:
            g = ev(dout.view(B, S, H, D).permute(0, 2, 1, 3))               # the hooks' own calls, torch's kernels
            dp, dv = g @ vq.transpose(2, 3), pq.transpose(2, 3) @ g
            if keep is not None:
                dp = torch.ops.aten.native_dropout_backward(dp, keep.bool(), ctx.drop_scale)
            ds = eq(torch.ops.aten._softmax_backward_data(dp, probs, -1, probs.dtype) * ctx.scaling)
            return ds @ kq, ds.transpose(2, 3) @ qq, dv, None, None, None, None, None, None, None, None
        st = _stream_ptr(qq)
        dq, dk, dv = (torch.empty((B, S, H, D), dtype=qq.dtype, device=dev) for _ in range(3))
        stages = (_native.QtChainStage * 2)()
        for i, (fq, _) in enumerate(members):
            if fq._observe:
                launch_scale_update(fq.amax_history, fq.scale, fq.quant_max, fq.force_scale_power_of_two, st)
            stages[i].scale_f32_dev = fq.scale.data_ptr()
            stages[i].amax_bits_dev = fq.amax_history.data_ptr() if fq._observe else None
            stages[i].out_dev = None
            stages[i].src = -1
        # g = ev(dO), dS and dS' = eq(dS) stay on the chip -- unless somebody hooked those fake-quantizers and wants to see the calls
        g = torch.empty((B, S, H, D), dtype=qq.dtype, device=dev) if _hooked(ev) else None
        ds, dsq = (torch.empty_like(probs), torch.empty_like(probs)) if _hooked(eq) else (None, None)
        if g is not None:
            stages[0].out_dev = g.data_ptr()
        if dsq is not None:
            stages[1].out_dev = dsq.data_ptr()
        # the query / key / value projections' own backward-pre quantizers (single-member chains with the bias gradient, `plan`) ride on
        # the gradients' way out: each finds its result when its hook is called on dQ / dK / dV viewed [B, S, H * D]
        L = _native.lib()
        gst = (_native.QtChainStage * 3)()
        couts = (ctypes.c_void_p * 3)()
        riders = []
        for i, lin in enumerate(ctx.lins or ()):
            head = _fq(getattr(lin, "error_pre_process", None), "0") if lin is not None else None
            mem, colsum = _grad_chain(head)
            if mem is None or len(mem) != 1 or _members_format(members + mem, dev) is None:
                continue
            fqg = mem[0][0]
            if fqg._observe:
                launch_scale_update(fqg.amax_history, fqg.scale, fqg.quant_max, fqg.force_scale_power_of_two, st)
            o = torch.empty((B, S, H * D), dtype=qq.dtype, device=dev)
            gst[i].scale_f32_dev = fqg.scale.data_ptr()
            gst[i].amax_bits_dev = fqg.amax_history.data_ptr() if fqg._observe else None
            gst[i].out_dev = o.data_ptr()
            gst[i].src = -1
            gb = None
            if colsum is not None and colsum[1].out_features == H * D:
                gb = torch.empty(H * D, dtype=torch.bfloat16, device=dev)
                couts[i] = gb.data_ptr()
            riders.append((mem, i, o, gb))
        ws = _chain_scratch(L.qt_attention_train_backward_ws_bytes(H), dev) if any(r[3] is not None for r in riders) else None
        _native.check(L.qt_attention_train_backward_bf16(dout.data_ptr(), qq.data_ptr(), kq.data_ptr(), vq.data_ptr(), qq.stride(0), qq.stride(2),
                                                         qq.stride(1), probs.data_ptr(), pq.data_ptr(), stages, ds.data_ptr() if ds is not None else None,
                                                         dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), gst if riders else None, couts if riders else None,
                                                         _format_max(ev), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                                         keep.data_ptr() if keep is not None else None, ctx.drop_scale, B, H, S, D, ctx.scaling,
                                                         ctypes.byref(fmt), _lut_ptr(ev, fmt), st),
                      "qt_attention_train_backward_bf16")
        STATS.attention += 1
        _served_call(ev, dout.view(B, S, H, D).permute(0, 2, 1, 3), g.permute(0, 2, 1, 3) if g is not None else None, dout.numel())
        _served_call(eq, ds, dsq, probs.numel())
        grads = (dq, dk, dv)
        for mem, i, o, gb in riders:
            _hand_over(mem, grads[i].view(B, S, H * D), [o])
            if gb is not None:
                put_colsum(o, gb)
        if len(riders) == 3 and [r[1] for r in riders] == [0, 1, 2]:
            group_qkv_backward(ctx.lins, [r[2] for r in riders])       # the projections' six backward products as two launches
        return dq.permute(0, 2, 1, 3), dk.permute(0, 2, 1, 3), dv.permute(0, 2, 1, 3), None, None, None, None, None, None, None, None


def attention_enabled():
    return producers_enabled() and _on("attention")


def attention_or_none(attn, query, key, value, attention_mask, scaling, dropout):
    """The attention core of a quantizable attention block inside a training step through _AttentionTrainFn -- the result in
    [B, S, H, D] -- or None (the caller takes the sub-modules one by one): head_dim 64 and 32..128 positions, q / k / v views of one
    layout, nothing but the reference's own hooks on qk_matmul / av_matmul (forward-pre on both inputs, backward-pre;
    quantize.py:143-148) and none on the scaling or the softmax, and every one of those fake-quantizers already created (the first
    step creates them)."""
    if not (attention_enabled() and torch.is_grad_enabled() and query.is_cuda and query.dtype == torch.bfloat16 and query.dim() == 4
            and key.shape == query.shape and value.shape == query.shape and key.dtype == torch.bfloat16 and value.dtype == torch.bfloat16
            and (query.requires_grad or key.requires_grad or value.requires_grad)):
        return None
    B, H, S, D = query.shape
    if not _native.lib().qt_attention_train_supported(B, H, S, D):
        return None
    if (query.stride() != key.stride() or query.stride() != value.stride() or query.stride(3) != 1 or any(s % 8 for s in query.stride()[:3])
            or any(t.data_ptr() % 16 for t in (query, key, value))):
        return None
    drop_p = float(dropout) if (dropout and attn.training) else 0.0
    if not 0.0 <= drop_p < 1.0:
        return None
    for name in ("attn_scaling", "softmax"):
        mod = getattr(attn, name, None)
        if mod is None or mod._forward_hooks or mod._forward_pre_hooks or mod._backward_hooks or mod._backward_pre_hooks \
                or getattr(mod, "activation_pre_process", None) is not None:
            return None
    if type(attn.softmax) is not torch.nn.Softmax:
        return None
    qk, av = getattr(attn, "qk_matmul", None), getattr(attn, "av_matmul", None)
    if qk is None or av is None:
        return None
    for mod in (qk, av):
        ha, he = getattr(mod, "activation_pre_process", None), getattr(mod, "error_pre_process", None)
        if (mod._forward_hooks or mod._backward_hooks or len(mod._forward_pre_hooks) != 1 or len(mod._backward_pre_hooks) != 1
                or getattr(mod, "error_post_process", None) is not None or ha is None or he is None or len(ha) != 2 or len(he) != 1
                or "0" not in ha or "1" not in ha or "0" not in he):
            return None
    fqs = [qk.activation_pre_process["0"], qk.activation_pre_process["1"], av.activation_pre_process["1"], av.activation_pre_process["0"]]
    efqs = (av.error_pre_process["0"], qk.error_pre_process["0"])
    dev = query.device
    if _members_format([(f, -1) for f in fqs], dev) is None or _members_format([(f, -1) for f in efqs], dev) is None:
        return None
    proj = attn.__dict__.get("_qt_out_proj")
    holder = getattr(proj, "activation_pre_process", None) if proj is not None else None
    fq_o = _fq(holder, "0") if holder is not None and len(holder) == 1 else None
    if fq_o is not None and (fq_o.__dict__.get("_qt_chain") is not None or _members_format([(f, -1) for f in fqs + [fq_o]], dev) is None):
        fq_o = None
    mask = None
    strides = (0, 0, 0)
    if attention_mask is not None:
        m = attention_mask[..., :S]
        if m.dtype != torch.bfloat16 or m.dim() != 4 or m.stride(-1) != 1 or m.device != dev or m.requires_grad:
            return None
        if m.shape[0] not in (1, B) or m.shape[1] not in (1, H) or m.shape[2] not in (1, S) or m.shape[3] != S:
            return None
        strides = (m.stride(0) if m.shape[0] == B and B > 1 else 0, m.stride(1) if m.shape[1] == H and H > 1 else 0,
                   m.stride(2) if m.shape[2] == S and S > 1 else 0)
        if (strides[0] | strides[1] | strides[2]) % 8 != 0 or m.data_ptr() % 16 != 0:
            return None
        mask = m
    lins = tuple(getattr(attn, n, None) for n in ("query", "key", "value"))
    _mark_qkv_members(attn, lins)
    return _AttentionTrainFn.apply(query, key, value, mask, strides, scaling, fqs, fq_o, efqs, lins, drop_p)


class _EmbeddingTrainFn(torch.autograd.Function):
    """nn.Embedding inside a training step: torch's lookup forward; the weight gradient by qt_embedding_backward_bf16 -- bit for bit
    torch's embedding_dense_backward (its <= 3072-index path: per 16-row chunk a partial sum per index, folded into the table row in
    bf16, chunk after chunk), two parallel launches instead of one workgroup's walk over all chunks (134 us per table, three tables)."""

    @staticmethod
    def forward(ctx, weight, ids, padding_idx):
        ctx.save_for_backward(ids)
        ctx.rows = weight.shape[0]
        ctx.pad = -1 if padding_idx is None else int(padding_idx)
        return torch.nn.functional.embedding(ids, weight, padding_idx)

    @staticmethod
    def backward(ctx, g):
        from .fake_quantize import _stream_ptr
        (ids,) = ctx.saved_tensors
        cols = g.shape[-1]
        g2 = g.contiguous().view(-1, cols)
        idf = ids.reshape(-1).contiguous()
        n = idf.numel()
        if not (g2.dtype == torch.bfloat16 and 0 < n <= 3072 and cols % 8 == 0 and g2.data_ptr() % 16 == 0 and idf.dtype == torch.int64):
            return torch.ops.aten.embedding_dense_backward(g, ids, ctx.rows, ctx.pad, False), None, None
        gw = torch.zeros((ctx.rows, cols), dtype=torch.bfloat16, device=g.device)
        part = torch.empty_like(g2)
        _native.check(_native.lib().qt_embedding_backward_bf16(g2.data_ptr(), idf.data_ptr(), n, cols, ctx.pad, ctx.rows, part.data_ptr(), gw.data_ptr(),
                                                               _stream_ptr(g2)), "qt_embedding_backward_bf16")
        STATS.embeddings += 1
        return gw, None, None


def embedding_or_none(emb, ids):
    """`emb(ids)` of a training step through _EmbeddingTrainFn, or None: a plain dense bf16 table on the device (no max_norm, no
    scale_grad_by_freq, not sparse), int64 indices, at most 3072 of them (beyond that torch's gradient takes another, sort-based path
    with other sums), nothing hooked onto the module."""
    w = emb.weight
    if not (producers_enabled() and _on("embedding") and torch.is_grad_enabled() and w.requires_grad and w.is_cuda
            and w.dtype == torch.bfloat16 and w.dim() == 2 and w.shape[1] % 8 == 0 and w.is_contiguous() and ids.is_cuda and ids.dtype == torch.int64
            and 0 < ids.numel() <= 3072 and emb.max_norm is None and not emb.scale_grad_by_freq and not emb.sparse
            and not emb._forward_hooks and not emb._forward_pre_hooks and not emb._backward_hooks and not emb._backward_pre_hooks):
        return None
    return _EmbeddingTrainFn.apply(w, ids, emb.padding_idx)
