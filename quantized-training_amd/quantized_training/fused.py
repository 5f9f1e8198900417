"""Fused fake-quant GEMM entry points used by the QAT Linear (and, later, the attention matmuls).

``fused_linear_or_none`` returns ``None`` whenever the fused HIP GEMM does not apply, and the
caller then runs the unfused sequence (HIP elementwise fake-quant + library GEMM); both are
device paths -- there is no CPU fallback for device tensors.
"""
import ctypes
import os

import torch

from . import _native
from .fake_quantize import (STATS, FusedAmaxObsFakeQuantFunction, FusedAmaxObsFakeQuantize, _stream_ptr, handover_valid,
                            launch_scale_update, materialize_lazy)
from .quantizer.quantizer import QScheme

_IDENTITY = _native.QtFormat(_native.QT_FMT_IDENTITY, 0, 0, 0.0, 0.0)


def fused_gemm_enabled():
    return os.environ.get("QT_FUSED_GEMM", "0") == "1"


def fp8_gemm_enabled():
    return os.environ.get("QT_FP8_GEMM", "1") != "0"


def fq8_gemm_mode():
    """QT_FQ8_GEMM: "auto" (default) -- per problem shape, whichever of the two routes measured faster on its first call;
    "1" -- always the hand-written FP8 GEMM with the weight fake-quantizer in its operand path (qt_linear_fq8_bf16);
    "0" -- always the weight pass + library GEMM pair."""
    v = os.environ.get("QT_FQ8_GEMM", "auto")
    return v if v in ("0", "1") else "auto"


def fq8_gemm_enabled():
    return fq8_gemm_mode() != "0"


_FQ8_CHOICE = {}          # (M, Ns, K, activation dtype, weight format, device) -> True: fused kernel, False: pass + library GEMM
ROUTES = {}               # what ran, for the bench line: "MxNxK" -> route name (first decision per shape)

# Measured on MI355X (tools/exp_linear_fq8.py, gpurun_out/fq8_routes.log -> profiles/r03_fq8_routes.txt): microseconds of the fused
# kernel against the weight pass + library GEMM, weights rotating beyond the Infinity Cache.  (M, sum N, K) -> fused kernel wins.
# The default route of a shape is this table, then the rule below: the same on every box, rank and run.
_FQ8_TABLE = {
    (1024, 11008, 4096): True,    # 54.7 / 69.1
    (1024, 4096, 11008): True,    # 68.0 / 72.8
    (1024, 4096, 4096): True,     # 27.9 / 32.9
    (1024, 12288, 4096): True,    # 54.7 / 70.5  (q / k / v)
    (1024, 32000, 4096): True,    # 159.8 / 169.8
    (1024, 8192, 4096): True,     # 44.9 / 50.2
    (1024, 6144, 4096): True,     # 40.7 / 43.1
    (512, 11008, 4096): True,     # 43.3 / 56.0
    (256, 11008, 4096): True,     # 30.0 / 43.2
    (2048, 11008, 4096): True,    # 102.5 / 105.3
    (2048, 4096, 4096): False,    # 45.5 / 45.0
    (4096, 4096, 4096): False,    # 87.6 / 57.6
    (6144, 768, 768): True,       # 16.7 / 27.2
    (6144, 2304, 768): False,     # 29.6 / 27.6  (q / k / v)
    (6144, 3072, 768): False,     # 35.2 / 30.1
    (6144, 768, 3072): False,     # 34.4 / 27.6
    (1024, 13824, 5120): False,   # 99.1 / 88.2
    (1024, 5120, 13824): False,   # 111.5 / 107.2
    (1024, 5120, 5120): True,     # 46.3 / 47.1
    (1024, 15360, 5120): False,   # 105.9 / 96.7 (q / k / v)
}


def _note_route(kind, M, ns, K, route):
    ROUTES.setdefault(f"{kind}:{M}x{sum(ns)}x{K}", route)


def routes_report():
    """The GEMM routes taken so far, per problem shape, and for library FP8 GEMMs which of the library's suggestions ran
    (bench.py puts them into its JSON line)."""
    out = dict(sorted(ROUTES.items()))
    out.update(sorted(LT_ALGOS.items()))
    from .modules.qat.linear import GEMM_ROUTES                # the training step's products (csrc/qt_train_gemm.hip or torch's GEMM)
    out.update(sorted(GEMM_ROUTES.items()))
    from .optim import ROUTES as OPT_ROUTES                    # clip_grad_norm_ + optimizer.step() (csrc/qt_optimizer.hip or torch's launches)
    out.update(sorted(OPT_ROUTES.items()))
    return out


def _fq8_heuristic(M, ns, K, device):
    """Shapes the table does not list: the fused kernel where it fills the chip in one round of 256-row tiles at least five column
    groups wide -- what the sweep in DESIGN.md 6b showed for M <= 1024."""
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    tiles_m = (M + 255) // 256
    groups = sum(ns) // 16
    tn_min = (groups + 11) // 12
    if tiles_m * tn_min > cus or tiles_m > cus:
        return False
    nt = groups / max(1, cus // tiles_m)
    return M <= 1024 and nt >= 5.0 and K <= 8192


def fq8_route_is_fused(x8, layers):
    """Route of one fake-quant Linear problem (x8 [M, K] FP8 codes x the bf16 weights of `layers`): QT_FQ8_GEMM=0 / 1 forces it; else
    the committed table, else the closed-form rule.  Never a measurement: a timing race would make the summation order -- and so
    the logits -- depend on the box, the run and the rank, and a lazily measured choice would need a collective that not every rank
    reaches (windows are dealt round-robin)."""
    K = x8.shape[-1]
    M = x8.numel() // K
    ns = tuple(l.weight.shape[0] for l in layers)
    mode = fq8_gemm_mode()
    if mode != "auto":
        hit = mode == "1"
    else:
        hit = _FQ8_TABLE.get((M, sum(ns), K))
        if hit is None:
            key = (M, ns, K, x8.dtype, layers[0].weight_fake_quant._qt_format.key(), x8.device.index)
            hit = _FQ8_CHOICE.get(key)
            if hit is None:
                hit = _FQ8_CHOICE[key] = _fq8_heuristic(M, ns, K, x8.device)
    _note_route("fq8", M, ns, K, "fused_fp8_gemm" if hit else "weight_pass+library_fp8_gemm")
    return hit


_F8_CODE = {torch.float8_e4m3fn: 0, torch.float8_e5m2: 1}


def hip_fq8_linear_or_none(x8, layers):
    """y[.., sum N] = x . [fq(W_0); fq(W_1); ...]^T + bias through qt_linear_fq8_bf16: x8 holds the FP8 codes of the already
    fake-quantized activation, the bf16 weights of `layers` (QAT Linears that share it, stateless FP8 weight fake-quantizers
    of one format) are converted inside the kernel.  None when the kernel does not take the problem."""
    K = x8.shape[-1]
    if x8.dtype not in _F8_CODE or K % 128 != 0 or not x8.is_contiguous() or x8.data_ptr() % 16 or not 1 <= len(layers) <= 4:
        return None
    wf = layers[0].weight_fake_quant._qt_format
    for l in layers:
        W, f = l.weight, l.weight_fake_quant._qt_format
        if (W.dtype != torch.bfloat16 or not W.is_contiguous() or W.shape[1] != K or W.shape[0] % 16 or W.data_ptr() % 16
                or W.device != x8.device or f.key() != wf.key()):
            return None
        if l.bias is not None and (l.bias.dtype != torch.bfloat16 or not l.bias.is_contiguous() or l.bias.data_ptr() % 8
                                   or l.bias.device != x8.device):
            return None
    n = len(layers)
    M = x8.numel() // K
    ns = [l.weight.shape[0] for l in layers]
    y = torch.empty((M, sum(ns)), dtype=torch.bfloat16, device=x8.device)
    wp = (ctypes.c_void_p * n)(*[l.weight.data_ptr() for l in layers])
    bp = (ctypes.c_void_p * n)(*[(l.bias.data_ptr() if l.bias is not None else None) for l in layers])
    nn = (ctypes.c_int * n)(*ns)
    rc = _native.lib().qt_linear_fq8_bf16(x8.data_ptr(), _F8_CODE[x8.dtype], wp, bp, nn, n, 1 if wf.p0 == 2 else 0, y.data_ptr(),
                                           M, K, _stream_ptr(x8))
    if rc in (_native.QT_ERR_BAD_ARG, _native.QT_ERR_UNALIGNED, _native.QT_ERR_BAD_DTYPE):
        return None
    _native.check(rc, "qt_linear_fq8_bf16")
    return y


def mark_fp8_producer(consumer, act_fq):
    """Called when the activation fake-quantizer of `consumer` is created: if `consumer` is a QAT Linear
    whose weight fake-quantizer and this activation fake-quantizer both produce exact FP8 values with
    scale 1 (e4m3 / e5m2 specs without `qs`), the activation pass also emits FP8 bytes for the GEMM."""
    from .modules.qat.linear import Linear as QATLinear
    if not (fp8_gemm_enabled() and isinstance(consumer, QATLinear)):
        return
    wfq = consumer.weight_fake_quant
    if isinstance(wfq, FusedAmaxObsFakeQuantize) and isinstance(act_fq, FusedAmaxObsFakeQuantize) \
            and wfq.fp8_exact() and act_fq.fp8_exact():
        act_fq._emit_fp8 = "both"


_WEIGHT_CACHE = {"on": False, "epoch": 0}      # harness.cache_quantized_weights(True) turns it on


def cache_quantized_weights(enable: bool = True):
    """Opt-in: in eval mode, with the weight fake-quantizer's observer off (stateless spec or frozen
    after calibration), fq(W) is a constant -- keep it instead of re-quantizing W on every forward.  Off by default
    because the reference re-quantizes every time (and the headline metric counts those elements).  Entries follow
    the tensors' version counters (optimizer steps, load_state_dict, in-place ops); a write through `.data` is not
    visible to them -- call this function again to drop every entry."""
    _WEIGHT_CACHE["on"] = bool(enable)
    _WEIGHT_CACHE["epoch"] += 1


def cached_weight(layer, kind, make):
    """make() -> quantized weight; cached on the layer while nothing it depends on changes."""
    fq = layer.weight_fake_quant
    W = layer.weight
    if (not _WEIGHT_CACHE["on"] or layer.training or getattr(fq, "_observe", True)
            or (torch.is_grad_enabled() and W.requires_grad)):
        return make()
    key = (kind, _WEIGHT_CACHE["epoch"], W.data_ptr(), W._version, fq.scale._version, bool(getattr(fq, "_quantize", True)),
           str(W.device))
    hit = layer.__dict__.get("_qt_wcache")
    if hit is not None and hit[0] == key:
        return hit[1]
    w = make()
    layer.__dict__["_qt_wcache"] = (key, w)
    return w


class SiblingGroup:
    """Linears that read the same tensor (q / k / v projections behind one RMSNorm).  When the first of them runs,
    every member's weight is fake-quantized to FP8 in one launch (qt_fake_quant_bf16_fp8_multi) into one [sum N, K]
    buffer and ONE FP8 GEMM multiplies the shared activation by all of them; the other members then return their
    column slice of that product.  Work per call is what the reference issues (each weight pass, each member's own
    input pass), only batched: three 22 us GEMMs + three 12.5 us weight passes become one 43 us GEMM + one 29 us pass.

    A member takes its slice only if its input is the fake-quantized image of the very tensor the leader saw
    (`_qt_origin` of the hook's output) and all members quantize inputs and weights with the same stateless FP8
    format, so the leader's FP8 activation IS theirs, byte for byte."""

    def __init__(self, layers, value_map_only=False):
        self.layers = list(layers)
        self.value_map_only = value_map_only   # gate / up: one launch only on the value-map GEMM (qt_linear_fqt_ws_bf16), see model_fusions
        self.buf = None
        self.bias = None                       # (key of the members' bias versions, concatenated bias [sum N])
        self.stash = None                      # (origin key, product [M, sum N], taken flags)

    def eligible(self):
        def fmt_key(f):
            return (f.kind, f.p0, f.p1, f.flo, f.fhi)
        K = self.layers[0].weight.shape[1]
        w_fmt = a_fmt = None
        with_bias = self.layers[0].bias is not None
        for l in self.layers:
            W, fq = l.weight, l.weight_fake_quant
            if (l.bias is not None) != with_bias or (with_bias and (l.bias.dtype != torch.bfloat16 or l.bias.device != W.device)):
                return False
            if not (isinstance(fq, FusedAmaxObsFakeQuantize) and fq.fp8_exact() and W.dtype == torch.bfloat16
                    and W.is_contiguous() and W.shape[1] == K and K % 16 == 0 and W.shape[0] % 16 == 0
                    and not (torch.is_grad_enabled() and W.requires_grad)):
                return False
            holder = getattr(l, "activation_pre_process", None)
            afq = holder["0"] if holder is not None and "0" in holder else None
            if not (isinstance(afq, FusedAmaxObsFakeQuantize) and afq.producer_fusable()):
                return False
            if w_fmt is None:
                w_fmt, a_fmt = fmt_key(fq._qt_format), fmt_key(afq._qt_format)
            elif fmt_key(fq._qt_format) != w_fmt or fmt_key(afq._qt_format) != a_fmt:
                return False
        return True

    def bias_or_none(self):
        """The members' biases as one [sum N] vector (BERT-style projections), rebuilt when any of them changes."""
        if self.layers[0].bias is None:
            return None
        key = tuple((l.bias.data_ptr(), l.bias._version) for l in self.layers)
        if self.bias is None or self.bias[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return False                   # no allocation that outlives the graph from inside a capture
            self.bias = (key, torch.cat([l.bias.detach() for l in self.layers]))
        return self.bias[1]


def value_key(value):
    return (value.data_ptr(), value._version, tuple(value.shape), value.stride())


def _origin_key(x):
    o = getattr(x, "_qt_origin", None)
    if o is not None:
        return o
    return (x.data_ptr(), x._version, tuple(x.shape))         # the leader's input was handed through by its hook


def hip_mlp_fq8_or_none(x8, gate, up, out_fq, codes_only=False):
    """(h, h8) = the gated MLP's front half in one launch (qt_mlp_fq8_bf16): fq_out(silu(gate(x)) * up(x)) as bf16 values and FP8
    codes from the FP8 codes of x and the UNQUANTIZED bf16 weights of the QAT Linears `gate` / `up`.  None when the kernel does
    not take the problem."""
    K = x8.shape[-1]
    Wg, Wu = gate.weight, up.weight
    fg, fu = gate.weight_fake_quant._qt_format, up.weight_fake_quant._qt_format
    if (x8.dtype not in _F8_CODE or K % 128 != 0 or not x8.is_contiguous() or x8.data_ptr() % 16 or Wg.shape != Wu.shape or fg.key() != fu.key()
            or Wg.shape[1] != K or Wg.shape[0] % 16):
        return None
    for W in (Wg, Wu):
        if W.dtype != torch.bfloat16 or not W.is_contiguous() or W.data_ptr() % 16 or W.device != x8.device:
            return None
    for l in (gate, up):
        if l.bias is not None and (l.bias.dtype != torch.bfloat16 or not l.bias.is_contiguous() or l.bias.data_ptr() % 8 or l.bias.device != x8.device):
            return None
    N = Wg.shape[0]
    M = x8.numel() // K
    h = torch.empty((M, N), dtype=torch.bfloat16, device=x8.device)
    h8 = torch.empty((M, N), dtype=torch.uint8, device=x8.device)
    rc = _native.lib().qt_mlp_fq8_bf16(x8.data_ptr(), _F8_CODE[x8.dtype], Wg.data_ptr(), Wu.data_ptr(),
                                        gate.bias.data_ptr() if gate.bias is not None else None, up.bias.data_ptr() if up.bias is not None else None,
                                        N, 1 if fg.p0 == 2 else 0, None if codes_only else h.data_ptr(), h8.data_ptr(),
                                        ctypes.byref(out_fq._qt_format), M, K, _stream_ptr(x8))      # codes_only: h stays unwritten (`_qt_lazy`)
    if rc in (_native.QT_ERR_BAD_ARG, _native.QT_ERR_UNALIGNED, _native.QT_ERR_BAD_DTYPE):
        return None
    _native.check(rc, "qt_mlp_fq8_bf16")
    return h, h8


_MLP_CHOICE = {}          # (M, N, K, activation dtype, formats, device) -> True: one launch, False: two fused GEMMs + SiLU * up


# (M, N, K) -> the gated MLP's front half as ONE launch (qt_mlp_fq8_bf16) beats two fused GEMMs + SiLU * up; measured on MI355X
# (profiles/r02_mlp_fq8.txt and the window profiles: 114 us inside the LLaMA-2-7B window against 132 us for the three launches)
_MLP_TABLE = {
    (1024, 11008, 4096): True,
    (512, 11008, 4096): True,     # 67.3 / 96.5
}


def mlp_route_is_one_launch(x8, gate, up, out_fq):
    """Whether qt_mlp_fq8_bf16 runs the gated MLP's front half on this problem: QT_FQ8_MLP=2 forces it; else the committed table;
    else one launch where it fills the chip in a single round (a rule, never a measurement: see fq8_route_is_fused)."""
    if os.environ.get("QT_FQ8_MLP", "1") == "2":              # always (tests, ablations)
        return True
    K = x8.shape[-1]
    M = x8.numel() // K
    N = gate.weight.shape[0]
    hit = _MLP_TABLE.get((M, N, K))
    if hit is None:
        key = (M, N, K, x8.dtype, gate.weight_fake_quant._qt_format.key(), out_fq._qt_format.key(), x8.device.index)
        hit = _MLP_CHOICE.get(key)
        if hit is None:
            cus = torch.cuda.get_device_properties(x8.device).multi_processor_count
            hit = _MLP_CHOICE[key] = ((M + 255) // 256) * ((N // 16 + 5) // 6) <= cus
    _note_route("mlp", M, [N, N], K, "one_launch_gate_up_silu" if hit else "two_gemms+silu_mul")
    return hit


# ---- every FP8 weight pass of an evaluation forward as ONE launch (harness.GraphedBatch / GraphedWindow) -------------------------
# The weight-pass + library-GEMM route issues one codes-only pass per Linear (or per q / k / v group): 36 launches of ~5 us for a
# BERT-base batch.  The weights are constants of a forward and these fake-quantizers are stateless, so a captured forward may run
# all of them first, as one launch (qt_fake_quant_multi_bf16_fp8); each call site then finds its codes (valid for the very next use,
# for those very tensors at those very versions) and counts its elements as before.
_W8_LOG = None            # while a warm-up forward is being recorded: [(owner, layers)] in call order
_W8_PRE = {}              # id(owner) -> (((data_ptr, version), ...), codes): what BatchedWeightCodes.launch() left for the next forward


def start_weight_pass_log():
    global _W8_LOG
    _W8_LOG = []


def stop_weight_pass_log():
    global _W8_LOG
    log, _W8_LOG = _W8_LOG, None
    return log or []


def _note_weight_pass(owner, layers):
    if _W8_LOG is not None and all(o is not owner for o, _ in _W8_LOG):
        _W8_LOG.append((owner, tuple(layers)))


def _take_weight_codes(owner, layers):
    pre = _W8_PRE.pop(id(owner), None)
    if pre is None:
        return None
    stamp, codes = pre
    if stamp != tuple((l.weight.data_ptr(), l.weight._version) for l in layers):
        return None
    return codes


class BatchedWeightCodes:
    """`entries`: [(owner, layers)] as logged by a warm-up forward -- owner is the QAT Linear itself or its SiblingGroup (whose members'
    codes are one [sum N, K] buffer).  One launch per weight format."""

    def __init__(self, entries, device):
        self.groups = {}
        for owner, layers in entries:
            fq = layers[0].weight_fake_quant
            if not all(l.weight.device == device and l.weight.dtype == torch.bfloat16 and l.weight.is_contiguous() and l.weight.numel() % 16 == 0
                       and l.weight.data_ptr() % 16 == 0 and l.weight_fake_quant._qt_format.key() == fq._qt_format.key() for l in layers):
                continue
            fq._move_to(device)
            K = layers[0].weight.shape[1]
            total = sum(l.weight.shape[0] for l in layers)
            codes = torch.empty((total, K), dtype=torch.uint8, device=device)
            fmt, members = self.groups.setdefault(fq._qt_format.key(), (fq._qt_format, []))
            members.append((owner, layers, codes))
        self.launches = []
        for fmt, members in self.groups.values():
            rows, tiles = [], 0
            for owner, layers, codes in members:
                off = 0
                for l in layers:
                    npair = l.weight.numel() // 16
                    rows.append([l.weight.data_ptr(), codes.data_ptr() + off, npair, tiles])
                    tiles += (npair + 1023) // 1024
                    off += l.weight.numel()
            self.launches.append((fmt, members, torch.tensor(rows, dtype=torch.int64, device=device), len(rows), tiles))

    def __len__(self):
        return sum(len(m) for _, m, _, _, _ in self.launches)

    def launch(self):
        for fmt, members, items, count, tiles in self.launches:
            _native.check(_native.lib().qt_fake_quant_multi_bf16_fp8(items.data_ptr(), count, tiles, ctypes.byref(fmt), _stream_ptr(items)),
                          "qt_fake_quant_multi_bf16_fp8")
            for owner, layers, codes in members:
                w8 = codes.view(torch.float8_e5m2 if fmt.p0 == 2 else torch.float8_e4m3fn)
                _W8_PRE[id(owner)] = (tuple((l.weight.data_ptr(), l.weight._version) for l in layers), w8)

    def forget(self):
        for _, members, _, _, _ in self.launches:
            for owner, _, _ in members:
                _W8_PRE.pop(id(owner), None)


def _sibling_linear_or_none(layer, x, x8):
    group = layer.__dict__.get("_qt_sibling_group")
    if group is None or group.value_map_only or os.environ.get("QT_SIBLING_GEMM", "1") == "0" or _WEIGHT_CACHE["on"]:
        return None
    idx = group.layers.index(layer)
    key = _origin_key(x)
    Ns = [l.weight.shape[0] for l in group.layers]
    off = sum(Ns[:idx])
    if idx > 0:
        st = group.stash
        if st is None or st[0] != key or st[2][idx]:
            return None                        # not the leader's tensor (or already taken): the ordinary single path
        st[2][idx] = True
        STATS.add(layer.weight.numel())        # this member's weight pass ran inside the batched launch
        return st[1][:, off:off + Ns[idx]].reshape(*x.shape[:-1], Ns[idx])
    group.stash = None
    if not group.eligible():
        return None
    K = layer.weight.shape[1]
    total = sum(Ns)
    dev = x.device
    n = len(group.layers)
    if fq8_gemm_enabled() and fq8_route_is_fused(x8.reshape(-1, K), group.layers):
        layer.weight_fake_quant._move_to(dev)
        y = hip_fq8_linear_or_none(x8.reshape(-1, K), group.layers)
        if y is not None:
            STATS.add(layer.weight.numel())
            group.stash = (key, y, [True] + [False] * (n - 1))
            return y[:, :Ns[0]].reshape(*x.shape[:-1], Ns[0])
    fq = layer.weight_fake_quant
    fq._move_to(dev)
    w8 = _take_weight_codes(group, group.layers)              # BatchedWeightCodes ran this group's pass in front of the forward
    if w8 is None:
        if group.buf is None or group.buf.device != dev or group.buf.shape != (total, K):
            if torch.cuda.is_current_stream_capturing():
                return None
            group.buf = torch.empty((total, K), dtype=torch.uint8, device=dev)
        L = _native.lib()
        xs = (ctypes.c_void_p * n)(*[l.weight.data_ptr() for l in group.layers])
        ns = (ctypes.c_size_t * n)(*[l.weight.numel() for l in group.layers])
        _native.check(L.qt_fake_quant_bf16_fp8_multi(xs, ns, n, group.buf.data_ptr(), ctypes.byref(fq._qt_format),
                                                     _stream_ptr(x)), "qt_fake_quant_bf16_fp8_multi")
        w8 = group.buf.view(torch.float8_e5m2 if fq._qt_format.p0 == 2 else torch.float8_e4m3fn)
        _note_weight_pass(group, group.layers)
    bias = group.bias_or_none()
    if bias is False:
        return None
    y = lt_fp8_gemm(x8.reshape(-1, K), w8, bias)
    if y is None:
        return None                            # library route unavailable: every member takes the ordinary path
    STATS.add(layer.weight.numel())
    group.stash = (key, y, [True] + [False] * (n - 1))
    return y[:, :Ns[0]].reshape(*x.shape[:-1], Ns[0])


def fp8_linear_or_none(layer, x):
    """E4M3 / E5M2 fake-quant Linear with scale 1 on the FP8 matrix cores: the activation pass already
    produced FP8 bytes (x._qt_fp8), the weight pass writes FP8 only (3 B/element of traffic instead of
    4 and nothing for the GEMM to re-read in bf16), and the products q_x * q_w are exactly the
    reference's bf16 products; fp32 accumulation, bf16 output.  Library FP8 GEMM (hipBLASLt through
    torch._scaled_mm) -- a plain GEMM on already-quantized operands."""
    x8 = getattr(x, "_qt_fp8", None) if handover_valid(x) else None
    fq = layer.weight_fake_quant
    if x8 is None or not fp8_gemm_enabled() or not isinstance(fq, FusedAmaxObsFakeQuantize) or not fq.fp8_exact():
        return None
    if torch.is_grad_enabled() and (layer.weight.requires_grad or x.requires_grad):
        return None
    W = layer.weight
    K = W.shape[1]
    if W.dtype != torch.bfloat16 or not W.is_contiguous() or K % 16 != 0 or W.shape[0] % 16 != 0:
        return None
    if layer.bias is not None and layer.bias.dtype != torch.bfloat16:
        return None
    fq._move_to(x.device)
    shared = _sibling_linear_or_none(layer, x, x8)
    if shared is not None:
        return shared
    if fq8_gemm_enabled() and not _WEIGHT_CACHE["on"] and fq8_route_is_fused(x8.reshape(-1, K), [layer]):
        y = hip_fq8_linear_or_none(x8.reshape(-1, K), [layer])
        if y is not None:
            STATS.add(W.numel())                   # the weight's fake-quant call, computed inside the GEMM
            return y.reshape(*x.shape[:-1], W.shape[0])
    def make():
        STATS.add(W.numel())
        return FusedAmaxObsFakeQuantFunction.apply(W.detach(), False, True, fq.qmap, fq.amax_history, fq.scale,
                                                   fq.amax_history_len, fq.quant_max, None, False, False,
                                                   fq._qt_format, "only")
    w8 = _take_weight_codes(layer, (layer,))                   # BatchedWeightCodes ran this pass in front of the forward
    if w8 is not None:
        STATS.add(W.numel())
    else:
        w8 = cached_weight(layer, "fp8", make)
        if not _WEIGHT_CACHE["on"]:
            _note_weight_pass(layer, (layer,))
    one = _one(x.device)
    x2 = x8.reshape(-1, K)
    y = lt_fp8_gemm(x2, w8, layer.bias)
    if y is None:
        y = torch._scaled_mm(x2, w8.t(), scale_a=one, scale_b=one, bias=layer.bias, out_dtype=torch.bfloat16)
    return y.reshape(*x.shape[:-1], W.shape[0])


_LT = {"ok": True, "ws": {}}                  # "ok" turns False when libhipblaslt cannot be resolved in the process

# Which of hipBLASLt's suggestions runs a problem: (batch, M, N, K, b_is_kn, bias) -> index into the library's ordered list.
# Measured once on MI355X by tools/tune_lt_algos.py (profiles/r05_lt_algos.txt: microseconds of every suggestion) and COMMITTED, so
# every process, rank and box runs the same library kernel -- the same order of fp32 additions -- for a shape; rounds 1-4 timed the
# suggestions in each process and two ranks could disagree.  Shapes not listed run the library's first suggestion (index 0).
_LT_ALGO_TABLE = {
    (1, 6144, 768, 3072, 0, 1): 3,        # BERT-base output dense: 20.2 us against 22.1 for the first suggestion
    (1, 2048, 4096, 4096, 0, 0): 1,       # 32.2 / 35.5
    (1, 1024, 5120, 13824, 0, 0): 5,      # 71.1 / 80.0
    (1, 1024, 5120, 5120, 0, 0): 2,       # 31.8 / 33.4
    (40, 1024, 1024, 128, 0, 0): 1,       # 13B Q.K^T chain: 26.9 / 27.8
    (40, 1024, 128, 1024, 1, 0): 2,       # 13B P.V chain: 19.8 / 31.6
    (192, 384, 384, 64, 0, 0): 12,        # BERT-base Q.K^T chain: 56.3 / 66.6
    (192, 384, 64, 384, 1, 0): 5,         # BERT-base P.V chain: 13.5 / 32.6
}
# The positions above are positions in the suggestion list of ONE build of the library (hipblasLtGetVersion; ROCm 7.2.0: hipblasLtGetVersion = 100000;
# 1.2.1).  Under any other build the same index would silently name another kernel, so the table is ignored there: every shape runs the
# library's first suggestion and routes_report() says so ("lt:library_version").
_LT_ALGO_TABLE_LIBRARY = 100000
LT_ALGOS = {}              # what ran: "bxMxNxK[kn][+bias]" -> index (routes_report)
_LT_LIBRARY = {"version": None}


def lt_table_applies():
    """True when the resolved hipBLASLt is the build _LT_ALGO_TABLE was measured on (or does not report a version at all)."""
    if _LT_LIBRARY["version"] is None:
        _LT_LIBRARY["version"] = int(_native.lib().qt_fp8_gemm_library_version())
        if _LT_LIBRARY["version"] not in (0, _LT_ALGO_TABLE_LIBRARY):
            LT_ALGOS["lt:library_version"] = f"{_LT_LIBRARY['version']} != table's {_LT_ALGO_TABLE_LIBRARY}: first suggestion everywhere"
    return _LT_LIBRARY["version"] in (0, _LT_ALGO_TABLE_LIBRARY)


def lt_algo_index(batch, M, N, K, b_is_kn, with_bias):
    forced = os.environ.get("QT_LT_ALGO")                     # tools only: the tuner's A/B runs
    idx = int(forced) if forced is not None else _LT_ALGO_TABLE.get((batch, M, N, K, int(bool(b_is_kn)), int(bool(with_bias))), 0)
    if forced is None and idx and not lt_table_applies():
        idx = 0
    LT_ALGOS.setdefault(f"lt:{batch}x{M}x{N}x{K}{'kn' if b_is_kn else ''}{'+bias' if with_bias else ''}", idx)
    return idx


def lt_fp8_gemm(a8, b8, bias=None, b_is_kn=False):
    """C = A . op(B) on FP8 operands through qt_fp8_gemm (hipBLASLt; which of its suggestions runs is a committed table, _LT_ALGO_TABLE).  a8 [.., M, K];
    b8 [N, K] (b_is_kn False) or [.., K, N] (True); leading dims of a8 / b8 are a batch.  None when the library route is
    unavailable for this problem (the caller falls back to torch._scaled_mm / bf16)."""
    if not _LT["ok"] or os.environ.get("QT_LT_GEMM", "1") == "0":
        return None
    dev = a8.device
    ws = _LT["ws"].get(dev)
    if ws is None:
        ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        _LT["ws"][dev] = ws
    fmt = {torch.float8_e4m3fn: 0, torch.float8_e5m2: 1}
    if a8.dtype not in fmt or b8.dtype not in fmt or not a8.is_contiguous() or not b8.is_contiguous():
        return None
    M, K = a8.shape[-2], a8.shape[-1]
    if b_is_kn:
        N = b8.shape[-1]
        if b8.shape[-2] != K or b8.shape[:-2] != a8.shape[:-2]:
            return None
    else:
        N = b8.shape[-2]
        if b8.shape[-1] != K or (b8.dim() > 2 and b8.shape[:-2] != a8.shape[:-2]):
            return None
    batch = 1
    for d in a8.shape[:-2]:
        batch *= d
    b_stride = 0 if b8.dim() == 2 else N * K
    out = torch.empty(a8.shape[:-1] + (N,), dtype=torch.bfloat16, device=dev)
    rc = _native.lib().qt_fp8_gemm(a8.data_ptr(), fmt[a8.dtype], b8.data_ptr(), fmt[b8.dtype], int(b_is_kn), out.data_ptr(),
                                   bias.data_ptr() if bias is not None else None, batch, M, N, K, M * K, b_stride, M * N,
                                   ws.data_ptr(), ws.numel(), lt_algo_index(batch, M, N, K, b_is_kn, bias is not None), _stream_ptr(a8))
    if rc != 0:
        if rc in (_native.QT_ERR_NO_DEVICE,):
            _LT["ok"] = False                      # the library cannot be resolved in this process: stop trying
        return None
    return out


_ONES = {}


def _one(device):
    t = _ONES.get(device)
    if t is None:
        t = torch.ones((), dtype=torch.float32, device=device)
        _ONES[device] = t
    return t


def _operand(fq, device):
    """qt_operand_q for a weight fake-quantizer (per-tensor only)."""
    op = _native.QtOperandQ()
    op.fmt = fq._qt_format
    op.lut_dev = fq.qmap.data_ptr() if fq._qt_format.kind == _native.QT_FMT_LUT else None
    op.scale_f32_dev = None if fq.qscheme is None and getattr(fq, "_scale_is_one", True) else fq.scale.data_ptr()
    op.amax_bits_dev = fq.amax_history.data_ptr() if fq._observe else None
    return op


def fused_linear_or_none(layer, x):
    """y = x @ fq(W)^T + b with W fake-quantized while its tiles are staged (qt_linear_fq_bf16).
    Applies to bf16 device tensors under no_grad with a per-tensor weight fake-quantizer."""
    out = fp8_linear_or_none(layer, x)
    if out is not None:
        return out
    materialize_lazy(x)                      # every other route reads the values: a producer may have written the FP8 codes only
    out = fqt_linear_or_none(layer, x)
    if out is not None:
        return out
    fq = layer.weight_fake_quant
    W = layer.weight
    if not (fused_gemm_enabled() and isinstance(fq, FusedAmaxObsFakeQuantize)):
        return None
    if torch.is_grad_enabled() and (W.requires_grad or x.requires_grad):
        return None
    if not (x.device.type == "cuda" and x.dtype == torch.bfloat16 and W.dtype == torch.bfloat16):
        return None
    if fq.is_per_channel or fq.qscheme in (QScheme.MICROSCALING, QScheme.GROUP_WISE_AFFINE):
        return None
    if fq.outlier_threshold is not None or fq.record_histogram or not fq._quantize:
        return None
    if fq._qt_format.kind == _native.QT_FMT_LUT:
        return None       # table formats: elementwise pass (LDS-staged table) + library GEMM is faster
    K = W.shape[1]
    if K % 8 != 0 or not W.is_contiguous() or (layer.bias is not None and layer.bias.dtype != torch.bfloat16):
        return None
    L = _native.lib()
    fq._move_to(x.device)
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M, N = x2.shape[0], W.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    st = _stream_ptr(x)
    if fq._observe:
        if fq.amax_history.numel() == 0:
            fq.amax_history.resize_((fq.amax_history_len,)).fill_(0.0)
            fq.scale.resize_(()).fill_(1.0)
        launch_scale_update(fq.amax_history, fq.scale, fq.quant_max, fq.force_scale_power_of_two, st)
    qw = _operand(fq, x.device)
    qx = _native.QtOperandQ()
    qx.fmt = _IDENTITY
    code = L.qt_linear_fq_bf16(x2.data_ptr(), W.data_ptr(),
                               layer.bias.data_ptr() if layer.bias is not None else None,
                               y.data_ptr(), M, N, K, ctypes.byref(qx), ctypes.byref(qw), st)
    _native.check(code, "qt_linear_fq_bf16")
    STATS.add(W.numel())
    return y.reshape(*x.shape[:-1], N)


# ---- any stateless value map on the weight: bf16 GEMM with the map applied in its operand path (qt_linear_fqt_bf16) --------
_FQT_TABLES = {}          # (dtype, device) -> dict: rows / map device tensors, signed_rows, sign_mask, usable


def fqt_gemm_mode():
    """QT_FQT_GEMM: "auto" (default) -- the fused kernel for the problem shapes where it measured faster than the weight pass +
    library GEMM pair (a fixed rule, `fqt_route_is_fused`: no timing at run time, every rank and every run takes the same route);
    "1" -- wherever the kernel takes the problem; "0" -- never."""
    v = os.environ.get("QT_FQT_GEMM", "auto")
    return v if v in ("0", "1") else "auto"


def fqt_tables(fq, device):
    """Row form of `fq`'s value map on `device` (built once per dtype and device): None when the map has flagged rows inside
    the range weights live in (2^-40 .. 2^15), where the kernel would keep falling back to its slow path."""
    key = (str(fq.dtype), str(device))
    hit = _FQT_TABLES.get(key)
    if hit is None:
        import numpy as np
        m = _native.build_map_u16(fq.dtype)
        rp = _native.build_rowparams(m)
        rows = np.ctypeslib.as_array(rp.row).reshape(2048).astype(np.uint32)
        flagged = np.ctypeslib.as_array(rp.flagged)
        lo, hi = 127 - 40, 127 + 15
        usable = not flagged[lo:hi].any() and not (rp.signed_rows and flagged[256 + lo:256 + hi].any())
        hit = {"rows": torch.from_numpy(rows.view(np.int32).copy()).to(device), "map": torch.from_numpy(m.view(np.int16).copy()).to(device),
               "signed": int(rp.signed_rows), "mask": int(rp.sign_mask), "usable": bool(usable)}
        _FQT_TABLES[key] = hit
    return hit if hit["usable"] else None


def fqt_plan(M, n_total, K):
    """(ksplit, workspace bytes, tickets) qt_linear_fqt_ws_bf16 would use for this problem (host-only query)."""
    ks, wb, nt = ctypes.c_int(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _native.check(_native.lib().qt_linear_fqt_plan(M, n_total, K, ctypes.byref(ks), ctypes.byref(wb), ctypes.byref(nt)), "qt_linear_fqt_plan")
    return ks.value, wb.value, nt.value


def fqt_route_is_fused(M, ns, K, device):
    """Fixed routing rule for qt_linear_fqt_ws_bf16, from same-box A/B runs of the whole configs[3] window (tools/ab_13b_routes.py,
    profiles/r04_13b_route_ab.txt; MI355X, posit(8,2)): the kernel takes every Linear whose 512-row tiles are at least five column
    groups wide on a full chip -- q / k / v as one launch, the lm head, and since round 4 gate / up too: alone it ties with the weight
    pass + library GEMM (161-183 against 173-179 us), inside the window it is worth 2.4 ms of 37 (the pair's pass and GEMM compete
    with their neighbours for HBM) -- and, with split-K, narrow outputs with a deep K (down: 1024 x 5120 x 13824, three workgroups
    per 512 x 128 tile, 144 k steps each: 167-177 against 181-200 us).  It loses where a split's k range is too short to pay for
    the hand-off of the fp32 partial sums (o: 1024 x 5120 x 5120, 83 against 70 us; +0.2 ms per window), which stays on the pair."""
    mode = fqt_gemm_mode()
    if mode != "auto":
        return mode == "1"
    if M <= 256 or K % 32:
        return False
    ksplit = fqt_plan(M, sum(ns), K)[0]
    if ksplit > 1:
        return (K // 32) // ksplit >= 128
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    groups = sum(ns) // 16
    tiles_m = (M + 511) // 512
    tn_min = (groups + 7) // 8
    rounds = (tiles_m * tn_min + cus - 1) // cus
    tn = max(tn_min, rounds * cus // tiles_m)
    return groups / tn >= 5.0 and tiles_m * tn >= 0.9 * rounds * cus


def _fqt_route(M, ns, K, device):
    hit = fqt_route_is_fused(M, ns, K, device)
    _note_route("fqt", M, ns, K, "fused_value_map_gemm" if hit else "weight_pass+library_bf16_gemm")
    return hit


def hip_fqt_linear_or_none(x2, layers, tables):
    """y[M, sum N] = x2 . [fq(W_0); fq(W_1); ...]^T + bias through qt_linear_fqt_bf16: x2 [M, K] holds the bf16 VALUES of the already
    fake-quantized activation, the bf16 weights of `layers` go through the row form of their value map inside the kernel."""
    M, K = x2.shape
    if x2.dtype != torch.bfloat16 or K % 32 or not x2.is_contiguous() or x2.data_ptr() % 16 or not 1 <= len(layers) <= 4:
        return None
    for l in layers:
        W = l.weight
        if (W.dtype != torch.bfloat16 or not W.is_contiguous() or W.shape[1] != K or W.shape[0] % 16 or W.data_ptr() % 16 or W.device != x2.device):
            return None
        if l.bias is not None and (l.bias.dtype != torch.bfloat16 or not l.bias.is_contiguous() or l.bias.data_ptr() % 8 or l.bias.device != x2.device):
            return None
    n = len(layers)
    ns = [l.weight.shape[0] for l in layers]
    y = torch.empty((M, sum(ns)), dtype=torch.bfloat16, device=x2.device)
    wp = (ctypes.c_void_p * n)(*[l.weight.data_ptr() for l in layers])
    bp = (ctypes.c_void_p * n)(*[(l.bias.data_ptr() if l.bias is not None else None) for l in layers])
    nn = (ctypes.c_int * n)(*ns)
    _native.note_device(x2.device.index)
    ws, tickets = _fqt_workspace(M, sum(ns), K, x2.device)
    rc = _native.lib().qt_linear_fqt_ws_bf16(x2.data_ptr(), wp, bp, nn, n, tables["rows"].data_ptr(), tables["signed"], tables["mask"],
                                              tables["map"].data_ptr(), y.data_ptr(), M, K,
                                              ws.data_ptr() if ws is not None else None, ws.numel() * 4 if ws is not None else 0,
                                              tickets.data_ptr() if tickets is not None else None, tickets.numel() if tickets is not None else 0,
                                              _stream_ptr(x2))
    if rc in (_native.QT_ERR_BAD_ARG, _native.QT_ERR_UNALIGNED, _native.QT_ERR_BAD_DTYPE):
        return None
    _native.check(rc, "qt_linear_fqt_ws_bf16")
    return y


# Split-K scratch (qt_linear_fqt_ws_bf16): an fp32 workspace and a ticket array that ONE ordered sequence of
# launches may share -- so they are keyed by (kernel, device, stream, capture): every eager stream and every captured graph has its
# own (include/qt_hip.h: "launches that share a workspace must be ordered on one stream"; a graph replay may overlap eager work on
# another stream, and a second capture may be replayed beside the first).  Inside a capture the buffers come from the graph's
# memory pool (the zero fill of the tickets becomes a node of the graph; every launch leaves them zero anyway).  Outgrown buffers
# stay allocated: a hipGraph captured earlier replays launches that hold their addresses.
_SPLITK_WS = {}
_SPLITK_WS_OLD = []


def _capture_id(stream_ptr):
    cid = ctypes.c_ulonglong(0)
    _native.check(_native.lib().qt_stream_capture_id(stream_ptr, ctypes.byref(cid)), "qt_stream_capture_id")
    return cid.value


def splitk_scratch(kind, wbytes, ntick, device):
    """(workspace, tickets) for the launch about to be issued on `device`'s current stream."""
    st = torch.cuda.current_stream(device).cuda_stream
    key = (kind, device.index if device.index is not None else torch.cuda.current_device(), st, _capture_id(ctypes.c_void_p(st)))
    ws, tickets = _SPLITK_WS.get(key, (None, None))
    if ws is None and len(_SPLITK_WS) > 8:
        # entries of captures that have ended: the graph's pool keeps their memory alive for its replays, this table need not
        for k in [k for k in _SPLITK_WS if k[3] not in (0, key[3])]:
            del _SPLITK_WS[k]
    if ws is None or ws.numel() * 4 < wbytes or tickets.numel() < ntick:
        if ws is not None:
            _SPLITK_WS_OLD.append((ws, tickets))
        ws = torch.empty(((wbytes + 3) // 4,), dtype=torch.float32, device=device)
        tickets = torch.zeros((max(ntick, 4096),), dtype=torch.int32, device=device)
        _SPLITK_WS[key] = (ws, tickets)
    return ws, tickets


def _fqt_workspace(M, n_total, K, device):
    ksplit, wbytes, ntick = fqt_plan(M, n_total, K)
    if ksplit <= 1:
        return None, None
    return splitk_scratch("fqt", wbytes, ntick, device)


def _fqt_weight_ok(layer):
    fq = layer.weight_fake_quant
    return (isinstance(fq, FusedAmaxObsFakeQuantize) and fq.stateless_map() and fq._qt_format.kind != _native.QT_FMT_IDENTITY
            and not fq.fp8_exact())


def fqt_linear_or_none(layer, x):
    """The fake-quant Linear of a stateless non-FP8 weight spec (posit, intN, fp6 / fp4 without `qs`) as ONE launch: the weight's
    value map is applied inside the bf16 GEMM, fq(W) is never written.  q / k / v projections that read the same fake-quantized
    tensor (same stateless input format) share a launch through their SiblingGroup.  None when the rule above prefers the weight
    pass + library GEMM, or the kernel does not take the problem."""
    if fqt_gemm_mode() == "0" or not _fqt_weight_ok(layer):
        return None
    W = layer.weight
    if torch.is_grad_enabled() and (W.requires_grad or x.requires_grad):
        return None
    if not (x.device.type == "cuda" and x.dtype == torch.bfloat16 and W.dtype == torch.bfloat16) or _WEIGHT_CACHE["on"]:
        return None
    K = W.shape[1]
    fq = layer.weight_fake_quant
    tables = fqt_tables(fq, x.device)
    if tables is None:
        return None
    x2 = x.reshape(-1, K)
    M = x2.shape[0]
    group = layer.__dict__.get("_qt_sibling_group")
    if group is not None and group.value_map_only and os.environ.get("QT_GATE_UP_GROUP", "1") == "0":      # A/B hook (tools/ab_13b_routes.py)
        group = None
    if group is not None and os.environ.get("QT_SIBLING_GEMM", "1") != "0":
        idx = group.layers.index(layer)
        Ns = [l.weight.shape[0] for l in group.layers]
        key = _origin_key(x)
        if idx > 0:
            st = group.stash
            if st is not None and st[0] == ("fqt",) + tuple(key) and not st[2][idx]:
                st[2][idx] = True
                STATS.add(W.numel())
                off = sum(Ns[:idx])
                return st[1][:, off:off + Ns[idx]].reshape(*x.shape[:-1], Ns[idx])
        else:
            group.stash = None
            same = all(_fqt_weight_ok(l) and str(l.weight_fake_quant.dtype) == str(fq.dtype) and l.weight.shape[1] == K
                       and (l.bias is None) == (layer.bias is None) for l in group.layers)
            afq = [getattr(l, "activation_pre_process", None) for l in group.layers]
            afq = [h["0"] if h is not None and "0" in h else None for h in afq]
            same = same and all(isinstance(f, FusedAmaxObsFakeQuantize) and f.stateless_map() and str(f.dtype) == str(afq[0].dtype) for f in afq)
            # x is the fake-quantized image of one tensor: a hook's output (`_qt_origin`), or a producer kernel's result that the leader's
            # hook handed through (model_fusions.rmsnorm_map; the siblings' hooks then name it as their origin)
            shared = getattr(x, "_qt_origin", None) is not None or (getattr(x, "_qt_fq_done_by", None) is not None and handover_valid(x))
            if same and shared and _fqt_route(M, Ns, K, x.device):
                if not x2.is_contiguous():
                    x2 = x2.contiguous()
                y = hip_fqt_linear_or_none(x2, group.layers, tables)
                if y is not None:
                    STATS.add(W.numel())
                    group.stash = (("fqt",) + tuple(key), y, [True] + [False] * (len(Ns) - 1))
                    return y[:, :Ns[0]].reshape(*x.shape[:-1], Ns[0])
    if not _fqt_route(M, [W.shape[0]], K, x.device):
        return None
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = hip_fqt_linear_or_none(x2, [layer], tables)
    if y is None:
        return None
    STATS.add(W.numel())
    return y.reshape(*x.shape[:-1], W.shape[0])


def _has_hooks(mod, name):
    holder = getattr(mod, name, None)
    return holder is not None


def fused_scores_to_probs_or_none(attn, scores, attention_mask, scaling, dropout, value):
    """scale -> +mask -> softmax -> fake-quant(probabilities) in ONE HIP pass (qt_softmax_fq_bf16), then
    av_matmul on the quantized probabilities.  Applies when nothing observes the intermediate tensors:
    `attn_scaling` and `softmax` carry no activation hooks (the `--quantize_forward gemm` default), no
    dropout is active, bf16 device tensors, and av_matmul's per-tensor fake-quantizers already exist
    (they are created by the first, unfused, call).  Returns (probs_q, attn_output) or None."""
    if os.environ.get("QT_FUSED_SOFTMAX", "1") == "0":
        return None
    if not (scores.device.type == "cuda" and scores.dtype == torch.bfloat16 and scores.dim() == 4):
        return None
    if torch.is_grad_enabled() and scores.requires_grad:
        return None
    if dropout and attn.training:
        return None
    if _has_hooks(attn.attn_scaling, "activation_pre_process") or _has_hooks(attn.softmax, "activation_pre_process"):
        return None
    if attn.attn_scaling._forward_hooks or attn.softmax._forward_hooks or attn.av_matmul._forward_hooks:
        return None
    holder = getattr(attn.av_matmul, "activation_pre_process", None)
    fq_p = fq_v = None
    if holder is not None:
        if "0" not in holder or "1" not in holder:
            return None                       # first call: let the hook create them
        fq_p, fq_v = holder["0"], holder["1"]
        if not isinstance(fq_p, FusedAmaxObsFakeQuantize) or fq_p.is_per_channel or fq_p.outlier_threshold is not None \
                or fq_p.record_histogram or fq_p.qscheme in (QScheme.MICROSCALING, QScheme.GROUP_WISE_AFFINE):
            return None
    if len(attn.av_matmul._forward_pre_hooks) > (1 if holder is not None else 0):
        return None
    B, H, Q, C = scores.shape
    if C % 8 != 0 or C > 4096 or not scores.is_contiguous():
        return None
    mask = None
    msb = msh = msq = 0
    if attention_mask is not None:
        m = attention_mask[..., :C]
        if m.dtype != torch.bfloat16 or m.dim() != 4 or m.stride(-1) != 1 or m.device != scores.device:
            return None
        if m.shape[0] not in (1, B) or m.shape[1] not in (1, H) or m.shape[2] not in (1, Q):
            return None
        msb = m.stride(0) if m.shape[0] == B and B > 1 else 0
        msh = m.stride(1) if m.shape[1] == H and H > 1 else 0
        msq = m.stride(2) if m.shape[2] == Q and Q > 1 else 0
        if (msb | msh | msq) % 8 != 0 or m.data_ptr() % 16 != 0:
            return None
        mask = m
    L = _native.lib()
    st = _stream_ptr(scores)
    fp8_out = _fp8_probs_times_v_or_none(L, st, scores, mask, msb, msh, msq, scaling, fq_p, fq_v, value, mask_owner=attention_mask)
    if fp8_out is not None:
        return None, fp8_out                  # the probabilities exist only as FP8 codes on this path
    out = torch.empty_like(scores)
    if fq_p is not None and (fq_p._observe or fq_p._quantize):
        fq_p._move_to(scores.device)
        fmt = fq_p._qt_format if fq_p._quantize else _IDENTITY
        if fq_p._observe:
            if fq_p.amax_history.numel() == 0:
                fq_p.amax_history.resize_((fq_p.amax_history_len,)).fill_(0.0)
                fq_p.scale.resize_(()).fill_(1.0)
            launch_scale_update(fq_p.amax_history, fq_p.scale, fq_p.quant_max, fq_p.force_scale_power_of_two, st)
        if fmt.kind == _native.QT_FMT_LUT:
            from .fake_quantize import _launch_format
            fmt = _launch_format(fmt, fq_p.qmap)             # the row form of the map where the allocation carries it
        lut = fq_p.qmap.data_ptr() if fmt.kind == _native.QT_FMT_LUT else None
        scale_ptr = fq_p.scale.data_ptr() if fq_p._quantize else None
        amax_ptr = fq_p.amax_history.data_ptr() if fq_p._observe else None
        STATS.add(scores.numel())
    else:
        fmt, lut, scale_ptr, amax_ptr = _IDENTITY, None, None, None
    _native.check(L.qt_softmax_fq_bf16(scores.data_ptr(), mask.data_ptr() if mask is not None else None, out.data_ptr(),
                                       B, H, Q, C, msb, msh, msq, float(scaling), ctypes.byref(fmt), lut, scale_ptr,
                                       amax_ptr, st), "qt_softmax_fq_bf16")
    v = fq_v(value) if fq_v is not None else value
    return out, torch.matmul(out, v)


def _mask_row_live(mask, owner, B, H, Q, C, st):
    """Per-row extent of the unmasked part of an additive mask (qt_mask_row_live), or None when the mask's rows are not evenly
    spaced.  Kept as an attribute of `owner` -- the tensor object the attention block was handed, of which `mask` is a view: the
    cached causal mask of a window evaluation is the same object for every layer and window, a mask Hugging Face builds per
    forward is the same object for every layer of that forward; a new object (or a new version of it) is scanned again.  The scan
    also leaves, behind the extents (element `rows` of the buffer), whether any row is NOT exactly "zeros, then the dtype's minimum":
    qt_attention_fp8 reads that on the device (no host read-back: also inside a stream capture).  Returns
    (row_live, stride_b, stride_h, stride_q) in rows."""
    if mask.dim() != 4:
        return None
    mb, mh, mq, _ = mask.shape
    rs = mask.stride(2) if mq > 1 else C
    if (mh > 1 and mask.stride(1) != mq * rs) or (mb > 1 and mask.stride(0) != mh * mq * rs) or rs < C:
        return None
    key = (mask.data_ptr(), owner._version, tuple(mask.shape), mask.stride())
    hit = getattr(owner, "_qt_row_live", None)
    if hit is not None and hit[0] == key:
        rl = hit[1]
    else:
        rows = mb * mh * mq
        rl = torch.empty(rows + 1, dtype=torch.int32, device=mask.device)
        _native.check(_native.lib().qt_mask_row_live_checked(mask.data_ptr(), rows, C, rs, rl.data_ptr(), rl.data_ptr() + 4 * rows, st),
                      "qt_mask_row_live_checked")
        owner._qt_row_live = (key, rl)
    return rl, (mh * mq if mb == B and B > 1 else 0), (mq if mh == H and H > 1 else 0), (1 if mq == Q and Q > 1 else 0)


def _fp8_probs_times_v_or_none(L, st, scores, mask, msb, msh, msq, scaling, fq_p, fq_v, value, mask_owner=None):
    """When the probabilities' and the values' fake-quantizers are stateless E4M3 / E5M2 ones, both tensors are exactly
    FP8: the score pass writes the probabilities' FP8 code only (1 B/element instead of 2), the value pass writes FP8
    next to bf16, and P.V runs as a batched FP8 GEMM (qt_fp8_gemm).  Same products, fp32 accumulation."""
    if (os.environ.get("QT_FP8_ATTENTION", "1") == "0" or os.environ.get("QT_LT_GEMM", "1") == "0" or not _LT["ok"]
            or fq_p is None or fq_v is None):
        return None
    if not (isinstance(fq_v, FusedAmaxObsFakeQuantize) and fq_p.producer_fusable() and fq_v.producer_fusable()):
        return None
    B, H, Q, C = scores.shape
    if value.dim() != 4 or value.shape[:3] != (B, H, C) or value.dtype != torch.bfloat16 or value.stride(-1) != 1:
        return None
    D = value.shape[-1]
    if D % 16 or C % 16 or any(s % 8 for s in value.stride()[:3]) or value.data_ptr() % 16 or B * H > 65535:
        return None
    if (B * H, Q, C, D) in _LT.setdefault("no_pv", set()):
        return None                            # the library had no kernel for this problem last time
    done = getattr(value, "_qt_fq_done_by", None) if handover_valid(value) else None
    v8 = getattr(value, "_qt_fp8", None)
    if not (done is fq_v and v8 is not None and value.is_contiguous()):
        v8u = torch.empty((B, H, C, D), dtype=torch.uint8, device=value.device)     # only the codes feed the FP8 P.V GEMM
        _native.check(L.qt_fake_quant_rows_bf16_fp8(value.data_ptr(), None, v8u.data_ptr(), B, H, C, D,
                                                    value.stride(0), value.stride(1), value.stride(2),
                                                    ctypes.byref(fq_v._qt_format), st), "qt_fake_quant_rows_bf16_fp8")
        v8 = v8u.view(torch.float8_e5m2 if fq_v._qt_format.p0 == 2 else torch.float8_e4m3fn)
    p8u = torch.empty((B, H, Q, C), dtype=torch.uint8, device=scores.device)
    # (qt_softmax_fq_bf16_fp8_live -- the same pass told each mask row's extent -- is bit-identical and gains a tenth alone on cold
    # buffers, nothing inside the window where the scores the Q.K^T GEMM just wrote are cache-resident: 12.97 against 12.88 ms; it
    # stays in the C ABI, the module path does not take it)
    _native.check(L.qt_softmax_fq_bf16_fp8(scores.data_ptr(), mask.data_ptr() if mask is not None else None, None,
                                           p8u.data_ptr(), B, H, Q, C, msb, msh, msq, float(scaling),
                                           ctypes.byref(fq_p._qt_format), st), "qt_softmax_fq_bf16_fp8")
    p8 = p8u.view(torch.float8_e5m2 if fq_p._qt_format.p0 == 2 else torch.float8_e4m3fn)
    out = lt_fp8_gemm(p8.view(B * H, Q, C), v8.view(B * H, C, D), None, b_is_kn=True)
    if out is None:
        _LT["no_pv"].add((B * H, Q, C, D))     # the bf16 path redoes (and counts) the two passes
        return None
    STATS.add(value.numel())                   # the two fake-quant calls the reference issues here (av_matmul's inputs)
    STATS.add(scores.numel())
    return out.view(B, H, Q, D)


def _mask_strides(attention_mask, B, H, Q, C, device, align):
    """(mask view, stride_b, stride_h, stride_q) of a broadcastable additive bf16 mask, or False if it cannot
    be consumed in place; None mask -> (None, 0, 0, 0)."""
    if attention_mask is None:
        return None, 0, 0, 0
    m = attention_mask[..., :C]
    if m.dtype != torch.bfloat16 or m.dim() != 4 or m.stride(-1) != 1 or m.device != device:
        return False
    if m.shape[0] not in (1, B) or m.shape[1] not in (1, H) or m.shape[2] not in (1, Q):
        return False
    sb = m.stride(0) if m.shape[0] == B and B > 1 else 0
    sh = m.stride(1) if m.shape[1] == H and H > 1 else 0
    sq = m.stride(2) if m.shape[2] == Q and Q > 1 else 0
    if (sb | sh | sq) % align != 0 or m.data_ptr() % (2 * align) != 0:
        return False
    return m, sb, sh, sq


def _attention_fp8_or_none(attn, query, key, value, attention_mask, scaling, fqs):
    """qt_attention_fp8: the attention core in one launch on FP8 codes (head_dim 128 or 64, keys in blocks of 128 up to 1024) when the
    four fake-quantizers around the two matmuls are stateless E4M3 / E5M2 ones of one format.  q / k either arrive with their codes (the
    rotary kernel attaches them) or are [B, H, S, D] views of the projections' outputs (BERT's transpose_for_scores), whose codes a
    codes-only pass over the view writes here -- that pass IS the fq_q / fq_k call.  Counts the four fake-quant calls the reference
    issues: q and k (handed through by their modules, or the passes just named), the value pass (qt_value_codes_t) and the
    probabilities inside the kernel.  Returns [B, Sq, H, D] or None."""
    fq_q, fq_k, fq_p, fq_v = fqs
    B, H, Q, D = query.shape
    C = key.shape[2]
    if D not in (64, 128) or C % 128 != 0 or C > 1024 or B * H > 65535 or value.dtype != torch.bfloat16 or value.stride(-1) != 1:
        return None
    if not all(isinstance(f, FusedAmaxObsFakeQuantize) and f.producer_fusable() for f in fqs):
        return None
    if len({f._qt_format.key() for f in fqs}) != 1 or any(s % 8 for s in value.stride()[:3]) or value.data_ptr() % 16:
        return None

    def handed(t, fq, rows):
        t8 = getattr(t, "_qt_fp8", None)
        if t8 is None or not handover_valid(t) or t._qt_fq_done_by is not fq or not t8.is_contiguous() or tuple(t8.shape) != (B, H, rows, D):
            return None
        return t8

    def view_ok(t):
        return t.dtype == torch.bfloat16 and t.stride(-1) == 1 and not any(s % 8 for s in t.stride()[:3]) and t.data_ptr() % 16 == 0

    q8, k8 = handed(query, fq_q, Q), handed(key, fq_k, C)
    if (q8 is None and not view_ok(query)) or (k8 is None and not view_ok(key)):
        return None
    mk = _mask_strides(attention_mask, B, H, Q, C, query.device, 4)
    if mk is False:
        return None
    mask, msb, msh, msq = mk
    L = _native.lib()
    st = _stream_ptr(query)
    rl_ptr, lsb, lsh, lsq, irregular_ptr = None, 0, 0, 0, None
    if mask is not None:
        live = _mask_row_live(mask, attention_mask, B, H, Q, C, st)
        if live is not None:
            rl, lsb, lsh, lsq = live
            rl_ptr = rl.data_ptr()
            irregular_ptr = rl_ptr + 4 * (rl.numel() - 1)    # 0: every row is "zeros, then the minimum" and the mask is not read

    def codes(t, fq, t8, rows):
        if t8 is not None:
            # hand-over, counted as the fake-quantizer's own forward would (calling it would also decode codes-only tensors, see
            # model_fusions.rope_fq: nothing here reads the bf16 values)
            fq.__dict__["_qt_calls"] = fq.__dict__.get("_qt_calls", 0) + 1
            STATS.add(t.numel())
            return t8
        t8 = torch.empty((B, H, rows, D), dtype=torch.uint8, device=t.device)
        _native.check(L.qt_fake_quant_rows_bf16_fp8(t.data_ptr(), None, t8.data_ptr(), B, H, rows, D, t.stride(0), t.stride(1), t.stride(2),
                                                    ctypes.byref(fq._qt_format), st), "qt_fake_quant_rows_bf16_fp8")
        STATS.add(t.numel())
        return t8

    def token_rows(t):
        """Row stride when `t` is the [B, H, S, D] view of a [B, S, H, D]-ordered buffer (transpose_for_scores of a projection), else None."""
        b_, h_, s_, d_ = t.shape
        rs = t.stride(2)
        return rs if (t.stride(3) == 1 and t.stride(1) == d_ and t.stride(0) == s_ * rs and rs % 8 == 0 and rs >= h_ * d_) else None

    vt8 = None
    if (q8 is None and k8 is None and Q == C and key.shape[1] == H and value.shape[1] == H and token_rows(query) is not None
            and token_rows(key) is not None and os.environ.get("QT_ROPE_VALUE_LAUNCH", "1") != "0"):
        # q and k codes and the value codes in ONE launch (qt_rope_fq_value without a rotation): the three calls fq_q, fq_k, fq_v
        q8 = torch.empty((B, H, Q, D), dtype=torch.uint8, device=query.device)
        k8 = torch.empty((B, H, C, D), dtype=torch.uint8, device=query.device)
        vt8 = torch.empty((B, H, D, C), dtype=torch.uint8, device=query.device)
        _native.check(L.qt_rope_fq_value(query.data_ptr(), key.data_ptr(), None, None, None, None, q8.data_ptr(), k8.data_ptr(), B, Q, H, H, D,
                                         token_rows(query), token_rows(key), ctypes.byref(fq_q._qt_format), ctypes.byref(fq_k._qt_format),
                                         value.data_ptr(), vt8.data_ptr(), value.stride(0), value.stride(1), value.stride(2),
                                         ctypes.byref(fq_v._qt_format), st), "qt_rope_fq_value")
        STATS.add(query.numel())
        STATS.add(key.numel())
    else:
        q8, k8 = codes(query, fq_q, q8, Q), codes(key, fq_k, k8, C)
    fmt = fq_v._qt_format
    early = attn.__dict__.pop("_qt_vt8", None)
    if vt8 is not None:
        pass                                                 # written by the launch above
    elif early is not None and early[0] == value_key(value) and early[1] is fq_v:
        vt8 = early[2]                                       # written by the launch that carried the rotary kernel (model_fusions.rope_fq)
    else:
        vt8 = torch.empty((B, H, D, C), dtype=torch.uint8, device=query.device)
        _native.check(L.qt_value_codes_t(value.data_ptr(), vt8.data_ptr(), B, H, C, D, value.stride(0), value.stride(1), value.stride(2),
                                         ctypes.byref(fmt), st), "qt_value_codes_t")
    STATS.add(value.numel())                                 # fq_v, evaluated by that pass
    STATS.add(B * H * Q * C)                                 # fq_p, evaluated inside the kernel
    out = torch.empty((B, Q, H, D), dtype=torch.bfloat16, device=query.device)
    # the output projection's stateless FP8 input fake-quantizer rides on the epilogue (as model_fusions.attention_output does for the
    # library-GEMM chain): HF reshapes the result before the projection's hook sees it, so the hand-over is an expectation
    from .model_fusions import consumer_fq
    proj = getattr(attn, "o_proj", None) or attn.__dict__.get("_qt_out_proj")       # LLaMA's own / BERT's BertSelfOutput.dense
    fq_o = consumer_fq(proj) if (proj is not None and os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0") else None
    out8 = torch.empty((B, Q, H, D), dtype=torch.uint8, device=query.device) if fq_o is not None else None
    _native.check(L.qt_attention_fp8(q8.data_ptr(), k8.data_ptr(), vt8.data_ptr(), 1 if fmt.p0 == 2 else 0,
                                     mask.data_ptr() if mask is not None else None, msb, msh, msq, rl_ptr, lsb, lsh, lsq, 0, irregular_ptr,
                                     out.data_ptr(), out8.data_ptr() if out8 is not None else None,
                                     ctypes.byref(fq_o._qt_format) if fq_o is not None else None, B, H, Q, C, D, float(scaling), st),
                  "qt_attention_fp8")
    if fq_o is not None:
        fq_o.expect_prequantized(out, out8.view(torch.float8_e5m2 if fq_o._qt_format.p0 == 2 else torch.float8_e4m3fn))
    return out


def fused_attention_or_none(attn, query, key, value, attention_mask, scaling, dropout):
    """The whole attention core in ONE HIP launch (qt_attention_fq_bf16): QK^T, scaling, mask, softmax,
    fake-quant of the probabilities and P.V on the matrix cores, the S x S tensor never written.  q, k, v go
    through their own per-tensor fake-quantizers first (elementwise passes that also write the contiguous
    [B, H, S, D] layout).  Same applicability rules as the fused score path, plus head_dim in {64, 128};
    returns the attention output already in [B, Sq, H, D], or None.

    QT_FUSED_ATTENTION: "0" never, "1" whenever applicable, unset = only head_dim 64, where the kernel measured
    faster than the library-GEMM chain (B16 H12 S384: 85 vs 113 us); at head_dim 128 the chain wins (92 vs 122 us at
    B1 H32 S1024), so the chain stays the default there."""
    mode = os.environ.get("QT_FUSED_ATTENTION", "auto")
    fp8_kernel = query.dim() == 4 and query.shape[-1] in (64, 128) and os.environ.get("QT_FP8_ATTENTION_KERNEL", "1") != "0"
    if mode == "0" or (mode != "1" and query.shape[-1] != 64 and not fp8_kernel):
        return None
    if not (query.device.type == "cuda" and query.dtype == torch.bfloat16 and query.dim() == 4):
        return None
    if torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad):
        return None
    if dropout and attn.training:
        return None
    B, H, Q, D = query.shape
    C = key.shape[2]
    if D not in (64, 128) or C % 4 != 0 or key.shape != (B, H, C, D) or value.shape != (B, H, C, D) or B * H > 65535:
        return None
    for mod in (attn.attn_scaling, attn.softmax):
        if _has_hooks(mod, "activation_pre_process") or mod._forward_hooks or mod._forward_pre_hooks:
            return None
    for mod in (attn.qk_matmul, attn.av_matmul):
        if mod._forward_hooks:
            return None
    hq, hv = getattr(attn.qk_matmul, "activation_pre_process", None), getattr(attn.av_matmul, "activation_pre_process", None)
    if (hq is None) != (hv is None):
        return None
    fq_q = fq_k = fq_p = fq_v = None
    if hq is not None:
        if not all(k in hq for k in ("0", "1")) or not all(k in hv for k in ("0", "1")):
            return None                       # first call: let the hooks create the fake-quantizers
        fq_q, fq_k, fq_p, fq_v = hq["0"], hq["1"], hv["0"], hv["1"]
        for f in (fq_q, fq_k, fq_p, fq_v):
            if not isinstance(f, FusedAmaxObsFakeQuantize) or f.is_per_channel or f.outlier_threshold is not None \
                    or f.record_histogram or f.qscheme in (QScheme.MICROSCALING, QScheme.GROUP_WISE_AFFINE):
                return None
        if len(attn.qk_matmul._forward_pre_hooks) != 1 or len(attn.av_matmul._forward_pre_hooks) != 1:
            return None
    elif attn.qk_matmul._forward_pre_hooks or attn.av_matmul._forward_pre_hooks:
        return None
    if fp8_kernel and fq_q is not None and os.environ.get("QT_FP8_ATTENTION", "1") != "0":
        out = _attention_fp8_or_none(attn, query, key, value, attention_mask, scaling, (fq_q, fq_k, fq_p, fq_v))
        if out is not None:
            return out
    table_p = (fq_p is not None and fq_p._quantize and fq_p._qt_format.kind == _native.QT_FMT_LUT)
    if mode != "1" and query.shape[-1] != 64 and not table_p:
        # head_dim 128 without the FP8 kernel: the library-GEMM chain (see above) -- except for table formats (posit, fpN, ...),
        # whose chain pays an LDS-table softmax pass and three strided fake-quant passes: LLaMA-2-13B posit8_2 window 42.6 -> 40.6 ms
        return None
    mk = _mask_strides(attention_mask, B, H, Q, C, query.device, 4)
    if mk is False:
        return None
    mask, msb, msh, msq = mk
    L = _native.lib()
    st = _stream_ptr(query)
    qq = (fq_q(query) if fq_q is not None else query).contiguous()
    kq = (fq_k(key) if fq_k is not None else key).contiguous()          # K, not K^T: elementwise, same statistics
    out = torch.empty((B, Q, H, D), dtype=torch.bfloat16, device=query.device)
    if fq_p is not None and (fq_p._observe or fq_p._quantize):
        fq_p._move_to(query.device)
        fmt = fq_p._qt_format if fq_p._quantize else _IDENTITY
        if fq_p._observe:
            if fq_p.amax_history.numel() == 0:
                fq_p.amax_history.resize_((fq_p.amax_history_len,)).fill_(0.0)
                fq_p.scale.resize_(()).fill_(1.0)
            launch_scale_update(fq_p.amax_history, fq_p.scale, fq_p.quant_max, fq_p.force_scale_power_of_two, st)
        lut = fq_p.qmap.data_ptr() if fmt.kind == _native.QT_FMT_LUT else None
        if lut is not None:
            from .fake_quantize import _launch_format
            fmt = _launch_format(fmt, fq_p.qmap)             # the row form of the map where the allocation carries it
        unit_scale = fq_p.qscheme is None and getattr(fq_p, "_scale_is_one", True)      # no scale tensor at all
        scale_ptr = fq_p.scale.data_ptr() if (fq_p._quantize and not unit_scale) else None
        amax_ptr = fq_p.amax_history.data_ptr() if fq_p._observe else None
        STATS.add(B * H * Q * C)
    else:
        fmt, lut, scale_ptr, amax_ptr = _IDENTITY, None, None, None
    # table formats: the output projection's stateless input fake-quantizer of the SAME format rides on the kernel's epilogue (its hook
    # then hands the result through: fake_quantize.expect_prequantized)
    fq_o = None
    if (table_p and scale_ptr is None and amax_ptr is None and (fmt.p1 & 1) and os.environ.get("QT_FUSED_PRODUCER_FQ", "1") != "0"
            and os.environ.get("QT_FUSED_PRODUCER_MAP", "1") != "0"):
        from .model_fusions import consumer_fq_map
        proj = getattr(attn, "o_proj", None) or attn.__dict__.get("_qt_out_proj")
        cand = consumer_fq_map(proj) if proj is not None else None
        if cand is not None and cand.dtype == fq_p.dtype:
            fq_o = cand
    if not (table_p and scale_ptr is None and amax_ptr is None and fq_v is not None and not _has_table_hooks(fq_v)
            and attention_rows_or_none(L, st, qq, kq, value, fq_v, mask, attention_mask, (msb, msh, msq), out, (B, H, Q, C, D), scaling, fmt, lut,
                                       fq_o is not None, attn=attn)):
        vq = (fq_v(value) if fq_v is not None else value).contiguous()
        launch_attention_fq(L, st, qq, kq, vq, mask, attention_mask, (msb, msh, msq), out, (B, H, Q, C, D), scaling, fmt, lut, scale_ptr, amax_ptr,
                            fq_o is not None)
    if fq_o is not None:
        fq_o.expect_prequantized(out, None)
    return out


def _has_table_hooks(fq):
    """A fake-quantizer somebody hooked (forward hooks / pre-hooks) must run as the module it is."""
    return bool(fq._forward_hooks or fq._forward_pre_hooks)


def attention_rows_or_none(L, st, qq, kq, value, fq_v, mask, mask_owner, mask_strides, out, dims, scaling, fmt, lut, out_fq, attn=None):
    """qt_attention_rows_bf16 (round 4): the table-format attention core with the score strip in registers -- head_dim 128, key counts in
    blocks of 128 up to 1024, the probabilities' fake-quantizer a stateless table format in its row form (fmt / lut as handed to
    launch_attention_fq), and `fq_v` a stateless table format in its row form too: its call IS the kernel's value pass
    (qt_value_t_rows: fq_v(value), transposed, keys in the k-slot order), counted here.  `value`: the UNQUANTIZED [B, H, C, D] view.
    Returns True when the launch was made; False: the caller takes the two-pass kernel."""
    B, H, Q, C, D = dims
    if D != 128 or C % 128 != 0 or C > 1024 or B * H > 65535 or not (fmt.kind == _native.QT_FMT_LUT and (fmt.p1 & 1)) or lut is None:
        return False
    if not (isinstance(fq_v, FusedAmaxObsFakeQuantize) and fq_v.stateless_map() and fq_v._qt_format.kind == _native.QT_FMT_LUT):
        return False
    from .fake_quantize import _launch_format
    fq_v._move_to(value.device)
    vfmt = _launch_format(fq_v._qt_format, fq_v.qmap)
    if not (vfmt.p1 & 1) or value.dtype != torch.bfloat16 or value.stride(3) != 1 or any(s_ % 8 for s_ in value.stride()[:3]) or value.data_ptr() % 16:
        return False
    msb, msh, msq = mask_strides
    # everything qt_attention_rows_bf16 would refuse, BEFORE the value pass is issued and counted
    if (qq.data_ptr() | kq.data_ptr() | lut) % 16 or out.data_ptr() % 8:
        return False
    if mask is not None and (mask.data_ptr() % 8 or (msb | msh | msq) % 4):
        return False
    live = _mask_row_live(mask, mask_owner, B, H, Q, C, st) if (mask is not None and mask_owner is not None) else None
    early = attn.__dict__.pop("_qt_vt_rows", None) if attn is not None else None
    if early is not None and early[0] == value_key(value) and early[1] is fq_v:
        vt = early[2]                                          # written by the launch that carried the rotary kernel (model_fusions.rope_map)
    else:
        vt = torch.empty((B, H, D, C), dtype=torch.bfloat16, device=value.device)
        _native.check(L.qt_value_t_rows(value.data_ptr(), vt.data_ptr(), B, H, C, D, value.stride(0), value.stride(1), value.stride(2),
                                        ctypes.byref(vfmt), fq_v.qmap.data_ptr(), st), "qt_value_t_rows")
    STATS.add(value.numel())                                   # the fq_v call
    fq_v.__dict__["_qt_calls"] = fq_v.__dict__.get("_qt_calls", 0) + 1
    if live is not None:
        rl, lsb, lsh, lsq = live
        rlp, irr = rl.data_ptr(), rl.data_ptr() + 4 * (rl.numel() - 1)
    else:
        rlp, irr, lsb, lsh, lsq = None, None, 0, 0, 0
    _native.check(L.qt_attention_rows_bf16(qq.data_ptr(), kq.data_ptr(), vt.data_ptr(), mask.data_ptr() if mask is not None else None, msb, msh, msq,
                                           rlp, lsb, lsh, lsq, irr, out.data_ptr(), int(bool(out_fq)), ctypes.byref(fmt), lut, B, H, Q, C, D,
                                           float(scaling), st), "qt_attention_rows_bf16")
    return True


def launch_attention_fq(L, st, qq, kq, vq, mask, mask_owner, mask_strides, out, dims, scaling, fmt, lut, scale_ptr, amax_ptr, out_fq):
    """qt_attention_fq_bf16 / _out_ / _live_: with a mask whose rows are evenly spaced the launch carries the mask's row extents
    (_mask_row_live: scanned once per mask object), and the kernel does not read a causal / right-padding mask at all."""
    B, H, Q, C, D = dims
    msb, msh, msq = mask_strides
    live = None
    if mask is not None and mask_owner is not None:
        live = _mask_row_live(mask, mask_owner, B, H, Q, C, st)
    if live is not None:
        rl, lsb, lsh, lsq = live
        _native.check(L.qt_attention_fq_live_bf16(qq.data_ptr(), kq.data_ptr(), vq.data_ptr(), mask.data_ptr(), out.data_ptr(), B, H, Q, C, D,
                                                  msb, msh, msq, float(scaling), ctypes.byref(fmt), lut, scale_ptr, amax_ptr, int(bool(out_fq)),
                                                  rl.data_ptr(), lsb, lsh, lsq, rl.data_ptr() + 4 * (rl.numel() - 1), st),
                      "qt_attention_fq_live_bf16")
    elif out_fq:
        _native.check(L.qt_attention_fq_out_bf16(qq.data_ptr(), kq.data_ptr(), vq.data_ptr(), mask.data_ptr() if mask is not None else None,
                                                 out.data_ptr(), B, H, Q, C, D, msb, msh, msq, float(scaling), ctypes.byref(fmt), lut, st),
                      "qt_attention_fq_out_bf16")
    else:
        _native.check(L.qt_attention_fq_bf16(qq.data_ptr(), kq.data_ptr(), vq.data_ptr(), mask.data_ptr() if mask is not None else None,
                                             out.data_ptr(), B, H, Q, C, D, msb, msh, msq, float(scaling), ctypes.byref(fmt), lut,
                                             scale_ptr, amax_ptr, st), "qt_attention_fq_bf16")
