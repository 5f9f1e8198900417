"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see qt_oracle.py's header for who may import this).

numpy restatement of the END of the reference's fine-tuning step (H3):

    examples/text_classification/run_glue_no_trainer.py:655-668
        accelerator.clip_grad_norm_(model.parameters(), 1.0); optimizer.step()          # optimizer = torch.optim.AdamW(...), :469-474

The arithmetic lives in a third-party dependency, torch (pinned here: 2.10.0), not in the reference's tree:

  * torch/nn/utils/clip_grad.py (_get_total_norm, _clip_grads_with_norm_): per-tensor 2-norms through torch._foreach_norm -- fp32
    accumulation, result in the gradients' dtype --, total = linalg.vector_norm over the stacked norms, clip_coef = max_norm /
    (total + 1e-6) evaluated as tensor.reciprocal() * max_norm (torch/_tensor.py __rdiv__), clamp(max=1), grads *= coef in place.
    On bf16 gradients every tensor in that chain is bf16: each step rounds once.
  * ATen/native/cuda/fused_adam_utils.cuh:27-98 (adam_math, ADAM_MODE::ADAMW): hyper-parameters are doubles, state fp32 ("opmath"), the
    mixed expressions are evaluated in double and assigned back to fp32; one rounding to bf16 per stored value; bias corrections
    1 - pow(beta, step) in double (:128-136), passed on as fp32.  The device compiler contracts `a * x + c` of those double expressions
    into one fused multiply-add (first product fused, second rounded); _fma64 below restates that -- it decides the result when
    beta1 * exp_avg and (1 - beta1) * grad cancel.

Pinned: tests/test_optimizer_cpu.py checks this restatement against torch's own CPU AdamW + clip_grad_norm_ on float64 tensors (the
algorithm, to 1e-12) -- and the GPU tests check it bit for bit against torch.optim.AdamW(fused=True) behind torch's clip on the device.
Values travel as bf16 bit patterns (uint16 arrays), like everywhere in the oracle.
"""
import math

import numpy as np

from .qt_oracle import bf16_to_f32, f32_to_bf16

F32 = np.float32
F64 = np.float64


def _rbf(x):
    """fp32 value(s) rounded to bf16, back as fp32 (a scalar stays a scalar)"""
    x = np.asarray(x, dtype=F32)
    r = bf16_to_f32(f32_to_bf16(x.reshape(-1))).reshape(x.shape)
    return F32(r) if r.ndim == 0 else r


def clip_coefficient_bf16(grad_bits_list, max_norm):
    """(total_norm, coef) as torch.nn.utils.clip_grad_norm_(..., max_norm) forms them on bf16 gradients; both are bf16 values held
    in fp32.  Per-tensor sums of squares in float64 (torch: fp32, order-dependent in the last bits; then one rounding to bf16)."""
    norms = []
    for bits in grad_bits_list:
        g = bf16_to_f32(np.asarray(bits, dtype=np.uint16).ravel()).astype(F64)
        norms.append(float(_rbf(F32(math.sqrt(float(np.sum(g * g)))))))
    total = _rbf(F32(math.sqrt(float(np.sum(np.asarray(norms, dtype=F64) ** 2)))))
    t1 = _rbf(F32(total) + F32(1e-6))
    r = _rbf(F32(1.0) / t1)
    coef = _rbf(r * F32(max_norm))
    coef = F32(1.0) if coef > 1.0 else F32(coef)            # clamp(max=1.0); NaN stays NaN
    return F32(total), coef


def _fma64(a, x, c):
    """fl64(a * x + c), one rounding (the device code contracts `a * x + c` into v_fma_f64, as torch's build of the same expression
    does: visible when the two terms cancel).  Exact rational arithmetic per element -- small arrays only."""
    from fractions import Fraction
    a = Fraction(float(a))
    x = np.asarray(x, dtype=F64)
    c = np.asarray(c, dtype=F64)
    out = np.empty(x.shape, dtype=F64)
    fin = np.isfinite(x) & np.isfinite(c)
    out[~fin] = float(a) * x[~fin] + c[~fin]
    flat_x, flat_c, flat_o, idx = x.reshape(-1), c.reshape(-1), out.reshape(-1), np.nonzero(fin.reshape(-1))[0]
    for i in idx:
        flat_o[i] = float(a * Fraction(float(flat_x[i])) + Fraction(float(flat_c[i])))
    return out


def adamw_fused_step(param_bits, grad_bits, exp_avg_bits, exp_avg_sq_bits, step, lr, beta1, beta2, eps, weight_decay, coef=None):
    """One AdamW update of one bf16 tensor, `step` = the count AFTER this update.  coef: the clip coefficient (the gradient is first
    multiplied by it and rounded to bf16, as the clip's in-place _foreach_mul_ does), None: no clipping.  Returns new (param, exp_avg,
    exp_avg_sq) bit patterns."""
    p = bf16_to_f32(param_bits).astype(F32)
    g = bf16_to_f32(grad_bits).astype(F32)
    m = bf16_to_f32(exp_avg_bits).astype(F32)
    v = bf16_to_f32(exp_avg_sq_bits).astype(F32)
    if coef is not None:
        g = _rbf(g * F32(coef))
    with np.errstate(over="ignore", invalid="ignore"):
        if weight_decay != 0:
            p = _fma64(-(lr * weight_decay), p.astype(F64), p.astype(F64)).astype(F32)          # param -= lr * weight_decay * param
        m = _fma64(beta1, m.astype(F64), (1 - beta1) * g.astype(F64)).astype(F32)              # beta1 * exp_avg + (1 - beta1) * grad
        v = _fma64(beta2, v.astype(F64), ((1 - beta2) * g.astype(F64)) * g.astype(F64)).astype(F32)
    bc1 = F32(1 - math.pow(beta1, step))
    bc2_sqrt = F32(math.sqrt(1 - math.pow(beta2, step)))
    step_size = F32(lr / float(bc1))
    with np.errstate(divide="ignore", invalid="ignore"):
        denom = ((np.sqrt(v) / bc2_sqrt).astype(F64) + eps).astype(F32)
        p = p - (step_size * m) / denom
    return f32_to_bf16(p), f32_to_bf16(m), f32_to_bf16(v)


def adamw_step_f64(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay, max_norm=None, all_grads=None):
    """The same update in plain float64 (no roundings): what the CPU pin compares with torch on float64 tensors."""
    if max_norm is not None:
        total = math.sqrt(sum(float(np.sum(np.asarray(x, dtype=F64) ** 2)) for x in all_grads))
        g = g * min(max_norm / (total + 1e-6), 1.0)
    if weight_decay != 0:
        p = p - lr * weight_decay * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    p = p - (lr / bc1) * m / (np.sqrt(v) / math.sqrt(bc2) + eps)
    return p, m, v
