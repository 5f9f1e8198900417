"""CPU checks of the step-end path (clip_grad_norm_ + AdamW): the oracle's restatement pinned against torch itself, the host plan of
the C ABI, and the host logic of quantized_training.optim (which keeps torch's own calls for everything the kernels do not cover).

Reference call site: examples/text_classification/run_glue_no_trainer.py:469-474, 655-668 (torch.optim.AdamW over two parameter groups,
accelerator.clip_grad_norm_(model.parameters(), 1.0), optimizer.step()).  The arithmetic is torch's (third party, pinned 2.10.0).
"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))

from oracle import optimizer_oracle as oo  # noqa: E402
from oracle.qt_oracle import bf16_to_f32, f32_to_bf16  # noqa: E402


def _bits(t):
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def test_oracle_f64_update_is_torchs_adamw_with_clip():
    """The algorithm, free of rounding questions: float64 tensors through torch's CPU AdamW + clip_grad_norm_ for three steps."""
    torch.manual_seed(0)
    ps = [torch.randn(37, 5, dtype=torch.float64), torch.randn(11, dtype=torch.float64)]
    params = [torch.nn.Parameter(p.clone()) for p in ps]
    wds = (0.01, 0.0)
    opt = torch.optim.AdamW([{"params": [params[0]], "weight_decay": wds[0]}, {"params": [params[1]], "weight_decay": wds[1]}], lr=1e-3)
    m = [np.zeros(p.shape) for p in ps]
    v = [np.zeros(p.shape) for p in ps]
    pp = [p.numpy().copy() for p in ps]
    for step in range(1, 4):
        gs = [torch.randn_like(p) * (3.0 if step != 2 else 1e-3) for p in ps]       # step 2: below the threshold, coefficient clamps to 1
        for q, g in zip(params, gs):
            q.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        for i in range(2):
            pp[i], m[i], v[i] = oo.adamw_step_f64(pp[i], gs[i].numpy(), m[i], v[i], step, 1e-3, 0.9, 0.999, 1e-8, wds[i], 1.0, [g.numpy() for g in gs])
            assert np.abs(pp[i] - params[i].detach().numpy()).max() < 1e-12
            assert np.abs(m[i] - opt.state[params[i]]["exp_avg"].numpy()).max() < 1e-12
            assert np.abs(v[i] - opt.state[params[i]]["exp_avg_sq"].numpy()).max() < 1e-12


@pytest.mark.parametrize("scale", [5.0, 1e-3, 300.0])
def test_oracle_clip_coefficient_is_torchs_on_bf16_gradients(scale):
    """clip_grad_norm_ on bf16 tensors: every intermediate is a bf16 tensor.  torch's CPU kernels and the restatement agree on the total
    norm and on every clipped gradient element."""
    torch.manual_seed(1)
    grads = [(torch.randn(n) * scale).bfloat16() for n in (4096, 33, 700, 1)]
    params = [torch.nn.Parameter(torch.zeros_like(g)) for g in grads]
    for p, g in zip(params, grads):
        p.grad = g.clone()
    total = torch.nn.utils.clip_grad_norm_(params, 1.0)
    t_or, coef = oo.clip_coefficient_bf16([_bits(g) for g in grads], 1.0)
    assert float(total.float()) == float(t_or)
    for p, g in zip(params, grads):
        want = f32_to_bf16(bf16_to_f32(_bits(g)) * np.float32(coef))
        assert np.array_equal(want, _bits(p.grad))


def test_oracle_bf16_update_stays_within_half_a_step_of_the_f64_update():
    """The mixed fp32 / fp64 restatement of torch's fused kernel against the plain float64 formula: each stored value is the bf16
    rounding of something within fp32 noise of the exact one."""
    rng = np.random.default_rng(0)
    n = 20000
    p = f32_to_bf16(rng.standard_normal(n).astype(np.float32))
    g = f32_to_bf16((rng.standard_normal(n) * 0.01).astype(np.float32))
    m = f32_to_bf16((rng.standard_normal(n) * 0.003).astype(np.float32))
    v = f32_to_bf16((rng.random(n) * 1e-4).astype(np.float32))
    args = (7, 2e-5, 0.9, 0.999, 1e-8, 0.01)
    pn, mn, vn = oo.adamw_fused_step(p, g, m, v, *args)
    pe, me, ve = oo.adamw_step_f64(*(bf16_to_f32(x).astype(np.float64) for x in (p, g, m, v)), *args)
    terms = sum(np.abs(bf16_to_f32(x).astype(np.float64)) for x in (p, g, m))       # where two terms cancel, float64 noise of the terms is all that is left
    for got, exact in ((pn, pe), (mn, me), (vn, ve)):
        gotf = bf16_to_f32(got).astype(np.float64)
        half_step = np.maximum(np.abs(exact), 1e-30) * 2.0 ** -8        # >= half a bf16 step at |exact|
        assert np.all(np.abs(gotf - exact) <= 1.0001 * half_step + 1e-15 * terms)


def test_plan_counts_chunks_per_tensor():
    from quantized_training import _native
    L = _native.lib()
    numels = [0, 1, 8192, 8193, 50265 * 768, 768]
    arr = (_native.QtAdamwTensor * len(numels))()
    for i, n in enumerate(numels):
        arr[i].numel = n
    total = L.qt_clip_adamw_plan(arr, len(numels), None, 0)
    want = [(n + 8191) // 8192 for n in numels]
    assert total == sum(want)
    cmap = (ctypes.c_int32 * total)()
    assert L.qt_clip_adamw_plan(arr, len(numels), cmap, total) == total
    firsts = [arr[i].first_chunk for i in range(len(numels))]
    assert firsts == [sum(want[:i]) for i in range(len(numels))]
    got = np.frombuffer(cmap, dtype=np.int32)
    assert np.array_equal(got, np.repeat(np.arange(len(numels)), want))
    assert L.qt_clip_adamw_ws_bytes(len(numels), total) >= 4 * total + 8 * len(numels) + 16
    arr[2].numel = -1
    assert L.qt_clip_adamw_plan(arr, len(numels), None, 0) == _native.QT_ERR_BAD_ARG


def test_clip_and_step_keeps_torchs_calls_on_the_host_and_says_so():
    """CPU tensors: the package does not touch them -- torch's own clip_grad_norm_ and optimizer.step() run, results identical to calling
    them directly, and the route table names the reason."""
    from quantized_training import optim
    torch.manual_seed(0)

    def make():
        torch.manual_seed(3)
        lin = torch.nn.Linear(16, 4)
        opt = torch.optim.AdamW(lin.parameters(), lr=1e-2)
        return lin, opt
    a, oa = make()
    b, ob = make()
    x = torch.randn(8, 16)
    for _ in range(2):
        for lin in (a, b):
            lin.zero_grad()
            (lin(x) ** 2).sum().backward()
        total = optim.clip_and_step(a.parameters(), oa, 1.0)
        want = torch.nn.utils.clip_grad_norm_(b.parameters(), 1.0)
        ob.step()
        assert torch.equal(total, want)
    assert torch.equal(a.weight, b.weight) and torch.equal(a.bias, b.bias)
    assert optim.ROUTES["train:clip + optimizer"].startswith("torch (")
    sgd = torch.optim.SGD(a.parameters(), lr=0.1)
    optim.clip_and_step(a.parameters(), sgd, None)
    assert "not torch.optim.AdamW" in optim.ROUTES["train:clip + optimizer"]
