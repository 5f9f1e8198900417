"""GPU parity of the step end (clip_grad_norm_ + AdamW in four launches, csrc/qt_optimizer.hip) through the C ABI and through
quantized_training.optim.clip_and_step.

Reference call site: examples/text_classification/run_glue_no_trainer.py:469-474, 655-668.  The arithmetic is torch's (third party,
pinned 2.10.0): the live oracle here is torch itself on the same device -- torch.nn.utils.clip_grad_norm_ followed by
torch.optim.AdamW(fused=True).step() -- bit for bit, plus the numpy restatement (oracle/optimizer_oracle.py) at sizes it finishes in
seconds.
"""
import os

import numpy as np
import pytest
import torch

from oracle import optimizer_oracle as oo
from quantized_training import optim

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

# RoBERTa-like mix: a big embedding, square and tall matrices, vectors, a 2-element bias, ragged ends, an empty tensor
SHAPES = [(5000, 768), (768, 768), (3072, 768), (768,), (2,), (8192 * 3 + 5,), (1, 13), (0,), (8193,)]


def _bits(t):
    return t.detach().contiguous().view(torch.int16).cpu().numpy().view(np.uint16).copy()


def _params(seed, shapes=SHAPES, unaligned=False):
    g = torch.Generator().manual_seed(seed)
    out = []
    for s in shapes:
        n = int(np.prod(s))
        if unaligned:
            base = torch.zeros(n + 8, dtype=torch.bfloat16, device=DEV)
            t = base[1:1 + n].view(s)                    # storage offset of one element: 2-byte aligned only
            t.copy_((torch.randn(s, generator=g) * 0.05).bfloat16())
            out.append(torch.nn.Parameter(t))
        else:
            out.append(torch.nn.Parameter((torch.randn(s, generator=g) * 0.05).bfloat16().to(DEV)))
    return out


def _grads(seed, shapes, scale):
    g = torch.Generator().manual_seed(1000 + seed)
    return [(torch.randn(s, generator=g) * scale).bfloat16().to(DEV) for s in shapes]


def _groups(params, **kw):
    decay = [p for p in params if p.dim() > 1]
    rest = [p for p in params if p.dim() <= 1]
    return torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": rest, "weight_decay": 0.0}], lr=2e-3, **kw)


def _assert_same_state(pa, oa, pb, ob):
    for a, b in zip(pa, pb):
        assert torch.equal(a, b), "parameters differ"
        if a in oa.state:
            sa, sb = oa.state[a], ob.state[b]
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
            assert float(sa["step"]) == float(sb["step"])


@pytest.mark.parametrize("scale", [0.5, 1e-4], ids=["clipped", "below-threshold"])
@pytest.mark.parametrize("capturable", [True, False])
def test_clip_and_step_is_torchs_fused_adamw_behind_torchs_clip(scale, capturable):
    """Four steps on a mix of shapes, two parameter groups.  Same total norm => every parameter, exp_avg, exp_avg_sq bit and every step
    count equal torch's.  (The total norm is a bf16 rounding of a sum whose last fp32 bits depend on the order of addition; if that
    rounding ever lands on the other side the step is compared against the restatement instead -- it never did in these seeds.)"""
    pa, pb = _params(0), _params(0)
    oa = _groups(pa, fused=True, capturable=capturable)
    ob = _groups(pb, fused=True, capturable=capturable)
    for step in range(4):
        gs = _grads(step, SHAPES, scale)
        for a, b, g in zip(pa, pb, gs):
            a.grad, b.grad = g.clone(), g.clone()
        got = optim.clip_and_step(pa, oa, 1.0)
        want = torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ob.step()
        assert optim.ROUTES["train:clip + optimizer"].startswith("in_tree_clip_adamw")
        assert got.dtype == want.dtype and float(got) == float(want), (float(got), float(want))
        _assert_same_state(pa, oa, pb, ob)
    assert (scale > 1e-2) == (float(want) > 1.0)


def test_step_without_clipping_and_host_step_counts():
    """max_norm None: no norm launches; a default-built AdamW (host step counts, torch would take its foreach path) advances its counts
    on the host and lands on the fused arithmetic."""
    pa, pb = _params(2), _params(2)
    oa, ob = _groups(pa), _groups(pb, fused=True)
    for step in range(3):
        gs = _grads(step, SHAPES, 0.01)
        for a, b, g in zip(pa, pb, gs):
            a.grad, b.grad = g.clone(), g.clone()
        assert optim.clip_and_step(pa, oa, None) is None
        ob.step()
        _assert_same_state(pa, oa, pb, ob)
    assert oa.state[pa[0]]["step"].device.type == "cpu" and float(oa.state[pa[0]]["step"]) == 3.0


def test_clip_and_step_against_the_restatement_and_on_unaligned_storage():
    """Small tensors, storage offset of one element (the kernels' element-wise path), against oracle/optimizer_oracle.py bit for bit."""
    shapes = [(257, 9), (33,), (4099,)]
    params = _params(5, shapes, unaligned=True)
    assert all(p.data_ptr() % 16 for p in params)
    opt = _groups(params, capturable=True)
    mv = [(np.zeros(int(np.prod(s)), np.uint16), np.zeros(int(np.prod(s)), np.uint16)) for s in shapes]
    pbits = [_bits(p).ravel() for p in params]
    for step in range(1, 4):
        gs = _grads(step, shapes, 0.3)
        for p, g in zip(params, gs):
            p.grad = g.clone()
        total = optim.clip_and_step(params, opt, 1.0)
        t_or, coef = oo.clip_coefficient_bf16([_bits(g) for g in gs], 1.0)
        assert float(total) == float(t_or)
        for i, (p, g) in enumerate(zip(params, gs)):
            wd = 0.01 if p.dim() > 1 else 0.0
            pbits[i], m, v = oo.adamw_fused_step(pbits[i], _bits(g).ravel(), mv[i][0], mv[i][1], step, 2e-3, 0.9, 0.999, 1e-8, wd, coef)
            mv[i] = (m, v)
            assert np.array_equal(pbits[i], _bits(p).ravel())
            assert np.array_equal(m, _bits(opt.state[p]["exp_avg"]).ravel()) and np.array_equal(v, _bits(opt.state[p]["exp_avg_sq"]).ravel())


def test_non_finite_norm_raises_before_anything_changes():
    params = _params(7, [(64, 64), (64,)])
    opt = _groups(params)
    for p in params:
        p.grad = torch.full_like(p, 0.01)
    optim.clip_and_step(params, opt, 1.0, error_if_nonfinite=True)
    before = [p.detach().clone() for p in params]
    m_before = [opt.state[p]["exp_avg"].clone() for p in params]
    params[0].grad[3, 3] = float("nan")
    with pytest.raises(RuntimeError, match="non-finite"):
        optim.clip_and_step(params, opt, 1.0, error_if_nonfinite=True)
    for p, b, m in zip(params, before, m_before):
        assert torch.equal(p, b) and torch.equal(opt.state[p]["exp_avg"], m)
        assert float(opt.state[p]["step"]) == 1.0
    # without the check torch scales by the NaN coefficient: everything the step touches becomes NaN, as with torch's own calls
    optim.clip_and_step(params, opt, 1.0, error_if_nonfinite=False)
    assert bool(torch.isnan(params[0]).all()) and float(opt.state[params[0]]["step"]) == 2.0


def test_state_dict_moves_between_the_kernels_and_torch():
    """Two steps here, state_dict into a fresh torch optimizer, two more steps with torch's own calls == four steps of torch."""
    pa, pb = _params(9), _params(9)
    oa, ob = _groups(pa, fused=True), _groups(pb, fused=True)
    for step in range(4):
        gs = _grads(step, SHAPES, 0.2)
        for a, b, g in zip(pa, pb, gs):
            a.grad, b.grad = g.clone(), g.clone()
        if step == 2:
            fresh = _groups(pa, fused=True)
            fresh.load_state_dict(oa.state_dict())
            oa = fresh
        if step < 2:
            optim.clip_and_step(pa, oa, 1.0)
        else:
            torch.nn.utils.clip_grad_norm_(pa, 1.0)
            oa.step()
        torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ob.step()
    _assert_same_state(pa, oa, pb, ob)


def test_captured_step_replays_what_eager_steps_compute():
    """A stream capture of clip_and_step (capturable optimizer, gradients written into fixed buffers): three replays equal three eager
    calls, step counts included; a learning rate held in a tensor is read at every replay."""
    shapes = [(300, 768), (768,), (9000,)]
    pa, pb = _params(11, shapes), _params(11, shapes)
    lr_a, lr_b = torch.tensor(2e-3, device=DEV), torch.tensor(2e-3, device=DEV)
    oa = torch.optim.AdamW(pa, lr=lr_a, weight_decay=0.01, capturable=True)
    ob = torch.optim.AdamW(pb, lr=lr_b, weight_decay=0.01, fused=True, capturable=True)
    for p in pa:
        p.grad = torch.zeros_like(p)
    warm = _grads(0, shapes, 0.4)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for p, g in zip(pa, warm):
            p.grad.copy_(g)
        optim.clip_and_step(pa, oa, 1.0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        optim.clip_and_step(pa, oa, 1.0)
    for b, g in zip(pb, warm):
        b.grad = g.clone()
    for _ in range(2):                                  # the eager twin takes the warm-up step and the captured one (a capture executes nothing)
        torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ob.step()
        for b, g in zip(pb, warm):
            b.grad = g.clone()
    # the capture itself did not run: redo it as a replay on the warm-up gradients
    graph.replay()
    for step in range(1, 4):
        gs = _grads(step, shapes, 0.4)
        lr_a.fill_(2e-3 / step)
        lr_b.fill_(2e-3 / step)
        for a, b, g in zip(pa, pb, gs):
            a.grad.copy_(g)
            b.grad = g.clone()
        graph.replay()
        torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ob.step()
    torch.cuda.synchronize()
    _assert_same_state(pa, oa, pb, ob)
    assert float(oa.state[pa[0]]["step"]) == 5.0


def test_uncovered_optimizers_keep_torchs_launches_and_the_debug_bit_switches_the_kernels_off():
    params = [torch.nn.Parameter(torch.randn(64, 64, device=DEV))]          # fp32 parameters
    opt = torch.optim.AdamW(params, lr=1e-3)
    params[0].grad = torch.randn_like(params[0])
    optim.clip_and_step(params, opt, 1.0)
    assert "not bf16" in optim.ROUTES["train:clip + optimizer"]
    pa, pb = _params(13, [(128, 64)]), _params(13, [(128, 64)])
    oa, ob = torch.optim.AdamW(pa, lr=1e-3, amsgrad=True), torch.optim.AdamW(pb, lr=1e-3, amsgrad=True)
    g = _grads(0, [(128, 64)], 0.1)[0]
    pa[0].grad, pb[0].grad = g.clone(), g.clone()
    optim.clip_and_step(pa, oa, 1.0)
    torch.nn.utils.clip_grad_norm_(pb, 1.0)
    ob.step()
    assert "amsgrad" in optim.ROUTES["train:clip + optimizer"] and torch.equal(pa[0], pb[0])
    os.environ["QT_TRAIN_DEBUG"] = "256"
    try:
        pc = _params(14, [(128, 64)])
        oc = torch.optim.AdamW(pc, lr=1e-3)
        pc[0].grad = g.clone()
        optim.clip_and_step(pc, oc, 1.0)
        assert "QT_TRAIN_DEBUG" in optim.ROUTES["train:clip + optimizer"]
    finally:
        del os.environ["QT_TRAIN_DEBUG"]


def test_full_size_roberta_parameter_set_matches_torch_and_takes_four_launches():
    """The configs[4] parameter set (RoBERTa-base: 124.6 M elements in ~200 tensors): one clipped step equals torch's; the route says
    four launches."""
    from transformers import RobertaConfig, RobertaForSequenceClassification
    torch.manual_seed(0)
    model = RobertaForSequenceClassification(RobertaConfig(num_labels=2)).bfloat16()
    shapes = [tuple(p.shape) for p in model.parameters()]
    del model
    pa, pb = _params(21, shapes), _params(21, shapes)
    oa, ob = _groups(pa, fused=True, capturable=True), _groups(pb, fused=True, capturable=True)
    for step in range(2):
        gs = _grads(step, shapes, 1e-3)
        for a, b, g in zip(pa, pb, gs):
            a.grad, b.grad = g, g.clone()
        got = optim.clip_and_step(pa, oa, 1.0)
        want = torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ob.step()
        assert float(got) == float(want) and float(want) > 1.0
    _assert_same_state(pa, oa, pb, ob)
    assert "4 launches" in optim.ROUTES["train:clip + optimizer"]
