import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_optimizer as T
from oracle import optimizer_oracle as oo
from oracle.qt_oracle import bf16_to_f32
from quantized_training import optim
shapes = [(257, 9), (33,), (4099,)]
params = T._params(5, shapes, unaligned=True)
opt = T._groups(params, capturable=True)
mv = [(np.zeros(int(np.prod(s)), np.uint16), np.zeros(int(np.prod(s)), np.uint16)) for s in shapes]
pbits = [T._bits(p).ravel() for p in params]
for step in range(1, 4):
    gs = T._grads(step, shapes, 0.3)
    for p, g in zip(params, gs):
        p.grad = g.clone()
    total = optim.clip_and_step(params, opt, 1.0)
    t_or, coef = oo.clip_coefficient_bf16([T._bits(g) for g in gs], 1.0)
    print("step", step, float(total), float(t_or), float(coef))
    for i, (p, g) in enumerate(zip(params, gs)):
        wd = 0.01 if p.dim() > 1 else 0.0
        gb = T._bits(g).ravel()
        pbits[i], m, v = oo.adamw_fused_step(pbits[i], gb, mv[i][0], mv[i][1], step, 2e-3, 0.9, 0.999, 1e-8, wd, coef)
        m0, v0 = mv[i]
        mv[i] = (m, v)
        for name, want, got in (("p", pbits[i], T._bits(p).ravel()), ("m", m, T._bits(opt.state[p]["exp_avg"]).ravel()), ("v", v, T._bits(opt.state[p]["exp_avg_sq"]).ravel())):
            bad = np.nonzero(want != got)[0]
            if len(bad):
                j = bad[0]
                print(" tensor", i, name, "mismatches", len(bad), "of", len(want), "first", j, hex(want[j]), hex(got[j]), "g", hex(gb[j]), float(bf16_to_f32(gb[j:j+1])[0]),
                      "m0", hex(m0[j]), "v0", hex(v0[j]))
        # continue from the device's state so that one mismatch does not cascade
        pbits[i] = T._bits(p).ravel(); mv[i] = (T._bits(opt.state[p]["exp_avg"]).ravel(), T._bits(opt.state[p]["exp_avg_sq"]).ravel())
