"""In-kernel clocks of one k step of qt_linear_fqt_bf16 (QT_FQT_ABLATE=9): where does a step's time go?"""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
os.environ["QT_FQT_STAMPS"] = hex(buf.data_ptr())
os.environ["QT_FQT_ABLATE"] = "9"
import exp_linear_fqt as E
f = E.Fmt("posit8_2")
M, N, K = 1024, 13824, 5120
x = f.fq(torch.randn(M, K, device="cuda").bfloat16())
W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
for _ in range(3):
    E.linear_fqt(x, [W], f)
torch.cuda.synchronize()
b = buf.cpu().tolist()
names = {29: "prev end", 31: "after top wait", 0: "after barrier", 1: "after requests", 2: "after frag reads issued", 3: "after first rows", 30: "step end"}
for wv in (0, 1):
    s = b[wv * 32: wv * 32 + 32]
    t0 = s[29]
    print("wave", wv * 4)
    order = [29, 31, 0, 1, 2, 3] + list(range(4, 4 + 12)) + [30]
    prev = t0
    for k in order:
        if s[k] == 0: continue
        nm = names.get(k) or ("group %d %s" % ((k - 4) // 3, ("after wait", "after units", "after mfma issue")[(k - 4) % 3]))
        print("  %-28s +%6d  (delta %5d)" % (nm, s[k] - t0, s[k] - prev)); prev = s[k]
