"""In-kernel clocks of four consecutive k steps of qt_linear_fqt_bf16 (QT_FQT_ABLATE=8: four stamps per step and wave; =9: every
phase, waves 0 and 4): where does a step's time go?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
buf = torch.zeros(8 * 4 * 32, dtype=torch.int64, device="cuda")
os.environ["QT_FQT_STAMPS"] = hex(buf.data_ptr())
mode = os.environ.setdefault("QT_FQT_ABLATE", "8")
import exp_linear_fqt as E
f = E.Fmt("posit8_2")
M, N, K = 1024, 13824, 5120
x = f.fq(torch.randn(M, K, device="cuda").bfloat16())
W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
for _ in range(3):
    E.linear_fqt(x, [W], f)
torch.cuda.synchronize()
b = buf.cpu().view(8, 4, 32)
t0 = int(b[0, 0, 29])
print("wave: per step [top-wait, barrier-wait, work] cycles; step start relative to wave 0 step 40")
for w in range(8):
    row = []
    for s in range(4):
        st = b[w, s]
        row.append("start %6d: wait %4d  barrier %4d  work %5d" % (int(st[29]) - t0, int(st[31] - st[29]), int(st[0] - st[31]), int(st[30] - st[0])))
    print("wave", w, " | ".join(row))
if mode == "9":
    names = {29: "prev end", 31: "after top wait", 0: "after barrier", 1: "after requests", 2: "after frag reads issued", 30: "step end"}
    for w in (0, 4):
        st = b[w, 0].tolist()
        prev = st[29]
        print("wave", w)
        for k in [29, 31, 0, 1, 2] + list(range(4, 16)) + [30]:
            if st[k] == 0: continue
            nm = names.get(k) or ("group %d %s" % ((k - 4) // 3, ("after wait", "after mfma issue", "after units+requests")[(k - 4) % 3]))
            print("  %-30s +%6d  (delta %5d)" % (nm, st[k] - st[29], st[k] - prev)); prev = st[k]
