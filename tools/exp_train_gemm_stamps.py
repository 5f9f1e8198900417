"""GPU experiment (tuning build): the timeline of one qt_train_gemm_bf16 launch, per workgroup.

    QT_HIP_LIB=tools/build/libqt_hip_tuning.so python tools/exp_train_gemm_stamps.py

Wave 0 of every workgroup stamps s_memrealtime (100 MHz, chip-wide) when it starts requesting operands, when its first k tile has
landed, when its k loop is done and when its results are stored (QT_TG_STAMPS).  The launch is replayed from a graph behind another
kernel (cold caches, as inside a step).  Printed per shape: the launch's span, and over workgroups in start order the quartiles of
(start - launch start), (first k tile - start), (k loop), (stores)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("QT_HIP_LIB", os.path.join(ROOT, "tools", "build", "libqt_hip_tuning.so"))
import torch  # noqa: E402
from quantized_training import _native  # noqa: E402

L = _native.lib()
DEV = torch.device("cuda:0")


def st():
    return ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)


def run(M, N, K, ta, tb, count=1):
    torch.manual_seed(0)
    As = [(torch.randn((K, M) if ta else (M, K), device=DEV) * 0.5).bfloat16() for _ in range(count)]
    Bs = [(torch.randn((K, N) if tb else (N, K), device=DEV) * 0.05).bfloat16() for _ in range(count)]
    Cs = [torch.empty((M, N), dtype=torch.bfloat16, device=DEV) for _ in range(count)]
    stamps = torch.zeros(4 * 8192, dtype=torch.int64, device=DEV)
    os.environ["QT_TG_STAMPS"] = hex(stamps.data_ptr())
    arr = (_native.QtGemmProblem * count)()
    for i in range(count):
        arr[i].a, arr[i].b, arr[i].bias, arr[i].c = As[i].data_ptr(), Bs[i].data_ptr(), None, Cs[i].data_ptr()
    junk = torch.randn(64 << 20, device=DEV)               # 256 MB: what runs in between evicts the operands from the caches

    def call():
        junk.mul_(1.0001)
        _native.check(L.qt_train_gemm_bf16(arr, count, ta, tb, M, N, K, As[0].stride(0), Bs[0].stride(0), N, st()), "qt_train_gemm_bf16")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        call()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            call()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
    del os.environ["QT_TG_STAMPS"]
    t = stamps.cpu().view(-1, 4).double() / 100.0
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    order = t[:, 0].argsort()
    t = t[order]
    q = lambda v: "%5.2f %5.2f %5.2f %5.2f" % tuple(float(x) for x in torch.quantile(v, torch.tensor([0.0, 0.5, 0.9, 1.0], dtype=torch.float64)))  # noqa: E731
    print(f"{M}x{N}x{K} ta={ta} tb={tb} x{count}: {t.shape[0]} workgroups, span {float(t[:, 3].max() - t0):6.2f} us")
    print("   start after launch (min med p90 max) ", q(t[:, 0] - t0))
    print("   first k tile lands after start        ", q(t[:, 1] - t[:, 0]))
    print("   k loop                                ", q(t[:, 2] - t[:, 1]))
    print("   result stores                         ", q(t[:, 3] - t[:, 2]))
    n = t.shape[0]
    for lo in range(0, n, max(n // 6, 1)):
        seg = t[lo:lo + max(n // 6, 1)]
        print(f"   workgroups {lo:4d}..: start {float((seg[:, 0] - t0).median()):6.2f}  end {float((seg[:, 3] - t0).median()):6.2f}")


SHAPES = ((2048, 3072, 768, 0, 0, 1), (2048, 768, 768, 0, 0, 3), (2048, 768, 768, 0, 0, 1), (2048, 768, 3072, 0, 0, 1), (3072, 768, 2048, 1, 1, 1))
if os.environ.get("SHAPES"):                    # e.g. SHAPES=2048x2048x2048 (forward layout, one problem)
    SHAPES = tuple(tuple(int(v) for v in sh.split("x")) + (0, 0, 1) for sh in os.environ["SHAPES"].split(","))
for shape in SHAPES:
    run(*shape)
