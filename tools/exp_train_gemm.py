"""GPU experiment: qt_train_gemm_bf16 (the in-tree bf16 GEMMs of a training step's Linear layers) against fp64 products and torch's GEMMs.

    python tools/exp_train_gemm.py [--iters 50]

Per layout (forward / dgrad / wgrad) and shape of a RoBERTa-base layer at [16, 128]: the error against the fp64 product of the same bf16
operands, run-to-run bit-identity, and microseconds per call (operands rotating over a pool) of the in-tree kernel -- single problem and
the three-problem query / key / value launch -- and of torch.matmul / F.linear (hipBLASLt)."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
sys.path.insert(0, ROOT)
from quantized_training import _native  # noqa: E402

L = _native.lib()
DEV = torch.device("cuda:0")


class Problem(ctypes.Structure):
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("c", ctypes.c_void_p)]


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)


TILES = os.environ.get("TILES", "").split(",") if os.environ.get("TILES") else []


def gemm(As, Bs, ta, tb, biases=None):
    """C_i = op(A_i) op(B_i): A [M][K] (ta 0) or [K][M]; B [N][K] (tb 0) or [K][N]."""
    n = len(As)
    A, B = As[0], Bs[0]
    M, K = (A.shape[1], A.shape[0]) if ta else (A.shape[0], A.shape[1])
    N = B.shape[1] if tb else B.shape[0]
    Cs = [torch.empty((M, N), dtype=torch.bfloat16, device=DEV) for _ in range(n)]
    arr = (Problem * n)()
    for i in range(n):
        arr[i].a, arr[i].b, arr[i].c = As[i].data_ptr(), Bs[i].data_ptr(), Cs[i].data_ptr()
        arr[i].bias = biases[i].data_ptr() if biases and biases[i] is not None else None
    _native.check(L.qt_train_gemm_bf16(arr, n, ta, tb, M, N, K, A.stride(0), B.stride(0), N, stream()), "qt_train_gemm_bf16")
    return Cs


def ref64(A, B, ta, tb, bias=None):
    a = (A.t() if ta else A).double()
    b = (B if tb else B.t()).double()
    r = a @ b
    return r + bias.double() if bias is not None else r, a.abs() @ b.abs()


def torch_gemm(A, B, ta, tb, bias=None):
    if not ta and not tb:
        return torch.nn.functional.linear(A, B, bias)
    if not ta and tb:
        return A.mm(B)
    return A.t().mm(B)


def timeit(fn, iters):
    """Microseconds per call inside a replayed hipGraph of `iters` calls (no host time between launches: what the training step sees)."""
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            keep = [fn(i) for i in range(iters)]
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(5):
        s.record()
        graph.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / iters)
    del keep
    return best


def operands(M, N, K, ta, tb, pool):
    As = [(torch.randn((K, M) if ta else (M, K), device=DEV) * 0.5).bfloat16() for _ in range(pool)]
    Bs = [(torch.randn((K, N) if tb else (N, K), device=DEV) * 0.05).bfloat16() for _ in range(pool)]
    return As, Bs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--skip-checks", action="store_true")
    args = ap.parse_args()
    ok = True
    if not args.skip_checks:
        torch.manual_seed(0)
        for (ta, tb, name) in ((0, 0, "forward"), (0, 1, "dgrad"), (1, 1, "wgrad"), (1, 0, "tn")):
            for (M, N, K) in ((2048, 768, 768), (2048, 3072, 768), (2048, 768, 3072), (768, 768, 2048), (3072, 768, 2048), (200, 72, 256), (8, 8, 256),
                              (136, 264, 320)):
                for tile in ("", "128x128", "128x64", "64x128", "64x64"):
                    if tile:
                        os.environ["QT_TRAIN_GEMM_TILE"] = tile
                    else:
                        os.environ.pop("QT_TRAIN_GEMM_TILE", None)
                    As, Bs = operands(M, N, K, ta, tb, 2)
                    bias = torch.randn(N, device=DEV).bfloat16()
                    c = gemm(As, Bs, ta, tb, [bias, None])
                    c2 = gemm(As, Bs, ta, tb, [bias, None])
                    for i in range(2):
                        ref, bound = ref64(As[i], Bs[i], ta, tb, bias if i == 0 else None)
                        err = (c[i].double() - ref).abs()
                        tol = ref.abs() * 2.0 ** -8 + bound * 2.0 ** -18 + 1e-30
                        good = bool((err <= tol).all()) and torch.equal(c[i].view(torch.int16), c2[i].view(torch.int16))
                        if not good:
                            print(f"FAIL {name} {M}x{N}x{K} tile {tile or 'auto'} problem {i}: worst err/tol {float((err / tol).max()):.3f}")
                        ok &= good
            print(f"checked {name}: {'ok' if ok else 'FAILED'}", flush=True)
        os.environ.pop("QT_TRAIN_GEMM_TILE", None)
    shapes = [("q/k/v/o forward", 0, 0, 2048, 768, 768), ("intermediate forward", 0, 0, 2048, 3072, 768), ("output forward", 0, 0, 2048, 768, 3072),
              ("q/k/v/o dgrad", 0, 1, 2048, 768, 768), ("intermediate dgrad", 0, 1, 2048, 768, 3072), ("output dgrad", 0, 1, 2048, 3072, 768),
              ("q/k/v/o wgrad", 1, 1, 768, 768, 2048), ("intermediate wgrad", 1, 1, 3072, 768, 2048), ("output wgrad", 1, 1, 768, 3072, 2048)]
    total_lib = total_ours = 0.0
    for name, ta, tb, M, N, K in shapes:
        pool = 12
        As, Bs = operands(M, N, K, ta, tb, pool)
        t_lib = timeit(lambda i: torch_gemm(As[i % pool], Bs[i % pool], ta, tb), args.iters)
        t_one = timeit(lambda i: gemm([As[i % pool]], [Bs[i % pool]], ta, tb), args.iters)
        line = f"{name:22s} {M}x{N}x{K}: torch {t_lib:6.2f} us   in-tree {t_one:6.2f} us"
        for tile in TILES:
            os.environ["QT_TRAIN_GEMM_TILE"] = tile
            line += f"  [{tile}: {timeit(lambda i: gemm([As[i % pool]], [Bs[i % pool]], ta, tb), args.iters):6.2f}]"
            os.environ.pop("QT_TRAIN_GEMM_TILE")
        if "q/k/v" in name:
            t3 = timeit(lambda i: gemm([As[i % pool], As[(i + 1) % pool], As[(i + 2) % pool]] if ta or tb else [As[i % pool]] * 3,
                                       [Bs[i % pool], Bs[(i + 1) % pool], Bs[(i + 2) % pool]], ta, tb), args.iters)
            line += f"   three problems in one launch {t3:6.2f} us (torch: 3 x {t_lib:.2f})"
            total_lib += 4 * t_lib
            total_ours += t3 + t_one
        else:
            total_lib += t_lib
            total_ours += t_one
        print(line, flush=True)
    print(f"one encoder layer, 18 products: torch {total_lib:.1f} us, in-tree {total_ours:.1f} us (12 launches)")
    print("ALL CHECKS", "PASSED" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
