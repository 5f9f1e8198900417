"""Tools only: which of hipBLASLt's suggestions is fastest for the FP8 GEMM shapes the package sends to the library
(qt_fp8_gemm_tune).  Prints microseconds per suggestion and the table to commit as fused._LT_ALGO_TABLE -- the product never times
anything itself (every process, rank and box must run the same library kernel for a shape).

    python tools/tune_lt_algos.py > gpurun_out/r05_lt_algos.txt
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
from quantized_training import _native  # noqa: E402

L = _native.lib()
DEV = torch.device("cuda:0")

# (batch, M, N, K, b_is_kn, bias)
SHAPES = [
    # BERT-base [16, 384] (configs[1]): q / k / v group, intermediate, output, attention output -- all with bias
    (1, 6144, 2304, 768, 0, 1), (1, 6144, 3072, 768, 0, 1), (1, 6144, 768, 3072, 0, 1), (1, 6144, 768, 768, 0, 1),
    # the pair route's LLaMA shapes (fused._FQ8_TABLE False entries) and the 13B widths
    (1, 2048, 4096, 4096, 0, 0), (1, 4096, 4096, 4096, 0, 0), (1, 1024, 13824, 5120, 0, 0), (1, 1024, 5120, 13824, 0, 0),
    (1, 1024, 15360, 5120, 0, 0), (1, 1024, 5120, 5120, 0, 0),
    # the headline's shapes (run by the fused kernel by default; listed so that QT_FQ8_GEMM=0 is deterministic too)
    (1, 1024, 11008, 4096, 0, 0), (1, 1024, 4096, 11008, 0, 0), (1, 1024, 4096, 4096, 0, 0), (1, 1024, 12288, 4096, 0, 0),
    (1, 1024, 32000, 4096, 0, 0),
    # attention chains neither attention kernel takes: Q.K^T and P.V, LLaMA-2-7B / 13B and BERT-base
    (32, 1024, 1024, 128, 0, 0), (32, 1024, 128, 1024, 1, 0), (40, 1024, 1024, 128, 0, 0), (40, 1024, 128, 1024, 1, 0),
    (192, 384, 384, 64, 0, 0), (192, 384, 64, 384, 1, 0),
]


def main():
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    st = ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    table = {}
    for rep in range(2):                                        # twice: the choice must be stable
        for (b, M, N, K, kn, bias) in SHAPES:
            a = torch.randint(0, 120, (b, M, K), dtype=torch.uint8, device=DEV)
            w = torch.randint(0, 120, (b, K, N) if kn else ((b, N, K) if b > 1 else (N, K)), dtype=torch.uint8, device=DEV)
            c = torch.empty((b, M, N), dtype=torch.bfloat16, device=DEV)
            bv = torch.zeros(N, dtype=torch.bfloat16, device=DEV) if bias else None
            best = ctypes.c_int(0)
            us = (ctypes.c_float * 16)()
            n = L.qt_fp8_gemm_tune(a.data_ptr(), 0, w.data_ptr(), 0, kn, c.data_ptr(), bv.data_ptr() if bias else None, b, M, N, K, M * K,
                                   (N * K if b > 1 else 0), M * N, ws.data_ptr(), ws.numel(), ctypes.byref(best), us, 16, st)
            if n < 1:
                print(f"{(b, M, N, K, kn, bias)}: no suggestion (rc {n})")
                continue
            times = [round(us[i], 1) for i in range(min(n, 16))]
            print(f"pass {rep} {(b, M, N, K, kn, bias)}: best {best.value}  us per suggestion {times}", flush=True)
            # keep the library's first suggestion unless another one is at least 3 % faster in BOTH passes
            first, bt = times[0], times[best.value]
            pick = best.value if (first < 0 or bt < 0.97 * first) else 0
            prev = table.get((b, M, N, K, kn, bias))
            table[(b, M, N, K, kn, bias)] = pick if (prev is None or prev == pick) else 0
    print("_LT_ALGO_TABLE = {")
    for k, v in table.items():
        if v:
            print(f"    {k}: {v},")
    print("}")


if __name__ == "__main__":
    main()
