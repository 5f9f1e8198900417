"""Times qt_mx_gemm's fp8 kernels on the M = 1024 LLaMA shapes under the tuning switches of the 256 x (16 nt) kernel
(QT_MX_WIDE, QT_MX_WIDE_DEBUG, QT_MX_WIDE_TILES_N are read once per process: one process per setting)."""
import ctypes
import os
import subprocess
import sys
import time

SHAPES = [(1024, 4096, 4096), (1024, 11008, 4096), (1024, 4096, 11008), (1024, 12288, 4096), (2048, 4096, 4096)]


def child():
    import torch
    sys.path.insert(0, "quantized-training_amd")
    from quantized_training import _native
    L = _native.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for M, N, K in SHAPES:
        a = torch.randint(0, 255, (M, K), device="cuda", dtype=torch.uint8) & 0x77
        b = torch.randint(0, 255, (N, K), device="cuda", dtype=torch.uint8) & 0x77
        sa = torch.full((M, K // 32), 127, device="cuda", dtype=torch.uint8)
        sb = torch.full((N, K // 32), 127, device="cuda", dtype=torch.uint8)
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        def run():
            _native.check(L.qt_mx_gemm(a.data_ptr(), sa.data_ptr(), 0, b.data_ptr(), sb.data_ptr(), 0, c.data_ptr(), 0, None, 1, M, N, K, 0, 0, st), "mx")
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(30):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 30
        out.append(f"{M}x{N}x{K} {dt * 1e6:6.1f} us")
    print("   ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    settings = [("old kernels", {"QT_MX_WIDE": "0"}), ("wide", {"QT_MX_WIDE": "1"}), ("wide, 256 rows", {"QT_MX_WIDE": "1", "QT_MX_WIDE_TM": "256"}),
                ("wide, 128 rows", {"QT_MX_WIDE": "1", "QT_MX_WIDE_TM": "128"}),
                ("wide, DMA only", {"QT_MX_WIDE": "1", "QT_MX_WIDE_DEBUG": "2"}),
                ("wide, no barrier (wrong results)", {"QT_MX_WIDE": "1", "QT_MX_WIDE_DEBUG": "32"})]
    for name, env in settings:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True, timeout=600)
        print(f"{name:32s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
