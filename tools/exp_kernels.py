"""GPU experiment: per-kernel rates of the elementwise pass and the fake-quant GEMM at LLaMA-2-7B /
BERT-base shapes (SURVEY.md section 8(d) microbench list).  Writes gpurun_out/exp_kernels.json."""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
import quantized_training as qt  # noqa: E402
from quantized_training import _native as nv  # noqa: E402

L = nv.lib()
dev = torch.device("cuda")


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def time_fq(shape, dtype_name, io, scale, observe, force_lut=False, iters=20):
    x = (torch.randn(*shape, device=dev) * 0.5)
    x = x.bfloat16() if io == "bf16" else x.float()
    y = torch.empty_like(x)
    lut = qt.get_quantization_map(dtype_name, dev)
    fmt = nv.QtFormat(0, 0, 0, 0.0, 0.0) if force_lut else nv.format_for(dtype_name)
    s = torch.tensor([scale], device=dev, dtype=torch.float32)
    hist = torch.zeros(16, device=dev, dtype=torch.float32)
    n = x.numel()
    fn = L.qt_fake_quant_bf16 if io == "bf16" else L.qt_fake_quant_f32

    def run():
        nv.check(fn(x.data_ptr(), y.data_ptr(), n, ctypes.byref(fmt), lut.data_ptr(), s.data_ptr(),
                    hist.data_ptr() if observe else None, stream()), "fq")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bpe = 4 if io == "bf16" else 8
    return {"shape": list(shape), "dtype": dtype_name, "io": io, "scale": scale, "observe": observe,
            "force_lut": force_lut, "ms": ms, "gelem_s": n / ms / 1e6, "GBps": n * bpe / ms / 1e6}


def time_gemm(M, N, K, wdtype, fused, iters=10):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wq = torch.empty_like(w)
    fmt = nv.format_for(wdtype)
    lut = qt.get_quantization_map(wdtype or 'e4m3', dev)
    qx = nv.QtOperandQ(); qx.fmt = nv.QtFormat(nv.QT_FMT_IDENTITY, 0, 0, 0.0, 0.0)
    qw = nv.QtOperandQ(); qw.fmt = fmt; qw.lut_dev = lut.data_ptr()
    sc = torch.tensor([0.02], device=dev)
    if wdtype == 'int8': qw.scale_f32_dev = sc.data_ptr()

    def run_fused():
        nv.check(L.qt_linear_fq_bf16(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), M, N, K,
                                     ctypes.byref(qx), ctypes.byref(qw), stream()), "gemm")

    def run_unfused():
        nv.check(L.qt_fake_quant_bf16(w.data_ptr(), wq.data_ptr(), w.numel(), ctypes.byref(fmt), lut.data_ptr(),
                                      None, None, stream()), "fq")
        torch.nn.functional.linear(x, wq)

    def run_plain():
        torch.nn.functional.linear(x, w)

    run = {"fused": run_fused, "unfused": run_unfused, "plain": run_plain}[fused]
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return {"M": M, "N": N, "K": K, "wdtype": wdtype, "mode": fused, "ms": ms, "TFLOPs": 2.0 * M * N * K / ms / 1e9}


def main():
    out = {"elementwise": [], "gemm": []}
    shapes = [(1024, 4096), (4096, 4096), (4096, 11008), (32000, 4096), (1, 32, 1024, 1024), (1, 32, 1024, 128),
              (6144, 768), (6144, 3072), (16, 12, 384, 384)]
    for shp in []:
        for dt, sc, obs, lutf in [("e4m3", 1.0, False, False), ("e4m3", 1.0, False, True), ("posit8_1", 1.0, False, False),
                                  ("fp8_e4m3", 0.01, True, False), ("int8", 0.01, True, False), ("e4m3", 0.01, True, False)]:
            r = time_fq(shp, dt, "bf16", sc, obs, lutf)
            out["elementwise"].append(r)
            print(r, flush=True)
    for shp in []:
        for dt, sc, obs in [("e4m3", 1.0, False), ("posit8_1", 1.0, False), ("int8", 0.01, True)]:
            r = time_fq(shp, dt, "f32", sc, obs)
            out["elementwise"].append(r)
            print(r, flush=True)
    for (M, N, K) in [(1024, 4096, 4096), (1024, 11008, 4096), (1024, 4096, 11008), (1024, 32000, 4096),
                      (6144, 768, 768), (6144, 3072, 768), (6144, 768, 3072)]:
        for mode, wd in (("fused", "e4m3"), ("fused", "int8"), ("fused", None), ("unfused", "e4m3"), ("plain", "e4m3")):
            r = time_gemm(M, N, K, wd, mode)
            out["gemm"].append(r)
            print(r, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "exp_kernels.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
