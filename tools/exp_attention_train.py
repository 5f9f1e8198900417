"""GPU experiment: the training attention kernels (qt_attention_train_bf16 / _backward_bf16) at the configs[4] shape -- launch time
back to back (HIP events) and, on the tuning build, s_memtime stamps of workgroup 0 per phase.

    make -C quantized-training_amd tuning
    QT_HIP_LIB=tools/build/libqt_hip_tuning.so python tools/exp_attention_train.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
DEV = torch.device("cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device=DEV)
os.environ["QT_AT_STAMPS"] = hex(stamps.data_ptr())
import quantized_training as qt  # noqa: E402
from quantized_training import _native  # noqa: E402
from quantized_training.fake_quantize import _launch_format  # noqa: E402

L = _native.lib()
FWD = ("start", "stage q' k' v'", "barrier", "S = q' k'^T", "barrier", "softmax + fq", "barrier", "O = P' v'", "barrier", "stores", "amax")
BWD = ("start", "stage g q' k' v' P'", "barrier", "dV, dP + tiles", "barrier", "softmax bwd + fq, dV out", "barrier", "dQ, dK + tiles", "barrier", "stores",
       "amax")


def main():
    B, H, S, D = 16, 12, 128, 64
    torch.manual_seed(0)

    def proj():
        return torch.randn(B, S, H * D, device=DEV).bfloat16().view(B, S, H, D).permute(0, 2, 1, 3)
    q, k, v = proj(), proj(), proj()
    mask = torch.zeros(B, 1, 1, S, device=DEV, dtype=torch.bfloat16)
    mask[::2, :, :, S - 24:] = torch.finfo(torch.bfloat16).min
    lut = qt.get_quantization_map("int8", DEV)
    fmt = _launch_format(_native.format_for("int8"), lut)
    lut5 = qt.get_quantization_map("fp8_e5m2", DEV)
    fmt5 = _launch_format(_native.format_for("fp8_e5m2"), lut5)
    sc = [torch.tensor([x], dtype=torch.float32, device=DEV) for x in (0.0317, 0.0291, 0.0333, 0.00787, 0.0171)]
    am = [torch.zeros(1, dtype=torch.float32, device=DEV) for _ in sc]
    qq, kq, vq = (torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=DEV) for _ in range(3))
    probs = torch.empty(B, H, S, S, dtype=torch.bfloat16, device=DEV)
    pq = torch.empty_like(probs)
    out = torch.empty(B, S, H * D, dtype=torch.bfloat16, device=DEV)
    oq = torch.empty_like(out)
    outs = [qq, kq, vq, pq, oq]
    stages = (_native.QtChainStage * 5)()
    for i in range(5):
        stages[i].scale_f32_dev, stages[i].amax_bits_dev, stages[i].out_dev, stages[i].src = sc[i].data_ptr(), am[i].data_ptr(), outs[i].data_ptr(), -1
    st = ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)

    def fwd():
        _native.check(L.qt_attention_train_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), q.stride(2), q.stride(1), mask.data_ptr(),
                                                mask.stride(0), 0, 0, stages, probs.data_ptr(), out.data_ptr(), None, 1.0, B, H, S, D, 0.125, ctypes.byref(fmt),
                                                lut.data_ptr(), st), "fwd")
    gy = (torch.randn(B, S, H, D, device=DEV) * 1e-3).bfloat16()
    esc = [torch.tensor([x], dtype=torch.float32, device=DEV) for x in (1.1e-7, 0.9e-7)]
    eam = [torch.zeros(1, dtype=torch.float32, device=DEV) for _ in esc]
    est = (_native.QtChainStage * 2)()
    for i in range(2):
        est[i].scale_f32_dev, est[i].amax_bits_dev, est[i].out_dev, est[i].src = esc[i].data_ptr(), eam[i].data_ptr(), None, -1
    dq, dk, dv = (torch.empty(B, S, H, D, dtype=torch.bfloat16, device=DEV) for _ in range(3))
    # the projections' backward-pre quantizers and bias gradients riding on the gradients' way out
    gsc = [torch.tensor([x], dtype=torch.float32, device=DEV) for x in (1.0e-7, 1.0e-7, 1.0e-7)]
    gam = [torch.zeros(1, dtype=torch.float32, device=DEV) for _ in gsc]
    gout = [torch.empty(B, S, H * D, dtype=torch.bfloat16, device=DEV) for _ in gsc]
    gb = [torch.empty(H * D, dtype=torch.bfloat16, device=DEV) for _ in gsc]
    gst = (_native.QtChainStage * 3)()
    couts = (ctypes.c_void_p * 3)()
    for i in range(3):
        gst[i].scale_f32_dev, gst[i].amax_bits_dev, gst[i].out_dev, gst[i].src = gsc[i].data_ptr(), gam[i].data_ptr(), gout[i].data_ptr(), -1
        couts[i] = gb[i].data_ptr()
    ws = torch.zeros(L.qt_attention_train_backward_ws_bytes(H), dtype=torch.uint8, device=DEV)

    def bwd():
        _native.check(L.qt_attention_train_backward_bf16(gy.data_ptr(), qq.data_ptr(), kq.data_ptr(), vq.data_ptr(), qq.stride(0), qq.stride(2), qq.stride(1),
                                                         probs.data_ptr(), pq.data_ptr(), est, None, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), gst, couts, 57344.0,
                                                         ws.data_ptr(), ws.numel(), None, 1.0, B, H, S, D, 0.125, ctypes.byref(fmt5), lut5.data_ptr(), st), "bwd")
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    for name, fn, labels, base in (("forward", fwd, FWD, 0), ("backward", bwd, BWD, 32)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        hot = e0.elapsed_time(e1) / 50 * 1e3
        cold = []
        for _ in range(5):
            flush.zero_()                              # evict L2 / MALL, and the instruction cache sees another kernel
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            cold.append(e0.elapsed_time(e1) * 1e3)
        print(f"{name}: {hot:.1f} us per launch back to back, {min(cold):.1f} us after a 512 MiB fill (event pair around one launch)")
        t = stamps.cpu().tolist()
        if any(t[base:base + 32]):
            for w, off in ((0, 0), (7, 16)):
                s_ = t[base + off:base + off + 11]
                print(f"  wave {w}: " + "  ".join(f"{labels[i]} {s_[i] - s_[i - 1]}" for i in range(1, 11)) + f"  | total {s_[10] - s_[0]}")


main()
