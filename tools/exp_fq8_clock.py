"""GPU experiment (tuning build): the clock the chip holds inside the fused FP8 GEMM's tiles, and the tile's length in CYCLES.

    QT_HIP_LIB=tools/build/libqt_hip_tuning.so python tools/exp_fq8_clock.py [--seconds 2.0]

Every workgroup of qt_linear_fq8_bf16 stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its tile
(QT_FQ8_STAMPS, tuning build only).  After `--seconds` of back-to-back launches on random data the last launch's stamps give
clock = cycles / ticks x 100 MHz (median over workgroups) -- MI355X_MICROARCH.md "DVFS give-back" item 6.  Printed per shape and
per ablation (QT_FQ8_ABLATE: 0 whole kernel, 5 fragment reads + multiplications + barriers with no operand traffic, 1 no
multiplications, 4 no fragment reads and no multiplications): wall us per launch, cycles per tile and per k step, clock.
"""
import argparse
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("QT_HIP_LIB", os.path.join(ROOT, "tools", "build", "libqt_hip_tuning.so"))

from quantized_training import _native  # noqa: E402

L = _native.lib()
DEV = torch.device("cuda:0")


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)


def fq8(x):
    fmt = _native.format_for("e4m3")
    y8 = torch.empty(x.shape, dtype=torch.uint8, device=DEV)
    one = torch.ones((), dtype=torch.float32, device=DEV)
    _native.check(L.qt_fake_quant_bf16_fp8(x.data_ptr(), None, y8.data_ptr(), x.numel(), ctypes.byref(fmt), one.data_ptr(), None, stream()), "fq8")
    return y8


def run(M, N, K, seconds, zeros=False):
    torch.manual_seed(0)
    pool = max(2, min(16, int(600e6 // (N * K * 2)) + 1))
    x = torch.zeros(M, K, device=DEV).bfloat16() if zeros else torch.randn(M, K, device=DEV).bfloat16()
    x8 = fq8(x)
    ws = [(torch.zeros(N, K, device=DEV) if zeros else torch.randn(N, K, device=DEV) * 0.05).bfloat16() for _ in range(pool)]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    stamps = torch.zeros(2 * 4096, dtype=torch.int64, device=DEV)
    os.environ["QT_FQ8_STAMPS"] = hex(stamps.data_ptr())
    ns = (ctypes.c_int * 1)(N)

    def call(i):
        wp = (ctypes.c_void_p * 1)(ws[i % pool].data_ptr())
        bp = (ctypes.c_void_p * 1)(None)
        _native.check(L.qt_linear_fq8_bf16(x8.data_ptr(), 0, wp, bp, ns, 1, 0, y.data_ptr(), M, K, stream()), "qt_linear_fq8_bf16")

    for i in range(5):
        call(i)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for i in range(200):
            call(n + i)
        n += 200
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(200):
        call(i)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 200
    st = stamps.cpu().view(-1, 2)
    st = st[st[:, 1] > 0]
    cyc, ticks = st[:, 0].double(), st[:, 1].double()
    clock = (cyc / ticks * 100.0).median().item()          # MHz
    return us, cyc.median().item(), cyc.max().item(), clock, len(st)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--shapes", default="1024x11008x4096,1024x4096x11008,1024x4096x4096,1024x12288x4096")
    ap.add_argument("--ablations", default="0,5,1,4")
    args = ap.parse_args()
    for shp in args.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        wide = N >= 8192
        for abl in args.ablations.split(","):
            if abl != "0" and not wide:
                env = {"1": None, "4": None, "5": "5"}.get(abl)       # the narrow-tile kernel has its own (smaller) ablation list
                if env is None:
                    continue
                os.environ["QT_FQ8_R2_ABLATE"] = env
            else:
                os.environ["QT_FQ8_ABLATE"] = abl
            for zeros in ((False, True) if abl == "0" else (False,)):
                us, cmed, cmax, clock, nwg = run(M, N, K, args.seconds, zeros)
                steps = K // 128
                print(f"{shp} ablate={abl} {'zeros ' if zeros else 'random'}: {us:7.2f} us/launch  tile {cmed:9.0f} cycles median ({cmax:9.0f} max) "
                      f"= {cmed / steps:7.1f} per k tile  clock {clock:6.0f} MHz  ({nwg} workgroups; tile = {cmed / clock:6.2f} us)", flush=True)
            os.environ.pop("QT_FQ8_ABLATE", None)
            os.environ.pop("QT_FQ8_R2_ABLATE", None)


if __name__ == "__main__":
    main()
