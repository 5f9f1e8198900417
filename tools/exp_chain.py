"""GPU experiment: qt_fake_quant_chain_bf16 (several fake-quantizer calls of a training step over one tensor in one launch, with the
bias gradient's column sums) against the launches it replaces, inside a replayed hipGraph with the amax slots zeroed before every
launch (what a step presents).

    python tools/exp_chain.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
import quantized_training as qt  # noqa: E402
from quantized_training import _native  # noqa: E402
from quantized_training.fake_quantize import _launch_format  # noqa: E402

DEV = torch.device("cuda:0")
L = _native.lib()


def graph_time(fn, n=32, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (n * reps)


def main():
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)   # noqa: E731
    for dtype, nstage, src in (("fp8_e5m2", 4, (-1, 0, 0, 1)), ("fp8_e5m2", 1, (-1,)), ("int8", 3, (-1, -1, -1))):
        fmt = _native.format_for(dtype)
        lut = qt.get_quantization_map(dtype, DEV)
        fmt = _launch_format(fmt, lut)
        for rows, cols in ((2048, 768), (2048, 3072)):
            x = (torch.randn(rows, cols, device=DEV) * 1e-4).bfloat16() if dtype != "int8" else torch.randn(rows, cols, device=DEV).bfloat16()
            sc = [torch.tensor([3e-9 if dtype != "int8" else 0.03], dtype=torch.float32, device=DEV) for _ in range(nstage)]
            am = torch.zeros(nstage, 32, dtype=torch.float32, device=DEV)
            outs = [torch.empty_like(x) for _ in range(nstage)]
            gb = torch.empty(cols, dtype=torch.bfloat16, device=DEV)
            wb = L.qt_fake_quant_chain_ws_bytes(rows, cols)
            ws = torch.zeros(max(wb, 16), dtype=torch.uint8, device=DEV)
            fmax = 57344.0 if dtype != "int8" else 127.0
            stages = (_native.QtChainStage * nstage)()
            for i in range(nstage):
                stages[i].scale_f32_dev, stages[i].amax_bits_dev, stages[i].out_dev, stages[i].src = sc[i].data_ptr(), am[i].data_ptr(), outs[i].data_ptr(), src[i]

            def zero():
                am.zero_()

            def chain(colsum):
                am.zero_()
                _native.check(L.qt_fake_quant_chain_bf16(x.data_ptr(), rows, cols, stages, nstage, ctypes.byref(fmt), lut.data_ptr(), nstage - 1 if colsum else -1,
                                                         fmax, gb.data_ptr(), ws.data_ptr(), wb, st()), "chain")

            def single(colsum):
                am.zero_()
                for i in range(nstage):
                    inp = x if src[i] < 0 else outs[src[i]]
                    _native.check(L.qt_fake_quant_bf16(inp.data_ptr(), outs[i].data_ptr(), inp.numel(), ctypes.byref(fmt), lut.data_ptr(), sc[i].data_ptr(),
                                                       am[i].data_ptr(), st()), "fq")
                if colsum:
                    _native.check(L.qt_colsum_bf16(outs[-1].data_ptr(), gb.data_ptr(), rows, cols, st()), "colsum")
            t0 = graph_time(zero)
            print(f"{dtype} x{nstage} [{rows}, {cols}]: chain {graph_time(lambda: chain(False)) - t0:6.2f} us, + column sums {graph_time(lambda: chain(True)) - t0:6.2f} us   |   "
                  f"single launches {graph_time(lambda: single(False)) - t0:6.2f} us, + qt_colsum_bf16 {graph_time(lambda: single(True)) - t0:6.2f} us   (zeroing launch {t0:.2f} us subtracted)", flush=True)


if __name__ == "__main__":
    main()
