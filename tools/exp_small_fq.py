"""GPU experiment: per-launch time of the fake-quant pass on the SMALL tensors of the RoBERTa training step (configs[4]), inside a
replayed hipGraph (what the step runs): int8 with a scale and the observer on, fp8_e5m2 (table format, row form) with scale + observer,
against a trivial torch kernel (the per-launch floor of a graph node) and a bf16 copy of the same size.

    python tools/exp_small_fq.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
import quantized_training as qt  # noqa: E402
from dataclasses import asdict  # noqa: E402

DEV = torch.device("cuda:0")


def graph_time(fn, n=64, reps=20):
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
    shapes = [(2048, 768), (2048, 3072), (16 * 12 * 128, 128), (768, 768), (768, 3072), (2048, 4096), (4096, 4096)]
    specs = ["int8,qs=per_tensor_symmetric", "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10", "e4m3"]
    tiny = torch.zeros(64, device=DEV)
    print(f"floor (torch add_ on 64 floats): {graph_time(lambda: tiny.add_(1.0)):.2f} us per launch")
    for shape in shapes:
        x = torch.randn(*shape, device=DEV).bfloat16()
        y = torch.empty_like(x)
        line = [f"{shape[0]}x{shape[1]} ({x.numel() * 4 / 1e6:.1f} MB moved): copy {graph_time(lambda: y.copy_(x)):.2f} us"]
        for spec in specs:
            m = qt.FusedAmaxObsFakeQuantize(**asdict(qt.QuantizationSpec.from_str(spec)), device=DEV)
            with torch.no_grad():
                m(x); m(x)
                t = graph_time(lambda: m(x))
            line.append(f"{spec.split(',')[0]}{'+obs' if 'qs=' in spec else ''} {t:.2f}")
            if "qs=" in spec:
                from quantized_training.fake_quantize import BatchedScaleUpdate
                with torch.no_grad():
                    b = BatchedScaleUpdate([m], DEV)

                    def pre():
                        b.launch()
                        return m(x)
                    t2 = graph_time(pre)
                    m.disable_observer()
                    t3 = graph_time(lambda: m(x))
                line.append(f"(batched update + call {t2:.2f}, frozen scale {t3:.2f})")
        print("   ".join(line), flush=True)


if __name__ == "__main__":
    main()


def abi_level():
    """The pieces of one observed call at the C ABI: the pass with / without the amax slot, with the slot zeroed before every pass (the
    state a training step presents: round 4 found the atomics' queue only there), the scale update alone."""
    import ctypes
    from quantized_training import _native
    L = _native.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    fmt = _native.format_for("int8")
    hist = torch.zeros(10, dtype=torch.float32, device=DEV)
    scale = torch.full((1,), 0.037, dtype=torch.float32, device=DEV)
    for shape in [(768, 768), (2048, 768), (2048, 3072)]:
        x = torch.randn(*shape, device=DEV).bfloat16()
        y = torch.empty_like(x)

        def run(amax):
            s = ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
            _native.check(L.qt_fake_quant_bf16(x.data_ptr(), y.data_ptr(), x.numel(), ctypes.byref(fmt), None, scale.data_ptr(),
                                               hist.data_ptr() if amax else None, s), "fq")

        def upd():
            s = ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
            _native.check(L.qt_scale_update(hist.data_ptr(), 10, 1, scale.data_ptr(), 127.0, 0, s), "upd")
        def run_fresh():
            # what a training step sees: the slot was zeroed since the last pass (here by a 4-byte memset node), so every workgroup can
            # raise the running maximum and issues its atomicMax; `run(True)` alone leaves the maximum in place and nobody does
            hist[:1].zero_()
            run(True)

        def zero_only():
            hist[:1].zero_()
        print(f"ABI {shape}: pass without observer {graph_time(lambda: run(False)):.2f} us   with amax slot {graph_time(lambda: run(True)):.2f} us   "
              f"with a freshly zeroed slot {graph_time(run_fresh) - graph_time(zero_only):.2f} us   scale update alone {graph_time(upd):.2f} us", flush=True)


if __name__ == "__main__" and os.environ.get("ABI", "1") == "1":
    abi_level()
