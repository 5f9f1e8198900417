"""GPU experiment: launch-geometry sweep of the streaming fake-quant pass (rotating pool > MALL)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
import torch
import quantized_training as qt
from quantized_training import _native as nv
L = nv.lib()
L.qt_internal_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]
L.qt_internal_set_variant.restype = None
dev = torch.device("cuda")
rows, cols, pool = 4096, 11008, 8
n = rows * cols
x = torch.empty(pool, rows, cols, device=dev, dtype=torch.bfloat16).normal_(0, 0.02)
y = torch.empty_like(x)
lut = qt.get_quantization_map("e4m3", dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
res = []
def run(dtype, variant, bpc, scale=None, amax=None, label=""):
    fmt = nv.format_for(dtype)
    L.qt_internal_set_variant(variant, bpc)
    if dtype is None:
        dtype = "e4m3"      # only for the (unused) table pointer
    ms = ctypes.c_float()
    for iters in (pool, 6 * pool):
        nv.check(L.qt_bench_fake_quant_bf16(x.data_ptr(), y.data_ptr(), n, ctypes.byref(fmt), qt.get_quantization_map(dtype, dev).data_ptr(),
                                            scale.data_ptr() if scale is not None else None,
                                            amax.data_ptr() if amax is not None else None, iters, n, pool, st, ctypes.byref(ms)), "b")
    r = {"dtype": dtype, "variant": variant, "blocks_per_cu": bpc, "label": label, "us": ms.value * 1e3, "GBps": n * 4 / ms.value / 1e6}
    res.append(r); print(r, flush=True)
# ceiling check: the same launch as a pure 16-B copy (identity format, no arithmetic), plus torch's own copy
for rep in range(3):
    run(None, 0, 32, label="identity copy")
    run("e4m3", 0, 32)
import time
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(2):
    e0.record()
    for i in range(48):
        y[i % pool].copy_(x[i % pool])
    e1.record(); torch.cuda.synchronize()
    print({"label": "torch copy_", "GBps": n * 4 / (e0.elapsed_time(e1) / 48) / 1e6}, flush=True)
s = torch.tensor([0.013], device=dev); h = torch.zeros(16, device=dev)
for v, bpc in ((0, 32), (0, 8), (5, 32), (0, 4)):
    for dt in ("e4m3", "int8", "posit8_1"):
        run(dt, v, bpc, None, None, "unit")
        run(dt, v, bpc, None, h, "unit+obs")
        run(dt, v, bpc, s, None, "scale")
        run(dt, v, bpc, s, h, "scale+obs")
L.qt_internal_set_variant(0, 8)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "exp_stream.json"), "w"), indent=1)
