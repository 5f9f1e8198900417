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
    ms = ctypes.c_float()
    for iters in (pool, 6 * pool):
        nv.check(L.qt_bench_fake_quant_bf16(x.data_ptr(), y.data_ptr(), n, ctypes.byref(fmt), qt.get_quantization_map(dtype, dev).data_ptr(),
                                            scale.data_ptr() if scale is not None else None,
                                            amax.data_ptr() if amax is not None else None, iters, n, pool, st, ctypes.byref(ms)), "b")
    r = {"dtype": dtype, "variant": variant, "blocks_per_cu": bpc, "label": label, "us": ms.value * 1e3, "GBps": n * 4 / ms.value / 1e6}
    res.append(r); print(r, flush=True)
for rep in range(2):
    for v in (0, 5, 6, 11, 12, 13, 14, 15, 16, 17):
        for bpc in (16, 32, 64, 128, 100000):
            run("e4m3", v, bpc)
s = torch.tensor([0.013], device=dev); h = torch.zeros(16, device=dev)
for v, bpc in ((0, 8), (0, 32), (6, 32)):
    run("e4m3", v, bpc, s, h, "scale+obs")
    run("int8", v, bpc, s, h, "scale+obs")
    run("posit8_1", v, bpc, None, None, "lut")
    run("posit8_1", v, bpc, s, h, "lut scale+obs")
L.qt_internal_set_variant(0, 8)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "exp_stream.json"), "w"), indent=1)
