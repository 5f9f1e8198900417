"""GPU experiment (tuning build): s_memtime stamps of qt_attention_rows_bf16's heaviest workgroup at the LLaMA-2-13B shape.

    make -C quantized-training_amd tuning
    QT_HIP_LIB=tools/build/libqt_hip_tuning.so python tools/exp_attention_rows_stamps.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-training_amd"))
DEV = torch.device("cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device=DEV)
os.environ["QT_AR_STAMPS"] = hex(stamps.data_ptr())
import quantized_training as qt  # noqa: E402
from quantized_training import _native  # noqa: E402
from quantized_training.fake_quantize import _launch_format  # noqa: E402

L = _native.lib()


def main():
    B, H, S, D = 1, 40, 1024, 128
    causal = "--nomask" not in sys.argv
    torch.manual_seed(0)
    m = qt.get_quantization_map("posit8_2", DEV)
    fmt = _launch_format(_native.format_for("posit8_2"), m)

    def on_grid(x):
        return m[(x.bfloat16().view(torch.int16).to(torch.int32) & 0xFFFF).long()]
    q = on_grid(torch.randn(B, H, S, D, device=DEV))
    k = on_grid(torch.randn(B, H, S, D, device=DEV))
    v = torch.randn(B, H, S, D, device=DEV).bfloat16()
    vt = torch.empty(B, H, D, S, device=DEV, dtype=torch.bfloat16)
    out = torch.empty(B, S, H, D, device=DEV, dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    mask = rl = None
    if causal:
        mask = torch.full((S, S), torch.finfo(torch.bfloat16).min, device=DEV).triu(1).bfloat16().contiguous()
        rl = torch.empty(S + 1, dtype=torch.int32, device=DEV)
        _native.check(L.qt_mask_row_live_checked(mask.data_ptr(), S, S, S, rl.data_ptr(), rl.data_ptr() + 4 * S, st), "live")
    _native.check(L.qt_value_t_rows(v.data_ptr(), vt.data_ptr(), B, H, S, D, H * S * D, S * D, D, ctypes.byref(fmt), m.data_ptr(), st), "value")
    for _ in range(3):
        _native.check(L.qt_attention_rows_bf16(
            q.data_ptr(), k.data_ptr(), vt.data_ptr(), mask.data_ptr() if causal else None, 0, 0, S,
            rl.data_ptr() if causal else None, 0, 0, 1, rl.data_ptr() + 4 * S if causal else None,
            out.data_ptr(), 1, ctypes.byref(fmt), m.data_ptr(), B, H, S, S, D, D ** -0.5, st), "rows")
    torch.cuda.synchronize()
    t = stamps.cpu().tolist()
    for grp in range(2):
        s = t[grp * 64:grp * 64 + 64]
        t0 = s[0]
        print(f"wave {4 * grp}: prologue {s[1] - t0}")
        for kb in range(8):
            a, b, c = s[2 + 3 * kb], s[3 + 3 * kb], s[4 + 3 * kb]
            prev = s[1] if kb == 0 else s[4 + 3 * (kb - 1)]
            print(f"  K  block {kb}: dma wait {a - prev:6d}  barrier {b - a:6d}  compute {c - b:6d}")
        print(f"  sweep 1 -> barrier {s[27] - s[26]}, max/exp/sum exchange {s[28] - s[27]}")
        for kb in range(8):
            a, b, c = s[29 + 3 * kb], s[30 + 3 * kb], s[31 + 3 * kb]
            prev = s[28] if kb == 0 else s[31 + 3 * (kb - 1)]
            print(f"  V  block {kb}: dma wait {a - prev:6d}  barrier {b - a:6d}  compute {c - b:6d}")
        print(f"  total to end of sweep 2: {s[53] - t0}")


main()
