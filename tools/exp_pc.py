"""Per-channel fake-quant pass on a LLaMA weight: vectorised kernel bandwidth."""
import sys, time, torch
sys.path.insert(0, "quantized-training_amd")
from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
from quantized_training.quantizer.quantizer import QScheme
for dtype, rng in (("int8", (-128.0, 127.0)), ("posit8_1", (-4096.0, 4096.0))):
    for obs in (True, False):
        fq = FusedAmaxObsFakeQuantize(dtype=dtype, qscheme=QScheme.PER_CHANNEL_SYMMETRIC, quant_min=rng[0], quant_max=rng[1],
                                      amax_history_len=4, ch_axis=0).cuda()
        w = [torch.randn(4096, 11008, device="cuda").bfloat16() for _ in range(4)]
        with torch.no_grad():
            for i in range(4): fq(w[i])
            if not obs: fq.disable_observer()
            torch.cuda.synchronize(); t = time.perf_counter()
            for i in range(20): fq(w[i % 4])
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        print(f"{dtype} observer={obs}: {dt*1e6:7.1f} us  {w[0].numel()*4/dt/1e12:5.2f} TB/s")
