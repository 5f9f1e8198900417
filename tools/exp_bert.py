"""BASELINE config 1 on one GPU: BERT-base-shaped QA model (random init), E4M3 activations + weights, SQuAD-style eval
batch [16, 384]; ms per batch and quantized elements per second (STATS counts elements as the hooks issue them)."""
import sys
import time

import torch

sys.path.insert(0, "quantized-training_amd")
import quantized_training as qt
from quantized_training.fake_quantize import STATS
from transformers import BertConfig, BertForQuestionAnswering

torch.manual_seed(0)
cfg = BertConfig()          # base: 768 hidden, 12 layers, 12 heads, 3072 FFN
m = BertForQuestionAnswering(cfg).cuda().eval()
qt.quantize(m, qt.add_qspec_args().parse_args(["--activation", "e4m3", "--weight", "e4m3", "--bf16", "--quantize_forward", "gemm"]))
ids = torch.randint(1000, 30000, (16, 384), device="cuda")
att = torch.ones_like(ids)
tt = torch.zeros_like(ids)
with torch.no_grad():
    for _ in range(3):
        m(input_ids=ids, attention_mask=att, token_type_ids=tt)
    STATS.reset()
    m(input_ids=ids, attention_mask=att, token_type_ids=tt)
    elems = STATS.elements
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        m(input_ids=ids, attention_mask=att, token_type_ids=tt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"eager: {dt*1e3:.2f} ms / batch, {elems} quantized elements -> {elems/dt/1e9:.1f} G elements/s")
    # graph replay
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            m(input_ids=ids, attention_mask=att, token_type_ids=tt)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = m(input_ids=ids, attention_mask=att, token_type_ids=tt)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        print(f"hipGraph replay: {dt*1e3:.2f} ms / batch -> {elems/dt/1e9:.1f} G elements/s")
    except Exception as e:  # noqa: BLE001
        print("graph capture failed:", type(e).__name__, str(e)[:200])

from torch.profiler import profile, ProfilerActivity
with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        m(input_ids=ids, attention_mask=att, token_type_ids=tt)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))
