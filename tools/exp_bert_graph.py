"""BERT-base QA batch [16, 384], E4M3: capture once, replay 20 times (for a kernel trace of the replayed graph)."""
import sys, time, torch
sys.path.insert(0, "quantized-training_amd")
import quantized_training as qt
from transformers import BertConfig, BertForQuestionAnswering
torch.manual_seed(0)
m = BertForQuestionAnswering(BertConfig()).cuda().eval()
qt.quantize(m, qt.add_qspec_args().parse_args(["--activation", "e4m3", "--weight", "e4m3", "--bf16", "--quantize_forward", "gemm"]))
ids = torch.randint(1000, 30000, (16, 384), device="cuda"); att = torch.ones_like(ids); tt = torch.zeros_like(ids)
with torch.no_grad():
    for _ in range(3): m(input_ids=ids, attention_mask=att, token_type_ids=tt)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): m(input_ids=ids, attention_mask=att, token_type_ids=tt)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): out = m(input_ids=ids, attention_mask=att, token_type_ids=tt)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print("replay ms", (time.perf_counter() - t) / 20 * 1e3)
