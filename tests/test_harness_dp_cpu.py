"""Data-parallel evaluation harness on CPU: window schedule, round-robin sharding and the metric
all_gather over a 2-rank gloo group reproduce the single-process perplexity exactly (stateless
fake-quant).  Mirrors the N > 1 path of bench.py / harness.evaluate_perplexity without a GPU."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import quantized_training as qt
from quantized_training import harness

G = os.path.join(os.path.dirname(__file__), "golden")


def test_window_schedule_matches_reference_loop():
    g = json.load(open(os.path.join(G, "wikitext_windows.json")))
    for key, v in g.items():
        seq, ml, st = map(int, key.split(","))
        rows = harness.wikitext_windows(seq, ml, st)
        assert len(rows) == v["n"] and sum(r[2] for r in rows) == v["sum_trg"]
        assert [list(r) for r in rows[:3]] == v["first"] and [list(r) for r in rows[-2:]] == v["last"]


def test_round_robin_shards_partition_the_windows():
    w = harness.wikitext_windows(341469, 1024, 512)
    parts = [harness.shard_round_robin(w, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == sorted(w)
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _model():
    m = harness.build_causal_lm("llama-tiny", device="cpu", dtype=torch.float32, seed=0)
    args = qt.add_qspec_args().parse_args(["--activation", "e4m3", "--weight", "posit8_1", "--quantize_forward", "gemm"])
    qt.quantize(m, args)
    return m


def _tokens():
    return torch.randint(0, 512, (1, 1100), generator=torch.Generator().manual_seed(3))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    m = _model()
    ppl, nlls = harness.evaluate_perplexity(m, _tokens(), max_length=128, stride=64, rank=rank, world=world)
    if rank == 0:
        torch.save({"ppl": ppl, "nlls": nlls}, out)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_two_rank_gloo_equals_single_process(tmp_path):
    torch.set_num_threads(2)
    ppl1, nlls1 = harness.evaluate_perplexity(_model(), _tokens(), max_length=128, stride=64)
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["nlls"].shape == nlls1.shape
    assert torch.equal(got["nlls"], nlls1)
    assert got["ppl"] == ppl1


def test_gather_in_order_single_rank_identity():
    x = torch.arange(5, dtype=torch.float32)
    assert torch.equal(harness.gather_in_order(x, 5, 0, 1), x)


def _qa_model():
    from transformers import BertConfig, BertForQuestionAnswering
    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=100,
                     max_position_embeddings=64)
    m = BertForQuestionAnswering(cfg).eval()
    qt.quantize(m, qt.add_qspec_args().parse_args(["--activation", "posit8_1", "--weight", "posit8_1"]))
    return m


def _qa_batches():
    g = torch.Generator().manual_seed(9)
    sizes = [4, 4, 4, 4, 3]                                   # ragged last batch, like a real dataloader
    return [{"input_ids": torch.randint(3, 100, (n, 32), generator=g),
             "attention_mask": torch.ones(n, 32, dtype=torch.long)} for n in sizes]


def _qa_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    s, e = harness.collect_qa_logits(_qa_model(), _qa_batches(), rank=rank, world=world)
    if rank == 1:
        torch.save((s, e), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_qa_logit_collection_two_ranks(tmp_path):
    torch.set_num_threads(2)
    s1, e1 = harness.collect_qa_logits(_qa_model(), _qa_batches())
    assert s1.shape == (19, 32) and s1.dtype == torch.float32
    out = str(tmp_path / "qa.pt")
    mp.spawn(_qa_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    s2, e2 = torch.load(out)
    assert torch.equal(s1, s2) and torch.equal(e1, e2)


def test_glue_style_training_loop_with_quantized_backward():
    from transformers import RobertaConfig, RobertaForSequenceClassification
    torch.manual_seed(0)
    cfg = RobertaConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=100,
                        max_position_embeddings=66, num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = RobertaForSequenceClassification(cfg)
    args = qt.add_qspec_args().parse_args([
        "--activation", "int8,qs=per_tensor_symmetric", "--weight", "int8,qs=per_tensor_symmetric",
        "--error", "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10",
        "--quantize_forward", "gemm", "--quantize_backprop", "gemm,residual"])
    qt.quantize(m, args)
    g = torch.Generator().manual_seed(1)
    batches = [{"input_ids": torch.randint(3, 100, (8, 16), generator=g), "labels": torch.randint(0, 2, (8,), generator=g)}
               for _ in range(6)]
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    losses = harness.train_steps(m, batches, opt)
    assert len(losses) == 6 and all(l == l and l < 10 for l in losses)
    mods = dict(m.named_modules())
    fq = mods["roberta.encoder.layer.0.attention.self.query.error_pre_process.0"]
    assert fq.dtype == "fp8_e5m2" and float(fq.amax_history.max()) > 0 and float(fq.scale) != 1.0
    assert "roberta.encoder.layer.0.attention.self.query.error_post_process.0" in mods       # residual-feeding layer
    assert "roberta.encoder.layer.0.output.residual.error_post_process.0" in mods


# ---- calibration flow, checkpoints, eval weight cache ---------------------------------------------
def _toy_qat(seed=0):
    import quantized_training as qt
    torch.manual_seed(seed)
    m = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8))
    args = qt.add_qspec_args().parse_args(["--activation", "int8,qs=per_tensor_symmetric,ahl=4",
                                           "--weight", "int8,qs=per_tensor_symmetric,ahl=4"])
    qt.quantize(m, args)
    return m


def _scales(m):
    return {k: v.clone() for k, v in m.state_dict().items() if k.endswith(".scale")}


def test_calibrate_then_observers_are_frozen():
    from quantized_training import harness
    m = _toy_qat()
    g = torch.Generator().manual_seed(1)
    batches = [torch.randn(4, 16, generator=g) * (i + 1) for i in range(6)]
    assert harness.calibrate(m, batches, steps=4) == 4
    for mod in m.modules():
        if isinstance(mod, torch.ao.quantization.FakeQuantizeBase):
            assert int(mod.observer_enabled[0]) == 0 and int(mod.fake_quant_enabled[0]) == 1
    before = _scales(m)
    assert any(float(v) != 1.0 for v in before.values())          # calibration moved the scales
    with torch.no_grad():
        m(batches[5] * 100)
    after = _scales(m)
    assert all(torch.equal(before[k], after[k]) for k in before)   # ... and they no longer follow the data


def test_checkpoint_round_trip_restores_lazy_fake_quant_state(tmp_path):
    from quantized_training import harness
    m = _toy_qat()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    x = torch.randn(4, 16, generator=torch.Generator().manual_seed(3))
    for _ in range(3):
        m(x).sum().backward(); opt.step(); opt.zero_grad()
    path = str(tmp_path / "checkpoint.tar")
    harness.save_checkpoint(path, m, opt, best_metric={"f1": 1.0})
    want = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = _toy_qat(seed=5)
    m2(x)                                                            # creates the per-argument fake-quantizers
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-3)
    ck = harness.load_checkpoint(path, m2, opt2)
    assert sorted(ck) == ["best_metric", "model_state_dict", "optimizer_state_dict", "run_id", "scheduler_state_dict"]
    got = m2.state_dict()
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].shape == want[k].shape and torch.equal(got[k], want[k]), k
    m.eval(); m2.eval()
    with torch.no_grad():
        assert torch.equal(m(x), m2(x))


def test_eval_weight_cache_is_bit_identical_and_skips_the_weight_pass():
    from quantized_training import harness
    from quantized_training.fake_quantize import STATS
    m = _toy_qat()
    x = torch.randn(4, 16, generator=torch.Generator().manual_seed(7))
    harness.calibrate(m, [x, 2 * x], steps=2)
    m.eval()
    with torch.no_grad():
        STATS.reset(); y0 = m(x); full = STATS.elements
        harness.cache_quantized_weights(True)
        try:
            m(x)
            STATS.reset(); y1 = m(x); cached = STATS.elements
            assert torch.equal(y0, y1)
            assert cached == full - sum(mod.weight.numel() for mod in m.modules() if hasattr(mod, "weight_fake_quant"))
            m[0].weight.mul_(2.0)                                   # an in-place update invalidates the entry
            STATS.reset(); y2 = m(x)
            assert STATS.elements == cached + m[0].weight.numel() and not torch.equal(y2, y1)
            m.train()                                               # training always re-quantizes
            STATS.reset(); m(x)
            assert STATS.elements == full
        finally:
            harness.cache_quantized_weights(False)


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (child process, before anything touches a GPU);
    with --dry-run the same sharding / barrier / max-over-ranks timing / metric gather runs on the CPU over gloo and rank 0
    prints the one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["valid"] is False and out["config"]["parallelism"].startswith("dp2")
    # a launcher that started a different number of ranks than --gpus says: clean non-zero exit, no traceback
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--dry-run"], capture_output=True, text=True,
                        timeout=300, env=env2)
    assert p2.returncode == 2 and "Traceback" not in p2.stderr
