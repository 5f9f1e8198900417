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
