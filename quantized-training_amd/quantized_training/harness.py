"""Evaluation loops around the hot path, data-parallel over the GPUs of one node.

Counterparts of the reference's drivers (which this repo does not copy):
  * WikiText sliding-window perplexity      examples/language_modeling/wikitext.py:138-167
  * SQuAD-style batched logits collection   examples/question_answering/run_qa_no_trainer.py:914-959
Windows / batches are independent units, so they are sharded round-robin over ranks with NO
collective on the data path; one all_gather of the per-unit metric at the end (RCCL over xGMI when
the process group is `nccl`, gloo on CPU).  Exact only when the fake-quantizers are stateless
(no `qs`) or their observers are frozen: with live delayed scaling every rank's amax history sees a
different subsequence of windows.
"""
import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

__all__ = ["GraphedTrainStep", "GraphedBatch", "wikitext_windows", "shard_round_robin", "window_nll", "evaluate_perplexity", "gather_in_order", "GraphedWindow",
           "collect_qa_logits", "train_steps", "calibrate", "freeze_observers", "save_checkpoint", "load_checkpoint",
           "cache_quantized_weights", "build_causal_lm", "LLAMA_SHAPES"]


def wikitext_windows(seq_len: int, max_length: int, stride: int) -> List[Tuple[int, int, int]]:
    """(begin, end, trg_len) per window, exactly the schedule of wikitext.py:143-165: windows start
    every `stride` tokens while begin < seq_len - max_length; only the last `trg_len` tokens of a
    window are scored (the first window scores all of them)."""
    rows, prev_end = [], 0
    for begin in range(0, seq_len - max_length, stride):
        end = min(begin + max_length, seq_len)
        rows.append((begin, end, end - prev_end))
        prev_end = end
        if end == seq_len:
            break
    return rows


def shard_round_robin(items: Sequence, rank: int, world: int) -> List:
    return [it for i, it in enumerate(items) if i % world == rank]


def _window_loss(model, input_ids, labels, **kwargs):
    """`model(input_ids, labels=labels).loss` of a causal LM.  On the device with bf16 logits the loss comes from ONE pass over the
    logits (qt_causal_lm_loss_bf16: the same shifted-label fp32 cross entropy, mean over the scored positions) instead of the bf16 ->
    fp32 copy + softmax + reduction transformers runs; anything else takes the model's own loss."""
    import ctypes
    import os
    if isinstance(model, torch.fx.GraphModule):
        # a PT2E graph (prepare_pt2e_causal_lm): exported with exactly (input_ids, labels=, use_cache=) -- mask, positions and the loss
        # are nodes of the graph (wikitext.py:83-96 exports the model's own forward)
        return model(input_ids, labels=labels, use_cache=False).loss.float()
    # the one-pass loss restates transformers' ForCausalLMLoss: only for models whose loss IS that function
    stock_loss = getattr(getattr(model, "loss_function", None), "__name__", "") == "ForCausalLMLoss"
    if input_ids.is_cuda and not torch.is_grad_enabled() and stock_loss:
        from . import _native
        out = model(input_ids, use_cache=False, **kwargs)
        logits = getattr(out, "logits", None)
        if (logits is not None and logits.dim() == 3 and logits.dtype == torch.bfloat16 and logits.is_cuda and logits.stride(2) == 1
                and logits.stride(0) == logits.shape[1] * logits.stride(1) and logits.stride(1) % 8 == 0 and logits.data_ptr() % 16 == 0
                and labels.dtype == torch.long and labels.is_contiguous() and labels.shape == logits.shape[:2]):
            B, S, V = logits.shape
            scratch = torch.empty(B * S + 1, dtype=torch.float32, device=logits.device)
            with torch.cuda.device(logits.device):
                _native.note_device(logits.device.index)
                stream = ctypes.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)
                _native.check(_native.lib().qt_causal_lm_loss_bf16(logits.data_ptr(), labels.data_ptr(), B, S, V, logits.stride(1), -100,
                                                                   scratch.data_ptr(), scratch.data_ptr() + 4 * B * S, stream),
                              "qt_causal_lm_loss_bf16")
            return scratch[B * S]
        from transformers.loss.loss_utils import ForCausalLMLoss
        return ForCausalLMLoss(logits, labels, vocab_size=logits.shape[-1]).float()
    return model(input_ids, labels=labels, use_cache=False, **kwargs).loss.float()


@torch.no_grad()
def window_nll(model, input_ids: torch.Tensor, trg_len: int) -> torch.Tensor:
    """Mean NLL over the scored labels of one window (labels masked to -100 outside the last
    trg_len positions; the model shifts labels internally), wikitext.py:146-158."""
    target = input_ids.clone()
    target[:, :-trg_len] = -100
    return _window_loss(model, input_ids, target)


class GraphedWindow:
    """One window forward (+ loss) captured into a hipGraph.  The path has no host synchronisation
    (scales, amax and enable flags stay on the device), so the ~2 500 kernel launches of a LLaMA
    window replay back-to-back instead of being issued one by one from Python.  The label mask
    (positions scored) is an input buffer, so windows with different `trg_len` share the graph."""

    def __init__(self, model, max_length, _unused=None, device=None):
        self.model = model
        self.device = device if device is not None else next(model.parameters()).device
        self.ids = torch.zeros((1, max_length), dtype=torch.long, device=self.device)
        self.labels = torch.zeros((1, max_length), dtype=torch.long, device=self.device)
        # HF builds the causal mask with a host->device scalar copy, which a stream capture forbids; a
        # ready-made 4-D additive mask (same values: 0 / finfo.min) is passed through unchanged instead.
        dtype = next(model.parameters()).dtype
        self.mask = torch.full((max_length, max_length), torch.finfo(dtype).min, dtype=torch.float32,
                               device=self.device).triu(1).to(dtype)[None, None]
        # the positions of a window are always 0 .. max_length - 1: one static tensor, so that what the model derives from them
        # (rotary tables) can be kept from forward to forward (model_fusions._rotary_table_forward)
        import inspect
        takes_positions = "position_ids" in inspect.signature(model.forward).parameters
        self.positions = torch.arange(max_length, device=self.device).unsqueeze(0) if takes_positions else None
        self.graph = None
        self.loss = None

    @torch.no_grad()
    def _forward(self):
        extra = {"position_ids": self.positions} if self.positions is not None else {}
        return _window_loss(self.model, self.ids, self.labels, attention_mask=self.mask, **extra)

    @torch.no_grad()
    def capture(self, example_ids):
        self._set(example_ids, example_ids.shape[1])
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._forward()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: a collective library's watchdog thread (multi-GPU runs) may issue its own event queries while
        # this thread captures; only this thread's calls belong to the graph
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = self._forward()

    def _set(self, ids, trg_len):
        self.ids.copy_(ids)
        self.labels.copy_(ids)
        self.labels[:, : ids.shape[1] - trg_len] = -100

    @torch.no_grad()
    def replay(self, ids, trg_len):
        self._set(ids, trg_len)
        self.graph.replay()
        return self.loss


def prepare_pt2e_causal_lm(model, activation, weight, max_length: int, bias=None, fuse=None):
    """The reference's current WikiText flow up to the evaluation loop (examples/language_modeling/wikitext.py:68-101): default
    quantizer with the rotary matmul excluded, `torch.export` with a dynamic sequence length, fake-quantizers inserted as `call_module`
    nodes, constructor nodes pinned to the model's device.  Returns the prepared GraphModule (called as `gm(input_ids, labels=...,
    use_cache=False)`); for a device model its chains run on the fused HIP kernels (pt2e_fusion, `fuse=False` keeps the plain graph)."""
    from . import pt2e_fusion, quantize_pt2e as qp
    quantizer = qp.get_default_quantizer(input_activation=activation, weight=weight, bias=bias)
    quantizer.set_module_name_object_type_order(r"model\.rotary_emb", torch.ops.aten.matmul.default, 0, None)
    device = next(model.parameters()).device
    ids = torch.randint(0, model.config.vocab_size, (1, max_length), device=device)
    seq = torch.export.Dim("seq_length", min=3, max=max_length)
    dynamic_shapes = {"input_ids": {1: seq}, "labels": {1: seq}, "use_cache": None}
    config = model.config
    with torch.no_grad():
        gm = qp.prepare_pt2e(model, quantizer, (ids,), {"labels": ids.clone(), "use_cache": False}, dynamic_shapes, fuse=False)
    for node in list(gm.graph.nodes):                       # the exporter does not record the inputs' device (wikitext.py:98-101)
        if "device" in node.kwargs:
            node.kwargs = dict(node.kwargs, device=device)
    gm.recompile()
    gm.config = config
    if fuse or (fuse is None and device.type == "cuda"):
        gm.fusion_counts = pt2e_fusion.fuse_prepared_graph(gm)
    return gm


def _group_ready() -> bool:
    return dist.is_available() and dist.is_initialized()


def gather_in_order(local: torch.Tensor, n_total: int, rank: int, world: int, group=None) -> torch.Tensor:
    """All ranks hold values for units rank, rank+world, ...; returns all n_total values in unit order.
    One all_gather of a padded fp32 vector (<= a few KB): latency-bound, not bandwidth-bound.  With an initialised process group
    the collective runs even for one rank (the same RCCL / gloo call path as N ranks); without one, a single rank returns its values."""
    if world == 1 and not _group_ready():
        return local
    per = (n_total + world - 1) // world
    buf = torch.full((per,), float("nan"), dtype=torch.float32, device=local.device)
    buf[: local.numel()] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = torch.empty(n_total, dtype=torch.float32, device=local.device)
    for r in range(world):
        cnt = len(range(r, n_total, world))
        out[r::world] = parts[r][:cnt]
    return out


@torch.no_grad()
def evaluate_perplexity(model, token_ids: torch.Tensor, max_length: int = 1024, stride: int = 512,
                        device=None, rank: int = 0, world: int = 1, group=None,
                        max_windows: Optional[int] = None, use_graph: bool = False):
    """Sliding-window perplexity, windows sharded over ranks.  Returns (ppl, nlls[all windows])
    with ppl = exp(unweighted mean of per-window NLLs), wikitext.py:167.  `use_graph` replays a captured
    hipGraph for every full-length window on a GPU (the first call of each fake-quantizer must already
    have happened, e.g. by one eager window, which this function runs itself)."""
    assert token_ids.dim() == 2 and token_ids.shape[0] == 1
    windows = wikitext_windows(token_ids.shape[1], max_length, stride)
    if max_windows is not None:
        windows = windows[:max_windows]
    mine = shard_round_robin(list(enumerate(windows)), rank, world)
    device = device if device is not None else next(model.parameters()).device
    local = torch.empty(len(mine), dtype=torch.float32, device=device)
    graphed = None
    for j, (_, (begin, end, trg_len)) in enumerate(mine):
        ids = token_ids[:, begin:end].to(device)
        if use_graph and torch.device(device).type == "cuda" and end - begin == max_length:
            if graphed is None and j > 0:           # window 0 ran eagerly and created the fake-quantizers
                graphed = GraphedWindow(model, max_length, None, torch.device(device))
                graphed.capture(ids)
            if graphed is not None:
                local[j] = graphed.replay(ids, trg_len)
                continue
        local[j] = window_nll(model, ids, trg_len)
    nlls = gather_in_order(local, len(windows), rank, world, group)
    return math.exp(nlls.double().mean().item()), nlls


# name -> (hidden, layers, heads, kv_heads, ffn, vocab)        SURVEY.md section 8 sizes
LLAMA_SHAPES = {
    "llama-2-7b": (4096, 32, 32, 32, 11008, 32000),
    "llama-2-13b": (5120, 40, 40, 40, 13824, 32000),
    "llama-tiny": (128, 2, 4, 4, 352, 512),
    "llama-mid": (1024, 8, 8, 8, 2816, 2048),          # head_dim 128 like the 7B / 13B shapes; tests of accumulated drift
}


def build_causal_lm(shape: str = "llama-2-7b", device="cuda", dtype=torch.bfloat16, seed: int = 0,
                    num_layers: Optional[int] = None):
    """A LLaMA-architecture model built from its config only (no checkpoint is available offline):
    weights ~ N(0, 0.02) from `seed`, eager attention like the reference driver (wikitext.py:60-65)."""
    from transformers import LlamaConfig, LlamaForCausalLM
    h, L, nh, nkv, ffn, vocab = LLAMA_SHAPES[shape]
    cfg = LlamaConfig(hidden_size=h, num_hidden_layers=num_layers or L, num_attention_heads=nh,
                      num_key_value_heads=nkv, intermediate_size=ffn, vocab_size=vocab,
                      max_position_embeddings=4096, rms_norm_eps=1e-5, tie_word_embeddings=False,
                      attn_implementation="eager")
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with torch.device(device):
            model = LlamaForCausalLM(cfg)
    finally:
        torch.set_default_dtype(prev)
    gen = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, 0.02, generator=gen)
    model.eval()
    return model


# ---- H2: SQuAD-style evaluation (run_qa_no_trainer.py:914-959) ---------------------------------------
class GraphedBatch:
    """Forward of one fixed-shape evaluation batch captured into a hipGraph (the SQuAD loop: 674 batches of [16, 384]); a
    BERT-base E4M3 batch is 9.8 ms launched eagerly and 3.2 ms replayed.  Needs frozen or stateless observers only in the
    sense every capture does -- no host reads -- which holds for this engine's fake-quantizers.  `batch_weight_passes`: the FP8 weight
    passes of the Linears on the weight-pass + library-GEMM route (stateless formats; logged during a warm-up forward) run as ONE
    launch in front of the captured forward (fused.BatchedWeightCodes: 36 launches -> 1 for BERT-base) -- same codes, same counts."""

    def __init__(self, model, example, extra=None, batch_weight_passes: bool = True):
        self.model = model
        self.static = {k: v.clone() for k, v in example.items()}
        self.extra = dict(extra or {})
        self.out = None
        device = next(iter(self.static.values())).device
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        from . import fused
        self.weight_codes = None
        with torch.cuda.stream(side), torch.no_grad():
            model(**self.static, **self.extra)
            fused.start_weight_pass_log()                       # which Linears run a separate FP8 weight pass (pair route)
            try:
                model(**self.static, **self.extra)
            finally:
                logged = fused.stop_weight_pass_log()           # (never left active: it holds strong references to the owners)
            batch = fused.BatchedWeightCodes(logged, device)
            if batch_weight_passes and len(batch) > 1:
                self.weight_codes = batch                       # ... all of them as ONE launch in front of the captured forward
                batch.launch()
                model(**self.static, **self.extra)              # (a warm-up of the path the capture will take)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            if self.weight_codes is not None:
                self.weight_codes.launch()
            self.out = model(**self.static, **self.extra)
        if self.weight_codes is not None:
            self.weight_codes.forget()

    def matches(self, batch):
        return batch.keys() == self.static.keys() and all(batch[k].shape == v.shape and batch[k].dtype == v.dtype
                                                         for k, v in self.static.items())

    def replay(self, batch):
        for k, v in batch.items():
            self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        return self.out


@torch.no_grad()
def collect_qa_logits(model, batches, device=None, rank: int = 0, world: int = 1, group=None, graph: bool = False):
    """Runs every batch (dict of tensors with `input_ids`, optional `attention_mask` / `token_type_ids`)
    through a question-answering model exactly as the reference's `run_eval` does -- `model.eval()`,
    `no_grad`, placeholder `start_positions` / `end_positions` of ones, logits collected as fp32 -- with the
    batches sharded round-robin over ranks and the fp32 logits all-gathered back into dataloader order.
    Returns (start_logits [N_features, S], end_logits [N_features, S]) on every rank; the HF
    post-processing / metric step consumes them unchanged.  With `graph=True` (device models) batches of the first
    batch's shape replay one captured forward (GraphedBatch); a ragged last batch runs eagerly.  The two warm-up passes of
    the capture run on the first batch, so with LIVE observers (delayed scaling still calibrating) use graph=False."""
    model.eval()
    device = device if device is not None else next(model.parameters()).device
    mine = shard_round_robin(list(enumerate(batches)), rank, world)
    starts, ends = [], []
    graphed = None
    for _, batch in mine:
        batch = {k: v.to(device) for k, v in batch.items()}
        bsz = batch["input_ids"].size(0)
        pos = torch.ones(bsz, dtype=torch.long, device=device)
        if graph and device.type == "cuda" and graphed is None:
            graphed = GraphedBatch(model, batch, {"start_positions": pos, "end_positions": pos.clone()})
        if graphed is not None and graphed.matches(batch):
            out = graphed.replay(batch)
            starts.append(out.start_logits.float().clone())
            ends.append(out.end_logits.float().clone())
            continue
        out = model(**batch, start_positions=pos, end_positions=pos.clone())
        starts.append(out.start_logits.float())
        ends.append(out.end_logits.float())
    if world == 1 and not _group_ready():
        return torch.cat(starts).cpu(), torch.cat(ends).cpu()
    seq = batches[0]["input_ids"].shape[1]
    sizes = [b["input_ids"].shape[0] for b in batches]
    per_rank_rows = max(sum(sizes[r::world]) for r in range(world))
    local = torch.zeros((2, per_rank_rows, seq), dtype=torch.float32, device=device)
    if starts:
        n = sum(t.shape[0] for t in starts)
        local[0, :n] = torch.cat(starts)
        local[1, :n] = torch.cat(ends)
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local, group=group)                      # one collective: <= 2 x N x S fp32 (~33 MB for SQuAD v1.1)
    out_s = [None] * len(batches)
    out_e = [None] * len(batches)
    for r in range(world):
        off = 0
        for i in range(r, len(batches), world):
            out_s[i] = parts[r][0, off:off + sizes[i]]
            out_e[i] = parts[r][1, off:off + sizes[i]]
            off += sizes[i]
    return torch.cat(out_s).cpu(), torch.cat(out_e).cpu()


# ---- H3: GLUE-style fine-tuning loop (run_glue_no_trainer.py:647-667) ---------------------------------
def train_steps(model, batches, optimizer, lr_scheduler=None, max_grad_norm: float = 1.0,
                gradient_accumulation_steps: int = 1):
    """The reference's inner training loop: forward (hooks fake-quantize activations and weights),
    `loss.backward()` (backward-pre hooks quantize incoming gradients, backward hooks the residual branch),
    clip_grad_norm_(1.0, error_if_nonfinite=True), optimizer / scheduler step.  Returns the per-step losses."""
    model.train()
    device = next(model.parameters()).device
    losses = []
    from . import optim, train_fusions
    for step, batch in enumerate(batches):
        train_fusions.ensure_planned(model)                 # launch fusions over the fake-quantizers that exist by now (values unchanged)
        batch = {k: v.to(device) for k, v in batch.items()}
        loss = model(**batch).loss
        losses.append(float(loss.detach().float()))
        (loss / gradient_accumulation_steps).backward()
        if step % gradient_accumulation_steps == 0 or step == len(batches) - 1:
            # clip_grad_norm_ + optimizer.step(): three launches over the optimizer's own state for torch.optim.AdamW on bf16 device
            # parameters (optim.py), torch's own calls otherwise
            optim.clip_and_step(model.parameters(), optimizer, max_grad_norm, error_if_nonfinite=True)
            if lr_scheduler is not None:
                lr_scheduler.step()
            optimizer.zero_grad()
    return losses


class GraphedTrainStep:
    """One training step -- forward with its fake-quant hooks, `loss.backward()` with the gradient fake-quantizers,
    clip_grad_norm_, optimizer step -- captured into a hipGraph and replayed per batch.  Nothing on this path needs the
    host: amax histories, scales and enable flags live on the device (`scale_update_kernel`), so the delayed-scaling
    state machine advances inside the graph exactly as it does eagerly, and the few hundred Python hook calls of a step
    (the eager step of a RoBERTa-base classifier is launch-bound at ~41 ms) are paid once.

    The optimizer must be built for capture (`torch.optim.AdamW(..., capturable=True)`, lr as a tensor if a scheduler
    changes it); `clip_grad_norm_` runs with `error_if_nonfinite=False` (the check is a host read).  Batches must keep one
    shape.  Warm-up steps run eagerly first (they also create the lazily built fake-quantizers) and DO train."""

    def __init__(self, model, optimizer, lr_scheduler=None, max_grad_norm: float = 1.0, batch_scale_updates: bool = True,
                 batch_weight_passes: bool = True):
        self.model, self.optimizer, self.lr_scheduler, self.max_grad_norm = model, optimizer, lr_scheduler, max_grad_norm
        self.batch_scale_updates = batch_scale_updates
        self.batch_weight_passes = batch_weight_passes
        self.weights = None               # fake_quantize.BatchedWeightFakeQuant over the QAT Linears' weight fake-quantizers
        self.graph = None
        self.static = None
        self.loss = None
        self.scales = None                # fake_quantize.BatchedScaleUpdate over the fake-quantizers a step calls

    def _step(self, batch):
        from . import optim
        if self.scales is not None:
            self.scales.launch()          # every delayed-scaling update of the step in one launch (356 -> 1 for RoBERTa-base)
        if self.weights is not None:
            self.weights.launch()         # every weight fake-quantizer of the step in one launch per format (74 -> 1): the weights
                                          # only change in optimizer.step() below
        loss = self.model(**batch).loss
        loss.backward()
        optim.clip_and_step(self.model.parameters(), self.optimizer, self.max_grad_norm, error_if_nonfinite=False)
        return loss.detach().float()

    def capture(self, example_batch, warmup: int = 3):
        """Runs `warmup` eager steps on `example_batch` (on a side stream, as stream capture requires), then captures."""
        self.model.train()
        device = next(self.model.parameters()).device
        self.static = {k: v.to(device).clone() for k, v in example_batch.items()}
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        from .fake_quantize import BatchedScaleUpdate, FusedAmaxObsFakeQuantize
        fqs = [mod for mod in self.model.modules() if isinstance(mod, FusedAmaxObsFakeQuantize)]
        with torch.cuda.stream(side):
            for i in range(max(warmup, 1)):
                for f in fqs:
                    f.__dict__["_qt_calls"] = 0
                self.optimizer.zero_grad(set_to_none=True)
                self._step(self.static)
                if self.lr_scheduler is not None:
                    self.lr_scheduler.step()
                if i == 0:                 # the first step created the lazily built fake-quantizers
                    fqs = [mod for mod in self.model.modules() if isinstance(mod, FusedAmaxObsFakeQuantize)]
                    from . import train_fusions
                    train_fusions.ensure_planned(self.model)      # chains of fake-quantizer calls as single launches (train_fusions.py)
            if self.batch_scale_updates:
                # those a step calls (at least once: a second call in the same step does its own update as before)
                self.scales = BatchedScaleUpdate([f for f in fqs if f.__dict__.get("_qt_calls", 0) >= 1], device)
            if self.batch_weight_passes and self.scales is not None:
                from .fake_quantize import BatchedWeightFakeQuant
                from .modules.qat.linear import Linear as QATLinear
                # weight fake-quantizers a step calls exactly once (a shared / tied Linear called twice keeps its own passes)
                pairs = [(m.weight_fake_quant, m.weight) for m in self.model.modules()
                         if isinstance(m, QATLinear) and m.weight_fake_quant.__dict__.get("_qt_calls", 0) == 1]
                batch = BatchedWeightFakeQuant(pairs, device)
                self.weights = batch if len(batch) else None
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.optimizer.zero_grad(set_to_none=True)          # gradients are (re)allocated inside the graph's pool
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = self._step(self.static)
        if self.scales is not None:
            self.scales.forget()          # nothing may stay marked "already updated" outside the captured step
        if self.weights is not None:
            self.weights.forget()
        return warmup

    def replay(self, batch):
        """Copies `batch` into the captured input buffers, replays the step, advances the scheduler; returns the loss
        tensor (device, overwritten by the next replay)."""
        for k, v in batch.items():
            self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        return self.loss


# ---- calibration flow and checkpoints (upstream examples/question_answering/run_qa_no_trainer.py:826-847, 961-990) ----
def calibrate(model, batches, steps: int, device=None):
    """Run `steps` forward passes in eval / no_grad so the delayed-scaling observers fill their amax histories, then
    freeze every observer (the reference's `calibrate()` + `disable_observer()` loop, :834-847).  `batches` yields
    dicts of tensors (a HF dataloader) or plain input-id tensors.  Returns the number of batches consumed."""
    was_training = model.training
    model.eval()
    n = 0
    with torch.no_grad():
        for batch in batches:
            if n >= steps:
                break
            if isinstance(batch, dict):
                model(**{k: (v.to(device) if device is not None else v) for k, v in batch.items()})
            else:
                model(batch.to(device) if device is not None else batch)
            n += 1
    freeze_observers(model)
    model.train(was_training)
    return n


def freeze_observers(model):
    """`module.disable_observer()` on every fake-quantizer: scales stay at their calibrated values."""
    for m in model.modules():
        if isinstance(m, torch.ao.quantization.FakeQuantizeBase):
            m.disable_observer()


def save_checkpoint(path, model, optimizer=None, lr_scheduler=None, best_metric=None, run_id=None):
    """`checkpoint.tar` with the reference's keys (:975-981), so either side can resume the other's runs."""
    torch.save({"model_state_dict": model.state_dict(),
                "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else None,
                "scheduler_state_dict": lr_scheduler.state_dict() if lr_scheduler is not None else None,
                "best_metric": best_metric, "run_id": run_id}, path)


def load_checkpoint(path, model, optimizer=None, lr_scheduler=None, map_location=None):
    """Counterpart of the reference's `load_state` (:985-990).  Lazily sized fake-quant buffers (`amax_history`,
    `scale`) take the checkpoint's shapes (fake_quantize.py:406-435); per-argument fake-quantizers that only exist
    after a first forward must have been created (run one batch first, as the reference does at :826-832)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None and ckpt.get("optimizer_state_dict") is not None:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if lr_scheduler is not None and ckpt.get("scheduler_state_dict") is not None:
        lr_scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    return ckpt


def cache_quantized_weights(enable: bool = True):
    """Opt-in eval optimisation: keep fq(W) of every QAT Linear while its inputs cannot change (see fused.py)."""
    from . import fused
    fused.cache_quantized_weights(enable)
