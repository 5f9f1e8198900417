"""The end of a fine-tuning step -- `clip_grad_norm_(max_norm)` + `optimizer.step()` -- on the in-tree kernels (csrc/qt_optimizer.hip).

Upstream (examples/text_classification/run_glue_no_trainer.py:469-474, 655-668) builds `torch.optim.AdamW` over two parameter groups
and calls `accelerator.clip_grad_norm_(model.parameters(), 1.0)`, `optimizer.step()`, `lr_scheduler.step()`, `optimizer.zero_grad()`.
On a RoBERTa-base classifier torch spends 33 launches and 0.95 ms on the first two (one multi-tensor launch per ~25 tensors for the
norms, the scaling and the update; ten scalar kernels for the coefficient).  `clip_and_step` does the same arithmetic in four launches
over the optimizer's OWN state: the `torch.optim.AdamW` object stays what the caller built -- its `state` / `state_dict()`, parameter
groups, learning-rate schedulers and checkpoints keep working -- only the launches differ.

Arithmetic: torch's (see oracle/optimizer_oracle.py for the restatement).  The clip follows torch.nn.utils.clip_grad_norm_ on bf16
gradients rounding for rounding; the update follows torch's FUSED AdamW kernel (fp32 state arithmetic, one rounding per stored value),
whichever of torch's three implementations (for-loop / foreach / fused) the optimizer object was built for -- they differ from each other
in the last bf16 bit, and the fused one is the most accurate.  Not covered (the call then runs torch's own clip + step): anything but
`torch.optim.AdamW` itself, amsgrad, maximize, differentiable, non-bf16 / non-contiguous / sparse tensors, a grad scaler.
QT_TRAIN_DEBUG bit 256 switches the kernels off.
"""
import ctypes

import numpy as np
import torch

from . import _native

# what ran last (bench.py's config.routes, tests)
ROUTES = {}

_TENSOR_DTYPE = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("step_dev", "<u8"), ("lr_dev", "<u8"),
                          ("numel", "<i8"), ("first_chunk", "<i8"), ("lr", "<f8"), ("beta1", "<f8"), ("beta2", "<f8"), ("eps", "<f8"),
                          ("weight_decay", "<f8"), ("step", "<f8")])
assert _TENSOR_DTYPE.itemsize == ctypes.sizeof(_native.QtAdamwTensor)

_PLANS = {}          # id(optimizer) -> _Plan (dropped with the optimizer: weakref.finalize)
_FROZEN = {}         # id(optimizer) -> plans a captured graph reads (kept alive as long as the optimizer)


def _enabled():
    from . import train_fusions
    return train_fusions._on("optimizer")


def _why_not(optimizer, entries):
    """None when the in-tree kernels cover this optimizer and these tensors, else the reason (a string)."""
    if type(optimizer) is not torch.optim.AdamW:
        return f"optimizer is {type(optimizer).__name__}, not torch.optim.AdamW"
    for group in optimizer.param_groups:
        if group.get("amsgrad") or group.get("maximize") or group.get("differentiable"):
            return "amsgrad / maximize / differentiable"
        if group.get("decoupled_weight_decay") is False:
            return "AdamW without decoupled weight decay"
        if isinstance(group["lr"], torch.Tensor) and (group["lr"].dtype != torch.float32 or group["lr"].numel() != 1):
            return "tensor lr that is not one fp32 value"
        if any(isinstance(b, torch.Tensor) for b in group["betas"]):
            return "tensor betas"
    if not entries:
        return "no parameter holds a gradient"
    dev = entries[0][0].device
    if dev.type != "cuda":
        return "parameters are not on the device"
    for p, _ in entries:
        g = p.grad
        if p.dtype != torch.bfloat16 or g.dtype != torch.bfloat16:
            return "a parameter or gradient is not bf16"
        if p.device != dev or g.device != dev:
            return "parameters on several devices"
        if g.is_sparse or not p.is_contiguous() or not g.is_contiguous():
            return "a sparse or non-contiguous tensor"
    return None


def _init_state(optimizer, p, group):
    """torch.optim.Adam._init_group's lazy state initialisation (torch/optim/adam.py), for a parameter the optimizer has not stepped yet"""
    state = optimizer.state[p]
    if len(state) == 0:
        on_device = bool(group.get("capturable") or group.get("fused"))
        state["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if on_device else torch.tensor(0.0, dtype=torch.float32)
        state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
    return state


class _Plan:
    """Device copies of the tensor table and the chunk map of one optimizer, rebuilt when a pointer or a hyper-parameter changes."""

    def __init__(self):
        self.key = None
        self.table_host = None       # pinned: a captured graph's copy node reads it at every replay
        self.table_dev = None
        self.chunks_dev = None
        self.ws = None
        self.total_norm = None
        self.ntensors = 0
        self.nchunks = 0
        self.frozen = False          # built inside a stream capture: the graph's copy node reads table_host at every replay -- never rewritten

    def build(self, rows, dev):
        n = len(rows)
        table = np.zeros(n, dtype=_TENSOR_DTYPE)
        for i, r in enumerate(rows):
            table[i] = r
        L = _native.lib()
        arr = table.view(np.uint8)
        ptr = arr.ctypes.data_as(ctypes.c_void_p)
        nchunks = L.qt_clip_adamw_plan(ptr, n, None, 0)
        if nchunks < 0:
            raise RuntimeError(f"qt_clip_adamw_plan failed ({nchunks})")
        resize = self.table_host is None or self.ntensors != n or self.nchunks != nchunks
        if resize:
            chunk_map = np.zeros(max(nchunks, 1), dtype=np.int32)
            L.qt_clip_adamw_plan(ptr, n, chunk_map.ctypes.data_as(ctypes.c_void_p), nchunks)
            self.table_host = torch.empty(arr.size, dtype=torch.uint8).pin_memory()
            self.table_dev = torch.empty(arr.size, dtype=torch.uint8, device=dev)
            self.chunks_dev = torch.from_numpy(chunk_map).to(dev)
            self.ws = torch.zeros(max(int(L.qt_clip_adamw_ws_bytes(n, nchunks)), 64), dtype=torch.uint8, device=dev)
            self.total_norm = torch.zeros((), dtype=torch.float32, device=dev)
            self.ntensors, self.nchunks = n, int(nchunks)
        if torch.cuda.is_current_stream_capturing():
            # pinned source + asynchronous copy: legal inside a stream capture, and the copy node reads the buffer at every replay -- the
            # plan is frozen from here on (clip_and_step), nothing rewrites it
            self.table_host.copy_(torch.from_numpy(arr.copy()))
            self.table_dev.copy_(self.table_host, non_blocking=True)
        else:
            # eager: a copy from pageable memory returns when the source has been read -- an asynchronous copy from the one pinned buffer
            # could still be queued when the NEXT step (new gradient addresses) rewrites that buffer
            self.table_dev.copy_(torch.from_numpy(arr.copy()))


def _plan_for(optimizer):
    plan = _PLANS.get(id(optimizer))
    if plan is None:
        import weakref
        plan = _PLANS[id(optimizer)] = _Plan()
        weakref.finalize(optimizer, _PLANS.pop, id(optimizer), None)
        weakref.finalize(optimizer, _FROZEN.pop, id(optimizer), None)
    return plan


def clip_and_step(parameters, optimizer, max_norm=None, error_if_nonfinite=False):
    """`torch.nn.utils.clip_grad_norm_(parameters, max_norm, error_if_nonfinite=...)` (skipped when max_norm is None) followed by
    `optimizer.step()`.  Returns the total norm (a 0-dim tensor, as clip_grad_norm_ does; None without clipping)."""
    parameters = list(parameters) if not isinstance(parameters, torch.Tensor) else [parameters]
    clipped = {id(p) for p in parameters if p.grad is not None}
    entries = [(p, group) for group in optimizer.param_groups for p in group["params"] if p.grad is not None]
    why = None if _enabled() else "QT_TRAIN_DEBUG bit 256"
    if why is None:
        why = _why_not(optimizer, entries)
    if why is None and max_norm is not None and clipped != {id(p) for p, _ in entries}:
        why = "the clipped parameters are not the optimizer's"           # (the norm would cover another set than the update)
    if why is None and torch.cuda.is_current_stream_capturing() and not all(g.get("capturable") for g in optimizer.param_groups):
        why = "stream capture with an optimizer that is not capturable"   # (its step counts live on the host)
    if why is not None:
        ROUTES["train:clip + optimizer"] = f"torch ({why})"
        total = None
        if max_norm is not None:
            total = torch.nn.utils.clip_grad_norm_(parameters, max_norm, error_if_nonfinite=error_if_nonfinite)
        optimizer.step()
        return total

    dev = entries[0][0].device
    rows, key = [], []
    for p, group in entries:
        state = _init_state(optimizer, p, group)
        step_t, lr = state["step"], group["lr"]
        if step_t.device.type == "cuda":
            step_dev, step = step_t.data_ptr(), 0.0
        else:                                          # host step count: advanced here, as torch's for-loop / foreach paths do
            step_t += 1
            step_dev, step = 0, float(step_t)
        lr_dev = lr.data_ptr() if isinstance(lr, torch.Tensor) and lr.device.type == "cuda" else 0
        b1, b2 = group["betas"]
        row = (p.data_ptr(), p.grad.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(), step_dev, lr_dev, p.numel(), 0,
               0.0 if lr_dev else float(lr), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step)
        if state["exp_avg"].dtype != torch.bfloat16 or state["exp_avg_sq"].dtype != torch.bfloat16 or not state["exp_avg"].is_contiguous() \
                or not state["exp_avg_sq"].is_contiguous():
            raise RuntimeError("quantized_training.optim: AdamW state of a bf16 parameter is not contiguous bf16 (was it loaded from another dtype?)")
        rows.append(row)
    key = tuple(rows)
    plan = _plan_for(optimizer)
    capturing = torch.cuda.is_current_stream_capturing()
    if plan.key != key:
        if plan.frozen:                                        # a captured graph replays that plan's table: leave it, size a new one like it
            _FROZEN.setdefault(id(optimizer), []).append(plan)
            if capturing:
                raise RuntimeError("quantized_training.optim: a second stream capture of one optimizer with other tensors; run one eager step "
                                   "with them first (the capture reuses the buffers that step sized)")
            plan = _PLANS[id(optimizer)] = _Plan()
        plan.build(rows, dev)
        plan.key = key
    if capturing:
        plan.frozen = True
    L = _native.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    norm = float(max_norm) if max_norm is not None else 0.0

    def launch(phases):
        _native.check(L.qt_clip_adamw_bf16(plan.table_dev.data_ptr(), plan.chunks_dev.data_ptr(), plan.ntensors, plan.nchunks, norm,
                                           plan.total_norm.data_ptr(), plan.ws.data_ptr(), plan.ws.numel(), phases, st), "qt_clip_adamw_bf16")

    if max_norm is not None and error_if_nonfinite:
        launch(1)
        if not bool(torch.isfinite(plan.total_norm)):           # (host read, as torch's own check)
            for p, group in entries:                           # nothing was updated: take the host step counts back
                if optimizer.state[p]["step"].device.type != "cuda":
                    optimizer.state[p]["step"] -= 1
            plan.key = None
            raise RuntimeError("The total norm of order 2.0 for gradients from `parameters` is non-finite, so it cannot be clipped. To disable "
                               "this error and scale the gradients by the non-finite norm anyway, set `error_if_nonfinite=False`")
        launch(2)
    else:
        launch(3)
    optimizer._opt_called = True                                # what a patched optimizer.step() tells torch's LR schedulers
    ROUTES["train:clip + optimizer"] = f"in_tree_clip_adamw, 4 launches ({plan.ntensors} tensors, {plan.nchunks} chunks)"
    return plan.total_norm.to(torch.bfloat16) if max_norm is not None else None
