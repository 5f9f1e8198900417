#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fake-quant hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: the forward of ONE WikiText-style window
(B=1, S=1024) through a LLaMA-2-7B-shaped model with E4M3 fake-quant on every GEMM input and
weight (`--activation e4m3 --weight e4m3 --bf16 --quantize_forward gemm`, the reference's
"full fusion" column), weights re-quantized every forward exactly as the reference does.
Metric (BASELINE.json): quantized elements / s = sum over fake-quant applications of numel(input)
/ wall time.  N > 1: one process per GPU, windows sharded round-robin, no data-path collective;
value = units of all ranks / max-over-ranks time (weak scaling).

Synthetic data: no checkpoint or dataset exists offline -> weights ~ N(0, 0.02) from a fixed seed,
token ids uniform random.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line (rank 0) with `roofline` (dominant elementwise pass, HIP events on the launch
stream, rotating over a > 256 MiB pool so HBM rather than Infinity-Cache traffic is timed) and
`cpu_baseline` (the C restatement of the reference path on this host's cores, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "quantized-training_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP8_PEAK_TFLOPS = 5000.0        # MI355X_MICROARCH.md: FP8 MFMA ~5 PF dense (sparse figures are never used)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


# BASELINE.json `configs` the driver can time: name -> (kind, model shape, activation spec, weight spec)
WORKLOADS = {
    "llama-7b-e4m3": ("llama", "llama-2-7b", "e4m3", "e4m3"),
    "llama-13b-posit8_2": ("llama", "llama-2-13b", "posit8_2", "posit8_2"),
    "bert-base-squad-e4m3": ("bert", "bert-base", "e4m3", "e4m3"),
    "roberta-mrpc-int8-e5m2-train": ("roberta", "roberta-base", "int8,qs=per_tensor_symmetric", "int8,qs=per_tensor_symmetric"),
    # the same step with HF's default dropout (0.1 on the hidden states and the attention probabilities), as a fine-tune from a checkpoint runs
    "roberta-mrpc-int8-e5m2-train-dropout": ("roberta", "roberta-base", "int8,qs=per_tensor_symmetric", "int8,qs=per_tensor_symmetric"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="llama-7b-e4m3", choices=sorted(WORKLOADS),
                    help="BASELINE.json configs: llama-7b-e4m3 = configs[2] (the headline, default), bert-base-squad-e4m3 = configs[1], "
                         "llama-13b-posit8_2 = configs[3] on the GPUs given, roberta-mrpc-int8-e5m2-train = configs[4] (dropout 0; -dropout: HF's default 0.1)")
    ap.add_argument("--model", default=None, help="(llama workloads) override the model shape")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (the line is then marked invalid)")
    ap.add_argument("--max_length", type=int, default=1024)
    ap.add_argument("--stride", type=int, default=512)
    ap.add_argument("--activation", default=None)
    ap.add_argument("--weight", default=None)
    ap.add_argument("--route", default="eager", choices=["eager", "pt2e"],
                    help="(llama workloads) eager: quantize() hooks + QAT modules; pt2e: the reference's current wikitext.py flow -- "
                         "torch.export + prepare_pt2e, chains rewritten to the fused kernels (pt2e_fusion)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--cache-eval-weights", action="store_true",
                    help="side experiment (line marked invalid): keep fq(W) across windows instead of re-quantizing, "
                         "which the reference does not do")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: initialise torch.distributed (nccl) and take the multi-rank code path even with one rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the three short secondary legs (the other BASELINE configs that fit one GPU) appended to the default line")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check without a GPU: tiny LLaMA on the CPU, gloo instead of RCCL, same sharding / timing / "
                         "gather code (the line is marked invalid)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py launches its own ranks")
    a = ap.parse_args()
    kind, model, act, wgt = WORKLOADS[a.workload]
    if a.model is not None and a.model != model and a.workload == "llama-7b-e4m3":
        # the pre-workload spelling: --model llama-2-13b --activation posit8_2 --weight posit8_2
        pass
    a.kind = kind
    a.model = a.model or model
    a.activation = a.activation or act
    a.weight = a.weight or wgt
    return a


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process (nothing
    in this process has touched the GPU yet -- a process that has must never be replaced by another program) and pass
    its exit code on.  Rank 0 of the child job prints the JSON line."""
    import socket
    import subprocess
    port = a.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


SECONDARY = ("bert-base-squad-e4m3", "llama-13b-posit8_2", "roberta-mrpc-int8-e5m2-train", "roberta-mrpc-int8-e5m2-train-dropout")


def secondary_legs(budget_s=190.0):
    """The other BASELINE.json configs that fit one GPU, three timed steps each, AFTER the headline leg (which alone is `value`): each
    one runs as a child process of its own (`bench.py --workload ...`: model built, timed, freed), so the driver's default command
    times them too.  A leg that fails or runs out of budget is reported as such, never fatal."""
    import subprocess
    out = {}
    t_all = time.perf_counter()
    for w in SECONDARY:
        left = budget_s - (time.perf_counter() - t_all)
        if left < 20.0:
            out[w] = {"error": "skipped: time budget of the secondary legs spent"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, "--steps", "3", "--warmup", "1", "--no-roofline", "--no-cpu-baseline"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=min(left, 90.0))
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                out[w] = {"error": f"rc {r.returncode}: {r.stderr.strip()[-200:]}"}
                continue
            j = json.loads(line[-1])
            out[w] = {"ms_per_step": round(j["ms_per_step"], 4), "steps": j["steps"], "elements_per_step": j["config"]["elements_per_step"],
                      "value": j["value"], "unit": j["unit"], "launch": j["config"]["launch"], "routes": j["config"].get("routes", {}),
                      "workload": j["config"]["workload"]}
            if "mean_window_nll" in j:
                out[w]["mean_window_nll"] = j["mean_window_nll"]
        except subprocess.TimeoutExpired:
            out[w] = {"error": "timeout"}
        except Exception as e:  # noqa: BLE001
            out[w] = {"error": f"{type(e).__name__}: {e}"}
    return out


def profiled_traffic(key):
    """HBM bytes per launch of the roofline kernels from the PMC passes kept under profiles/ (rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE in separate passes, FETCH_SIZE x 2 on gfx950 as the microarchitecture guide prescribes).  Collected by
    tools/gpu_session_prof.sh on the same launch, NOT in this run: the JSON names the file next to the number."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                node = json.load(open(path))
                for part in key:
                    node = node[part]
                return node, "profiles/" + name
            except Exception:  # noqa: BLE001
                continue
    return None, None


def roofline_leg(device, weight_dtype="e4m3", model_shape=(4096, 11008)):
    """The fake-quant pass over a weight of the model (bf16 [hidden, intermediate]; 4096 x 11008 for LLaMA-2-7B) -- the dominant
    elementwise kernel of the window, HBM-bound.  Since round 2 the q / k / v, gate and up projections run the weight's
    fake-quantizer inside their GEMM ("fused_gemm" below, the window's largest kernel by time, MFMA / L2-bound); the separate pass
    still serves the o and down projections and every spec whose values are not FP8 codes.
      * e4m3 / e5m2 without `qs` (the FP8 GEMM route): `fq8_kernel` with FP8-only output; algorithmic bytes 3 B/element
        (2 read as bf16 + 1 written as the FP8 code) -- less than SURVEY 8(d)'s 4 B/element because the bf16 copy of the
        quantized weight is never needed.  The bf16 -> bf16 pass (4 B/element) is timed next to it ("bf16_out").
      * every other dtype (posit(8,2) of BASELINE configs[3], ...): the bf16 -> bf16 pass with the value map staged in LDS,
        4 B/element."""
    from quantized_training import _native as nv
    import quantized_training as qt
    L = nv.lib()
    rows, cols, pool = model_shape[0], model_shape[1], 8    # 7B: 8 x 90 MB in + 8 x 90 MB out = 1.44 GB, beyond the Infinity Cache
    n = rows * cols
    x = torch.empty(pool, rows, cols, device=device, dtype=torch.bfloat16).normal_(0.0, 0.02)
    y = torch.empty_like(x)
    base = weight_dtype.split(",")[0]
    fmt = nv.format_for(base)
    lut = qt.get_quantization_map(base, device)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    fp8_route = base in ("e4m3", "e5m2") and "qs=" not in weight_dtype
    ms, ms8 = ctypes.c_float(0.0), ctypes.c_float(0.0)
    y8 = torch.empty(pool, rows, cols, device=device, dtype=torch.uint8) if fp8_route else None
    for iters in (pool, 5 * pool):                           # warm-up pass, then the timed region
        nv.check(L.qt_bench_fake_quant_bf16(x.data_ptr(), y.data_ptr(), n, ctypes.byref(fmt), lut.data_ptr(),
                                            None, None, iters, n, pool, st, ctypes.byref(ms)), "bench")
        if fp8_route:
            nv.check(L.qt_bench_fake_quant_bf16_fp8(x.data_ptr(), None, y8.data_ptr(), n, ctypes.byref(fmt), None, None,
                                                    iters, n, pool, st, ctypes.byref(ms8)), "bench fp8")
    achieved = n * 4 / (ms.value * 1e-3) / 1e9
    shape = f"{rows}x{cols}"
    del x, y, y8
    torch.cuda.empty_cache()
    is_7b = (rows, cols) == (4096, 11008)
    t4, src4 = profiled_traffic(("hbm_bytes_per_launch",)) if (is_7b and base == "e4m3") else (None, None)
    bf16_out = {"kernel": f"fq_kernel<bf16> {base} {shape} ({'closed form' if fmt.kind != nv.QT_FMT_LUT else 'value map in LDS'})",
                "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBPS, 4), "ms_per_launch": round(ms.value, 5),
                "algorithmic_bytes_per_launch": n * 4, "traffic": t4, "traffic_source": src4}
    if not fp8_route:
        out = {"bound": "hbm", "achieved": bf16_out["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": bf16_out["frac"],
               "traffic": t4, "traffic_source": src4, "kernel": bf16_out["kernel"] + " (weight pass, bf16 GEMM route)",
               "ms_per_launch": bf16_out["ms_per_launch"], "algorithmic_bytes_per_launch": n * 4}
        return out
    achieved8 = n * 3 / (ms8.value * 1e-3) / 1e9
    t3, src3 = profiled_traffic(("fp8_only", "hbm_bytes_per_launch")) if (is_7b and base == "e4m3") else (None, None)
    pass_leg = {"bound": "hbm", "achieved": round(achieved8, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved8 / HBM_PEAK_GBPS, 4), "traffic": t3, "traffic_source": src3,
                "kernel": f"fq8_kernel<obs off, fp8 only> {base} {shape} (weight pass of the FP8 GEMM route)",
                "ms_per_launch": round(ms8.value, 5), "algorithmic_bytes_per_launch": n * 3, "bf16_out": bf16_out}
    fused = fused_gemm_leg(device) if is_7b else None
    lib = gemm_leg(device)
    if fused is None:
        pass_leg["gemm"] = lib
        return pass_leg
    # Since round 2 every Linear of the headline window runs the weight fake-quantizer inside its GEMM (QT_FQ8_GEMM=auto picked the
    # fused kernel for all four shapes), so the window's dominant kernel is that GEMM; the separate weight pass (the dominant kernel of
    # round 1, still what every other spec and the pair route run) and the library GEMM it fed are reported beside it.
    tf_, srcf = profiled_traffic(("fused_gemm", "hbm_bytes_per_launch"))
    fused["traffic"], fused["traffic_source"] = tf_, srcf
    # The matrix core's share of the launch by HARDWARE COUNTER (north_star: "rocprof showing ... MFMA utilisation for the GEMM"):
    # SQ_VALU_MFMA_BUSY_CYCLES per SIMD (a property of the kernel: instructions x 32 cycles, measured by rocprofv3 --pmc) and the clock the
    # chip holds inside this kernel on random data (in-kernel stamps), both from profiles/r06_linear_fq8_pmc.json, against THIS run's
    # launch duration.  `frac` above prices flops / time against 5 PF at the nominal 2.4 GHz; at the clock the part actually holds the
    # peak is lower by clock / 2400, which `frac_at_held_clock` applies.
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_linear_fq8_pmc.json")))
        busy_us = pmc["mfma_busy_cycles_per_simd"] / pmc["in_kernel_clock_mhz_random"]
        fused["mfma_busy"] = {"share_of_launch": round(busy_us / (fused["ms_per_launch"] * 1e3), 4), "busy_us": round(busy_us, 2),
                              "busy_cycles_per_simd": pmc["mfma_busy_cycles_per_simd"], "in_kernel_clock_mhz": pmc["in_kernel_clock_mhz_random"],
                              "in_kernel_clock_mhz_on_zero_operands": pmc["in_kernel_clock_mhz_zeros"],
                              "frac_at_held_clock": round(fused["frac"] * 2400.0 / pmc["in_kernel_clock_mhz_random"], 4),
                              "source": "profiles/r06_linear_fq8_pmc.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES; s_memtime / s_memrealtime stamps)"}
    except Exception:  # noqa: BLE001
        fused["mfma_busy"] = None
    fused["elementwise_pass"] = pass_leg
    fused["library_gemm"] = lib
    return fused


def gemm_leg(device):
    """MFMA side of the window: the FP8 GEMM that consumes the weight pass's codes, at the gate/up projection's
    shape (tokens 1024 x out 11008 x in 4096; one of the 225 linear GEMMs of a window, 2 M N K = 92.3 GFLOP), through
    qt_fp8_gemm (hipBLASLt, algorithm chosen by measurement), timed with events on the launch stream while rotating
    over 8 weight tensors (8 x 45 MB, more than the Infinity Cache).  Peak: the guide's dense FP8 figure (~5 PFLOP/s)."""
    from quantized_training.fused import lt_fp8_gemm
    M, N, K, pool = 1024, 11008, 4096, 8
    a8 = torch.randn(M, K, device=device).to(torch.float8_e4m3fn)
    b8 = (torch.randn(pool, N, K, device=device) * 0.02).to(torch.float8_e4m3fn)
    if lt_fp8_gemm(a8, b8[0]) is None:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for iters in (pool, 5 * pool):
        e0.record()
        for i in range(iters):
            lt_fp8_gemm(a8, b8[i % pool])
        e1.record()
        e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    del a8, b8
    torch.cuda.empty_cache()
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": FP8_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / FP8_PEAK_TFLOPS, 4), "ms_per_launch": round(ms, 5),
            "kernel": "FP8 E4M3 GEMM 1024x11008x4096 (gate/up projection), fp32 accumulate, bf16 out",
            "flops_per_launch": 2 * M * N * K}


def fused_gemm_leg(device):
    """The kernel most of the window's Linears run on (QT_FQ8_GEMM=auto picks it for the q / k / v, gate and up projections):
    qt_linear_fq8_bf16 -- FP8 GEMM with the weight's E4M3 fake-quantizer in its operand path -- at 1024 x 11008 x 4096, rotating
    over 8 bf16 weights (8 x 90 MB, more than the Infinity Cache), HIP events on the launch stream.  Two views of the same launch:
    matrix-core rate (2 M N K flops against the dense FP8 peak) and the bytes it has to move at least (weights 2 B/element once,
    FP8 activations once, bf16 output once) against the HBM peak."""
    import ctypes
    from quantized_training import _native
    L = _native.lib()
    M, N, K, pool = 1024, 11008, 4096, 8
    x8 = torch.randn(M, K, device=device).to(torch.float8_e4m3fn)
    w = (torch.randn(pool, N, K, device=device) * 0.02).bfloat16()
    y = torch.empty(M, N, dtype=torch.bfloat16, device=device)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def run(i):
        wp = (ctypes.c_void_p * 1)(w[i % pool].data_ptr())
        return L.qt_linear_fq8_bf16(x8.data_ptr(), 0, wp, None, (ctypes.c_int * 1)(N), 1, 0, y.data_ptr(), M, K, st)
    if run(0) != 0:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(pool):                                     # warm-up
        run(i)
    samples = []
    for _ in range(3):                                        # three timed repeats of 5 x pool launches: the median is reported
        e0.record()
        for i in range(5 * pool):
            run(i)
        e1.record()
        e1.synchronize()
        samples.append(e0.elapsed_time(e1) / (5 * pool))
    ms = sorted(samples)[1]
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    nbytes = N * K * 2 + M * K + M * N * 2
    gbs = nbytes / (ms * 1e-3) / 1e9
    del x8, w, y
    torch.cuda.empty_cache()
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": FP8_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP8_PEAK_TFLOPS, 4),
            "ms_per_launch": round(ms, 5), "ms_per_launch_min": round(min(samples), 5), "ms_per_launch_max": round(max(samples), 5),
            "timing": "median of 3 x 40 launches, HIP events on the launch stream", "flops_per_launch": 2 * M * N * K,
            "kernel": "linear_fq8r_kernel: FP8 E4M3 GEMM 1024x11008x4096 with the bf16 weight fake-quantized in its operand path",
            "hbm_view": {"algorithmic_bytes_per_launch": nbytes, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBPS, 4)}}


def linears_in_window_leg(model, device, M):
    """Every Linear launch of one headline window -- 32 x (q / k / v as one launch, o, gate + up + SiLU * up as one launch, down) and the lm
    head -- on the model's own weights in the window's order, back to back in one hipGraph WITHOUT the kernels between them: the
    flop-weighted matrix-core fraction of the window's GEMMs (the single-shape `roofline.frac` is its widest member alone)."""
    from quantized_training import fused
    try:
        layers = model.model.layers
        K = model.config.hidden_size
        I = model.config.intermediate_size
        x8 = torch.randn(M, K, device=device).to(torch.float8_e4m3fn)
        h8 = torch.randn(M, I, device=device).to(torch.float8_e4m3fn)
        flops = 0

        def forward():
            nonlocal flops
            flops = 0
            for l in layers:
                a, m = l.self_attn, l.mlp
                ok = fused.hip_fq8_linear_or_none(x8, [a.q_proj, a.k_proj, a.v_proj]) is not None
                ok = ok and fused.hip_fq8_linear_or_none(x8, [a.o_proj]) is not None
                ok = ok and fused.hip_mlp_fq8_or_none(x8, m.gate_proj, m.up_proj, m.down_proj.activation_pre_process["0"], codes_only=True) is not None
                ok = ok and fused.hip_fq8_linear_or_none(h8, [m.down_proj]) is not None
                if not ok:
                    return False
                flops += 2 * M * K * (a.q_proj.weight.shape[0] + a.k_proj.weight.shape[0] + a.v_proj.weight.shape[0] + K + 3 * I)
            if fused.hip_fq8_linear_or_none(x8, [model.lm_head]) is None:
                return False
            flops += 2 * M * K * model.lm_head.weight.shape[0]
            return True
        with torch.no_grad():
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                if not forward():
                    return None
            torch.cuda.current_stream(device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                forward()
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            samples = []
            for _ in range(3):
                e0.record()
                g.replay()
                e1.record()
                e1.synchronize()
                samples.append(e0.elapsed_time(e1))
        ms = sorted(samples)[1]
        tf = flops / (ms * 1e-3) / 1e12
        return {"what": "all Linear launches of one window back to back (model weights, window order, one hipGraph, no other kernels)",
                "launches": 4 * len(layers) + 1, "flops": flops, "ms": round(ms, 4), "ms_min": round(min(samples), 4), "ms_max": round(max(samples), 4),
                "achieved": round(tf, 1), "peak": FP8_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP8_PEAK_TFLOPS, 4)}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def native_library_note():
    """{} for the product library; names the file when QT_HIP_LIB made the package load another one (tools/ only)."""
    from quantized_training import _native
    return {"native_library_override": _native.LIB_PATH} if _native.LIB_OVERRIDDEN else {}


def fused_routes():
    from quantized_training import fused
    return fused.routes_report()


def cpu_baseline_leg(dtype="e4m3", shape=(4096, 11008)):
    """The reference-semantics path (C restatement, oracle/qt_oracle.c, validated against the
    reference's golden vectors) on this host's cores, bounded sample of the same workload:
    a bf16 weight-shaped tensor of the workload fake-quantized with its dtype's map repeatedly for about 10-20 s."""
    import numpy as np
    from oracle import c_oracle, qt_oracle
    rng = np.random.default_rng(0)
    n = shape[0] * shape[1]
    dtype = dtype.split(",")[0]
    x = qt_oracle.f32_to_bf16((rng.standard_normal(n) * 0.02).astype(np.float32))
    y = np.empty_like(x)
    qmap = qt_oracle.get_quantization_map(dtype)
    one = int(qt_oracle.f32_to_bf16(np.array([1.0], np.float32))[0])
    c_oracle.fake_quant_bf16(x[: 1 << 20], qmap, one)
    t0 = time.perf_counter()
    reps = 0
    while True:
        c_oracle.fake_quant_bf16(x, qmap, one, out=y)
        reps += 1
        el = time.perf_counter() - t0
        if el > 12.0:
            break
    return {"value": reps * n / el, "unit": "elements/s", "cores": c_oracle.num_threads(), "kind": "port",
            "sample": f"{reps} x bf16[{shape[0]},{shape[1]}] {dtype} fake-quant passes (scale 1), {el:.1f} s, OpenMP static"}


def fqt_gemm_leg(device, dtype, M, Ns, K):
    """qt_linear_fqt_bf16 -- bf16 GEMM with the weight's value map (`dtype`) applied in its operand path -- at one problem shape,
    rotating over weights beyond the Infinity Cache, HIP events on the launch stream; against the dense bf16 matrix-core peak."""
    from quantized_training import fused
    import quantized_training as qt

    class _L:                                              # what hip_fqt_linear_or_none reads of a layer
        def __init__(self, w):
            self.weight, self.bias = w, None
    fq = qt.FusedAmaxObsFakeQuantize(dtype=dtype).to(device)
    tables = fused.fqt_tables(fq, device)
    if tables is None:
        return None
    N = sum(Ns)
    pool = max(2, min(8, int(600e6 // (N * K * 2)) + 1))
    x = fq((torch.randn(M, K, device=device)).bfloat16())
    ws = [[_L((torch.randn(n, K, device=device) * 0.02).bfloat16()) for n in Ns] for _ in range(pool)]
    if fused.hip_fqt_linear_or_none(x, ws[0], tables) is None:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for iters in (pool, 5 * pool):
        e0.record()
        for i in range(iters):
            fused.hip_fqt_linear_or_none(x, ws[i % pool], tables)
        e1.record()
        e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    nbytes = N * K * 2 + M * K * 2 + M * N * 2
    del x, ws
    torch.cuda.empty_cache()
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
            "ms_per_launch": round(ms, 5), "flops_per_launch": 2 * M * N * K, "traffic": None,
            "kernel": f"linear_fqt_kernel: bf16 GEMM {M}x{N}x{K} with the {dtype} value map applied to the bf16 weight in its operand path",
            "hbm_view": {"algorithmic_bytes_per_launch": nbytes, "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}}


def lib_fp8_gemm_leg(device, M, N, K):
    """Library FP8 GEMM (qt_fp8_gemm, hipBLASLt) at one shape: the route the committed table keeps for the BERT-base widths."""
    from quantized_training.fused import lt_fp8_gemm
    pool = 8
    a8 = torch.randn(M, K, device=device).to(torch.float8_e4m3fn)
    b8 = (torch.randn(pool, N, K, device=device) * 0.02).to(torch.float8_e4m3fn)
    if lt_fp8_gemm(a8, b8[0]) is None:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for iters in (pool, 5 * pool):
        e0.record()
        for i in range(iters):
            lt_fp8_gemm(a8, b8[i % pool])
        e1.record()
        e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": FP8_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP8_PEAK_TFLOPS, 4),
            "ms_per_launch": round(ms, 5), "flops_per_launch": 2 * M * N * K, "traffic": None,
            "kernel": f"FP8 E4M3 GEMM {M}x{N}x{K} (BERT-base intermediate dense), fp32 accumulate, bf16 out"}


def encoder_workload(a, device, world, rank, multi, sync):
    """BASELINE.json configs[1] and configs[4] as bench workloads (one step = one batch; every rank its own batches, no data-path
    collective).  configs[1]: BERT-base QA model, SQuAD-style eval batches [16, 384] (run_squad.py:29-47 -> run_qa_no_trainer.py:914-959),
    E4M3 activations + weights, forward replayed from a hipGraph.  configs[4]: RoBERTa-base classifier, MRPC-style batches [16, 128]
    (run_glue_no_trainer.py:647-700), int8 activations + weights with delayed scaling, E5M2 gradients through the backward hooks,
    clip 1.0 + AdamW, the whole step replayed from a hipGraph.  Returns the JSON dict on rank 0 (None elsewhere)."""
    import quantized_training as qt
    from quantized_training import fused, harness
    from quantized_training.fake_quantize import STATS
    torch.manual_seed(0)
    need = a.warmup + a.steps
    g = torch.Generator().manual_seed(1 + rank)
    if a.kind == "bert":
        from transformers import BertConfig, BertForQuestionAnswering
        model = BertForQuestionAnswering(BertConfig()).to(device).eval()
        qt.quantize(model, qt.add_qspec_args().parse_args(["--activation", a.activation, "--weight", a.weight, "--bf16", "--quantize_forward", "gemm"]))
        B, S = 16, 384
        batches = [{"input_ids": torch.randint(1000, 30000, (B, S), generator=g).to(device), "attention_mask": torch.ones(B, S, dtype=torch.long, device=device),
                    "token_type_ids": torch.zeros(B, S, dtype=torch.long, device=device)} for _ in range(need)]
        with torch.no_grad():
            for _ in range(2):
                model(**batches[0])
            STATS.reset()
            model(**batches[0])
            sync()
            elems, calls = STATS.elements, STATS.calls
            step = harness.GraphedBatch(model, batches[0])
            run = lambda b: step.replay(b)                      # noqa: E731
            for i in range(a.warmup):
                run(batches[i])
            if multi:
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            for i in range(a.steps):
                run(batches[a.warmup + i])
            sync()
            if multi:
                dist.barrier()
            sync()
            el = time.perf_counter() - t0
        what = (f"BERT-base QA model (12 layers, hidden 768, random init) SQuAD-style eval batch [{B}, {S}], fake-quant activation={a.activation} "
                f"weight={a.weight}, --quantize_forward gemm (weights re-quantized every forward)")
        launch = "hipGraph replay"
        shape_w = (768, 3072)
    else:
        from transformers import RobertaConfig, RobertaForSequenceClassification
        # dropout 0: the fused training paths (train_fusions.py) cover the deterministic step; QT_BENCH_DROPOUT=p runs HF's default-style
        # dropout instead (an experiment: the attention core, the softmax kernel and the four-member gradient chains then decline)
        drop = float(os.environ.get("QT_BENCH_DROPOUT", "0") or 0) or (0.1 if a.workload.endswith("-dropout") else 0.0)
        model = RobertaForSequenceClassification(RobertaConfig(num_labels=2, hidden_dropout_prob=drop, attention_probs_dropout_prob=drop)).to(device).bfloat16()
        qt.quantize(model, qt.add_qspec_args().parse_args(["--activation", a.activation, "--weight", a.weight, "--error",
                                                           "fp8_e5m2,qs=per_tensor_symmetric,qmax=57344,ahl=10", "--quantize_forward", "gemm",
                                                           "--quantize_backprop", "gemm,residual", "--bf16"]))
        B, S = 16, 128
        batches = [{"input_ids": torch.randint(3, 50000, (B, S), generator=g).to(device), "labels": torch.randint(0, 2, (B,), generator=g).to(device)}
                   for _ in range(need + 3)]
        opt = torch.optim.AdamW(model.parameters(), lr=2e-5, fused=True, capturable=True)
        harness.train_steps(model, batches[:2], torch.optim.AdamW(model.parameters(), lr=2e-5))
        STATS.reset()
        harness.train_steps(model, batches[2:3], torch.optim.AdamW(model.parameters(), lr=2e-5))
        sync()
        elems, calls = STATS.elements, STATS.calls
        experiment = None
        if os.environ.get("QT_BENCH_NO_OBSERVE") == "1":
            # experiment only (tools/, profiles/): the step with every observer frozen -- no amax atomics, no scale updates -- prices the
            # delayed-scaling bookkeeping; the line says so and is no measurement of the workload
            from quantized_training.fake_quantize import FusedAmaxObsFakeQuantize
            for mod in model.modules():
                if isinstance(mod, FusedAmaxObsFakeQuantize):
                    mod.disable_observer()
            experiment = "observers frozen (QT_BENCH_NO_OBSERVE=1): NOT the workload"
        step = harness.GraphedTrainStep(model, opt)
        step.capture(batches[0], warmup=3)
        for i in range(a.warmup):
            step.replay(batches[3 + i])
        if multi:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step.replay(batches[3 + a.warmup + i])
        sync()
        if multi:
            dist.barrier()
        sync()
        el = time.perf_counter() - t0
        what = (f"RoBERTa-base classifier (12 layers, hidden 768, random init) MRPC-style TRAINING step [{B}, {S}]: fake-quant activation={a.activation} "
                f"weight={a.weight}, E5M2 gradients (--quantize_backprop gemm,residual), dropout {drop:g}, clip 1.0, AdamW")
        launch = "hipGraph replay (forward + backward + optimizer)"
        if experiment:
            what = "[" + experiment + "] " + what
        shape_w = (768, 3072)
    t = torch.tensor([el], device=device, dtype=torch.float64)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    if rank != 0:
        return None
    total = elems * a.steps * world
    out = {"metric": "quantized_elements_per_sec", "value": total / el, "unit": "elements/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": what, "elements_per_step": elems, "fake_quant_calls_per_step": calls,
                      "parallelism": f"dp{world} (one batch per rank and step, no data-path collective)", "launch": launch, "valid": True,
                      "routes": fused.routes_report()},
           "fraction_of_hbm_roofline": round(total / el * 4.0 / (HBM_PEAK_GBPS * 1e9 * world), 4)}
    del model
    torch.cuda.empty_cache()
    if not a.no_roofline:
        if a.kind == "bert":
            leg = lib_fp8_gemm_leg(device, 6144, 3072, 768)
            ew = roofline_leg(device, a.weight, shape_w)
            if leg is not None:
                leg["elementwise_pass"] = ew
            out["roofline"] = leg if leg is not None else ew
        else:
            out["roofline"] = roofline_leg(device, a.weight, shape_w)
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_leg(a.weight, shape_w)
    return out


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:      # decided before anything initialises the GPU
        sys.exit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        print(f"[bench] --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    multi = world > 1 or a.force_dist
    if a.dry_run:
        device = torch.device("cpu")
        a.model, a.max_length, a.stride, a.no_graph, a.no_roofline, a.no_cpu_baseline = "llama-tiny", 64, 32, True, True, True
        if multi:
            dist.init_process_group("gloo")
    else:
        if not torch.cuda.is_available():
            print("[bench] no GPU visible (use --dry-run for the CPU plumbing check)", file=sys.stderr)
            sys.exit(3)
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
        if multi:
            dist.init_process_group("nccl", device_id=device)

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize()

    import quantized_training as qt
    from quantized_training import harness
    from quantized_training.fake_quantize import STATS

    if a.kind != "llama" and not a.dry_run:
        out = encoder_workload(a, device, world, rank, multi, sync)
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return

    model = harness.build_causal_lm(a.model, device=device, seed=0, num_layers=a.layers,
                                    dtype=torch.float32 if a.dry_run else torch.bfloat16)
    fusion_counts = None
    if a.route == "pt2e":
        model = harness.prepare_pt2e_causal_lm(model, a.activation, a.weight, a.max_length)
        fusion_counts = getattr(model, "fusion_counts", None)
    else:
        qargs = qt.add_qspec_args().parse_args(["--activation", a.activation, "--weight", a.weight, "--quantize_forward", "gemm"]
                                               + ([] if a.dry_run else ["--bf16"]))
        qt.quantize(model, qargs)
    if a.cache_eval_weights:
        model.eval()
        harness.cache_quantized_weights(True)

    vocab = model.config.vocab_size
    gen = torch.Generator().manual_seed(0)
    tokens = torch.randint(0, vocab, (1, 341_469), generator=gen)         # wikitext-2 test length in LLaMA tokens
    windows = harness.wikitext_windows(tokens.shape[1], a.max_length, a.stride)
    mine = harness.shard_round_robin(windows, rank, world)
    need = a.warmup + a.steps
    assert len(mine) >= need, "not enough windows"
    batches = [(tokens[:, b:e].to(device), t) for (b, e, t) in mine[:need]]   # resident before timing

    nlls = []
    graph_used = False
    with torch.no_grad():
        for i in range(max(a.warmup, 1)):             # first call creates the per-tensor fake-quantizers
            harness.window_nll(model, *batches[min(i, need - 1)])
        STATS.reset()
        harness.window_nll(model, *batches[0])
        sync()
        elems_per_step = STATS.elements
        calls_per_step = STATS.calls
        # The whole window forward is launch-bound on the host (~2 500 small launches), and nothing in
        # it synchronises with the host, so it is captured once into a hipGraph and replayed per window.
        step = None
        if not a.no_graph:
            try:
                step = harness.GraphedWindow(model, a.max_length, batches[0][1] if False else None, device)
                step.capture(batches[0][0])
                graph_used = True
            except Exception as e:  # noqa: BLE001
                import traceback
                traceback.print_exc(file=sys.stderr)
                print(f"[bench] hipGraph capture failed ({type(e).__name__}); running eagerly", file=sys.stderr)
                step = None

        def run_window(ids, trg_len):
            if step is not None and ids.shape[1] == a.max_length:
                return step.replay(ids, trg_len)
            return harness.window_nll(model, ids, trg_len)

        for i in range(a.warmup):
            run_window(*batches[i])
        if multi:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(a.steps):
            nlls.append(run_window(*batches[a.warmup + i]).clone())
        sync()
        if multi:
            dist.barrier()
        sync()
        el = time.perf_counter() - t0
    t = torch.tensor([el], device=device, dtype=torch.float64)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    local = torch.stack(nlls)
    allnll = harness.gather_in_order(local, a.steps * world, rank, world) if multi else local

    out = None
    if rank == 0:
        total_elems = elems_per_step * a.steps * world
        hidden, layers = model.config.hidden_size, model.config.num_hidden_layers
        model_shape = (model.config.hidden_size, model.config.intermediate_size)
        full = a.layers is None
        out = {
            "metric": "quantized_elements_per_sec",
            "value": total_elems / el,
            "unit": "elements/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.dry_run else "bf16", "data": "synthetic",
            "config": {"workload": f"{a.model}-shaped LLaMA ({layers} layers, hidden {hidden}, random init) "
                                   f"WikiText-style window eval B=1 S={a.max_length} stride {a.stride}, "
                                   f"fake-quant activation={a.activation} weight={a.weight}, "
                                   + ("--quantize_forward gemm " if a.route == "eager" else
                                      "PT2E route (torch.export + prepare_pt2e, wikitext.py:60-136; fused prepared graph) ")
                                   + ("(weights re-quantized every forward)" if not a.cache_eval_weights
                                      else "(EXPERIMENT: quantized weights cached across windows)"),
                       "route": a.route, **({"pt2e_fusions": fusion_counts} if fusion_counts is not None else {}),
                       "elements_per_step": elems_per_step, "fake_quant_calls_per_step": calls_per_step,
                       "parallelism": f"dp{world} (windows round-robin, metric all_gather only)",
                       "launch": "hipGraph replay" if graph_used else "eager",
                       "valid": bool(full) and not a.cache_eval_weights and not a.dry_run,
                       "routes": fused_routes(), **native_library_note()},
            "mean_window_nll": float(allnll.double().mean().item()),
            # north_star: elements/s "as absolute and as fraction of HBM roofline" -- SURVEY 8(d)'s 4 B per quantized element
            # (bf16 in + bf16 out) against the HBM peak of the GPUs used
            "fraction_of_hbm_roofline": None if a.dry_run else round(total_elems / el * 4.0 / (HBM_PEAK_GBPS * 1e9 * world), 4),
        }
    if rank == 0:                                   # outside the timed region; the other ranks wait at the barrier below
        in_window = None
        if (not a.no_roofline and not a.dry_run and a.route == "eager" and a.weight in ("e4m3", "e5m2") and a.activation == a.weight
                and a.layers is None and a.model == "llama-2-7b"):
            in_window = linears_in_window_leg(model, device, a.max_length)
        del model
        if device.type == "cuda":
            torch.cuda.empty_cache()
        if not a.no_roofline:
            out["roofline"] = roofline_leg(device, a.weight, model_shape)
            if in_window is not None:
                out["roofline"]["linears_in_window"] = in_window
            if a.weight.split(",")[0] not in ("e4m3", "e5m2") and "qs=" not in a.weight:
                # every other stateless spec: the widest Linears (q / k / v as one launch, the lm head) run the value map inside a bf16 GEMM
                leg = fqt_gemm_leg(device, a.weight, a.max_length, [hidden, hidden, hidden], hidden)
                if leg is not None:
                    leg["elementwise_pass"] = out["roofline"]
                    out["roofline"] = leg
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_leg(a.weight, model_shape)
        if (world == 1 and not multi and not a.no_secondary and not a.dry_run and a.workload == "llama-7b-e4m3" and a.route == "eager"
                and a.layers is None and a.model == "llama-2-7b" and not a.cache_eval_weights):
            out["secondary"] = secondary_legs()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
