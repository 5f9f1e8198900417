"""Command-line flags of the quantization drivers (flag-compatible with upstream
src/quantized_training/training_args.py:36-256, so reference command lines parse unchanged)."""
import argparse

from .quantizer.quantizer import QuantizationSpec
from .utils import SLURM_ARGS

__all__ = ["add_qspec_args"]

SPEC_HELP = """
Comma-separated list: the dtype first, then key=value pairs (full names or abbreviations).
  qs=qscheme  qmax=quant_max  qmin=quant_min  ahl=amax_history_len  ax=ch_axis  bs=block_size
  scale=scale_dtype  outlier=outlier_threshold
dtype examples: int8, int4, e4m3, e5m2, fp8_e4m3, fp8_e5m2, fp6_e3m2, fp4_e2m1, posit8_1
example: --activation int8,qs=per_tensor_symmetric,qmax=127,ahl=50
"""

_csv = lambda s: s.split(",")  # noqa: E731

# (flag, kwargs) in upstream order.  Grouped: logging / W&B, training, quantization.
_FLAGS = [
    ("--project", dict(default=None, help="W&B project name.")),
    ("--run_name", dict(default=None, help="Display name of this run.")),
    ("--run_id", dict(default=None, help="Unique run id, used for resuming.")),
    ("--sweep_config", dict(default=None, help="Path to a JSON W&B sweep configuration.")),
    ("--sweep_id", dict(default=None, help="Identifier of an existing sweep.")),
    ("--max_trials", dict(type=int, default=None, help="Number of sweep trials to run.")),
    ("--log_level", dict(choices=["DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL"], default="WARNING",
                         help="Logging level.")),
    ("--log_file", dict(default=None, help="Log file ('datetime' = logs/<timestamp>.log); default stdout.")),
    ("--gpu", dict(type=int, default=None, help="GPU to use.")),
    ("--do_train", dict(action="store_true", help="Run training.")),
    ("--sgd", dict(action="store_true", help="Use the SGD optimizer.")),
    ("--warmup_ratio", dict(type=float, default=0.0, help="Warm-up fraction of the lr schedule.")),
    ("--bf16", dict(action="store_true", help="Run the model in bfloat16 instead of float32.")),
    ("--num_hidden_layers", dict(type=int, default=None, help="Number of Transformer layers to keep.")),
    ("--lora_rank", dict(type=int, default=0, help="LoRA rank (0 = no LoRA).")),
    ("--lora_alpha", dict(type=int, default=8, help="LoRA scaling factor.")),
    ("--target_modules", dict(type=_csv, default="query,value", help="Modules that receive LoRA updates.")),
    ("--peft_model_id", dict(default=None, help="Pre-trained PEFT adapter.")),
    ("--pt2e", dict(action="store_true", help="Use the torch.export (PT2E) quantization flow.")),
    ("--activation", dict(default=None, help="Activation quantization spec." + SPEC_HELP)),
    ("--output_activation", dict(default=None, help="Output-activation quantization spec (same format).")),
    ("--weight", dict(default=None, help="Weight quantization spec (same format).")),
    ("--bias", dict(default=None, help="Bias quantization spec (same format).")),
    ("--error", dict(default=None, type=QuantizationSpec.from_str,
                     help="Activation-gradient quantization spec (same format).")),
    ("--quantize_forward", dict(default="gemm",
                                help="Forward ops to quantize: gemm, residual, activation, layernorm, scaling.")),
    ("--quantize_backprop", dict(default="gemm",
                                 help="Backward ops to quantize: gemm, residual, activation, layernorm, scaling.")),
    ("--force_scale_power_of_two", dict(action="store_true", help="Round scales up to a power of two.")),
    ("--calibration_steps", dict(type=int, default=0, help="Calibration steps for post-training quantization.")),
    ("--convert_model", dict(action="store_true", help="Convert the model to a quantized model.")),
    ("--compile", dict(action="store_true", help="Generate an accelerator program (not supported here).")),
    ("--op_fusion", dict(type=_csv, default=None,
                         help="Module-name substrings whose inputs stay unquantized (fused with the previous GEMM).")),
    ("--posit_exp", dict(action="store_true", help="Posit-approximated exp in softmax.")),
    ("--posit_exp_shifted", dict(action="store_true", help="Shifted posit-approximated exp in softmax.")),
    ("--posit_reciprocal", dict(action="store_true", help="Posit-approximated reciprocal in softmax.")),
    ("--record_histogram", dict(action="store_true", help="Record exponent histograms of quantized tensors.")),
    ("--bank_width", dict(type=int, default=None, help="Memory bank width in bytes (accelerator planning).")),
]


def add_qspec_args(parser=None):
    if parser is None:
        parser = argparse.ArgumentParser(description="Run quantized inference or training.")
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    sub = parser.add_subparsers(help="sub-command help", dest="action")
    slurm = sub.add_parser("slurm", help="slurm command help")
    for name, kw in SLURM_ARGS.items():
        slurm.add_argument("--" + name, **kw)
    sub.add_parser("bash", help="bash command help")
    return parser
