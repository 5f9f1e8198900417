"""Quantization spec mini-language: ``dtype[,key=value]*``.

Mirrors the reference's ``QuantizationSpec`` / ``QScheme`` / ``get_quant_min_max``
(src/quantized_training/quantizer/quantizer.py:18-146 upstream): same keys, abbreviations,
defaults and error types, so ``--activation int8,qs=per_tensor_symmetric`` strings keep working.
"""
import re
from dataclasses import dataclass
from enum import Enum
from typing import Callable, List, Optional, Tuple, Union

from torch.ao.quantization.quantizer.quantizer import QuantizationSpecBase

__all__ = ["QScheme", "QuantizationSpec", "DerivedQuantizationSpec", "get_quant_min_max"]


class QScheme(Enum):
    PER_TENSOR_SYMMETRIC = "per_tensor_symmetric"
    PER_CHANNEL_SYMMETRIC = "per_channel_symmetric"
    MICROSCALING = "microscaling"
    GROUP_WISE_AFFINE = "group_wise_affine"


def _int_or_tuple(text: str):
    text = text.strip()
    if text[:1] == "(" and text[-1:] == ")":
        return tuple(int(tok.strip()) for tok in text[1:-1].split(","))
    return int(text)


# full key -> (abbreviation, parser)                      upstream: ABBREV_MAP / PARAMS_TYPE (:24-51)
_SPEC_KEYS = {
    "quant_min": ("qmin", float),
    "quant_max": ("qmax", float),
    "qscheme": ("qs", QScheme),
    "amax_history_len": ("ahl", int),
    "ch_axis": ("ax", _int_or_tuple),
    "block_size": ("bs", _int_or_tuple),
    "scale_dtype": ("scale", str),
    "outlier_threshold": ("outlier", float),
}
_ABBREV = {abbr: full for full, (abbr, _) in _SPEC_KEYS.items()}

_RE_INT = re.compile(r"int(\d+)", re.IGNORECASE)
_RE_UINT = re.compile(r"uint(\d+)", re.IGNORECASE)
_RE_FLOAT = re.compile(r"fp(\d+)_e(\d+)m(\d+)", re.IGNORECASE)
_RE_POSIT = re.compile(r"posit(\d+)_(\d+)", re.IGNORECASE)
_RE_NF = re.compile(r"nf(\d+)(?:_(\d+))?", re.IGNORECASE)


def get_quant_min_max(dtype: str):
    """(min, max) representable value of ``dtype`` (upstream quantizer.py:53-94)."""
    if m := _RE_INT.fullmatch(dtype):
        n = int(m.group(1))
        return -(2 ** (n - 1)), 2 ** (n - 1) - 1
    if m := _RE_UINT.fullmatch(dtype):
        return 0, 2 ** int(m.group(1)) - 1
    if m := _RE_FLOAT.fullmatch(dtype):
        ebits, mant = int(m.group(2)), int(m.group(3)) + 2
        emax = 2 ** (ebits - 1) - 1 if ebits > 4 else 2 ** (ebits - 1)
        if dtype.lower() == "fp8_e4m3":
            top = 2 ** emax * 1.75
        else:
            top = 2 ** emax * (2 ** (mant - 1) - 1) / 2 ** (mant - 2)
        return -top, top
    if m := _RE_POSIT.fullmatch(dtype):
        nbits, es = int(m.group(1)), int(m.group(2))
        top = (2 ** (2 ** es)) ** (nbits - 2)
        return -top, top
    if m := _RE_NF.fullmatch(dtype):
        top = 2 ** (int(m.group(2)) - 1) - 1 if m.group(2) is not None else 1
        return -top, top
    raise ValueError(f"Unsupported dtype: {dtype}")


def _default_ctr(*args, **kwargs):
    from ..fake_quantize import FusedAmaxObsFakeQuantize
    return FusedAmaxObsFakeQuantize(*args, **kwargs)


@dataclass(eq=True)
class QuantizationSpec(QuantizationSpecBase):
    """How to quantize one tensor (upstream quantizer.py:96-146)."""

    dtype: str
    observer_or_fake_quant_ctr: Callable = None
    quant_min: Optional[float] = None
    quant_max: Optional[float] = None
    qscheme: Optional[QScheme] = None
    amax_history_len: Optional[int] = None
    ch_axis: Optional[Union[int, List[int]]] = None
    block_size: Optional[Union[int, List[int]]] = None
    scale_dtype: Optional[str] = None
    outlier_threshold: Optional[float] = None
    is_dynamic: bool = False

    def __post_init__(self):
        if self.observer_or_fake_quant_ctr is None:
            from ..fake_quantize import FusedAmaxObsFakeQuantize
            self.observer_or_fake_quant_ctr = FusedAmaxObsFakeQuantize
        if self.qscheme is not None and self.quant_max is None:
            raise ValueError("quant_max is required for quantization.")
        if self.qscheme in (QScheme.MICROSCALING, QScheme.GROUP_WISE_AFFINE) and self.block_size is None:
            raise ValueError("block_size is required for microscaling.")

    @staticmethod
    def from_str(s):
        if isinstance(s, QuantizationSpec):   # argparse may already have converted it (upstream bug)
            return s
        if not s:
            raise ValueError("String quantization_spec is None or empty")
        head, *rest = re.split(r",(?![^()]*\))", s)
        fields = {"dtype": head}
        for item in rest:
            if "=" not in item:
                raise ValueError(f"Expected key=value format but got '{item}'")
            key, value = item.split("=")
            key = _ABBREV.get(key, key)
            if key not in _SPEC_KEYS:
                raise ValueError(f"Unknown argument '{key}'. Valid keys: {', '.join(_SPEC_KEYS)}")
            fields[key] = _SPEC_KEYS[key][1](value)
        scheme = fields.get("qscheme")
        if scheme is not None:
            lo, hi = get_quant_min_max(fields["dtype"])
            fields.setdefault("quant_min", float(lo))
            fields.setdefault("quant_max", float(hi))
            if scheme in (QScheme.PER_TENSOR_SYMMETRIC, QScheme.PER_CHANNEL_SYMMETRIC):
                fields.setdefault("amax_history_len", 16)
        return QuantizationSpec(**fields)


@dataclass(eq=True)
class DerivedQuantizationSpec(QuantizationSpecBase):
    """Spec whose qparams derive from other tensors' observers (upstream quantizer.py:150-159)."""

    derived_from: list
    derive_qparams_fn: Callable
    dtype: str
    quant_min: Optional[int] = None
    quant_max: Optional[int] = None
    qscheme: Optional[QScheme] = None
