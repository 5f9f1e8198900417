"""QConfig: which fake-quantizer to build for activations, weights and errors (gradients).

Mirrors upstream src/quantized_training/qconfig.py:13-58.  ``get_qconfig`` accepts spec
strings or already-parsed ``QuantizationSpec`` objects (argparse converts ``--error`` itself,
upstream training_args.py:174, which the reference then fails to re-parse).
"""
import copy
from collections import namedtuple
from dataclasses import asdict

from torch import nn

from .fake_quantize import FusedAmaxObsFakeQuantize
from .quantizer import QuantizationSpec

__all__ = ["QConfig", "get_qconfig"]


class QConfig(namedtuple("QConfig", ["activation", "weight", "error"])):
    """Constructors (not instances) of the fake-quantizers for one layer: the preparation code
    instantiates them once per tensor."""

    def __new__(cls, activation, weight, error):
        return super().__new__(cls, activation, weight, error)


def _fake_quant_ctr(spec, record_histogram, force_scale_power_of_two):
    if spec is None:
        return nn.Identity
    fields = copy.deepcopy(asdict(QuantizationSpec.from_str(spec)))
    return FusedAmaxObsFakeQuantize.with_args(
        **fields, record_histogram=record_histogram, force_scale_power_of_two=force_scale_power_of_two)


def get_qconfig(activation, weight, error, record_histogram=False, force_scale_power_of_two=False):
    opts = dict(record_histogram=record_histogram, force_scale_power_of_two=force_scale_power_of_two)
    return QConfig(
        activation=_fake_quant_ctr(activation, **opts),
        weight=_fake_quant_ctr(weight, **opts),
        error=_fake_quant_ctr(error, **opts),
    )
