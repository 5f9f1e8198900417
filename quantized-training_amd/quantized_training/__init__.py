"""quantized_training -- MI355X-native fake-quantization engine.

Drop-in for the hot path of jeffreyyu0602/quantized-training: the same ``quantize()`` /
``add_qspec_args()`` / ``FusedAmaxObsFakeQuantize`` / ``quantized_ops`` surface
(upstream src/quantized_training/__init__.py:23-67), with the rounding, observer and GEMM work
done by hand-written HIP kernels for gfx950 behind the C ABI of ``include/qt_hip.h``.
"""
import torch

from . import _native  # noqa: F401
from .quantizer import QScheme, QuantizationSpec, DerivedQuantizationSpec, get_quant_min_max  # noqa: F401
from .fake_quantize import (  # noqa: F401
    FusedAmaxObsFakeQuantize,
    FusedAmaxObsFakeQuantFunction,
    _DerivedObserverOrFakeQuantize,
    get_quantization_map,
)
from .decomposed import vmap, quantize as quantize_op, dequantize, expand  # noqa: F401
from .fp8 import quantize_to_fp8_e4m3, quantize_to_fp8_e5m2  # noqa: F401
from .posit import quantize_to_posit  # noqa: F401
from .normal_float import quantize_to_nf, create_normal_map  # noqa: F401
from .qconfig import QConfig, get_qconfig  # noqa: F401
from .quantize import (  # noqa: F401
    convert, get_quantized_model, prepare, propagate_config, quantize, replace_softmax, swap_module,
)
from .quantize_pt2e import (  # noqa: F401
    convert_pt2e, derive_bias_qparams_fn, export_model, get_default_quantizer, prepare_pt2e,
)
from .quantizer import XNNPACKQuantizer, QuantizationConfig, get_node_name_to_scope  # noqa: F401
from .training_args import add_qspec_args  # noqa: F401
from .utils import setup_logging  # noqa: F401
from . import modules  # noqa: F401

__all__ = [
    "FusedAmaxObsFakeQuantize", "QConfig", "QuantizationSpec", "add_qspec_args", "convert",
    "get_qconfig", "get_quantized_model", "prepare", "propagate_config", "quantize",
    "quantize_to_fp8_e4m3", "quantize_to_fp8_e5m2", "quantize_to_posit", "quantize_to_nf", "replace_softmax",
    "setup_logging", "get_quantization_map", "vmap", "dequantize",
    "get_default_quantizer", "prepare_pt2e", "convert_pt2e", "export_model", "derive_bias_qparams_fn",
    "get_node_name_to_scope",
]


class qscheme:  # noqa: N801  (upstream name)
    ...


# upstream __init__.py:64-67
per_tensor_symmetric = QScheme.PER_TENSOR_SYMMETRIC
per_channel_symmetric = QScheme.PER_CHANNEL_SYMMETRIC
microscaling = QScheme.MICROSCALING
group_wise_affine = QScheme.GROUP_WISE_AFFINE

aten = torch.ops.aten
quantized_ops = torch.ops.quantized_ops
