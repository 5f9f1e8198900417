from .quantizer import *  # noqa: F401,F403
from .quantizer import QScheme, QuantizationSpec, DerivedQuantizationSpec, get_quant_min_max  # noqa: F401
from .xnnpack_quantizer_utils import QuantizationConfig  # noqa: F401
from .xnnpack_quantizer import XNNPACKQuantizer, get_node_name_to_scope  # noqa: F401
