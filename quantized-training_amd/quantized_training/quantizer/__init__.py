from .quantizer import *  # noqa: F401,F403
from .quantizer import QScheme, QuantizationSpec, DerivedQuantizationSpec, get_quant_min_max  # noqa: F401
