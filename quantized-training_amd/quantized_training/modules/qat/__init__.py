from .linear import Linear  # noqa: F401
from .lora import LoraLinear, is_lora_linear  # noqa: F401
