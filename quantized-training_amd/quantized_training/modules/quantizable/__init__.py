from .functional_modules import AddFunctional, MulFunctional, MatmulFunctional  # noqa: F401
from .attention import (  # noqa: F401
    QuantizableAttentionCore,
    BertSelfAttention, BertSelfOutput, BertOutput,
    MobileBertSelfAttention, MobileBertSelfOutput, FFNOutput, MobileBertOutput,
    LlamaAttention,
)
