from .flash_attention import (FlashAttentionConfig, FlashAttention3, FlashAttentionLayer,  # noqa: F401
                              FlashSelfAttention, ModelConverter)
from .ring_attention import (RingAttention, RingAttentionConfig, RingCrossAttention,  # noqa: F401
                             RingSelfAttention)
