from .flash_attention import (FlashAttentionConfig, FlashAttention3, FlashAttentionLayer,  # noqa: F401
                              FlashSelfAttention, ModelConverter)
