from .fused_mlp import (FusedMLPConfig, FusedMLP, FusedMLPGeluTanh, FusedMLPSwiGLU, FusedMLPReLU,  # noqa: F401
                        FusedTransformerMLP, MLPConverter)
