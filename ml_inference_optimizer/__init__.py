"""Import shim so `from ml_inference_optimizer import Optimizer` (reference README.md:57) resolves to the
MI355X implementation in ml-inference-optimizer_amd/mio."""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-inference-optimizer_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from mio.optimizer import Optimizer, apply_flash_attention, apply_fused_mlp  # noqa: E402,F401
