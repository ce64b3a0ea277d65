"""fa3_fwd5_kernel epilogue: 16-byte stores after the row-pair exchange (shipped) vs the 8-byte form (diagnostic library,
mio_dbg_set(3, 4)); same process, interleaved, C2 shape, pre-scaled K; outputs must be bit-identical."""
import os, sys
os.environ["MIO_LIB_DBG"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
from tools.kbench import timeit
B, S, H, D = 8, 4096, 16, 64
torch.manual_seed(0)
qkv = torch.randn(B, S, 3 * H * D, device="cuda", dtype=torch.bfloat16)
q, k, v = (qkv[..., i * H * D:(i + 1) * H * D].view(B, S, H, D) for i in range(3))
res = {}
outs = {}
for rd in range(3):
    for mode in (0, 4):
        _lib.lib.mio_dbg_set(3, mode)
        for causal in (True, False):
            outs[(mode, causal)] = ops.fa3_fwd(q, k, v, causal=causal, k_prescaled=True)
            t = timeit(lambda: ops.fa3_fwd(q, k, v, causal=causal, k_prescaled=True), 20)
            res.setdefault((mode, causal), []).append(t)
_lib.lib.mio_dbg_set(3, 0)
for causal in (True, False):
    assert torch.equal(outs[(0, causal)], outs[(4, causal)]), "store forms differ"
    for mode in (0, 4):
        ts = sorted(res[(mode, causal)])
        print(f"causal={causal} {'16-byte' if mode == 0 else ' 8-byte'} stores: median {ts[1]*1e3:.4f} ms min {ts[0]*1e3:.4f}")
