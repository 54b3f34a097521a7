"""attention kernel A/B on the GPU box: 32 vs 64 queries per wave (development aid)"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
for (B, T) in [(32, 499), (8, 2999)]:
    q = torch.randn(B, T, 768, device="cuda"); k = torch.randn(B, T, 768, device="cuda"); v = torch.randn(B, T, 768, device="cuda")
    o = torch.empty(B, T, 768, device="cuda")
    outs = {}
    for qw in (1, 2):
        for _ in range(2): lib.sylber_op_attention(p(q), p(k), p(v), None, p(o), B, T, 0, 32 * qw, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 5
        for _ in range(n): lib.sylber_op_attention(p(q), p(k), p(v), None, p(o), B, T, 0, 32 * qw, None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        outs[qw] = o.clone()
        print("B=%d T=%d qw=%d: %.1f us per call (includes pack + unpack kernels)" % (B, T, qw, dt * 1e6))
    print("   max |qw1 - qw2| = %.3e" % (outs[1] - outs[2]).abs().max().item())
