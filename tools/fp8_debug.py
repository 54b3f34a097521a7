import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sylber_amd import _lib
from oracle import mxfp8_ref as Q
lib = _lib.load()
def _p(t): return ctypes.c_void_p(t.data_ptr())
def run(a, w):
    M, K = a.shape; N = w.shape[0]
    ad, wd = torch.from_numpy(a).cuda(), torch.from_numpy(w).cuda()
    out = torch.empty(M, N, device="cuda")
    _lib.check(lib.sylber_op_linear(_p(ad), _p(wd), None, _p(out), M, N, K, 0, 2, -1, None), "lin")
    return out.cpu().numpy().astype(np.float64)
rng = np.random.default_rng(0)
for name, M, N, K, spread in [("narrow", 256, 256, 128, 0), ("narrowK1024", 256, 256, 1024, 0), ("wide", 256, 256, 128, 2), ("ints", 256, 256, 128, -1)]:
    if spread == -1:
        a = rng.integers(-3, 4, (M, K)).astype(np.float32); w = rng.integers(-2, 3, (N, K)).astype(np.float32)
    else:
        a = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-spread, spread, (M, 1)))).astype(np.float32)
        w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    got = run(a, w)
    aq, wq = Q.dequantize(*Q.quantize(a)), Q.dequantize(*Q.quantize(w))
    exp = aq @ wq.T
    mag = np.abs(aq) @ np.abs(wq).T
    err = np.abs(got - exp)
    print(name, "max err/mag", (err / mag).max(), "max err/|exp|max", err.max() / np.abs(exp).max(), "mean err/mag", (err / mag).mean())
    # f32 sequential emulation
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 32):
        acc = (acc.astype(np.float64) + aq[:, k0:k0+32] @ wq[:, k0:k0+32].T).astype(np.float32)
    print("   f32-per-block emulation err/mag", (np.abs(acc - exp) / mag).max())
