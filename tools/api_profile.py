"""API-level (PCIe-inclusive) timing breakdown of Segmenter.__call__ on the GPU box (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sylber_amd import Segmenter
from sylber_amd.synth import noise_batch
from sylber_amd.weights import synthetic_state_dict

sd = synthetic_state_dict(0)
S = Segmenter(model_ckpt=sd)
B, N = 32, 160000
wavs = [w[None, :].clone() for w in noise_batch(B, N, seed=1)]
def T(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("call total            %.2f ms" % T(lambda: S(wav=wavs)))
print("encode_batch (H2D+fwd) %.2f ms" % T(lambda: S.encode_batch(wavs)))
hid, _ = S.encode_batch(wavs)
print("hidden.cpu()           %.2f ms" % T(lambda: hid.cpu()))
pin = torch.empty(hid.shape, dtype=torch.float32).pin_memory()
print("copy to pinned + sync  %.2f ms" % T(lambda: (pin.copy_(hid, non_blocking=True), torch.cuda.synchronize())))
print("pinned numpy copy      %.2f ms" % T(lambda: pin.numpy().copy()))
pg = torch.empty(hid.shape, dtype=torch.float32)
print("pageable numpy copy    %.2f ms" % T(lambda: pg.numpy().copy()))
stage = torch.empty(B, N).pin_memory()
def fill():
    for i, w in enumerate(wavs): stage[i, :N] = w[0]
print("fill pinned stage      %.2f ms" % T(fill))
st2 = torch.empty(B, N)
def fill2():
    for i, w in enumerate(wavs): st2[i, :N] = w[0]
print("fill pageable stage    %.2f ms" % T(fill2))
d = torch.empty(B, N, device="cuda")
print("H2D pinned             %.2f ms" % T(lambda: d.copy_(stage, non_blocking=True)))
print("H2D pageable           %.2f ms" % T(lambda: d.copy_(st2)))
