"""API-level (PCIe-inclusive) timing breakdown of Segmenter.__call__ on the GPU box (development aid): where do the
milliseconds of one call go, and how repeatable is each stage?"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402
from sylber_amd import Segmenter  # noqa: E402
from sylber_amd.synth import noise_batch  # noqa: E402
from sylber_amd.weights import synthetic_state_dict  # noqa: E402

sd = synthetic_state_dict(0)
S = Segmenter(model_ckpt=sd)
B, N = 32, 160000
wavs = [w[None, :].clone() for w in noise_batch(B, N, seed=1)]
dev = S.speech_model.device


def stats(name, xs):
    xs = [x * 1e3 for x in xs]
    print("%-34s min %8.2f  med %8.2f  max %8.2f ms" % (name, min(xs), statistics.median(xs), max(xs)), flush=True)


def rep(name, f, n=12):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    stats(name, ts)


print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count())
rep("call total", lambda: S(wav=wavs))
rep("encode_batch (stage+H2D+fwd)", lambda: S.encode_batch(wavs))
hid, _ = S.encode_batch(wavs)
torch.cuda.synchronize()
x_dev = torch.empty(B, N, device=dev)
rep("forward only (device input)", lambda: S.speech_model.forward(x_dev, None))
rep("segment only", lambda: S.speech_model.segment(hid, 2.6, 0.8))
pin = torch.empty(hid.shape, dtype=torch.float32, pin_memory=True)
rep("D2H 49 MB -> pinned", lambda: pin.copy_(hid, non_blocking=True))
small = torch.empty(1 << 20, dtype=torch.uint8, pin_memory=True)
small_d = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
rep("D2H 1 MB -> pinned", lambda: small.copy_(small_d, non_blocking=True))
rep("hidden.cpu() (pageable)", lambda: hid.cpu())
stage = torch.empty(B, N, pin_memory=True)


def fill():
    for i, w in enumerate(wavs):
        stage[i, :N] = w[0]


rep("fill pinned stage (32 row copies)", fill)
st2 = torch.empty(B, N)


def fill2():
    for i, w in enumerate(wavs):
        st2[i, :N] = w[0]


rep("fill pageable stage", fill2)
rep("H2D 20 MB pinned", lambda: x_dev.copy_(stage, non_blocking=True))
rep("H2D 20 MB pageable", lambda: x_dev.copy_(st2))
rep("pinned -> numpy copy 49 MB", lambda: pin.numpy().copy())
for nt in (1, 8):
    torch.set_num_threads(nt)
    rep("fill pinned stage, %d torch threads" % nt, fill)
    rep("call total, %d torch threads" % nt, lambda: S(wav=wavs))
