"""attention kernel A/B inside the real forward (per-kernel HIP-event profile) (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.synth import noise_batch
from sylber_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
for B, secs in [(32, 10), (8, 60)]:
    x = noise_batch(B, secs * 16000, 1).cuda()
    ref = None
    for qw in (0, 32, 64):
        e = HubertEncoderHIP(sd)
        e.set_option(2, qw)
        for _ in range(2): h = e.forward(x, None)
        if ref is None: ref = h.clone()
        e.set_profiling(True)
        for _ in range(5): e.forward(x, None)
        torch.cuda.synchronize()
        p = e.get_profile(); e.set_profiling(False)
        T = e.num_frames(secs * 16000)
        fl = 9 * 4.0 * T * T * 64 * 12 * B
        print("B=%d %ds qw=%s: attention %.3f ms/forward  %.0f TF   max|h - h(default)| %.2e" % (
            B, secs, qw or "auto", p["attention"] / 5, fl / (p["attention"] / 5 * 1e-3) / 1e12, (h - ref).abs().max().item()))
        del e
