"""SYLBER_FP8 forward with the attention core on MXFP8 operands (default) against the same forward with the bf16 core
(SYLBER_OPT_FP8_ATTENTION = -1) and against the bf16 mode: hidden-state differences on synthetic clips."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
x = torch.cat([syllable_wave(160000, seed=9000 + i) for i in range(8)], 0).float().cuda()
ref = HubertEncoderHIP(sd, precision="fp32").forward(x, None)
def rel(a, b): return float(((a - b).pow(2).mean() / b.pow(2).mean()).sqrt())
b16 = HubertEncoderHIP(sd).forward(x, None)
e = HubertEncoderHIP(sd, precision="fp8")
new = e.forward(x, None).clone()
e.set_option(7, -1)
old = e.forward(x, None).clone()
print("finite:", bool(torch.isfinite(new).all()), bool(torch.isfinite(old).all()))
print("hidden rel-rms vs fp32:  bf16 %.3e | fp8, bf16 attention core %.3e | fp8, fp8 attention core %.3e" % (rel(b16, ref), rel(old, ref), rel(new, ref)))
print("fp8 attention core vs bf16 attention core (both fp8 weight GEMMs): %.3e" % rel(new, old))
lens = [160000, 120000, 90000, 160000, 50000, 160000, 33000, 160000]
e.set_option(7, 0); r1 = e.forward(x, lens).clone(); e.set_option(7, -1); r0 = e.forward(x, lens).clone()
print("ragged lengths: finite %s, fp8 core vs bf16 core %.3e" % (bool(torch.isfinite(r1).all()), rel(r1, r0)))
