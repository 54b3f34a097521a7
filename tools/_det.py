import os, sys
sys.path.insert(0, os.getcwd())
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.synth import syllable_wave
from sylber_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
e = HubertEncoderHIP(sd)
def batch(ls, seed):
    b = torch.zeros(len(ls), max(ls))
    for i, n in enumerate(ls): b[i, :n] = syllable_wave(n, seed + i)[0]
    return b.cuda()
LENS = [24000, 16000, 31000, 9000, 20000]
A = batch(LENS, 40); Bb = batch(LENS[1:4], 50); C = batch(LENS[:2], 70)
def f(x, ls): 
    h = e.forward(x, ls).clone(); torch.cuda.synchronize(); return h
a1 = f(A, LENS); a2 = f(A, LENS)
print("same shape repeat equal:", torch.equal(a1, a2))
f(Bb, LENS[1:4]); a3 = f(A, LENS)
print("after other shape equal:", torch.equal(a1, a3), (a1 - a3).abs().max().item())
f(C, LENS[:2]); a4 = f(A, LENS)
print("after other shape 2 equal:", torch.equal(a1, a4), (a1 - a4).abs().max().item())
for st in (1, 2, 3, 4, 7, 11):
    x1 = e.forward(A, LENS, stop_stage=st).clone(); e.forward(C, LENS[:2]); x2 = e.forward(A, LENS, stop_stage=st).clone()
    d = (x1 - x2).abs()
    print("stage", st, torch.equal(x1, x2), d.max().item(), "first bad row (b,t):", (d.amax(-1) > 0).nonzero()[:3].tolist())
