import os, sys
sys.path.insert(0, os.getcwd())
import torch
from sylber_amd import HubertEncoderHIP
from sylber_amd.synth import syllable_wave, noise_batch
from sylber_amd.weights import synthetic_state_dict
sd = synthetic_state_dict(0)
E = [HubertEncoderHIP(sd), HubertEncoderHIP(sd)]
S = [torch.cuda.Stream(), torch.cuda.Stream()]
def batch(ls, seed):
    b = torch.zeros(len(ls), max(ls))
    for i, n in enumerate(ls): b[i, :n] = syllable_wave(n, seed + i)[0]
    return b.cuda()
LENS = [24000, 16000, 31000, 9000, 20000]
cases = [(batch(LENS, 40), LENS), (batch(LENS[1:4], 50), LENS[1:4]), (noise_batch(32, 160000, 3).cuda(), None)]
for stage in (0, 1, 2, 3):
    for ci, (x, ls) in enumerate(cases):
        ref = [E[k].forward(x, ls, stop_stage=stage).clone() for k in (0, 1)]
        torch.cuda.synchronize()
        print("stage", stage, "case", ci, "engines agree sequentially:", torch.equal(ref[0], ref[1]), end="  ")
        bad = 0
        for it in range(6):
            outs = []
            for k in (0, 1):
                with torch.cuda.stream(S[k]):
                    outs.append(E[k].forward(x, ls, stop_stage=stage))
            torch.cuda.synchronize()
            for k in (0, 1):
                if not torch.equal(outs[k], ref[0]):
                    bad += 1
                    d = (outs[k] - ref[0]).abs()
                    if bad == 1: print("\n   first mismatch: engine", k, "max", d.max().item(), "rows", (d.amax(-1) > 0).nonzero()[:4].tolist(), end="")
        print(" concurrent mismatches:", bad, "/ 12")
