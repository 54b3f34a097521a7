"""GPU box: measured deviations of the HIP path from the reference goldens / fp32 oracle -> markdown."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hubert_ref, segment_oracle
from sylber_amd import HubertEncoderHIP
from sylber_amd.synth import noise_batch, syllable_wave
from sylber_amd.weights import synthetic_state_dict

def rel_rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))

sd = synthetic_state_dict(0)
g = np.load(os.path.join(ROOT, "tests/golden/encoder_stages.npz"))
wav = torch.from_numpy(g["wav"]).cuda(); lengths = [int(x) for x in g["lengths"]]
print("# Parity report (MI355X) — HIP path vs reference goldens (tests/golden/encoder_stages.npz)\n")
print("| stage | bf16 rel-RMS | bf16 max-abs | fp16 rel-RMS | fp16 max-abs | fp32 rel-RMS | fp32 max-abs | fp8 (configs[4]) rel-RMS | fp8 max-abs |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|")
encs = {p: HubertEncoderHIP(sd, precision=p) for p in ("bf16", "fp16", "fp32", "fp8")}
for name, stage, key in [("conv stack", 1, "conv6"), ("encoder input (proj+pos-conv+LN)", 2, "enc_in"), ("layer 0", 3, "layer0"),
                         ("layer 4", 7, "layer4"), ("layer 8 = hidden_states", 0, "layer8")]:
    ref = g[key].transpose(0, 2, 1) if key == "conv6" else g[key]
    row = []
    for p in ("bf16", "fp16", "fp32", "fp8"):
        out = encs[p].forward(wav, lengths, stop_stage=stage).cpu().numpy()
        row += ["%.2e" % rel_rms(out, ref), "%.2e" % np.abs(out - ref).max()]
    print("| %s | %s |" % (name, " | ".join(row)))
# segment agreement of the end-to-end paths with the fp32 reference segmentation
for prec in ("bf16", "fp16", "fp8"):
    tot = agree = 0; nb = mb = 0
    for seed in range(16):
        x = syllable_wave(80000, 500 + seed)
        ref_h = hubert_ref.forward(sd, x, None)["hidden"][0].numpy()
        ref_s = segment_oracle.get_segment(ref_h, 2.6, 0.8).reshape(-1, 2)
        h = encs[prec].forward(x.cuda().contiguous())
        seg, nseg, _ = encs[prec].segment(h, 2.6, 0.8)
        s = seg[0, : int(nseg[0])].cpu().numpy()
        tot += 1; agree += int(s.shape == ref_s.shape and np.array_equal(s, ref_s))
        rb, gb = set(ref_s.reshape(-1).tolist()), set(s.reshape(-1).tolist())
        nb += len(rb); mb += len(rb & gb)
    print("\n%s end-to-end vs fp32 reference segmentation on 16 synthetic 5 s clips: %d/%d tables identical, %.1f %% of boundaries identical" % (prec, agree, tot, 100.0 * mb / max(nb, 1)))
