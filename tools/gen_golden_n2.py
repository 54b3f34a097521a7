"""Golden vector for row N2 (survey container only): the REFERENCE's own ``Sylber.segment``
(sylber/model/sylber.py:208-247) on a ragged two-clip batch with the seeded weights, stored in
tests/golden/sylber_segment.npz (inputs by seed, outputs: segments, zero-padded avg_fts, a hidden-state checksum).
Contains no reference code; never runs on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sylber_amd.synth import syllable_wave                # noqa: E402
from sylber_amd.weights import synthetic_state_dict      # noqa: E402
from tools import ref_shim                                # noqa: E402

LENS = [30000, 22000]
SEEDS = [901, 902]


def main():
    ref, _, cfg_dir = ref_shim.load()
    torch.manual_seed(0)
    model = ref.Sylber(speech_upstream=cfg_dir, encoding_layer=9).eval()
    sd = synthetic_state_dict(0)
    missing = model.speech_model.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if "masked_spec_embed" not in k], missing
    batch = torch.zeros(2, max(LENS))
    mask = torch.zeros(2, max(LENS), dtype=torch.long)
    for i, (n, s) in enumerate(zip(LENS, SEEDS)):
        batch[i, :n] = syllable_wave(n, s)[0]
        mask[i, :n] = 1
    with torch.no_grad():
        feats, segments, avg_fts = model.segment(input_values=batch, attention_mask=mask, mergethreshold=0.8, normthreshold=2.6)
    out = {"lens": np.array(LENS), "seeds": np.array(SEEDS), "avg_fts": avg_fts.numpy(),
           "hidden_first_last": feats[:, [0, -1]].numpy(), "hidden_abs_mean": np.float64(feats.abs().mean())}
    for i, sg in enumerate(segments):
        out["segments%d" % i] = np.asarray(sg)
    print("segments per clip:", [len(s) for s in segments], "avg_fts", tuple(avg_fts.shape))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sylber_segment.npz"), **out)
    print("wrote tests/golden/sylber_segment.npz")
    check_file_branch(ref, cfg_dir, sd)


def check_file_branch(ref, cfg_dir, sd):
    """The reference's FILE branch (sylber.py:79-87: torchaudio.load -> normalise -> batch) run for real on
    samples/sample.wav and on a two-file list, with ``torchaudio.load`` provided by a stdlib ``wave`` reader of
    16-bit PCM (int16 / 32768, what torchaudio.load returns); must reproduce the committed e2e golden, which was
    generated through the tensor branch with the same normalisation ops."""
    import tempfile
    import wave
    ta = sys.modules["torchaudio"]

    def load(path):
        with wave.open(str(path), "rb") as w:
            assert w.getsampwidth() == 2
            x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
            return torch.from_numpy(x.reshape(-1, w.getnchannels()).T.copy()), w.getframerate()
    ta.load = load
    tmp = os.path.join(tempfile.gettempdir(), "sylber_synth_state.pt")
    torch.save(sd, tmp)
    S = ref.Segmenter(model_ckpt=tmp, speech_upstream=cfg_dir, device="cpu")
    path = os.path.join(ref_shim.REFERENCE_ROOT, "samples", "sample.wav")
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e.npz"))
    out = S(wav_file=path, in_second=False)
    assert np.array_equal(out["segments"], g["sample_segments"])
    assert np.abs(out["hidden_states"] - g["sample_hidden"]).max() < 2e-5
    outs = S(wav_file=[path, path], in_second=True)
    assert isinstance(outs, list) and np.array_equal(outs[1]["segments"], g["sample_segments_sec"])
    print("file branch of the reference reproduces the e2e golden: hidden max abs",
          float(np.abs(out["hidden_states"] - g["sample_hidden"]).max()))


if __name__ == "__main__":
    main()
