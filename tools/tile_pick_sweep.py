"""In-forward time of each GEMM launch group (sequential profile, cold activations) with ONE tile id forced on every launch that has it, over a few batch
shapes: the ground truth the cost model of gemm_bf16.hip::launch_f is checked against (its hot micro-benchmark, tools/tail_sweep.py, flatters
the hipcc-scheduled tiles).  `auto` = the cost model's own choice.

    python tools/tile_pick_sweep.py [--shapes 8x60,24x10,...] [--tiles 91,51,...] > profiles/r06_tile_pick.md
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="8x60,24x10,48x10,32x15,24x15,16x30,32x10")
    ap.add_argument("--tiles", default="-1,91,51,97,57,85,10,4,3")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--convs", action="store_true", help="also the 3-tap conv layers (conv1-4)")
    args = ap.parse_args()
    import torch
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.synth import noise_batch
    from sylber_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(0)
    keys = ("gemm_conv5", "gemm_conv6", "gemm_qkv", "gemm_out", "gemm_ffn1", "gemm_ffn2")
    if args.convs:
        keys = ("gemm_conv1", "gemm_conv2", "gemm_conv3", "gemm_conv4") + keys
    tiles = [int(t) for t in args.tiles.split(",")]
    print("ms per forward of each launch group (sequential profile), one tile id forced on every GEMM launch that has it (`auto` = the cost model); "
          "`*` = the fastest forced tile of the column")
    for sh in args.shapes.split(","):
        b, s = (int(x) for x in sh.split("x"))
        n = s * 16000
        x = noise_batch(b, n, seed=0).cuda()
        rows = {}
        frames = 0
        for t in tiles:
            enc = HubertEncoderHIP(sd, precision="bf16")
            frames = enc.num_frames(n)
            for kv in args.opt:
                k, v = (int(z) for z in kv.split("="))
                enc.set_option(k, v)
            if t >= 0:
                enc.set_option(1, t)
            for _ in range(2):
                enc.forward(x, None)
            torch.cuda.synchronize()
            enc.set_profiling(True)
            for _ in range(3):
                enc.forward(x, None)
            torch.cuda.synchronize()
            prof = enc.get_profile()
            enc.set_profiling(False)
            rows[t] = {k: prof.get(k, 0.0) / 3 for k in keys}
            rows[t]["sum"] = sum(rows[t].values())
            del enc
            torch.cuda.empty_cache()
        print()
        print("### %d x %d s (%d frames per clip, %d rows)" % (b, s, frames, b * ((frames + 31) // 32 * 32)))
        print()
        print("| tile | " + " | ".join(keys) + " | sum |")
        print("|---|" + "---:|" * (len(keys) + 1))
        allk = list(keys) + ["sum"]
        best = {k: min([rows[t][k] for t in tiles if t >= 0] or [0.0]) for k in allk}
        for t in tiles:
            print("| %s | " % ("auto" if t < 0 else str(t)) + " | ".join("%.4f%s" % (rows[t][k], "*" if t >= 0 and rows[t][k] == best[k] else "") for k in allk) + " |")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
