#!/usr/bin/env python
"""Throughput benchmark of the Segmenter hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 with no WORLD_SIZE in the environment re-executes itself under ``python -m torch.distributed.run``
(one rank per GPU, rendezvous on 127.0.0.1) and fails loudly when fewer than N GPUs are visible; launched
by an external ``torch.distributed.run`` it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

A step = one pass of the hot path over one batch of synthetic input: waveform batch resident in HBM
-> 7-layer conv frontend -> 9-layer HuBERT encoder -> boundary detection + segment mean-pool, outputs
(hidden_states, segments, segment_features) left in HBM.  Workload = BASELINE.json configs[1]:
32 x 10 s x 16 kHz random waveforms per GPU, synthetic seeded weights of the sylber_base geometry
(no network for the real checkpoint), bf16 MFMA compute with fp32 accumulation / residual stream.

N = 1: the shard is resident in HBM when the timed region starts.
N > 1 (BASELINE configs[2], weak scaling, 32 clips per GPU): the job's N x 32 clips are resident on RANK 0 and every
timed step contains the RCCL exchange of sylber_amd/dist.py::ShardedSegmenter.run_stream: root scatters the
waveform blocks over xGMI, every rank encodes + segments its block, root gathers hidden states, segment tables,
counts and pooled features; the gather of step i travels under the compute of step i+1.  The resident-shard
figure (no data-path collective) is reported next to it as ``resident_shards``.  ``--no-exchange`` swaps the two.
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# ROCm multiplexes a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue run
# strictly one after the other (sylber_amd/streams.py).  The N > 1 step keeps two engine streams, two side streams and RCCL's own
# stream busy: with 4 queues RCCL's gather would share one with a compute stream and hold its kernels back.  Must be set before
# HIP initialises; harmless at N = 1 (measured: 4.87 vs 4.87 ms).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

CLIP_SAMPLES = 160000
CLIP_SECONDS = 10.0
BATCH_PER_GPU = 32
MFMA_BF16_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16
MFMA_FP8_PEAK_TFLOPS = 5000.0       # same guide: ~5 PF dense fp8 (the MXFP8 launches of precision="fp8" answer to THIS peak)
ENCODER_GEMMS = ("gemm_qkv", "gemm_out", "gemm_ffn1", "gemm_ffn2")
HBM_PEAK_GBS = 8000.0               # same guide: 8 TB/s spec (6.3 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exchange", action="store_true",
                    help="N>1: time resident shards (no data-path collective) as `value`; the exchange figure moves to `exchange`")
    ap.add_argument("--no-overlap", action="store_true", help="one batch in flight, segmenter on the forward stream")
    ap.add_argument("--inflight", type=int, default=2, help="batches in flight (independent handles/streams)")
    ap.add_argument("--precision", choices=["bf16", "fp8", "fp32", "fp16", "mixed16", "split16"], default="bf16",
                    help="bf16 = BASELINE configs[1] (the headline); fp8 = configs[4] (MXFP8 weight GEMMs); fp32 = the exact "
                         "parity mode (f32 MFMA), reported so that its cost is a number")
    ap.add_argument("--graph", action="store_true", help="replay the forward from a captured hipGraph (small, launch-bound batches)")
    ap.add_argument("--ragged", action="store_true",
                    help="clip lengths U[2 s, clip-seconds] (seeded), padded to the batch maximum like sylber.py:93-118; value "
                         "counts VALID audio only -> the padding overhead of the reference's batching contract (not the headline)")
    ap.add_argument("--clip-seconds", type=float, default=CLIP_SECONDS,
                    help="clip length; 10 = BASELINE configs[1] (default), 60 with --batch 8 = configs[3] (long-form)")
    ap.add_argument("--exchange-selftest", action="store_true",
                    help="N = 1 only: run the N>1 code path (one-rank RCCL communicator, scatter / gather to self inside every "
                         "step) -- the only way to exercise it on a one-GPU box; the line is labelled accordingly")
    ap.add_argument("--exchange-timeout", type=int, default=180,
                    help="watchdog for the exchange phase (s): on expiry rank 0 prints the resident-shard line with `exchange_error`")
    ap.add_argument("--ingest", choices=["scatter", "per-rank"], default="scatter",
                    help="N>1 exchange step: 'scatter' = the job's clips live on rank 0 and are scattered over RCCL every step "
                         "(default, BASELINE configs[2]); 'per-rank' = every rank's shard lives in its own page-locked host memory "
                         "and crosses that GPU's own PCIe link every step (SURVEY.md §8(e)), gather unchanged")
    ap.add_argument("--gather", choices=["root", "none"], default="root",
                    help="N>1 exchange step: 'root' = hidden states, tables, counts and features are gathered on rank 0 over RCCL every step "
                         "(default, north_star's scatter/gather); 'none' = results stay on the rank that computed them (a corpus job whose "
                         "ranks write their own shards).  With 'root' the 'none' figure is measured too and reported as `results_left_sharded`")
    ap.add_argument("--fuse-ln", choices=["auto", "on", "off"], default="auto",
                    help="attention out-projection + LayerNorm as one launch (SYLBER_OPT_FUSE_OUTPROJ_LN); A/B switch")
    ap.add_argument("--gemm-tile", type=int, default=-1, help="force one GEMM tile id for every launch that has it (SYLBER_OPT_GEMM_TILE); A/B switch")
    ap.add_argument("--cu-split", choices=["none", "xcd", "half"], default="none",
                    help="A/B: give each in-flight batch's main stream its own CUs (hipExtStreamCreateWithCUMask): 'xcd' = whole XCDs "
                         "(mask bit i -> XCD i %% 8), 'half' = every other CU of every XCD; 2 batches in flight only")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="sylber_set_option(KEY, VALUE) on every handle (integers; A/B switches, include/sylber_hip.h)")
    ap.add_argument("--conv0-valu", action="store_true", help="conv layer 0 on the VALU kernel (SYLBER_OPT_CONV0_VALU); A/B switch")
    ap.add_argument("--out-sets", type=int, default=0,
                    help="output buffer sets of the resident steps (default = 2 x batches in flight; minimum = batches in flight): with "
                         "only one set per handle, a forward waits for the boundary detection of the step that last used its handle")
    ap.add_argument("--plain-streams", action="store_true", help="A/B switch: take the pipeline's streams from the pool without the hardware-queue probe")
    ap.add_argument("--skip-segment", action="store_true",
                    help="A/B switch: leave boundary detection out of the timed steps (the line is marked `invalid`)")
    ap.add_argument("--no-api", action="store_true", help="skip the API-level (PCIe-inclusive) Segmenter.__call__ timing")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` block (BASELINE configs[3] long-form, configs[4] fp8 and the split16 parity mode, "
                         "10 steps each in this same process after the headline run; N = 1, default workload only)")
    ap.add_argument("--exchange-root-copies", action="store_true", help="A/B: root's own share goes through the collectives' device-to-device copies as in rounds 2-6a "
                    "(default since round 6: root computes on a view of its chunk and writes its results straight into the gathered tensors; ShardedSegmenter.inplace_root)")
    ap.add_argument("--exchange-lookahead", type=int, default=0, help="A/B: batches ahead the input exchange is issued (0 = default = engines in flight; 1 = round 5)")
    ap.add_argument("--exchange-ingest-stream", choices=["default", "own", "engine"], default="default",
                    help="A/B: the input exchange under its own stream (default) or under the consuming engine's stream (round 5)")
    ap.add_argument("--gc-in-timing", action="store_true", help="A/B: leave Python's cyclic garbage collector enabled inside the timed regions")
    ap.add_argument("--exchange-side-delay", type=int, default=-1, help="A/B: iterations by which run_stream issues a batch's side-stream work late (default 1; 0 = rounds 2-5)")
    ap.add_argument("--exchange-consumer-stream", action="store_true",
                    help="A/B: the exchange loop's consumer (the waits for each batch's gather, the events that time the steps) under a stream of its own "
                         "(ShardedSegmenter.consumer_stream; rounds 6a) instead of the process's current stream (default since the second half of round 6: with "
                         "root's share in place the dedicated stream costs +5-7 %% against +2.6-3.3 %%, profiles/r06_exchange.md section 7)")
    ap.add_argument("--exchange-default-stream", action="store_true", help="(the default now; kept so that older command lines still parse)")
    ap.add_argument("--exchange-fresh-results", action="store_true",
                    help="A/B: the gathered tensors of every step freshly allocated (run_stream's default) instead of from its buffer ring "
                         "(reuse_results=True: valid until 2 x engines - 1 further batches have been yielded)")
    ap.add_argument("--no-exchange-rehearsal", action="store_true",
                    help="skip the one-rank RCCL self-test child of `other_configs` (the N > 1 code path on this one GPU)")
    ap.add_argument("--agreement-clips", type=int, default=0,
                    help="also report bf16-vs-fp32 segment agreement on this many synthetic clips (fp32 parity mode as truth)")
    return ap.parse_args()


def maybe_spawn(args):
    """`python bench.py --gpus N` without a launcher: become N ranks."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible to PyTorch-ROCm; refusing to "
                         "oversubscribe (one rank per GPU)\n" % (args.gpus, have))
        sys.exit(2)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def gemm_flops_per_forward(B: int, n_samples: int = CLIP_SAMPLES) -> dict:
    """Algorithmic FLOPs (2*MAC) of the launches of the bf16 MFMA GEMM kernel family
    in one forward over B clips of n_samples, keyed by the launch names api.hip uses."""
    K = [10, 3, 3, 3, 3, 2, 2]
    S = [5, 2, 2, 2, 2, 2, 2]
    L, n = [], n_samples
    for k, st in zip(K, S):
        n = (n - k) // st + 1
        L.append(n)
    T = L[-1]
    f = {}
    for i in range(1, 7):
        f[f"gemm_conv{i}"] = 2.0 * L[i] * 512 * 512 * K[i] * B
    f["gemm_proj"] = 2.0 * T * 512 * 768 * B
    f["gemm_qkv"] = 9 * 3 * 2.0 * T * 768 * 768 * B
    f["gemm_out"] = 9 * 2.0 * T * 768 * 768 * B
    f["gemm_ffn1"] = 9 * 2.0 * T * 768 * 3072 * B
    f["gemm_ffn2"] = 9 * 2.0 * T * 768 * 3072 * B
    return f


# launch names that share FLOPs with an entry of gemm_flops_per_forward (older / alternative launch splits)
GEMM_ALIASES = {"gemm_qkv": ("gemm_qkv", "gemm_qk", "gemm_v")}


def csrc_sha16() -> str:
    """Identity of the kernel sources a committed PMC figure was collected with."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sylber_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    for g in ("gen_gemm_asm.py", "gen_attn_asm.py"):        # the inline-asm loops are generated at build time: the generators ARE kernel source
        gp = os.path.join(ROOT, "tools", g)
        if os.path.exists(gp):
            h.update(g.encode())
            h.update(open(gp, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(sd, seconds_budget=25.0):
    """The CPU restatement of the same path (oracle/: torch fp32 ops in the reference's order +
    C get_segment), timed on this box's host cores on a bounded sample of the same workload."""
    import torch
    from oracle.segmenter_ref import SegmenterRef
    from sylber_amd.synth import noise_batch
    ref = SegmenterRef(sd)
    B = 4
    wavs = [w[None, :] for w in noise_batch(B, CLIP_SAMPLES, seed=0)]
    # torch's CPU conv/GEMM stop scaling (and then regress badly) long before a 256-thread host is
    # full: probe a few thread counts on a short clip and time the best one
    probe = [w[:, :32000] for w in wavs]
    best, cores = None, 1
    for nt in sorted({min(n, os.cpu_count() or 1) for n in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        ref(probe, in_second=False)
        t0 = time.perf_counter()
        ref(probe, in_second=False)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, nt
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    ref(wavs, in_second=False)               # warm-up (also bounds the loop below)
    warm = time.perf_counter() - t0
    iters = max(1, min(5, int(seconds_budget / max(warm, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(iters):
        ref(wavs, in_second=False)
    dt = (time.perf_counter() - t0) / iters
    return {"value": round(B * CLIP_SECONDS / dt, 2), "unit": "audio-sec/s", "cores": cores, "kind": "port",
            "sample": "%d iterations of batch %d x 10 s (same generator as the GPU workload), fp32, torch CPU ops + "
                      "C get_segment; %d torch threads = fastest of a {8,16,32,64} probe on a %d-cpu host"
                      % (iters, B, cores, os.cpu_count() or 1)}


def attention_roofline(kernels: dict, B: int, T: int, precision: str, layers: int = 9):
    """the attention core's own line: algorithmic FLOPs 4 B 12 T^2 64 per layer (0.765 GFLOP per 10 s clip and layer) over the
    summed duration of its launches; MXFP8 operands (precision="fp8") answer to the fp8 peak"""
    ms = kernels.get("attention", 0.0)
    if ms <= 0:
        return None
    fl = layers * 4.0 * B * 12 * T * T * 64
    peak = MFMA_FP8_PEAK_TFLOPS if precision == "fp8" else MFMA_BF16_PEAK_TFLOPS
    tf = fl / (ms * 1e-3) / 1e12
    return {"bound": "mfma (issue / power limited: profiles/r05_attention.md)", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s",
            "frac": round(tf / peak, 4), "ms_per_forward": round(ms, 4), "launches_per_forward": layers}


def roofline_by_peak(kernels: dict, B: int, clip_samples: int, precision: str) -> dict:
    """Per-dtype MFMA roofline of one forward: every GEMM launch is scored against the dense peak of the operand type it
    ACTUALLY runs on -- precision="fp8" runs q,k,v / out-proj / FFN1 / FFN2 on v_mfma_scale_f32_32x32x64_f8f6f4 (5 PF) and the
    conv stack + projection on the 16-bit MFMA (2.5 PF); every other 16-bit mode runs all of them on the 16-bit MFMA."""
    fl = gemm_flops_per_forward(B, clip_samples)
    ms_of = {k: sum(kernels.get(a, 0.0) for a in GEMM_ALIASES.get(k, (k,))) for k in fl}
    groups = {"conv_stack_and_projection": [k for k in fl if k not in ENCODER_GEMMS], "encoder_gemms": list(ENCODER_GEMMS)}
    out = {}
    for name, keys in groups.items():
        ms = sum(ms_of[k] for k in keys)
        if ms <= 0:
            continue
        f8 = precision == "fp8" and name == "encoder_gemms"
        peak = MFMA_FP8_PEAK_TFLOPS if f8 else MFMA_BF16_PEAK_TFLOPS
        tf = sum(fl[k] for k in keys) / (ms * 1e-3) / 1e12
        passes = 3 if precision == "split16" else 1          # hi.hi + lo.hi + hi.lo: three MFMA passes per algorithmic contraction
        out[name] = {"bound": "mfma", "operands": "mxfp8 (e4m3 + E8M0)" if f8 else ("f16 hi/lo pairs, 3 passes" if passes == 3 else "16-bit"),
                     "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                     "ms_per_forward": round(ms, 4),
                     "per_launch_tflops": {k: round(fl[k] / (ms_of[k] * 1e-3) / 1e12, 1) for k in keys if ms_of[k] > 0}}
        if passes == 3:
            out[name]["issued_frac"] = round(3 * tf / peak, 4)
    return out


def exchange_rehearsal(timeout_s: int = 150) -> dict:
    """BASELINE configs[2] cannot be measured on a one-GPU box; what CAN be is the N > 1 code path itself: a child process runs this script
    with --exchange-selftest (one-rank RCCL communicator, scatter / gather to self inside every step) and its figure is reported beside the
    resident one of the same child -- what the collectives cost the persistent GEMM workgroups they share the chip with, with zero
    remote bytes.  A child, not this process: a communicator that hangs must not take the headline line with it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--exchange-selftest", "--no-api", "--no-cpu-baseline", "--no-other-configs", "--steps", "10",
           "--warmup", "3", "--exchange-timeout", str(max(30, timeout_s - 60))]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "the self-test child did not finish within %d s" % timeout_s}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "self-test child exited %d: %s" % (r.returncode, (r.stderr or "")[-300:])}
    d = json.loads(lines[-1])
    res = d.get("resident_shards") or {}
    out = {"value": d.get("value"), "unit": "audio-sec/s", "ms_per_step": d.get("ms_per_step"), "steps": d.get("steps"),
           "resident_ms_per_step_same_process": res.get("ms_per_step"),
           "what": "one-rank RCCL self-test of the N > 1 step (ShardedSegmenter.run_stream: scatter + asynchronous gather to self in every step, "
                   "two engines); NOT a multi-GPU measurement", "child_wall_s": round(time.perf_counter() - t0, 1)}
    if res.get("ms_per_step") and d.get("ms_per_step"):
        out["overhead_vs_resident"] = round(d["ms_per_step"] / res["ms_per_step"] - 1.0, 4)
    for k in ("exchange_detail", "exchange_error", "results_left_sharded"):
        if d.get(k) is not None:
            out[k] = d[k]
    return out


def measure_other_config(torch, dev, sd, streams4, precision: str, B: int, clip_seconds: float, steps: int = 10, warmup: int = 3,
                         agreement_clips: int = 0) -> dict:
    """One more BASELINE configuration in THIS process, timed like the headline (two batches in flight on independent handles,
    boundary detection on side streams, inputs resident in HBM, device-wide synchronize on both sides of exactly `steps` steps),
    followed by a profiling pass for its own roofline."""
    from sylber_amd import HubertEncoderHIP
    from sylber_amd.synth import noise_batch
    clip_samples = int(round(clip_seconds * 16000))
    encs = [HubertEncoderHIP(sd, device=str(dev), precision=precision) for _ in range(2)]
    for e_ in encs:
        e_.set_batches_in_flight(2)
    T_frames = encs[0].num_frames(clip_samples)
    batch = noise_batch(B, clip_samples, seed=0).to(dev)
    mains, sides = streams4[:2], streams4[2:4]
    bufs = [(torch.empty(B, T_frames, 768, device=dev),
             (torch.empty(B, T_frames, 2, dtype=torch.int64, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
              torch.empty(B, T_frames, 768, device=dev))) for _ in range(4)]
    seg_done = [None] * 4
    st = {"i": 0}

    def run(n):
        for _ in range(n):
            k, ks = st["i"] % 2, st["i"] % 4
            st["i"] += 1
            hidden, seg_out = bufs[ks]
            with torch.cuda.stream(mains[k]):
                if seg_done[ks] is not None:
                    mains[k].wait_event(seg_done[ks])
                encs[k].forward(batch, None, out=hidden)
                ready = torch.cuda.Event()
                ready.record(mains[k])
            with torch.cuda.stream(sides[k]):
                sides[k].wait_event(ready)
                encs[k].segment(hidden, 2.6, 0.8, out=seg_out)
                ev = torch.cuda.Event()
                ev.record(sides[k])
            seg_done[ks] = ev

    run(warmup)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    enc = encs[0]
    enc.set_batches_in_flight(1)                      # the per-kernel pass: this handle alone on the chip (see main())
    enc.forward(batch, None)
    enc.set_profiling(True)
    nprof = 3
    for _ in range(nprof):
        h = enc.forward(batch, None)
        enc.segment(h, 2.6, 0.8)
    torch.cuda.synchronize(dev)
    kernels = {k: round(v / nprof, 4) for k, v in enc.get_profile().items()}
    enc.set_profiling(False)
    out = {"value": round(B * clip_seconds * steps / dt, 1), "unit": "audio-sec/s", "ms_per_step": round(1e3 * dt / steps, 3),
           "steps": steps, "warmup": warmup, "precision": precision, "batch": B, "clip_seconds": clip_seconds, "frames_per_clip": T_frames,
           "roofline": dict(roofline_by_peak(kernels, B, clip_samples, precision), attention=attention_roofline(kernels, B, T_frames, precision)),
           "kernel_ms_per_forward": {k: kernels[k] for k in ("attention", "conv0_gn_gelu", "posconv", "layernorm", "segment") if k in kernels},
           "workspace_gb": round(enc.workspace_bytes() / 2 ** 30, 2)}
    if precision in ("fp16", "mixed16"):
        # does anything come near the format's limit?  (the audit launches run outside every timed region)
        enc.fp16_audit(start=True)
        enc.forward(batch, None)
        aud = enc.fp16_audit()
        enc.set_option(10, 0)
        out["fp16_audit"] = {"saturated_values": sum(v["saturated"] for v in aud.values()), "largest_magnitude": max([v["max_abs"] for v in aud.values()] or [0.0]),
                             "limit": 65504.0, "stage_of_largest": max(aud, key=lambda k: aud[k]["max_abs"]) if aud else None,
                             "what": "sylber_get_fp16_audit over one forward of this workload (synthetic weights): values clamped at +-65504 and the largest "
                                     "16-bit activation magnitude over all stages"}
    if agreement_clips > 0:
        from sylber_amd.agreement import segment_agreement
        out["segment_agreement"] = segment_agreement(sd, enc, agreement_clips, device=str(dev))
    del encs, enc, bufs, batch
    torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    maybe_spawn(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s)\n" % (args.gpus, world))
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback on the product path)"
    rccl_ranks = 1
    selftest = args.exchange_selftest and world == 1
    # RCCL's own warnings go to a per-rank file (unless the caller configured NCCL_DEBUG already): a failed or hung exchange
    # quotes its tail in `exchange_error`
    nccl_log = None
    if (selftest or world > 1) and "NCCL_DEBUG" not in os.environ:
        import tempfile
        nccl_log = os.path.join(tempfile.gettempdir(), "sylber_bench_rccl_%d_%%h_%%p.log" % os.getpid())
        os.environ["NCCL_DEBUG"] = "WARN"
        os.environ["NCCL_DEBUG_FILE"] = nccl_log

    def rccl_warnings(limit=600):
        if not nccl_log:
            return ""
        import glob
        txt = ""
        for f in sorted(glob.glob(nccl_log.replace("%h", "*").replace("%p", "*"))):
            try:
                txt += open(f, errors="replace").read()
            except OSError:
                pass
        txt = " ".join(ln.strip() for ln in txt.splitlines() if "WARN" in ln or "error" in ln.lower())
        return txt[-limit:]
    if selftest:
        import datetime
        import socket
        with socket.socket() as _s:
            _s.bind(("127.0.0.1", 0))
            _port = _s.getsockname()[1]
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank),
                                timeout=datetime.timedelta(seconds=300))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # SYLBER_DIST_BACKEND=gloo is a development aid only (several ranks sharing the one GPU of a development box);
        # one rank per GPU is enforced otherwise
        backend = os.environ.get("SYLBER_DIST_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if backend == "nccl" and ndev < world:
            sys.stderr.write("bench.py: %d ranks but %d visible GPU(s)\n" % (world, ndev))
            sys.exit(2)
        local_rank %= max(1, ndev)
        torch.cuda.set_device(local_rank)
        import datetime
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=300))
            rccl_ranks = dist.get_world_size()
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            rccl_ranks = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from sylber_amd import HubertEncoderHIP
    from sylber_amd.dist import ShardedSegmenter
    from sylber_amd.synth import noise_batch
    from sylber_amd.weights import synthetic_state_dict

    clip_seconds = args.clip_seconds
    clip_samples = int(round(clip_seconds * 16000))
    sd = synthetic_state_dict(0)
    B = args.batch
    NPIPE = 1 if args.no_overlap else max(1, args.inflight)
    encs = [HubertEncoderHIP(sd, device=str(dev), precision=args.precision) for _ in range(NPIPE)]
    enc = encs[0]
    for e_ in encs:                                  # the timed steps keep NPIPE batches in flight: the handles share the chip (GEMM tile choice only)
        e_.set_batches_in_flight(NPIPE)
    if args.graph:
        for e_ in encs:
            e_.set_graph_mode(True)
    if args.conv0_valu:
        for e_ in encs:
            e_.set_option(5, 1)
    if args.gemm_tile >= 0:
        for e_ in encs:
            e_.set_option(1, args.gemm_tile)
    for kv in args.opt:
        key, val = (int(x) for x in kv.split("="))
        for e_ in encs:
            e_.set_option(key, val)
    if args.fuse_ln != "auto":
        for e_ in encs:
            e_.set_option(4, 1 if args.fuse_ln == "on" else -1)
    T_frames = enc.num_frames(clip_samples)
    on_cpu_group = world > 1 and dist.get_backend() != "nccl"

    # inputs resident in HBM before the timed region: rank r holds clips [r*B, (r+1)*B) of the seeded job; for the
    # exchange the root additionally holds the whole job (N x B clips) and scatters it every step
    # (torch.Generator().manual_seed(0), randn(B, N) -- rank r of a sharded job draws seed r)
    my_batch = noise_batch(B, clip_samples, seed=rank).to(dev)
    lengths = None
    valid_seconds = B * clip_seconds
    if args.ragged:
        import numpy as _np
        rng = _np.random.default_rng(2000 + rank)
        lengths = [int(x) for x in rng.integers(2 * 16000, clip_samples + 1, B)]
        lengths[0] = clip_samples                              # the batch maximum stays the nominal clip length
        for i, n in enumerate(lengths):
            my_batch[i, n:] = 0.0                              # right zero padding (sylber.py:104-106)
        valid_seconds = sum(lengths) / 16000.0

    def barrier():
        torch.cuda.synchronize(dev)               # device-wide: every stream of this rank has drained
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if on_cpu_group else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(x):
        """[x of rank 0, x of rank 1, ...] on every rank"""
        if world == 1:
            return [float(x)]
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if on_cpu_group else dev)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    per_rank = {}

    def timed(run_steps, warmup=None):
        """contract: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize, MAX over ranks.
        run_steps(n) returns one timing-enabled event per step (recorded where the step's last kernel was issued)."""
        import gc
        run_steps(args.warmup if warmup is None else warmup)
        barrier()
        # no cyclic garbage collection inside the timed region (A/B: --gc-in-timing): the exchange loop creates a few hundred container objects per
        # step, which triggers a full collection every few steps, and a full collection of this process's heap (torch, numpy, the weights' Python
        # side) stops the host thread for 40-120 ms -- the GPU drains and idles (profiles/r06_exchange.md: the "erratic" one-rank self-test)
        if not args.gc_in_timing:
            gc.collect()
            gc.disable()
        try:
            t0 = time.perf_counter()
            evs = run_steps(args.steps)
            torch.cuda.synchronize(dev)
            mine = time.perf_counter() - t0              # this rank's own K steps (before the closing barrier)
        finally:
            gc.enable()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        per_rank["last"] = [round(1e3 * x / args.steps, 3) for x in all_ranks(mine)]
        # per-step time from the completion events: with P batches in flight completions arrive in bursts, so a step's
        # time is the distance to the completion P steps later, divided by P
        P = max(1, NPIPE)
        gaps = [evs[i].elapsed_time(evs[i + P]) / P for i in range(len(evs) - P)] if evs and len(evs) > P else []
        if os.environ.get("SYLBER_BENCH_DEBUG") and evs:
            sys.stderr.write("step completion times (ms after the first): %s\n" % [round(evs[0].elapsed_time(e), 2) for e in evs])
        return elapsed, (statistics.median(gaps) if gaps else None)

    # ---- resident shards: NPIPE encoder handles (own workspace + HIP stream) take the steps round-robin, so kernels
    # of consecutive batches overlap on the chip; boundary detection of batch i (one workgroup per utterance) runs on a
    # side stream.  Every step is one full pass over one B-clip batch; all work is complete before the closing
    # device synchronize of the timed region.
    # HIP streams are multiplexed onto 4 hardware queues; two streams on the same queue run one after the other, so the
    # streams of the pipeline are PROBED for concurrency (sylber_amd/streams.py) instead of taken as they come
    from sylber_amd.streams import concurrent_streams
    pool_st = concurrent_streams(2 * NPIPE, dev) if not args.plain_streams else [torch.cuda.Stream(device=dev) for _ in range(2 * NPIPE)]
    streams = pool_st[:NPIPE]
    if args.cu_split != "none" and NPIPE == 2:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        streams = []
        for k in range(2):
            bits = [(i % 8) // 4 == k if args.cu_split == "xcd" else (i // 8) % 2 == k for i in range(ncu)]
            words = (ctypes.c_uint32 * ((ncu + 31) // 32))()
            for i, b in enumerate(bits):
                if b:
                    words[i // 32] |= 1 << (i % 32)
            sp = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), len(words), words)
            assert rc == 0, "hipExtStreamCreateWithCUMask: %d" % rc
            streams.append(torch.cuda.ExternalStream(sp.value, device=dev))
    sides = pool_st[NPIPE:]
    # two output sets per batch in flight: with one, forward(i + NPIPE) has to wait for the boundary detection of step i (a
    # ~0.25 ms latency-bound kernel on 32 CUs) before it may overwrite hidden_states -- measured 4.87 -> 4.81 ms per step
    NSETS = max(NPIPE, args.out_sets if args.out_sets > 0 else 2 * NPIPE)
    bufs = [(torch.empty(B, T_frames, 768, device=dev),
             (torch.empty(B, T_frames, 2, dtype=torch.int64, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
              torch.empty(B, T_frames, 768, device=dev))) for _ in range(NSETS)]
    seg_done = [None] * NSETS
    state = {"i": 0}

    def resident_steps(n):
        evs = []
        for _ in range(n):
            k = state["i"] % NPIPE
            ks = state["i"] % NSETS
            state["i"] += 1
            hidden, seg_out = bufs[ks]
            main_s, side = streams[k], sides[k]
            with torch.cuda.stream(main_s):
                if seg_done[ks] is not None:
                    main_s.wait_event(seg_done[ks])     # the segmenter that last read this buffer set has finished
                encs[k].forward(my_batch, lengths, out=hidden)
                if args.no_overlap:
                    if not args.skip_segment:
                        encs[k].segment(hidden, 2.6, 0.8, out=seg_out)
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(main_s)
                    evs.append(ev)
                    continue
                ready = torch.cuda.Event()
                ready.record(main_s)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                if not args.skip_segment:
                    encs[k].segment(hidden, 2.6, 0.8, out=seg_out)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(side)
            seg_done[ks] = ev
            evs.append(ev)
        return evs

    # ---- exchange (N > 1 default): root scatter + gather over RCCL inside every step
    sharded = ShardedSegmenter(encs, always_collective=selftest, streams=pool_st if len(pool_st) >= 2 * len(encs) else None)
    if args.exchange_side_delay >= 0:
        sharded.side_delay = args.exchange_side_delay
    if args.exchange_lookahead > 0:
        sharded.lookahead = args.exchange_lookahead
    if args.exchange_root_copies:
        sharded.inplace_root = False
    if args.exchange_ingest_stream != "default":
        sharded.ingest_stream = args.exchange_ingest_stream == "own"
    root_batch = None
    if selftest:
        root_batch = my_batch
    if world > 1 and rank == 0:
        root_batch = torch.cat([noise_batch(B, clip_samples, seed=r) for r in range(world)], 0).to(dev)

    host_shard = None
    if (world > 1 or selftest) and args.ingest == "per-rank":
        host_shard = torch.empty(B, clip_samples, dtype=torch.float32, pin_memory=True)
        host_shard.copy_(my_batch)
        torch.cuda.synchronize(dev)

    gather_mode = {"m": args.gather}

    host_issue = {"s": 0.0, "n": 0}

    def exchange_steps(n):
        evs = []
        src = [root_batch] * n if rank == 0 else [None] * n
        t_issue0 = time.perf_counter()
        # consumed under a stream of its own: the default stream shares a hardware queue with one of the pipeline's streams more often than not, and
        # every batch handed over puts a wait for its gather into the consumer's stream (ShardedSegmenter.consumer_stream)
        import contextlib as _cl
        cons = sharded.consumer_stream() if args.exchange_consumer_stream else None
        with (torch.cuda.stream(cons) if cons is not None else _cl.nullcontext()):
            _exchange_loop(src, n, evs)
        host_issue["s"], host_issue["n"] = time.perf_counter() - t_issue0, n
        return evs

    def _exchange_loop(src, n, evs):
        for _o in sharded.run_stream(src, None, max_segments=min(T_frames, 192), ingest=args.ingest,
                                     host_shards=[host_shard] * n if host_shard is not None else None, gather=gather_mode["m"],
                                     reuse_results=not args.exchange_fresh_results):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))
            evs.append(ev)

    ex_txt = ("root scatter over RCCL in every step, results left on their ranks (gather=none)" if args.gather == "none" else
              "root scatter + gather over RCCL in every step (run_stream: gather(i) overlapped with compute(i+1))"
              if args.ingest == "scatter" else
              "per-rank H2D ingest over each GPU's own PCIe link + gather over RCCL in every step (run_stream: gather(i) "
              "overlapped with compute(i+1))")
    res_txt = "shards resident per rank, no data-path collective"

    def build_line(value_, elapsed_, med_, exchange_first_, roofline_=None, frontend_=None, cpu_=None, api_=None, kernels_=None,
                   seg_stats_=None):
        dtype = {"bf16": "bf16", "fp8": "bf16 + mxfp8 (e4m3, E8M0 block scales) weight GEMMs and attention core", "fp32": "f32", "fp16": "f16 (IEEE half operands, f32 accumulate)",
                 "mixed16": "f16 conv stack + bf16 encoder (f32 accumulate)",
                 "split16": "f16 hi/lo operand pairs, three MFMA passes per contraction (f32 accumulate, erf GELU)"}[args.precision]
        # which BASELINE.json configuration this run IS (the label follows the arguments, not the default)
        is10 = clip_samples == CLIP_SAMPLES and not args.ragged
        if is10 and B == BATCH_PER_GPU and args.precision == "bf16":
            cfg_name = "BASELINE.json configs[1]" if world == 1 else "BASELINE.json configs[2]: configs[1] sharded over %d GPUs" % world
        elif is10 and B == BATCH_PER_GPU and args.precision == "fp8":
            cfg_name = "BASELINE.json configs[4] (fp8) on the configs[1] batch"
        elif abs(clip_seconds - 60.0) < 1e-9 and B == 8 and not args.ragged:
            cfg_name = "BASELINE.json configs[3] (long-form)" + ("" if args.precision == "bf16" else ", precision %s" % args.precision)
        else:
            cfg_name = "not a BASELINE.json configuration: batch %d x %g s%s, precision %s" % (
                B, clip_seconds, " ragged" if args.ragged else "", args.precision)
        clips_txt = ("batched %g s clips" % clip_seconds) if not args.ragged else ("ragged clips up to %g s" % clip_seconds)
        line = {
            "metric": "audio-sec/s encoded (sylber_base, 16 kHz, %s)" % clips_txt,
            "value": round(value_, 1), "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed_ / args.steps, 3),
            "ms_per_step_median": None if med_ is None else round(med_, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "Segmenter forward (conv frontend + HuBERT-9L encoder + boundary detection + "
                                   "segment mean-pool), batch %d x %g s 16 kHz random waveforms per GPU, random-init "
                                   "sylber_base weights (%s)" % (B, clip_seconds, cfg_name),
                       "global_batch": world * B, "clip_seconds": clip_seconds,
                       "ragged": ("lengths U[2 s, %g s], %.1f valid s of %g padded s per batch" % (clip_seconds, valid_seconds, B * clip_seconds))
                                 if args.ragged else None, "frames_per_clip": T_frames,
                       "parallelism": "utterance-sharded x%d, %s%s" % (world, ex_txt if exchange_first_ else res_txt,
                                                                        " [one-rank RCCL self-test of the N>1 path]" if selftest else ""),
                       "rccl_ranks": rccl_ranks,
                       "pipelining": "%d batch(es) in flight on independent handles/streams%s" % (
                           NPIPE, "; segmenter on a side stream" if not args.no_overlap else "") + (
                           "" if (exchange_first_ or args.no_overlap) else ", %d output sets" % NSETS),
                       "gflop_per_clip": 124.65 if clip_samples == CLIP_SAMPLES else None},
            "roofline": roofline_, "roofline_frontend": frontend_, "cpu_baseline": cpu_, "api_level": api_,
            "kernel_ms_per_forward": kernels_ or {},
            "workspace_gb": round(enc.workspace_bytes() / 2 ** 30, 2), "segments_per_clip": seg_stats_,
        }
        return line

    # Resident shards are ALWAYS measured first; the exchange (the default N > 1 step) is measured after them under a
    # watchdog: the RCCL path cannot be exercised on the one-GPU development boxes beyond a one-rank group, so if its
    # first multi-GPU run hangs in a collective, rank 0 still prints a contract-complete line (resident-shard value,
    # "exchange_error" saying what happened) instead of losing the whole measurement.
    want_exchange = world > 1 or selftest                      # --no-exchange only swaps which figure is `value`
    r_elapsed, r_med = timed(resident_steps)
    r_per_rank = per_rank.get("last")
    x_info = None
    total_audio = world * valid_seconds * args.steps           # (ragged: rank 0's draw stands for every rank)
    elapsed, med_ms = r_elapsed, r_med
    exchange_first = False
    secondary = None
    sharded_none = None
    exchange_error = None
    if want_exchange:
        import threading

        xdone = {}

        def on_timeout():
            if rank == 0:
                if "exchange" in xdone and not args.no_exchange:
                    # the exchange itself was measured; what hung is the optional results-left-sharded pass behind it
                    xe, xm = xdone["exchange"]
                    fb = build_line(total_audio / xe, xe, xm, True)
                    fb["resident_shards"] = {"value": round(total_audio / r_elapsed, 1), "unit": "audio-sec/s", "ms_per_step": round(1e3 * r_elapsed / args.steps, 3)}
                    fb["results_left_sharded"] = {"error": "did not finish within the watchdog's %d s; rank 0 was in: %s" % (args.exchange_timeout, sharded.phase)}
                    sys.stdout.flush()
                    print(json.dumps(fb), flush=True)
                    os._exit(0)
                fb = build_line(total_audio / r_elapsed, r_elapsed, r_med, False)
                fb["exchange_error"] = ("exchange phase did not finish within %d s (watchdog); value = resident shards; rank 0 was in: %s"
                                        % (args.exchange_timeout, sharded.phase))
                w = rccl_warnings()
                if w:
                    fb["exchange_error"] += "; RCCL: " + w
                sys.stdout.flush()
                print(json.dumps(fb), flush=True)
            os._exit(0)

        dog = threading.Timer(args.exchange_timeout + (0 if rank == 0 else 5), on_timeout)
        dog.daemon = True
        dog.start()
        try:
            exchange_steps(args.warmup)                  # (its own warm-up, so that the counters below cover K steps only)
            sharded.reset_stats()
            ms0_ = torch.cuda.memory_stats(dev)
            x_elapsed, x_med = timed(exchange_steps, warmup=0)
            ms1_ = torch.cuda.memory_stats(dev)
            st_ = sharded.stats
            x_info = {"per_rank_ms_per_step": per_rank.get("last"),
                      "root_wait_ms_per_step": round(1e3 * st_["wait_s"] / max(st_["steps"], 1), 3),
                      "wait_ms_per_step_by_rank": [round(1e3 * x / max(st_["steps"], 1), 3) for x in all_ranks(st_["wait_s"])],
                      "scatter_bytes_per_step_root": st_["scatter_bytes"] // max(st_["steps"], 1),
                      "gather_bytes_per_step_root": st_["gather_bytes"] // max(st_["steps"], 1),
                      "h2d_bytes_per_step_rank0": st_["h2d_bytes"] // max(st_["steps"], 1),
                      "ingest": args.ingest, "root_share": "through the collectives' copies" if args.exchange_root_copies else "in place (no copy of root's own rows)",
                      "host_issue_ms_per_step_by_phase_rank0": {k_: round(1e3 * v_ / max(st_["steps"], 1), 3) for k_, v_ in st_["host_s"].items()},
                      # a hipMalloc inside the timed steps synchronises the device (the caching allocator could not reuse a block that another
                      # stream still holds): must be 0 in steady state
                      "device_mallocs_during_timing_rank0": int(ms1_.get("num_device_alloc", 0) - ms0_.get("num_device_alloc", 0)),
                      "alloc_retries_during_timing_rank0": int(ms1_.get("num_alloc_retries", 0) - ms0_.get("num_alloc_retries", 0)),
                      "reserved_gb_rank0": round(ms1_.get("reserved_bytes.all.current", 0) / 2 ** 30, 2)}
            if args.no_exchange and not selftest:
                secondary = ("exchange", x_elapsed, x_med)
            else:
                elapsed, med_ms = x_elapsed, x_med
                exchange_first = True
                secondary = ("resident_shards", r_elapsed, r_med)
            xdone["exchange"] = (x_elapsed, x_med)
            if args.gather == "root":
                # the same step with the results left on their ranks: what the root gather costs, and the realistic corpus deployment.
                # OPTIONAL: a failure (or a hang: the watchdog below) here must not cost the exchange figure measured above
                try:
                    gather_mode["m"] = "none"
                    sharded.phase = "results-left-sharded pass"
                    n_elapsed, n_med = timed(exchange_steps, warmup=1)
                    sharded_none = {"value": round(total_audio / n_elapsed, 1), "unit": "audio-sec/s",
                                    "ms_per_step": round(1e3 * n_elapsed / args.steps, 3),
                                    "ms_per_step_median": None if n_med is None else round(n_med, 3),
                                    "parallelism": "input scatter over RCCL in every step, results stay on the rank that computed them (gather=none)"}
                except Exception as e:  # noqa: BLE001
                    sharded_none = {"error": "%s: %s; rank %d was in: %s" % (type(e).__name__, e, rank, sharded.phase)}
                finally:
                    gather_mode["m"] = "root"
        except Exception as e:  # noqa: BLE001 - keep a measurable line; the failure is reported in the line itself
            exchange_error = "%s: %s; rank %d was in: %s" % (type(e).__name__, e, rank, sharded.phase)
            w = rccl_warnings()
            if w:
                exchange_error += "; RCCL: " + w
        finally:
            dog.cancel()
    value = total_audio / elapsed

    # ---- per-kernel device time with HIP events on the launch stream (separate pass: event records
    # perturb the launch stream slightly, so they are kept out of the throughput timing above)
    roofline = None
    kernels = {}
    frontend = None
    if rank == 0 and args.precision == "fp32":
        # the exact mode: every contraction on v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s dense, MI355X_MICROARCH.md)
        enc.set_profiling(True)
        nprof = 3
        for _ in range(nprof):
            h = enc.forward(my_batch, None)
            enc.segment(h, 2.6, 0.8)
        torch.cuda.synchronize(dev)
        prof = enc.get_profile()
        enc.set_profiling(False)
        kernels = {k: round(v / nprof, 4) for k, v in prof.items()}
        gemm_fl = sum(gemm_flops_per_forward(B, clip_samples).values())
        if kernels.get("gemm_f32"):
            tf = gemm_fl / (kernels["gemm_f32"] * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "gemm_f32_tiled_kernel (all 43 GEMM launches of the fp32 parity forward, v_mfma_f32_32x32x2_f32)",
                        "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4), "traffic": None,
                        "traffic_note": "no PMC pass for the parity mode", "flops_per_forward": gemm_fl,
                        "ms_per_forward": kernels["gemm_f32"]}
    if rank == 0 and args.precision != "fp32":
        # the per-kernel pass runs ONE handle with nothing else on the chip: it tells the library so (SYLBER_OPT_GEMM_MODEL: the tiles a
        # synchronous caller gets), where the timed steps above declared NPIPE batches in flight; `roofline.tile_selection` says which
        if not any(kv.startswith("12=") for kv in args.opt):
            enc.set_batches_in_flight(1)
            enc.forward(my_batch, None)                  # (first use of a tile: function attributes, instruction cache)
        enc.set_profiling(True)
        nprof = max(3, min(args.steps, 10))
        for _ in range(nprof):
            h = enc.forward(my_batch, None)
            enc.segment(h, 2.6, 0.8)
        torch.cuda.synchronize(dev)
        prof = enc.get_profile()
        enc.set_profiling(False)
        if not any(kv.startswith("12=") for kv in args.opt):
            enc.set_batches_in_flight(NPIPE)
        kernels = {k: round(v / nprof, 4) for k, v in prof.items()}      # ms per forward
        fl = gemm_flops_per_forward(B, clip_samples)
        ms_of = {k: sum(kernels.get(a, 0.0) for a in GEMM_ALIASES.get(k, (k,))) for k in fl}
        gemm_ms = sum(ms_of.values())
        gemm_fl = sum(fl.values())
        n_launch = sum(1 for k in kernels if k.startswith("gemm_conv") or k == "gemm_proj") + 9 * sum(
            1 for k in ("gemm_qkv", "gemm_qk", "gemm_v", "gemm_out", "gemm_ffn1", "gemm_ffn2") if k in kernels)
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        traffic, traffic_note = None, "no PMC pass committed for these kernel sources"
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))       # the newest round's PMC passes
        tpath = cands[-1] if cands else os.path.join(ROOT, "profiles", "none.json")
        tname = os.path.basename(tpath)
        if os.path.exists(tpath) and B == BATCH_PER_GPU and clip_samples == CLIP_SAMPLES and args.precision == "bf16":
            # HBM bytes per GEMM launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE x2 +
            # WRITE_SIZE, separate passes; tools/pmc_traffic.py).  PMC cannot be read live, so the figure is only
            # repeated when the kernel sources are the ones it was collected with.
            tj = json.load(open(tpath))
            if tj.get("csrc_sha16") == csrc_sha16():
                traffic = round(tj["gemm_family_bytes_per_launch"])
                traffic_note = "HBM bytes per launch, rocprofv3 PMC (profiles/%s), same kernel sources" % tname.replace(".json", ".md")
            else:
                traffic_note = "profiles/%s was collected with other kernel sources (csrc sha %s != %s)" % (
                    tname, tj.get("csrc_sha16"), csrc_sha16())
        roofline = {"bound": "mfma", "kernel": "bf16 MFMA GEMM family (all %d launches per forward: 6 implicit-GEMM convs, "
                    "projection, 9 x {qkv, out, ffn1, ffn2})" % n_launch,
                    "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "avg_launch_ms": round(gemm_ms / max(n_launch, 1), 4), "flops_per_forward": gemm_fl,
                    "per_launch_tflops": {k: round(fl[k] / (ms_of[k] * 1e-3) / 1e12, 1) for k in fl if ms_of[k] > 0}}
        roofline["tile_selection"] = ("per-kernel pass: one handle alone on the chip, GEMM tiles as the library picks them for one batch in flight "
                                      "(SYLBER_OPT_GEMM_MODEL 0); the timed steps keep %d batches in flight and declare it (model %s). Same kernels and "
                                      "bits, the tile of a launch may differ where the batch shape leaves a partial round" % (NPIPE, "5" if NPIPE >= 2 else "0"))
        roofline["by_operand_type"] = roofline_by_peak(kernels, B, clip_samples, args.precision)
        roofline["attention"] = attention_roofline(kernels, B, T_frames, args.precision)
        if args.precision == "fp8":
            roofline["peak_note"] = ("`frac` of this block divides the whole GEMM family by the 16-bit peak (2.5 PF); the MXFP8 launches are "
                                     "scored against the fp8 peak (5 PF) in by_operand_type.encoder_gemms")
        enc_keys = list(ENCODER_GEMMS)
        enc_ms = sum(ms_of[k] for k in enc_keys)
        if enc_ms > 0:
            enc_tf = sum(fl[k] for k in enc_keys) / (enc_ms * 1e-3) / 1e12
            enc_peak = MFMA_FP8_PEAK_TFLOPS if args.precision == "fp8" else MFMA_BF16_PEAK_TFLOPS
            roofline["encoder_gemms"] = {"achieved": round(enc_tf, 1), "peak": enc_peak, "frac": round(enc_tf / enc_peak, 4),
                                         "ms_per_forward": round(enc_ms, 4)}
        # the single dominant instantiation, the rows rocprofv3 prints as gemmc_bf16_kernel<0, 1, 0, *, 4> (csrc/gemm_asm16.hip, tile 47): the
        # hand-scheduled 256x256 kernel on eight waves with the GELU epilogue, K loop on v_mfma_f32_16x16x32 = conv1..conv5 + 9 x FFN1,
        # persistent with the next tile's first K step requested before the epilogue (rounds 3-6a: gemmb_bf16_kernel<0, 1, 0, 0, 2, 8, true, 0, 3>,
        # the same tile on v_mfma_f32_32x32x16: --opt 13=-1)
        if clip_samples == CLIP_SAMPLES and B == BATCH_PER_GPU and args.precision == "bf16" and args.gemm_tile < 0 and not args.opt:
            big = ["gemm_conv1", "gemm_conv2", "gemm_conv3", "gemm_conv4", "gemm_conv5", "gemm_ffn1"]
            n_big = 5 + 9
            ms_big = sum(kernels.get(k, 0.0) for k in big)
            fl_big = sum(fl[k] for k in big)
            if ms_big > 0:
                roofline["dominant_instantiation"] = {
                    "kernel": "gemmc_bf16_kernel<0, 1, 0, *, 4> (tile 47: 256x256, 8 waves x 128x64, generated K loop on v_mfma_f32_16x16x32 over 128-byte LDS rows, "
                              "GELU + bf16 epilogue, persistent; conv1-4 in the 3-tap K order)", "launches_per_forward": n_big,
                    "avg_launch_ms": round(ms_big / n_big, 4), "achieved": round(fl_big / (ms_big * 1e-3) / 1e12, 1),
                    "frac": round(fl_big / (ms_big * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
        # the HBM-bound end of the path (north_star: "achieved HBM GB/s on the conv frontend"): conv0 + GroupNorm + GELU
        # writes the channels-last bf16 activation once and reads the waveform once
        if kernels.get("conv0_gn_gelu"):
            rows0 = enc.padded_frames(clip_samples) * 64       # R_0 = Tp * 2^6 rows per utterance
            fe_bytes = B * (rows0 * 512 * 2 + clip_samples * 4)
            fe_gbs = fe_bytes / (kernels["conv0_gn_gelu"] * 1e-3) / 1e9
            frontend = {"bound": "hbm", "kernel": "conv0_mfma_kernel (Conv1d(1->512,k10,s5) taps on the matrix pipe + GroupNorm + GELU, 16-bit channels-last out)",
                        "achieved": round(fe_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fe_gbs / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": fe_bytes}

    # ---- API level, PCIe inclusive (never `value`): Segmenter.__call__ on host tensors -> numpy dicts
    api = None
    if rank == 0 and world == 1 and not args.no_api:
        from sylber_amd import Segmenter
        seg_api = Segmenter(model_ckpt=sd, device=str(dev), precision=args.precision)
        host_wavs = [w[None, :].clone() for w in noise_batch(B, clip_samples, seed=0)]
        seg_api(wav=host_wavs, in_second=True)
        torch.cuda.synchronize(dev)
        seg_api(wav=host_wavs, in_second=True)                  # second warm-up: the output pool reaches its steady state
        torch.cuda.synchronize(dev)
        n_api = 20
        t_api = []
        allocs0 = seg_api.out_pool.allocations
        for _ in range(n_api):
            t0 = time.perf_counter()
            seg_api(wav=host_wavs, in_second=True)
            t_api.append(time.perf_counter() - t0)
        if os.environ.get("SYLBER_BENCH_DEBUG"):
            sys.stderr.write("api call times (ms): %s\n" % [round(x * 1e3, 1) for x in t_api])
        dt = statistics.median(t_api)
        # the same batches through Segmenter.stream: padding + H2D of batch i + 1 and D2H + slicing of batch i - 1 under batch i's forward
        for _o in seg_api.stream([host_wavs] * 40, in_second=True):      # (warm-up: the page-locked output blocks are allocated one by one,
            pass                                                          #  tens of ms each, until the pool holds what the loop keeps in flight)
        n_stream = 40
        if os.environ.get("SYLBER_BENCH_DEBUG"):
            seg_api._trace = []
        t0 = time.perf_counter()
        for _o in seg_api.stream([host_wavs] * n_stream, in_second=True):
            pass
        dt_stream = (time.perf_counter() - t0) / n_stream
        if os.environ.get("SYLBER_BENCH_DEBUG"):
            evs_ = [m[2] for m in seg_api._trace if m[0] == "compute issued"]
            sys.stderr.write("stream GPU gaps (ms): %s\n" % [round(evs_[i].elapsed_time(evs_[i + 1]), 1) for i in range(len(evs_) - 1)])
            seg_api._trace = None
        del _o
        # the same call for a consumer of the tables / pooled features only (tokenisation, the resynthesis front half): outputs= without
        # "hidden_states" skips the 49 MB D2H and its page-locked block
        lean = Segmenter(model_ckpt=sd, device=str(dev), precision=args.precision, outputs=("segments", "segment_features"))
        for _ in range(2):
            lean(wav=host_wavs, in_second=True)
        t_lean = []
        for _ in range(n_api):
            t0 = time.perf_counter()
            lean(wav=host_wavs, in_second=True)
            t_lean.append(time.perf_counter() - t0)
        dt_lean = statistics.median(t_lean)
        del lean
        # the reference's own example: ONE clip through the synchronous call (BASELINE configs[0]'s shape, on the GPU): latency, not throughput
        for _ in range(3):
            seg_api(wav=host_wavs[0], in_second=True)
        t_one = []
        for _ in range(30):
            t0 = time.perf_counter()
            seg_api(wav=host_wavs[0], in_second=True)
            t_one.append(time.perf_counter() - t0)
        dt_one = statistics.median(t_one)
        api = {"value": round(B * clip_seconds / dt, 1), "unit": "audio-sec/s", "ms_per_call": round(dt * 1e3, 2),
               "single_clip": {"ms_per_call": round(dt_one * 1e3, 3), "ms_min": round(min(t_one) * 1e3, 3), "value": round(clip_seconds / dt_one, 1), "unit": "audio-sec/s",
                               "what": "Segmenter.__call__(wav=one %g s host tensor) -> dict: the reference's README usage; every GEMM of it is a launch of fewer tiles than CUs "
                                       "(small two-per-CU tiles, the 16-bit-output ones on eight waves; profiles/r06_small_tiles.md); 30 back-to-back calls" % clip_seconds},
               "without_hidden_states": {"value": round(B * clip_seconds / dt_lean, 1), "unit": "audio-sec/s", "ms_per_call": round(dt_lean * 1e3, 2),
                                         "what": "the same call with Segmenter(outputs=('segments', 'segment_features')): opt-in, the default returns the reference's three keys"},
               "stream": {"value": round(B * clip_seconds / dt_stream, 1), "unit": "audio-sec/s", "ms_per_batch": round(dt_stream * 1e3, 2),
                          "what": "Segmenter.stream over %d such batches (host tensors in, numpy dicts out): copies and host work of "
                                  "neighbouring batches overlap the forward" % n_stream},
               "ms_min": round(min(t_api) * 1e3, 2), "ms_median": round(dt * 1e3, 2), "ms_max": round(max(t_api) * 1e3, 2),
               "calls": n_api, "pinned_allocations_during_timing": seg_api.out_pool.allocations - allocs0,
               "what": "Segmenter.__call__(wav=[%d host tensors]) -> list of numpy dicts: one batched H2D from a pinned staging "
                       "ring, forward + segmentation, D2H of hidden states / segments / features into a leased page-locked "
                       "block (PinnedOutputPool), per-utterance slicing; %d back-to-back calls" % (B, n_api)}
        del seg_api

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd)

    agreement = None
    if rank == 0 and world == 1 and args.agreement_clips > 0 and args.precision != "fp32":
        from sylber_amd.agreement import segment_agreement
        agreement = segment_agreement(sd, enc, args.agreement_clips, device=str(dev))

    # ---- the other BASELINE configurations, timed by the same command (N = 1, default workload only): the driver runs ONE
    # command, so configs[3] (long-form), configs[4] (fp8) and the mode that meets north_star's "boundaries bit-identical"
    # (split16) are measured here, each with its own roofline -- fp8 launches against the fp8 peak
    other = None
    if (rank == 0 and world == 1 and not args.no_other_configs and args.precision == "bf16" and B == BATCH_PER_GPU
            and clip_samples == CLIP_SAMPLES and not args.ragged and not args.opt and args.gemm_tile < 0):
        other = {}
        for name, (prec, b_, secs, agree) in {"configs[3] long-form 8 x 60 s, bf16": ("bf16", 8, 60.0, 0),
                                              "configs[4] fp8 (MXFP8 attention + FFN / projection GEMMs), 32 x 10 s": ("fp8", BATCH_PER_GPU, CLIP_SECONDS, 0),
                                              "fp16 (IEEE half operands, same MFMA rate; 5x the table agreement of bf16), 32 x 10 s": ("fp16", BATCH_PER_GPU, CLIP_SECONDS, 64),
                                              "split16 (segment tables bit-identical to the fp32 reference), 32 x 10 s": ("split16", BATCH_PER_GPU, CLIP_SECONDS, 0)}.items():
            try:
                other[name] = measure_other_config(torch, dev, sd, pool_st if len(pool_st) >= 4 else concurrent_streams(4, dev), prec, b_, secs,
                                                   agreement_clips=agree)
            except Exception as e:  # noqa: BLE001 - the headline line must survive a failing side measurement
                other[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_exchange_rehearsal:
            torch.cuda.synchronize(dev)
            other["configs[2] rehearsal: the N > 1 step on a one-rank RCCL communicator, 32 x 10 s"] = exchange_rehearsal()

    # ---- box-speed normaliser: boxes of the pool differ by +-4-5 % (more than a round's gains), and every figure above moves with the box.
    # One fixed, shape-independent launch series of the same library -- the 4096^3 bf16 GEMM loop, hot operands -- lets a reader
    # normalise a slow box: value / box_speed is comparable across runs
    box_speed = None
    if rank == 0 and world == 1:
        try:
            import ctypes as _ct
            from sylber_amd import _lib as _l
            torch.cuda.synchronize(dev)
            ms_, ms47_ = _ct.c_float(), _ct.c_float()
            # the yardstick stays the kernel it has been since round 5 (tile 97: cfg + 1000000 keeps a 16-bit-output launch on the 32x32x16 kernels);
            # the same launch on the tile the library picks by itself since round 6 (47: the same geometry on v_mfma_f32_16x16x32) is reported beside it
            _l.check(_l.load().sylber_debug_gemm_bench(4096, 4096, 4096, 4096, 0, 0, 1000097, 30, _ct.byref(ms_)), "gemm_bench")
            _l.check(_l.load().sylber_debug_gemm_bench(4096, 4096, 4096, 4096, 0, 0, -1, 30, _ct.byref(ms47_)), "gemm_bench")
            box_speed = {"gemm_4096_cubed_tflops": round(2.0 * 4096 ** 3 / (ms_.value * 1e-3) / 1e12, 1), "us_per_launch": round(ms_.value * 1e3, 1),
                         "automatic_tile_tflops": round(2.0 * 4096 ** 3 / (ms47_.value * 1e-3) / 1e12, 1),
                         "what": "30 back-to-back launches of this library's 4096^3 bf16 GEMM on tile 97 (plain 16-bit epilogue, hot operands; the 32x32x16 "
                                 "kernel of rounds 3-6a, kept as THE per-box yardstick: round-5/6 boxes read 1290-1345 TF) right after the measurements above; "
                                 "automatic_tile_tflops = the same launch as the library runs it now (tile 47, v_mfma_f32_16x16x32)"}
        except Exception as e:  # noqa: BLE001
            box_speed = {"error": "%s: %s" % (type(e).__name__, e)}

    seg_stats = None
    if rank == 0:
        h = enc.forward(my_batch, None)
        _, nseg_t, _ = enc.segment(h, 2.6, 0.8)
        nn_ = nseg_t.float()
        seg_stats = {"mean": round(float(nn_.mean()), 1), "min": int(nn_.min()), "max": int(nn_.max())}
    if rank == 0:
        line = build_line(value, elapsed, med_ms, exchange_first, roofline, frontend, cpu, api, kernels, seg_stats)
        if secondary is not None:
            name, s_el, s_med = secondary
            line[name] = {"value": round(total_audio / s_el, 1), "unit": "audio-sec/s",
                          "ms_per_step": round(1e3 * s_el / args.steps, 3),
                          "ms_per_step_median": None if s_med is None else round(s_med, 3),
                          "parallelism": res_txt if name == "resident_shards" else ex_txt}
        if x_info is not None:
            line["exchange_detail"] = x_info
        if sharded_none is not None:
            line["results_left_sharded"] = sharded_none
        line["resident_per_rank_ms_per_step"] = r_per_rank
        if exchange_error is not None:
            line["exchange_error"] = exchange_error
        if agreement is not None:
            line["segment_agreement"] = agreement
        if box_speed is not None:
            line["box_speed"] = box_speed
        if other is not None:
            line["other_configs"] = other
        if args.skip_segment:
            line["invalid"] = "--skip-segment: boundary detection left out of the timed steps (A/B measurement only)"
    if world > 1 or selftest:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out last: the communicator is gone, and RCCL's NCCL_DEBUG=VERSION banner (C stdio,
        # otherwise flushed at exit, i.e. AFTER this line) is pushed out first
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
