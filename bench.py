#!/usr/bin/env python
"""Throughput benchmark of the Segmenter hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch of synthetic input: waveform batch resident in HBM
-> 7-layer conv frontend -> 9-layer HuBERT encoder -> boundary detection + segment mean-pool, outputs
(hidden_states, segments, segment_features) left in HBM.  Workload = BASELINE.json configs[1]:
32 x 10 s x 16 kHz random waveforms per GPU, synthetic seeded weights of the sylber_base geometry
(no network for the real checkpoint), bf16 MFMA compute with fp32 accumulation/residual stream.
Weak scaling: every rank processes its own resident 32-clip shard (cfg3 = 256 clips on 8 GPUs); the path
shards over utterances with no collective inside it, so the timed steps contain none (only the closing
barrier / max-reduce of the contract).  `--exchange` adds the root scatter of waveforms and the gather of all
outputs over RCCL to every step (sylber_amd/dist.py).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CLIP_SAMPLES = 160000
CLIP_SECONDS = 10.0
BATCH_PER_GPU = 32
MFMA_BF16_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16


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
    f["gemm_qk"] = 9 * 2 * 2.0 * T * 768 * 768 * B
    f["gemm_v"] = 9 * 2.0 * T * 768 * 768 * B
    f["gemm_out"] = 9 * 2.0 * T * 768 * 768 * B
    f["gemm_ffn1"] = 9 * 2.0 * T * 768 * 3072 * B
    f["gemm_ffn2"] = 9 * 2.0 * T * 768 * 3072 * B
    return f


def cpu_baseline(sd, seconds_budget=25.0):
    """The CPU restatement of the same path (oracle/: torch fp32 ops in the reference's order +
    C get_segment), timed on this box's host cores on a bounded sample of the same workload."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", action="store_true",
                    help="N>1: include the root scatter of waveforms and the gather of all outputs (RCCL) in every step")
    ap.add_argument("--no-overlap", action="store_true", help="run the segmenter on the forward stream (no pipelining)")
    ap.add_argument("--inflight", type=int, default=2, help="batches in flight (independent handles/streams)")
    ap.add_argument("--gemm-wg-per-cu", type=int, default=0,
                    help="4-wave GEMM launches: 0 = one workgroup per tile, k = persistent k x 256 workgroups")
    ap.add_argument("--precision", choices=["bf16", "fp8"], default="bf16",
                    help="bf16 = BASELINE configs[1] (the headline); fp8 = configs[4]: FFN GEMMs on MXFP8 operands (not the headline)")
    ap.add_argument("--graph", action="store_true", help="replay the forward from a captured hipGraph (small, launch-bound batches)")
    ap.add_argument("--ragged", action="store_true",
                    help="clip lengths U[2 s, clip-seconds] (seeded), padded to the batch maximum like sylber.py:93-118; value "
                         "counts VALID audio only -> the padding overhead of the reference's batching contract (not the headline)")
    ap.add_argument("--clip-seconds", type=float, default=CLIP_SECONDS,
                    help="clip length; 10 = BASELINE configs[1] (default), 60 with --batch 8 = configs[3] (long-form)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # SYLBER_DIST_BACKEND=gloo is a development aid: it lets the N>1 control flow (rendezvous, barriers,
        # max-over-ranks) be exercised with several ranks sharing the one GPU of a development box
        backend = os.environ.get("SYLBER_DIST_BACKEND", "nccl")
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            except Exception as e:  # noqa: BLE001 - the default line has no data-path collective: keep it measurable
                print("bench.py: RCCL initialisation failed (%s); control barrier / max-reduce fall back to gloo" % (e,),
                      file=sys.stderr, flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29511")) + 1)
                dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback on the product path)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from sylber_amd import HubertEncoderHIP
    from sylber_amd.dist import ShardedSegmenter
    from sylber_amd.synth import noise_batch
    from sylber_amd.weights import synthetic_state_dict

    clip_seconds = args.clip_seconds
    clip_samples = int(round(clip_seconds * 16000))
    sd = synthetic_state_dict(0)
    enc = HubertEncoderHIP(sd, device=str(dev), precision=args.precision)
    if args.gemm_wg_per_cu:
        enc.lib.sylber_debug_force_gemm_cfg(-200 - args.gemm_wg_per_cu)
    sharded = ShardedSegmenter(enc)
    B = args.batch
    exchange = world > 1 and args.exchange
    # inputs resident in HBM before the timed region: rank r holds clips [r*B, (r+1)*B) of the seeded job;
    # with the exchange enabled the root additionally holds the whole job and scatters it every step
    my_batch = noise_batch(B, clip_samples, seed=1000 + rank).to(dev)
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
    root_batch = None
    if exchange and rank == 0:
        root_batch = torch.cat([noise_batch(B, clip_samples, seed=1000 + r) for r in range(world)], 0).to(dev)

    # Pipelining across steps (a serving loop keeps more than one batch in flight): NPIPE encoder handles, each
    # with its own workspace and HIP stream, take the steps round-robin, so kernels of consecutive batches
    # overlap on the chip — the one-round launches (500 tiles on 512 workgroup slots) leave their prologue /
    # epilogue phases uncovered otherwise — and the boundary detection of batch i (one workgroup per utterance,
    # 32 of 256 CUs) runs on a side stream.  Every step is still one full pass over one 32-clip batch; all work
    # is complete before the closing device synchronize of the timed region.
    T_frames = enc.num_frames(clip_samples)
    NPIPE = 1 if (exchange or args.no_overlap) else args.inflight
    encs = [enc] + [HubertEncoderHIP(sd, device=str(dev), precision=args.precision) for _ in range(NPIPE - 1)]
    if args.graph:
        for e_ in encs:
            e_.set_graph_mode(True)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NPIPE)]
    sides = [torch.cuda.Stream(device=dev) for _ in range(NPIPE)]
    bufs = [(torch.empty(B, T_frames, 768, device=dev),
             (torch.empty(B, T_frames, 2, dtype=torch.int64, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
              torch.empty(B, T_frames, 768, device=dev))) for _ in range(NPIPE)]
    seg_done = [None] * NPIPE
    state = {"i": 0}

    def step():
        if exchange:
            return sharded.step(root_batch, None)
        if args.no_overlap:
            hidden = enc.forward(my_batch, lengths, out=bufs[0][0])
            return (hidden,) + tuple(enc.segment(hidden, 2.6, 0.8, out=bufs[0][1]))
        k = state["i"] % NPIPE
        state["i"] += 1
        hidden, seg_out = bufs[k]
        main, side = streams[k], sides[k]
        with torch.cuda.stream(main):
            if seg_done[k] is not None:
                main.wait_event(seg_done[k])      # the segmenter that last read this buffer set has finished
            encs[k].forward(my_batch, lengths, out=hidden)
            ready = torch.cuda.Event()
            ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            encs[k].segment(hidden, 2.6, 0.8, out=seg_out)
            ev = torch.cuda.Event()
            ev.record(side)
        seg_done[k] = ev
        return (hidden,) + seg_out

    def barrier():
        torch.cuda.synchronize(dev)               # device-wide: every stream of this rank has drained
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def run_steps(n):
        if exchange:
            # root scatter + gather inside every step, software-pipelined (dist.py run_stream): the gather of step i
            # rides under the compute of step i+1
            for _o in sharded.run_stream([root_batch] * n if rank == 0 else [None] * n, None, max_segments=160):
                pass
        else:
            for _ in range(n):
                step()

    run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_audio = world * valid_seconds * args.steps           # (ragged: rank 0's draw stands for every rank)
    value = total_audio / elapsed

    # ---- per-kernel device time with HIP events on the launch stream (separate pass: event records
    # perturb the launch stream slightly, so they are kept out of the throughput timing above)
    roofline = None
    kernels = {}
    if rank == 0:
        enc.set_profiling(True)
        nprof = max(3, min(args.steps, 10))
        for _ in range(nprof):
            h = enc.forward(my_batch, None)
            enc.segment(h, 2.6, 0.8)
        torch.cuda.synchronize(dev)
        prof = enc.get_profile()
        enc.set_profiling(False)
        kernels = {k: round(v / nprof, 4) for k, v in prof.items()}      # ms per forward
        fl = gemm_flops_per_forward(B, clip_samples)
        gemm_ms = sum(kernels.get(k, 0.0) for k in fl)
        gemm_fl = sum(fl.values())
        n_launch = 6 + 1 + 9 * 5
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tpath) and B == BATCH_PER_GPU and clip_samples == CLIP_SAMPLES:
            # HBM bytes per GEMM launch from the committed rocprofv3 PMC passes of this same command
            # (FETCH_SIZE x2 + WRITE_SIZE, separate passes; tools/pmc_traffic.py) -- PMC cannot be read live
            traffic = round(json.load(open(tpath))["gemm_family_bytes_per_launch"])
        roofline = {"bound": "mfma", "kernel": "gemm_bf16_kernel (all %d launches per forward: 6 implicit-GEMM convs, "
                    "projection, 9 x {qk, v, out, ffn1, ffn2})" % n_launch,
                    "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch, rocprofv3 PMC (profiles/r01_hbm_traffic.md); algorithmic minimum "
                                    "~160 MB per launch (8.3 GB of operands/outputs over 52 launches)",
                    "avg_launch_ms": round(gemm_ms / n_launch, 4), "flops_per_forward": gemm_fl,
                    "per_launch_tflops": {k: round(fl[k] / (kernels[k] * 1e-3) / 1e12, 1) for k in fl if kernels.get(k)}}
        # the single dominant instantiation (40 % of the device time): the 8-wave 256x256 kernel with the GELU epilogue =
        # conv1..conv5 + 9 x FFN1; its rocprofv3 row is "gemm8_bf16_kernel<4, 2, 2, 4, 0, 1>" (profiles/r01_kernel_stats_sequential.csv)
        if clip_samples == CLIP_SAMPLES and B == BATCH_PER_GPU and args.precision == "bf16":
            big = ["gemm_conv1", "gemm_conv2", "gemm_conv3", "gemm_conv4", "gemm_conv5", "gemm_ffn1"]
            n_big = 5 + 9
            ms_big = sum(kernels.get(k, 0.0) for k in big)
            fl_big = sum(fl[k] for k in big)
            if ms_big > 0:
                roofline["dominant_instantiation"] = {
                    "kernel": "gemm8_bf16_kernel<4, 2, 2, 4, 0, 1>", "launches_per_forward": n_big,
                    "avg_launch_ms": round(ms_big / n_big, 4), "achieved": round(fl_big / (ms_big * 1e-3) / 1e12, 1),
                    "frac": round(fl_big / (ms_big * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}

    # the HBM-bound end of the path (north_star: "achieved HBM GB/s on the conv frontend"): conv0 + GroupNorm + GELU
    # writes the channels-last bf16 activation once and reads the waveform once
    frontend = None
    if rank == 0 and kernels.get("conv0_gn_gelu"):
        rows0 = ((T_frames + 3) // 4 * 4) * 64                 # R_0 = Tp * 2^6 rows per utterance (Tp = frames rounded up to 4)
        fe_bytes = B * (rows0 * 512 * 2 + clip_samples * 4)
        fe_gbs = fe_bytes / (kernels["conv0_gn_gelu"] * 1e-3) / 1e9
        frontend = {"bound": "hbm", "kernel": "conv0_gn_gelu_kernel (Conv1d(1->512,k10,s5) + GroupNorm + GELU, bf16 channels-last out)",
                    "achieved": round(fe_gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(fe_gbs / 8000.0, 4),
                    "algorithmic_bytes_per_launch": fe_bytes}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sd)

    seg_stats = None
    if rank == 0:
        h = enc.forward(my_batch, None)
        _, nseg_t, _ = enc.segment(h, 2.6, 0.8)
        nn_ = nseg_t.float()
        seg_stats = {"mean": round(float(nn_.mean()), 1), "min": int(nn_.min()), "max": int(nn_.max())}
    if rank == 0:
        line = {
            "metric": "audio-sec/s encoded (sylber_base, 16 kHz, batched 10 s clips)",
            "value": round(value, 1), "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "bf16 + mxfp8 (e4m3, E8M0 block scales) FFN GEMMs",
            "data": "synthetic",
            "config": {"workload": "Segmenter forward (conv frontend + HuBERT-9L encoder + boundary detection + "
                                   "segment mean-pool), batch %d x %g s 16 kHz random waveforms per GPU, random-init "
                                   "sylber_base weights (BASELINE.json configs[1]%s)" % (B, clip_seconds, "; configs[2] sharding" if world > 1 else ""),
                       "global_batch": world * B, "clip_seconds": clip_seconds,
                       "ragged": ("lengths U[2 s, %g s], %.1f valid s of %g padded s per batch" % (clip_seconds, valid_seconds, B * clip_seconds))
                                 if args.ragged else None, "frames_per_clip": T_frames,
                       "parallelism": "utterance-sharded x%d, %s" % (world, "root scatter + gather over RCCL in every step, gather(i) overlapped with compute(i+1)" if exchange
                                                                      else "shards resident per rank, no data-path collective"),
                       "pipelining": "none" if (exchange or args.no_overlap) else
                                     "%d batches in flight on independent handles/streams; segmenter on a side stream" % NPIPE,
                       "gflop_per_clip": 124.65 if clip_samples == CLIP_SAMPLES else None},
            "roofline": roofline, "roofline_frontend": frontend, "cpu_baseline": cpu, "kernel_ms_per_forward": kernels,
            "workspace_gb": round(enc.workspace_bytes() / 2 ** 30, 2), "segments_per_clip": seg_stats,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
