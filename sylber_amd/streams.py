"""HIP streams that really run concurrently.

ROCm multiplexes a process's HIP streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default); two streams that
land on the SAME hardware queue execute strictly one after the other, whatever the program's dependencies say.  Which streams
share a queue depends on the order in which the process first used its streams (measured on MI355X / ROCm 7.2,
tools/stream_queue_probe.py: of ten pool streams, {0,7} {1,6,null} {2,5,9} {3,4,8} serialise).  A two-batches-in-flight
pipeline whose two compute streams share a queue silently degenerates into one batch at a time: the first ``Segmenter`` of a
process ran its two half-batches back to back (8.0 ms per call), every later one overlapped them (7.1 ms).

``concurrent_streams`` therefore PROBES: it draws streams from PyTorch's pool and keeps those on which two spin kernels
(``torch.cuda._sleep``) finish in the time of one against every stream already kept.
"""
from __future__ import annotations

import time
from typing import List, Optional, Sequence

import torch

_PROBE_CYCLES = 250_000          # ~0.1 ms on MI355X: long against launch + synchronise jitter, short enough to be free


def _pair_ms(a: torch.cuda.Stream, b: Optional[torch.cuda.Stream], device) -> float:
    best = None
    for _ in range(3):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            torch.cuda._sleep(_PROBE_CYCLES)
        if b is not None:
            with torch.cuda.stream(b):
                torch.cuda._sleep(_PROBE_CYCLES)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    return best


# probe results are per process and device: which pool streams share a hardware queue does not change once they have been used
_PAIR = {}            # (device index, stream id a, stream id b) -> serialised?
_SETS = {}            # device index -> mutually independent streams found so far
_EXHAUSTED = {}       # device index -> a full draw found no further independent stream
_LEASES = {}          # (device index, stream id) -> how many callers hold this stream (concurrent_streams hands out the least used first)


def _index(device) -> int:
    """device index of ``device``; an index-less 'cuda' means the CURRENT device, not device 0"""
    d = torch.device(device)
    return torch.cuda.current_device() if d.index is None else d.index


def _key(a: torch.cuda.Stream, b: torch.cuda.Stream, device):
    ia, ib = int(a.cuda_stream), int(b.cuda_stream)
    return (_index(device), min(ia, ib), max(ia, ib))


def release_streams(streams: Sequence[torch.cuda.Stream], device=None) -> None:
    """give streams obtained from ``concurrent_streams`` back (their owner is done with them): the next caller gets them before it has to
    share a stream somebody still uses"""
    for s in streams:
        k = (_index(s.device if device is None else device), int(s.cuda_stream))
        if _LEASES.get(k, 0) > 0:
            _LEASES[k] -= 1


def serialised(a: torch.cuda.Stream, b: torch.cuda.Stream, device=None) -> bool:
    """True when kernels on ``a`` and ``b`` run one after the other (same hardware queue); probed once per pair and process"""
    device = a.device if device is None else device
    k = _key(a, b, device)
    if k not in _PAIR:
        # the spin kernel counts shader cycles: on a GPU that idled before this call the first launches run at a lower clock than
        # the later ones, so the single-stream time is taken on both sides of the pair (and after one discarded warm-up launch)
        _pair_ms(a, None, device)
        one = _pair_ms(a, None, device)
        pair = _pair_ms(a, b, device)
        one = min(one, _pair_ms(a, None, device))
        _PAIR[k] = pair > 1.5 * one
    return _PAIR[k]


def concurrent_streams(n: int, device, avoid: Sequence[torch.cuda.Stream] = (), candidates: int = 16) -> List[torch.cuda.Stream]:
    """``n`` streams on ``device`` that run concurrently with each other and with every stream in ``avoid``.  Best effort: with
    fewer hardware queues than requested (or without ``torch.cuda._sleep``) the remaining slots are filled with plain pool
    streams -- correct, just not concurrent.  The probe (timed spin kernels between device-wide synchronisations: it stalls
    whatever else the process has in flight, and a busy GPU can fool it) runs once per process and device; later calls are
    served from its result.  ``SYLBER_NO_STREAM_PROBE=1`` skips it altogether.

    The independent set is shared by the whole process, so two owners (two ``Segmenter.stream`` loops, a ``ShardedSegmenter`` beside
    them) would be handed the SAME streams and serialise against each other: streams are leased, the least-used ones go out first, and a
    stream is shared only when every independent one is already held (``release_streams`` returns them)."""
    import os
    if n <= 0:
        return []
    di = _index(device)
    device = torch.device("cuda", di)
    if not hasattr(torch.cuda, "_sleep") or os.environ.get("SYLBER_NO_STREAM_PROBE"):
        return [torch.cuda.Stream(device=device) for _ in range(n)]
    avoid = list(avoid)

    def lease(streams):
        for s_ in streams:
            _LEASES[(di, int(s_.cuda_stream))] = _LEASES.get((di, int(s_.cuda_stream)), 0) + 1
        return streams

    def by_use(streams):                                  # stable: equally used streams keep their probe order
        return sorted(streams, key=lambda s_: _LEASES.get((di, int(s_.cuda_stream)), 0))

    def unused(streams):
        return [s_ for s_ in streams if _LEASES.get((di, int(s_.cuda_stream)), 0) == 0]

    with torch.cuda.device(device):
        have = _SETS.setdefault(di, [])
        ok = [s for s in have if all(not serialised(s, o, device) for o in avoid)]
        if len(unused(ok)) >= n:
            return lease(unused(ok)[:n])
        if len(ok) >= n and _EXHAUSTED.get(di):
            return lease(by_use(ok)[:n])                  # every independent queue is known already: share the least used
        # (first call on this device, or more independent streams wanted than found so far): draw and probe pool streams
        drawn = [torch.cuda.Stream(device=device) for _ in range(max(candidates, n))]
        for s in drawn:                                   # a stream gets its hardware queue when it is first used
            with torch.cuda.stream(s):
                torch.cuda._sleep(1000)
        torch.cuda.synchronize(device)
        for s in drawn:
            if len(unused(ok)) >= n:
                break
            if any(s.cuda_stream == o.cuda_stream for o in have):
                continue
            if all(not serialised(s, o, device) for o in have):
                have.append(s)
                if all(not serialised(s, o, device) for o in avoid):
                    ok.append(s)
        else:
            _EXHAUSTED[di] = True                         # a whole draw without reaching n unused ones: no more hardware queues to find
        kept = (unused(ok) + [s for s in by_use(ok) if s not in unused(ok)])[:n]
        for s in drawn:                                   # not enough independent queues: fill up
            if len(kept) == n:
                break
            if s not in kept:
                kept.append(s)
    return lease(kept)
