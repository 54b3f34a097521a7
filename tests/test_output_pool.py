"""PinnedOutputPool (sylber_amd/segmenter.py): the leased host blocks behind the numpy results of Segmenter.__call__
(the reference's `.cpu().numpy()` hand-over, sylber/model/sylber.py:122-138).  Host logic only: pageable memory is injected
in place of page-locked memory, so this runs without a GPU; the Segmenter-level behaviour is in tests/test_gpu_e2e.py."""
import gc
import threading

import numpy as np
import torch

from sylber_amd.segmenter import PinnedOutputPool


def _pool(n=2):
    return PinnedOutputPool(max_leased=n, alloc=lambda cap: torch.empty(cap, dtype=torch.uint8))


def test_blocks_are_reused_and_allocations_stop_growing():
    pool = _pool(2)
    for i in range(10):
        owner, blk = pool.lease(3 << 20)
        owner[:4] = i
        assert owner.nbytes >= 3 << 20 and owner.nbytes % PinnedOutputPool.GRANULE == 0
        del owner, blk
        gc.collect()
    assert pool.allocations == 1 and pool.leased == 0


def test_results_stay_intact_while_a_view_is_alive():
    pool = _pool(2)
    owner, _ = pool.lease(1 << 20)
    view = owner[16:32].view(np.float32)          # what a caller keeps: a numpy view whose .base chain ends at the owner
    view[:] = 7.0
    del owner, _
    gc.collect()
    assert pool.leased == 1                       # the view keeps the block out
    other, _b = pool.lease(1 << 20)               # a later call gets ANOTHER block
    other[:] = 0
    assert np.all(view == 7.0) and pool.allocations == 2
    del view
    gc.collect()
    assert pool.leased == 1
    del other, _b
    gc.collect()
    assert pool.leased == 0


def test_fallback_beyond_max_leased_and_recovery():
    pool = _pool(2)
    a = pool.lease(100)
    b = pool.lease(100)
    assert pool.lease(100) is None                # the caller falls back to pageable copies
    del a
    gc.collect()
    c = pool.lease(100)
    assert c is not None and pool.allocations == 2
    del b, c
    gc.collect()
    assert pool.leased == 0


def test_too_small_block_is_dropped_not_hoarded():
    pool = _pool(1)
    a = pool.lease(1 << 20)
    del a
    gc.collect()
    b = pool.lease(4 << 20)                       # the free 1-MiB block cannot serve this: replaced, not kept beside it
    assert b is not None and b[0].nbytes >= 4 << 20 and pool.allocations == 2
    del b
    gc.collect()
    assert pool.leased == 0 and len(pool._free) == 1


def test_release_from_a_finalizer_inside_lease_does_not_deadlock():
    """owner arrays caught in a reference cycle are freed by a collection that can start at any allocation, e.g. inside
    lease() while it holds the pool's lock: the release path must not take that lock"""
    pool = _pool(4)

    class Box:
        pass
    box = Box()
    box.owner, box.blk = pool.lease(100)
    box.me = box                                   # cycle: only the cycle collector frees it
    del box
    real_alloc = pool._alloc

    def alloc_with_gc(cap):
        gc.collect()                               # the collection runs the finalizer on THIS thread (no lock held: allocation is outside it)
        return real_alloc(cap)
    pool._alloc = alloc_with_gc
    done = []

    def run():
        with pool._lock:                           # worst case: a finalizer firing while the lock is held
            gc.collect()
        done.append(pool.lease(200))
    t = threading.Thread(target=run)
    t.start()
    t.join(20)
    assert not t.is_alive(), "deadlock: _release waited for the pool lock"
    assert done and done[0] is not None and pool.leased == 1


def test_trim_after_a_temporarily_raised_budget():
    """Segmenter.stream raises max_leased by its batches in flight and restores it: the extra free blocks must not stay hoarded"""
    pool = _pool(2)
    pool.max_leased = 5
    leases = [pool.lease(1 << 20) for _ in range(5)]
    assert all(l is not None for l in leases) and pool.allocations == 5
    del leases
    gc.collect()
    assert pool.leased == 0 and len(pool._free) == 5
    pool.max_leased = 2
    pool.trim()
    assert len(pool._free) == 2
    a = pool.lease(1 << 20); b = pool.lease(1 << 20)
    assert a is not None and b is not None and pool.lease(1 << 20) is None and pool.allocations == 5


def test_segment_slot_sizing_decays_with_the_recent_batches():
    """Segmenter._note_segments: the block size follows the recent per-batch maximum (`_kcap_seen` only grew)"""
    import collections
    from sylber_amd.segmenter import Segmenter
    s = Segmenter.__new__(Segmenter)
    s._kcap_seen, s._kcap_recent = 128, collections.deque(maxlen=16)
    s._note_segments(40)
    assert s._kcap_seen == 128                     # floor
    s._note_segments(700)                          # one long-clip batch
    assert s._kcap_seen == 704
    for _ in range(16):
        s._note_segments(45)
    assert s._kcap_seen == 128                     # ... forgotten 16 batches later
