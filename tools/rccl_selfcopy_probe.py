"""Does a one-rank c10d collective block the HOST until the GPU reaches it?  (round 6: the one-rank self-test of the N > 1 step ran at 6-8 ms per step against
4.8 resident although its copies take 72 us of GPU time per step: the host thread spends 2.9 ms per step inside the four gathers.)  Queues ~5 ms of GPU work on a
stream, then times the HOST duration of one operation issued behind it: an op that returns in microseconds is asynchronous, one that takes ~5 ms waited for the GPU."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = str(port)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
src = torch.randn(32, 499, 768, device=dev)
dst = torch.empty_like(src)
small = torch.randn(32, device=dev)
small_dst = torch.empty_like(small)
cyc = 10_000_000          # ~5 ms of spinning


def probe(name, fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            torch.cuda._sleep(cyc)
            t0 = time.perf_counter()
            w = fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        del w
    print("%-58s host ms: %s" % (name, " ".join("%.3f" % t for t in ts)), flush=True)


torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(side):
    torch.cuda._sleep(cyc)
torch.cuda.synchronize()
print("the queued GPU work alone: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
probe("torch copy_ 49 MB (non_blocking) on the stream", lambda: dst.copy_(src, non_blocking=True))
probe("dist.gather 49 MB async_op=True", lambda: dist.gather(src, [dst], dst=0, async_op=True))
probe("dist.gather 128 B async_op=True", lambda: dist.gather(small, [small_dst], dst=0, async_op=True))
probe("dist.scatter 49 MB (synchronous op)", lambda: dist.scatter(dst, [src], src=0))
probe("dist.all_reduce 49 MB async_op=True", lambda: dist.all_reduce(dst, async_op=True))
probe("dist.broadcast 128 B", lambda: dist.broadcast(small, src=0))
if hasattr(dist, "batch_isend_irecv"):
    probe("batch_isend_irecv to self 49 MB", lambda: dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]))

# several collectives back to back behind the same pending work (what one step of run_stream issues: four gathers)
def burst(name, mk, count=4, reps=4):
    for _ in range(reps):
        torch.cuda.synchronize()
        ts = []
        with torch.cuda.stream(side):
            torch.cuda._sleep(cyc)
            ws = []
            for j in range(count):
                t0 = time.perf_counter()
                ws.append(mk(j))
                ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        print("%-58s host ms per call: %s" % (name, " ".join("%.3f" % t for t in ts)), flush=True)


dsts = [torch.empty_like(src) for _ in range(4)]
burst("4 x dist.gather 49 MB async, same stream", lambda j: dist.gather(src, [dsts[j]], dst=0, async_op=True))
burst("4 x torch copy_ 49 MB, same stream", lambda j: dsts[j].copy_(src, non_blocking=True))
side2 = torch.cuda.Stream(device=dev)


def alt(j):
    with torch.cuda.stream(side if j % 2 == 0 else side2):
        return dist.gather(src, [dsts[j]], dst=0, async_op=True)


burst("4 x dist.gather 49 MB async, alternating two streams", alt)
burst("8 x dist.gather 128 B async, same stream", lambda j: dist.gather(small, [small_dst], dst=0, async_op=True), count=8)
dist.destroy_process_group()
