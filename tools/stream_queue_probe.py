"""Which of a process's HIP streams share a hardware queue?  Pairwise: two spin kernels on streams i and j, time relative to one
kernel (1.0 = concurrent, 2.0 = serialised).  See sylber_amd/streams.py."""
import time, torch
torch.cuda.init()
ss = [torch.cuda.Stream() for _ in range(10)]
cyc = 400000
def pair(a, b):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(a): torch.cuda._sleep(cyc)
    if b is not None:
        with torch.cuda.stream(b): torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for s in ss: pair(s, None)
one = min(pair(ss[0], None) for _ in range(5))
print("one sleep kernel: %.3f ms" % one)
for i in range(10):
    row = []
    for j in range(10):
        if i == j: row.append("  . "); continue
        t = min(pair(ss[i], ss[j]) for _ in range(3))
        row.append("%4.1f" % (t / one))
    print(i, " ".join(row))
nul = torch.cuda.default_stream()
print("null vs", " ".join("%4.1f" % (min(pair(nul, s) for _ in range(3)) / one) for s in ss))
