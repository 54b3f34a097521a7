// Are a kernel's stores visible to the NEXT kernel of the same stream when a second stream runs its own chain
// concurrently?  Each stream loops: K1 writes buf[i] = tag (one XCD mapping), K2 copies buf -> chk through the
// opposite workgroup -> address mapping (so the reader CU sits on another XCD than the writer), host checks chk.
// Buffers are private per stream.  A stale value = the previous iteration's tag.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/xq_visibility.hip -o tools/ubench/xq_visibility
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_write(unsigned* __restrict__ buf, size_t n, unsigned tag) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = tag + (unsigned)i;
}
__global__ void k_copy_rev(const unsigned* __restrict__ buf, unsigned* __restrict__ chk, size_t n) {
    const size_t nb = gridDim.x;
    const size_t b = nb - 1 - blockIdx.x;            // reversed block -> address mapping: another XCD reads what one wrote
    for (size_t i = b * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) chk[i] = buf[i];
}

// the same copy with the read done by LDS-DMA (global_load_lds_dwordx4), the way the GEMM kernels stage operands
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;
__global__ __launch_bounds__(256) void k_copy_rev_dma(const unsigned* __restrict__ buf, unsigned* __restrict__ chk, size_t n) {
    __shared__ __attribute__((aligned(1024))) unsigned stage[4][256];           // one 1-KiB piece per wave
    const size_t nb = gridDim.x;
    const size_t b = nb - 1 - blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (size_t i0 = b * 1024; i0 < n; i0 += nb * 1024) {                       // 1024 words = 4 KiB per block step
        const size_t base = i0 + (size_t)wave * 256;
        __builtin_amdgcn_global_load_lds((glb_vptr)(buf + base + lane * 4), (lds_vptr)&stage[wave][0], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 v = *(const uint4*)&stage[wave][lane * 4];
        *(uint4*)(chk + base + lane * 4) = v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

int main(int argc, char** argv) {
    const int nstreams = argc > 1 ? atoi(argv[1]) : 2;
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    const size_t n = (size_t)1 << 20;                 // 4 MB per buffer: small kernels that really overlap
    const int blocks = argc > 3 ? atoi(argv[3]) : 61;
    const int dma = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<hipStream_t> st(nstreams);
    std::vector<unsigned*> buf(nstreams), chk(nstreams);
    std::vector<std::vector<unsigned>> host(nstreams, std::vector<unsigned>(n));
    for (int s = 0; s < nstreams; ++s) {
        hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking);
        hipMalloc(&buf[s], n * 4); hipMalloc(&chk[s], n * 4);
        hipMemset(buf[s], 0, n * 4); hipMemset(chk[s], 0, n * 4);
    }
    hipDeviceSynchronize();
    long bad_words = 0, bad_iters = 0;
    for (int it = 1; it <= iters; ++it) {
        for (int s = 0; s < nstreams; ++s) {
            const unsigned tag = (unsigned)it * 0x01000000u + (unsigned)s * 0x00100000u;
            hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, st[s], buf[s], n, tag);
            if (dma) hipLaunchKernelGGL(k_copy_rev_dma, dim3(blocks), dim3(256), 0, st[s], buf[s], chk[s], n);
            else hipLaunchKernelGGL(k_copy_rev, dim3(blocks), dim3(256), 0, st[s], buf[s], chk[s], n);
        }
        for (int s = 0; s < nstreams; ++s) {
            hipMemcpyAsync(host[s].data(), chk[s], n * 4, hipMemcpyDeviceToHost, st[s]);
        }
        hipDeviceSynchronize();
        for (int s = 0; s < nstreams; ++s) {
            const unsigned tag = (unsigned)it * 0x01000000u + (unsigned)s * 0x00100000u;
            long b = 0; size_t first = 0;
            for (size_t i = 0; i < n; ++i) if (host[s][i] != tag + (unsigned)i) { if (!b) first = i; ++b; }
            if (b) { ++bad_iters; bad_words += b; if (bad_iters <= 5) printf("  iter %d stream %d: %ld stale words, first at %zu: got %08x want %08x\n", it, s, b, first, host[s][first], tag + (unsigned)first); }
        }
    }
    printf("dma %d streams %d blocks %d: %ld bad (stream,iter) pairs of %d, %ld stale words\n", dma, nstreams, blocks, bad_iters, iters * nstreams, bad_words);
    return 0;
}
