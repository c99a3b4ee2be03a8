// membench.hip -- measures what an HBM *read* stream can reach on this MI355X with the access shapes
// the demod kernel uses, so roofline.frac can be read against a measured ceiling as well as the
// 8 TB/s spec.  Not part of the product; build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o membench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef NTAUX
#define NTAUX 2      /* cache policy of the XCD-aware variant's loads: 2 = nt, as in the product */
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// (a) grid-stride 16 B/lane register loads, xor-reduced so nothing is dead
__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ p, size_t n16, unsigned* sink)
{
    unsigned acc = 0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// (b) one tile of `tile_bytes` per block, register loads (LOADS x 16 B per lane), no LDS
template <int LOADS>
__global__ void __launch_bounds__(256) k_tile_reg(const uint4* __restrict__ p, unsigned* sink)
{
    const uint4* src = p + (size_t)blockIdx.x * LOADS * 256 + threadIdx.x;
    uint4 v[LOADS];
#pragma unroll
    for (int l = 0; l < LOADS; ++l) v[l] = src[l * 256];
    unsigned acc = 0;
#pragma unroll
    for (int l = 0; l < LOADS; ++l) acc ^= v[l].x ^ v[l].y ^ v[l].z ^ v[l].w;
    if (acc == 0x12345678u) sink[0] = acc;
}

// (c) one tile per block through LDS-DMA + barrier + one LDS read pass (the demod kernel's skeleton)
template <int LOADS>
__global__ void __launch_bounds__(256) k_tile_dma(const unsigned char* __restrict__ p, unsigned* sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    const unsigned char* src = p + (size_t)blockIdx.x * LOADS * 4096 + 16u * tid;
#pragma unroll
    for (int l = 0; l < LOADS; ++l)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4096u * l),
                                         (__attribute__((address_space(3))) void*)(smem + 4096u * l + 1024u * wave), 16, 0, 0);
    __syncthreads();
    unsigned acc = 0;
    const unsigned* w = reinterpret_cast<const unsigned*>(smem);
#pragma unroll
    for (int l = 0; l < LOADS * 4; ++l) acc ^= w[tid * 5 % (LOADS * 1024) + l * 256 % 7];
    if (acc == 0x12345678u) sink[0] = acc;
}

// (c') the same with the XCD-aware block order of the product kernels: grid (8, tiles per eighth), blockIdx.x is the
// XCD (workgroups go to XCDs round-robin in dispatch order) and XCD k streams its own contiguous eighth of the buffer
template <int LOADS>
__global__ void __launch_bounds__(256) k_tile_dma_xcd(const unsigned char* __restrict__ p, unsigned* sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    const size_t tile = (size_t)blockIdx.x * gridDim.y + blockIdx.y;
    const unsigned char* src = p + tile * LOADS * 4096 + 16u * tid;
#pragma unroll
    for (int l = 0; l < LOADS; ++l)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4096u * l),
                                         (__attribute__((address_space(3))) void*)(smem + 4096u * l + 1024u * wave), 16, 0, NTAUX);
    __syncthreads();
    unsigned acc = 0;
    const unsigned* w = reinterpret_cast<const unsigned*>(smem);
#pragma unroll
    for (int l = 0; l < LOADS * 4; ++l) acc ^= w[tid * 5 % (LOADS * 1024) + l * 256 % 7];
    if (acc == 0x12345678u) sink[0] = acc;
}

// (d) window-shaped register loads: every lane reads its own two 20-byte windows (stride 20 B across lanes,
// 64 windows apart) as dwordx4 + dword, `rounds` consecutive 2400-byte rounds per wave, next round's loads
// issued before the current round is consumed.  No LDS, no barriers.
template <int DEPTH>
__global__ void __launch_bounds__(256) k_window_regs(const unsigned char* __restrict__ p, size_t total_rounds, int rounds, unsigned* sink)
{
    typedef const __attribute__((address_space(1))) unsigned char* gp;
    const unsigned lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    size_t r = wave * (size_t)rounds;
    if (r + (size_t)rounds > total_rounds) return;          // whole spans only (the bench tolerates the missing tail)
    const gp base = (gp)p + r * 2400 + 20u * lane;
    const unsigned offB = lane < 57u ? 1280u : 1280u - 20u * (lane - 56u);
    unsigned acc = 0;
    typedef unsigned v4u __attribute__((ext_vector_type(4))); v4u a[DEPTH], b[DEPTH]; unsigned a4[DEPTH], b4[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const gp q = base + 2400u * d;
        a[d] = *(const __attribute__((address_space(1))) v4u*)q; a4[d] = *(const __attribute__((address_space(1))) unsigned*)(q + 16);
        b[d] = *(const __attribute__((address_space(1))) v4u*)(q + offB); b4[d] = *(const __attribute__((address_space(1))) unsigned*)(q + offB + 16);
    }
    for (int it = 0; it < rounds; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const v4u ca = a[d], cb = b[d]; const unsigned c4 = a4[d], d4 = b4[d];
            if (it + d + DEPTH < rounds) {
                const gp q = base + 2400u * (unsigned)(it + d + DEPTH);
                a[d] = *(const __attribute__((address_space(1))) v4u*)q; a4[d] = *(const __attribute__((address_space(1))) unsigned*)(q + 16);
                b[d] = *(const __attribute__((address_space(1))) v4u*)(q + offB); b4[d] = *(const __attribute__((address_space(1))) unsigned*)(q + offB + 16);
            }
            acc ^= ca.x ^ ca.y ^ ca.z ^ ca.w ^ c4 ^ cb.x ^ cb.y ^ cb.z ^ cb.w ^ d4;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F>
static double time_ms(F launch, int iters)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char** argv)
{
    const size_t bytes = 1ull << 30;
    const int nbuf = 3, iters = argc > 1 ? atoi(argv[1]) : 60;
    std::vector<unsigned char*> bufs(nbuf);
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0x5a, bytes)); }
    // random-ish fill so DVFS sees realistic toggling
    {
        std::vector<unsigned> h(bytes / 4);
        unsigned s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
        for (auto& b : bufs) CK(hipMemcpy(b, h.data(), bytes, hipMemcpyHostToDevice));
    }
    unsigned* sink; CK(hipMalloc(&sink, 64));
    auto report = [&](const char* name, double ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); };

    for (int blocks : {2048, 4096, 8192, 16384}) {
        char nm[64]; snprintf(nm, sizeof nm, "stream grid-stride x4, %d blocks", blocks);
        report(nm, time_ms([&](int i) { hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, (const uint4*)bufs[i % nbuf], bytes / 16, sink); }, iters));
    }
    report("tile reg 4 x 16 B/lane (16 KiB tiles)", time_ms([&](int i) { hipLaunchKernelGGL(k_tile_reg<4>, dim3(bytes / 16384), dim3(256), 0, 0, (const uint4*)bufs[i % nbuf], sink); }, iters));
    report("tile reg 5 x 16 B/lane (20 KiB tiles)", time_ms([&](int i) { hipLaunchKernelGGL(k_tile_reg<5>, dim3(bytes / 20480), dim3(256), 0, 0, (const uint4*)bufs[i % nbuf], sink); }, iters));
    report("tile reg 8 x 16 B/lane (32 KiB tiles)", time_ms([&](int i) { hipLaunchKernelGGL(k_tile_reg<8>, dim3(bytes / 32768), dim3(256), 0, 0, (const uint4*)bufs[i % nbuf], sink); }, iters));
    report("tile LDS-DMA 5 x 16 B/lane + barrier", time_ms([&](int i) { hipLaunchKernelGGL(k_tile_dma<5>, dim3(bytes / 20480), dim3(256), 20480, 0, bufs[i % nbuf], sink); }, iters));
    report("tile LDS-DMA 5 x 16 B/lane + barrier, XCD-aware order, nt", time_ms([&](int i) { hipLaunchKernelGGL(k_tile_dma_xcd<5>, dim3(8, bytes / 20480 / 8), dim3(256), 20480, 0, bufs[i % nbuf], sink); }, iters));
    report("tile LDS-DMA 8 x 16 B/lane + barrier", time_ms([&](int i) { hipLaunchKernelGGL(k_tile_dma<8>, dim3(bytes / 32768), dim3(256), 32768, 0, bufs[i % nbuf], sink); }, iters));
    report("tile LDS-DMA 2 x 16 B/lane + barrier", time_ms([&](int i) { hipLaunchKernelGGL(k_tile_dma<2>, dim3(bytes / 8192), dim3(256), 8192, 0, bufs[i % nbuf], sink); }, iters));
    {
        const size_t total_rounds = bytes / 2400 - 2;
        for (int rounds : {8, 16, 32, 110}) {
            const size_t waves = (total_rounds + rounds - 1) / rounds;
            const unsigned blocks = (unsigned)((waves + 3) / 4);
            char nm[80];
            snprintf(nm, sizeof nm, "window regs stride-20, depth 1, %d rounds/wave", rounds);
            report(nm, time_ms([&](int i) { hipLaunchKernelGGL(k_window_regs<1>, dim3(blocks), dim3(256), 0, 0, bufs[i % nbuf], total_rounds, rounds, sink); }, iters));
            snprintf(nm, sizeof nm, "window regs stride-20, depth 2, %d rounds/wave", rounds);
            report(nm, time_ms([&](int i) { hipLaunchKernelGGL(k_window_regs<2>, dim3(blocks), dim3(256), 0, 0, bufs[i % nbuf], total_rounds, rounds, sink); }, iters));
        }
    }
    return 0;
}
