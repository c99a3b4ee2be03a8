// salubench: is the scalar ALU a per-SIMD or a per-CU resource, and what does a scalar instruction cost next to a vector one?
// Long kernels (clock settled), 8 independent chains per wave, W waves per SIMD (blocks of 256 threads = one wave per SIMD).
// Reports ns per wave-instruction per SIMD for s_add_u32 and for v_add_f32 at W = 1, 2, 4, 8, and a MIXED kernel (8 s_add + 8
// v_add per iteration): if the mixed kernel takes max(s, v) the two pipes issue side by side, if s + v they share the slot.
// Build: hipcc -O2 --offload-arch=gfx950 salubench.hip -o salubench
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_salu(unsigned* out, int iters, unsigned seed)
{
    unsigned a0 = seed, a1 = seed * 3, a2 = seed * 5, a3 = seed * 7, a4 = seed + 11, a5 = seed + 99, a6 = seed + 5, a7 = seed * 9, b = seed | 1;
    for (int it = 0; it < iters; ++it)
        asm volatile("s_add_u32 %0, %0, %8\ns_add_u32 %1, %1, %8\ns_add_u32 %2, %2, %8\ns_add_u32 %3, %3, %8\n"
                     "s_add_u32 %4, %4, %8\ns_add_u32 %5, %5, %8\ns_add_u32 %6, %6, %8\ns_add_u32 %7, %7, %8\n"
                     : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4), "+s"(a5), "+s"(a6), "+s"(a7) : "s"(b) : "scc");
    if (threadIdx.x == 0) out[blockIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void __launch_bounds__(256) k_valu(unsigned* out, int iters, unsigned seed)
{
    float a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 99, a6 = a0 + 5, a7 = a0 * 9, b = 1.0000001f;
    for (int it = 0; it < iters; ++it)
        asm volatile("v_add_f32 %0, %0, %8\nv_add_f32 %1, %1, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\n"
                     "v_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_add_f32 %6, %6, %8\nv_add_f32 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void __launch_bounds__(256) k_mixed(unsigned* out, int iters, unsigned seed)
{
    float a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 99, a6 = a0 + 5, a7 = a0 * 9, b = 1.0000001f;
    unsigned s0 = seed, s1 = seed * 3, s2 = seed * 5, s3 = seed * 7, s4 = seed + 11, s5 = seed + 99, s6 = seed + 5, s7 = seed * 9, sb = seed | 1;
    for (int it = 0; it < iters; ++it)
        asm volatile("v_add_f32 %0, %0, %16\ns_add_u32 %8, %8, %17\nv_add_f32 %1, %1, %16\ns_add_u32 %9, %9, %17\n"
                     "v_add_f32 %2, %2, %16\ns_add_u32 %10, %10, %17\nv_add_f32 %3, %3, %16\ns_add_u32 %11, %11, %17\n"
                     "v_add_f32 %4, %4, %16\ns_add_u32 %12, %12, %17\nv_add_f32 %5, %5, %16\ns_add_u32 %13, %13, %17\n"
                     "v_add_f32 %6, %6, %16\ns_add_u32 %14, %14, %17\nv_add_f32 %7, %7, %16\ns_add_u32 %15, %15, %17\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                       "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "v"(b), "s"(sb) : "scc");
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
}
typedef void (*kern_t)(unsigned*, int, unsigned);
static unsigned* g_out;
static double run_ms(kern_t k, int waves_per_simd, int iters)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256 * waves_per_simd), dim3(256), 0, 0, g_out, iters, 3u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms;
}
int main()
{
    if (hipMalloc(&g_out, 256 * 8 * 256 * 4) != hipSuccess) return 1;
    const int iters = 400000;
    for (int i = 0; i < 6; ++i) run_ms(k_valu, 4, iters);
    for (int w : {1, 2, 4, 8}) {
        const double s = run_ms(k_salu, w, iters), v = run_ms(k_valu, w, iters), m = run_ms(k_mixed, w, iters);
        const double n = (double)w * iters * 8.0;            // instructions of one kind per SIMD
        printf("waves/SIMD %d: s_add_u32 %.3f ns/instr/SIMD   v_add_f32 %.3f   mixed (8 + 8 per iteration) %.3f per PAIR   [ms %.1f %.1f %.1f]\n",
               w, s * 1e6 / n, v * 1e6 / n, m * 1e6 / n, s, v, m);
    }
    return 0;
}
