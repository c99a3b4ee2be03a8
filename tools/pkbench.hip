// pkbench: issue rate of the packed-f32 VALU instructions against the scalar ones, measured so that the clock state cannot
// fool the ratio: long kernels (tens of ms each, after a 0.3 s warm-up), every candidate bracketed by the baseline kernel
// (base, X, base) and reported as time(X) / mean(time(base before), time(base after)).  4 waves per SIMD, 8 independent
// dependency chains per wave.  Build: hipcc -O2 --offload-arch=gfx950 pkbench.hip -o pkbench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define K32(NAME, BODY)                                                                                  \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters, float seed)                        \
    {                                                                                                     \
        float a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 99, a6 = a0 + 5, a7 = a0 * 9; \
        float b = 1.0000001f, c = 1e-9f;                                                                  \
        for (int it = 0; it < iters; ++it) { asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); } \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                      \
    }
#define K64(NAME, BODY)                                                                                  \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters, float seed)                        \
    {                                                                                                     \
        typedef float f2 __attribute__((ext_vector_type(2)));                                             \
        f2 a0 = {threadIdx.x + seed, 1.f}, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, a4 = a0 + 11.f, a5 = a0 + 99.f, a6 = a0 + 5.f, a7 = a0 * 9.f; \
        f2 b = {1.0000001f, 0.9999999f}, c = {1e-9f, 2e-9f};                                              \
        for (int it = 0; it < iters; ++it) { asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); } \
        f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                                                  \
    }
#define R8_3(OP) OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"
#define R8_2(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
K32(k_add_f32, R8_2("v_add_f32"))
K32(k_mul_f32, R8_2("v_mul_f32"))
K32(k_fma_f32, R8_3("v_fma_f32"))
K32(k_xor_b32, R8_2("v_xor_b32"))
K32(k_rndne, "v_rndne_f32 %0, %0\nv_rndne_f32 %1, %1\nv_rndne_f32 %2, %2\nv_rndne_f32 %3, %3\nv_rndne_f32 %4, %4\nv_rndne_f32 %5, %5\nv_rndne_f32 %6, %6\nv_rndne_f32 %7, %7\n")
K32(k_rcp, "v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\nv_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7\n")
K32(k_dot4, R8_3("v_dot4_i32_i8"))
K32(k_bfi, R8_3("v_bfi_b32"))
K32(k_bitop3, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x6c\nv_bitop3_b32 %1, %1, %8, %9 bitop3:0x6c\nv_bitop3_b32 %2, %2, %8, %9 bitop3:0x6c\nv_bitop3_b32 %3, %3, %8, %9 bitop3:0x6c\nv_bitop3_b32 %4, %4, %8, %9 bitop3:0x6c\nv_bitop3_b32 %5, %5, %8, %9 bitop3:0x6c\nv_bitop3_b32 %6, %6, %8, %9 bitop3:0x6c\nv_bitop3_b32 %7, %7, %8, %9 bitop3:0x6c\n")
K32(k_cvt_i32, "v_cvt_i32_f32 %0, %0\nv_cvt_i32_f32 %1, %1\nv_cvt_i32_f32 %2, %2\nv_cvt_i32_f32 %3, %3\nv_cvt_i32_f32 %4, %4\nv_cvt_i32_f32 %5, %5\nv_cvt_i32_f32 %6, %6\nv_cvt_i32_f32 %7, %7\n")
K32(k_mov_dpp, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
K32(k_dot4c, "v_dot4c_i32_i8 %0, %8, %9\nv_dot4c_i32_i8 %1, %8, %9\nv_dot4c_i32_i8 %2, %8, %9\nv_dot4c_i32_i8 %3, %8, %9\nv_dot4c_i32_i8 %4, %8, %9\nv_dot4c_i32_i8 %5, %8, %9\nv_dot4c_i32_i8 %6, %8, %9\nv_dot4c_i32_i8 %7, %8, %9\n")
K32(k_fmac, "v_fmac_f32 %0, %8, %9\nv_fmac_f32 %1, %8, %9\nv_fmac_f32 %2, %8, %9\nv_fmac_f32 %3, %8, %9\nv_fmac_f32 %4, %8, %9\nv_fmac_f32 %5, %8, %9\nv_fmac_f32 %6, %8, %9\nv_fmac_f32 %7, %8, %9\n")
K32(k_add_abs, "v_add_f32 %0, |%0|, |%8|\nv_add_f32 %1, |%1|, |%8|\nv_add_f32 %2, |%2|, |%8|\nv_add_f32 %3, |%3|, |%8|\nv_add_f32 %4, |%4|, |%8|\nv_add_f32 %5, |%5|, |%8|\nv_add_f32 %6, |%6|, |%8|\nv_add_f32 %7, |%7|, |%8|\n")
K32(k_perm, R8_3("v_perm_b32"))
K32(k_mov, "v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\nv_mov_b32 %4, %8\nv_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8\n")
K64(k_pk_add_f32, R8_2("v_pk_add_f32"))
K64(k_pk_mul_f32, R8_2("v_pk_mul_f32"))
K64(k_pk_fma_f32, R8_3("v_pk_fma_f32"))

typedef void (*kern_t)(float*, int, float);
static float* g_out;
static double run_ms(kern_t k, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256 * 4), dim3(256), 0, 0, g_out, iters, 2.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}

int main()
{
    CK(hipMalloc(&g_out, 256 * 4 * 256 * 4));
    const int iters = 400000;                                 // 3.2 M instructions per wave
    for (int i = 0; i < 6; ++i) run_ms(k_add_f32, iters);     // warm-up: the clock settles
    struct { const char* name; kern_t k; } tests[] = {
        {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_xor_b32", k_xor_b32}, {"v_fma_f32", k_fma_f32}, {"v_dot4_i32_i8", k_dot4},
        {"v_rndne_f32", k_rndne}, {"v_rcp_f32", k_rcp}, {"v_bfi_b32", k_bfi}, {"v_bitop3_b32", k_bitop3}, {"v_cvt_i32_f32", k_cvt_i32},
        {"v_mov_b32_dpp", k_mov_dpp}, {"v_dot4c_i32_i8", k_dot4c}, {"v_fmac_f32", k_fmac}, {"v_add_f32 |a|,|b|", k_add_abs}, {"v_perm_b32", k_perm}, {"v_mov_b32", k_mov}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_fma_f32", k_pk_fma_f32}};
    for (auto& t : tests) {
        for (int rep = 0; rep < 2; ++rep) {
            const double b0 = run_ms(k_add_f32, iters), x = run_ms(t.k, iters), b1 = run_ms(k_add_f32, iters);
            const double base = 0.5 * (b0 + b1);
            // cycles per wave-instruction at 4 waves / SIMD if the baseline is 2: ratio * 2
            printf("%-14s %8.2f ms  base %8.2f / %8.2f ms  ratio %.3f   ns per instruction per SIMD %.3f\n", t.name, x, b0, b1, x / base,
                   x * 1e6 / (4.0 * iters * 8.0));
        }
    }
    return 0;
}
