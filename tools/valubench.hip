// valubench.hip -- relative issue cost of the VALU instructions the demod kernel is made of (gfx950).
// Tuning aid, not part of the product.  Each kernel runs N iterations of 32 independent instances of one
// instruction (8 chains x 4) on every SIMD; reported: ns per wave-instruction per SIMD at the given occupancy.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(S) S S S S S S S S
#define BODY(INSTR)                                                                                     \
    for (int it = 0; it < iters; ++it) {                                                                \
        asm volatile(REP8(INSTR) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(b), "v"(c), "s"(s0) : "vcc", "s20", "s21");                                                        \
    }

#define KERNEL(NAME, INSTR)                                                                             \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, int iters, unsigned seed)                \
    {                                                                                                   \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 ^ 99, a6 = a0 + 5, a7 = a0 * 9; \
        unsigned b = a0 * 17 + 1, c = a0 * 29 + 3;                                                      \
        unsigned s0 = seed * 7 + 1;                                                                     \
        BODY(INSTR)                                                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                    \
    }

// 8 instructions per asm block, each on its own accumulator -> independent chains
#define I8(OP, TAIL) OP " %0, " TAIL "\n" OP " %1, " TAIL "\n" OP " %2, " TAIL "\n" OP " %3, " TAIL "\n" \
                     OP " %4, " TAIL "\n" OP " %5, " TAIL "\n" OP " %6, " TAIL "\n" OP " %7, " TAIL "\n"
#define I8S(OP, MID, TAIL) OP " %0, %0" MID TAIL "\n" OP " %1, %1" MID TAIL "\n" OP " %2, %2" MID TAIL "\n" OP " %3, %3" MID TAIL "\n" \
                           OP " %4, %4" MID TAIL "\n" OP " %5, %5" MID TAIL "\n" OP " %6, %6" MID TAIL "\n" OP " %7, %7" MID TAIL "\n"

KERNEL(k_add_u32,   I8S("v_add_u32", ", ", "%8"))
KERNEL(k_xor_b32,   I8S("v_xor_b32", ", ", "%8"))
KERNEL(k_sub_u32,   I8S("v_sub_u32", ", ", "%8"))
KERNEL(k_max_i32,   I8S("v_max_i32", ", ", "%8"))
KERNEL(k_lshl,      I8S("v_lshlrev_b32", ", ", "%8"))
KERNEL(k_dot4c,     I8("v_dot4c_i32_i8", "%8, %9"))
KERNEL(k_dot2c,     I8("v_dot2c_i32_i16", "%8, %9"))
KERNEL(k_dot4,      I8S("v_dot4_i32_i8", ", %8, ", "%9"))
KERNEL(k_cndmask,   I8S("v_cndmask_b32", ", %8, ", "vcc"))
KERNEL(k_cmp_cnd_vcc, "v_cmp_lt_u32 vcc, %8, %9\n" I8S("v_cndmask_b32", ", %8, ", "vcc"))   // 9 instructions per block
KERNEL(k_cmp_cnd_e64, "v_cmp_lt_u32 s[20:21], %8, %9\n" I8S("v_cndmask_b32_e64", ", %8, ", "s[20:21]"))
KERNEL(k_cmp_cnd_const, "v_cmp_lt_u32 s[20:21], %8, %9\n" I8S("v_cndmask_b32_e64", ", 0, ", "s[20:21]"))
KERNEL(k_cmp_cnd_e64vcc, "v_cmp_lt_u32 vcc, %8, %9\n" I8S("v_cndmask_b32_e64", ", %8, ", "vcc"))
KERNEL(k_cmp8_vcc,  REP8("v_cmp_lt_u32 vcc, %8, %9\n"))
KERNEL(k_cmp8_sgpr, REP8("v_cmp_lt_u32 s[20:21], %8, %9\n"))
KERNEL(k_mul_lo,    I8S("v_mul_lo_u32", ", ", "%8"))
KERNEL(k_mul_i24,   I8S("v_mul_i32_i24", ", ", "%8"))
KERNEL(k_mad_u24,   I8S("v_mad_u32_u24", ", %8, ", "%9"))
KERNEL(k_add3,      I8S("v_add3_u32", ", %8, ", "%9"))
KERNEL(k_lshl_add,  I8S("v_lshl_add_u32", ", 2, ", "%9"))
KERNEL(k_and_or,    I8S("v_and_or_b32", ", %8, ", "%9"))
KERNEL(k_alignbit,  I8S("v_alignbit_b32", ", %8, ", "16"))
KERNEL(k_perm,      I8S("v_perm_b32", ", %8, ", "%9"))
KERNEL(k_bfe_i32,   I8S("v_bfe_i32", ", 3, ", "16"))
KERNEL(k_cvt_f32_u32, I8("v_cvt_f32_u32", "%8"))
KERNEL(k_cvt_u32_f32, I8("v_cvt_u32_f32", "%8"))
KERNEL(k_rcp_f32,   I8("v_rcp_f32", "%8"))
KERNEL(k_mul_f32,   I8S("v_mul_f32", ", ", "%8"))
KERNEL(k_fma_f32,   I8S("v_fma_f32", ", %8, ", "%9"))
KERNEL(k_add_f32,   I8S("v_add_f32", ", ", "%8"))
KERNEL(k_pk_add_u16, I8S("v_pk_add_u16", ", ", "%8"))
KERNEL(k_pk_mul_lo_u16, I8S("v_pk_mul_lo_u16", ", ", "%8"))
KERNEL(k_sad_u32,   I8S("v_sad_u32", ", %8, ", "%9"))
KERNEL(k_mov_dpp,   I8("v_mov_b32_dpp", "%8 wave_shr:1 row_mask:0xf bank_mask:0xf"))
// round 2: candidates for a float-domain discriminator
KERNEL(k_and_b32,   I8S("v_and_b32", ", ", "%8"))
KERNEL(k_or_b32,    I8S("v_or_b32", ", ", "%8"))
KERNEL(k_ashr,      I8S("v_ashrrev_i32", ", ", "%8"))
KERNEL(k_lshr,      I8S("v_lshrrev_b32", ", ", "%8"))
KERNEL(k_mov,       I8("v_mov_b32", "%8"))
KERNEL(k_floor_f32, I8("v_floor_f32", "%8"))
KERNEL(k_trunc_f32, I8("v_trunc_f32", "%8"))
KERNEL(k_rndne_f32, I8("v_rndne_f32", "%8"))
KERNEL(k_cvt_f32_i32, I8("v_cvt_f32_i32", "%8"))
KERNEL(k_cvt_i32_f32, I8("v_cvt_i32_f32", "%8"))
KERNEL(k_min_f32,   I8S("v_min_f32", ", ", "%8"))
KERNEL(k_max_f32,   I8S("v_max_f32", ", ", "%8"))
KERNEL(k_med3_f32,  I8S("v_med3_f32", ", %8, ", "%9"))
KERNEL(k_bfi_b32,   I8S("v_bfi_b32", ", %8, ", "%9"))
KERNEL(k_sub_f32,   I8S("v_sub_f32", ", ", "%8"))
KERNEL(k_add_f32_abs, I8("v_add_f32_e64", "|%8|, |%9|"))
KERNEL(k_add_f32_clamp, I8("v_add_f32_e64", "%8, %9 clamp"))
KERNEL(k_fma_f32_neg, I8S("v_fma_f32", ", -%8, ", "|%9|"))
KERNEL(k_mul_f32_e64, I8("v_mul_f32_e64", "|%8|, %9"))
KERNEL(k_cmp8_f32,  REP8("v_cmp_ge_f32 s[20:21], %8, %9\n"))
KERNEL(k_mul_u24,   I8S("v_mul_u32_u24", ", ", "%8"))
KERNEL(k_mad_i24,   I8S("v_mad_i32_i24", ", %8, ", "%9"))
KERNEL(k_xad_u32,   I8S("v_xad_u32", ", %8, ", "%9"))
KERNEL(k_ldexp_f32, I8S("v_ldexp_f32", ", ", "%8"))
KERNEL(k_max_u32,   I8S("v_max_u32", ", ", "%8"))
KERNEL(k_min_u32,   I8S("v_min_u32", ", ", "%8"))
KERNEL(k_mul_hi_u32, I8S("v_mul_hi_u32", ", ", "%8"))
KERNEL(k_cvt_pk_i16, I8S("v_cvt_pk_i16_i32", ", ", "%8"))
KERNEL(k_add_lshl,  I8S("v_add_lshl_u32", ", %8, ", "2"))
KERNEL(k_lshl_or,   I8S("v_lshl_or_b32", ", 16, ", "%9"))

// packed f32 (two floats per 64-bit register pair).  Round 4's versions of these three kernels typed the accumulators `double`
// and reported 0.28x an add -- not a plausible issue rate (profiles/archive/r04_valubench.txt); round 5: float2 operands as in
// tools/pkbench.hip, finite values that stay finite (b = 1 +- 2^-23, c tiny), and tools/valubench_check.sh asserts from the
// emitted ISA that each loop body holds exactly eight of the named instruction.
typedef float vb_f2 __attribute__((ext_vector_type(2)));
#define KERNEL64(NAME, BODY)                                                                            \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, int iters, unsigned seed)                \
    {                                                                                                   \
        vb_f2 a0 = {(float)(threadIdx.x + seed), 1.f}, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, a4 = a0 + 11.f, a5 = a0 + 99.f, a6 = a0 + 5.f, a7 = a0 * 9.f; \
        vb_f2 b = {1.0000001f, 0.9999999f}, c = {1e-9f, 2e-9f};                                         \
        for (int it = 0; it < iters; ++it) {                                                            \
            asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                               \
        const vb_f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                          \
        out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(s.x + s.y);                                    \
    }
#define P8_3(OP) OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"
#define P8_2(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
KERNEL64(k_pk_fma_f32x2, P8_3("v_pk_fma_f32"))
KERNEL64(k_pk_add_f32x2, P8_2("v_pk_add_f32"))
KERNEL64(k_pk_mul_f32x2, P8_2("v_pk_mul_f32"))
// the same three with the modifiers a paired discriminator would use: a negated operand half (neg_lo / neg_hi), the clamp bit,
// and a swapped source half (op_sel) -- none of them may change the issue cost
KERNEL64(k_pk_add_f32_neg, "v_pk_add_f32 %0, %0, %8 neg_lo:[0,1] neg_hi:[0,1]\nv_pk_add_f32 %1, %1, %8 neg_lo:[0,1] neg_hi:[0,1]\nv_pk_add_f32 %2, %2, %8 neg_lo:[0,1] neg_hi:[0,1]\nv_pk_add_f32 %3, %3, %8 neg_lo:[0,1] neg_hi:[0,1]\n"
                           "v_pk_add_f32 %4, %4, %8 neg_lo:[0,1] neg_hi:[0,1]\nv_pk_add_f32 %5, %5, %8 neg_lo:[0,1] neg_hi:[0,1]\nv_pk_add_f32 %6, %6, %8 neg_lo:[0,1] neg_hi:[0,1]\nv_pk_add_f32 %7, %7, %8 neg_lo:[0,1] neg_hi:[0,1]\n")
KERNEL64(k_pk_fma_f32_clamp, "v_pk_fma_f32 %0, %0, %8, %9 clamp\nv_pk_fma_f32 %1, %1, %8, %9 clamp\nv_pk_fma_f32 %2, %2, %8, %9 clamp\nv_pk_fma_f32 %3, %3, %8, %9 clamp\n"
                             "v_pk_fma_f32 %4, %4, %8, %9 clamp\nv_pk_fma_f32 %5, %5, %8, %9 clamp\nv_pk_fma_f32 %6, %6, %8, %9 clamp\nv_pk_fma_f32 %7, %7, %8, %9 clamp\n")
KERNEL64(k_pk_mul_f32_opsel, "v_pk_mul_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0]\nv_pk_mul_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0]\nv_pk_mul_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0]\nv_pk_mul_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
                             "v_pk_mul_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0]\nv_pk_mul_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0]\nv_pk_mul_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0]\nv_pk_mul_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0]\n")
// a DEPENDENT chain (one accumulator, eight deep) against the same of v_add_f32: latency, not issue
KERNEL64(k_pk_add_f32_chain, "v_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %0, %0, %8\n")
KERNEL(k_add_f32_chain, "v_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\n")

template <typename K>
static void run(const char* name, K kern, int blocks_per_cu, int iters, double base_ns)
{
    unsigned* out;
    const int cus = 256;
    const int blocks = cus * blocks_per_cu;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 2u + r);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    // per SIMD: blocks_per_cu waves (one wave of each block per SIMD), each iters * 8 instructions, 3 launches
    const double instr_per_simd = 3.0 * blocks_per_cu * (double)iters * 8.0;
    const double ns = ms * 1e6 / instr_per_simd;
    printf("%-18s waves/SIMD=%d  %7.3f ns/instr/SIMD  (%.2fx v_add_u32)\n", name, blocks_per_cu, ns, base_ns > 0 ? ns / base_ns : 1.0);
    CK(hipFree(out));
}

int main()
{
    const int iters = 4000;
    for (int occ : {1, 2, 4, 8}) run("v_add_u32", k_add_u32, occ, iters, 0);
    const int occ = 4;
    // baseline
    unsigned* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_add_u32, dim3(256 * occ), dim3(256), 0, 0, out, iters, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_add_u32, dim3(256 * occ), dim3(256), 0, 0, out, iters, 2u + r);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double base = ms * 1e6 / (3.0 * occ * iters * 8.0);
#define R(K) run(#K, K, occ, iters, base)
    R(k_add_u32); R(k_xor_b32); R(k_sub_u32); R(k_max_i32); R(k_lshl); R(k_dot4c); R(k_dot2c); R(k_dot4); R(k_cndmask);
    R(k_cmp_cnd_e64); R(k_cmp_cnd_const); R(k_cmp_cnd_vcc); R(k_cmp8_vcc); R(k_cmp8_sgpr);
    R(k_mul_lo); R(k_mul_i24); R(k_mad_u24); R(k_add3); R(k_lshl_add); R(k_and_or); R(k_alignbit); R(k_perm); R(k_bfe_i32);
    R(k_cvt_f32_u32); R(k_cvt_u32_f32); R(k_rcp_f32); R(k_mul_f32); R(k_fma_f32); R(k_add_f32); R(k_pk_add_u16);
    R(k_pk_mul_lo_u16); R(k_sad_u32); R(k_mov_dpp);
    R(k_and_b32); R(k_or_b32); R(k_ashr); R(k_lshr); R(k_mov); R(k_floor_f32); R(k_trunc_f32); R(k_rndne_f32);
    R(k_cvt_f32_i32); R(k_cvt_i32_f32); R(k_min_f32); R(k_max_f32); R(k_med3_f32); R(k_bfi_b32); R(k_sub_f32);
    R(k_cvt_pk_i16); R(k_add_lshl); R(k_lshl_or);
    R(k_add_f32_abs); R(k_add_f32_clamp); R(k_fma_f32_neg); R(k_mul_f32_e64); R(k_cmp8_f32);
    R(k_pk_fma_f32x2); R(k_pk_add_f32x2); R(k_pk_mul_f32x2);
    R(k_pk_add_f32_neg); R(k_pk_fma_f32_clamp); R(k_pk_mul_f32_opsel); R(k_pk_add_f32_chain); R(k_add_f32_chain);
    R(k_mul_u24); R(k_mad_i24); R(k_xad_u32); R(k_ldexp_f32); R(k_max_u32); R(k_min_u32); R(k_mul_hi_u32);
    return 0;
}
