// fmd_fir.hip -- generalised tapped decimating FIR over the rotated + centred IQ stream (SURVEY 8a row G',
// BASELINE config 4), gfx950.  Not a reference interface: see include/fmd.h for the definition
//     y[m] = sum_{t < T} h[t] * x[M*m + t].
//
// Data path.  rotate_90 (simple_fm.rs:276-299) + `- 127` (:258) are folded into the taps: for an aligned
// dword (bytes b0..b3 = two complex samples) whose first sample has stream index = 0 mod 4 ("even" dword)
//     re = h0*(b0 - 127) + h1*(128 - b3)      im = h0*(b1 - 127) + h1*(b2 - 127)
// and for the "odd" dword (first sample = 2 mod 4) every sign flips and 127 <-> 128 swap.  So with the bytes
// zero-extended into 16-bit pairs (one v_perm_b32 each: (b0, b3) and (b1, b2)) a dword costs two
// v_dot2_i32_i16 against packed tap pairs that already carry the alternating sign, plus one constant per
// window parity.  Bound: VALU (about 2.5 instructions per tap per output), not HBM -- this kernel exists for
// LDS-window / tap-table sizing, not for the 70 % HBM target.
//
// MFMA form (the default whenever decim <= 64): the filter IS a contraction, so it runs on the matrix cores.
// Four consecutive outputs (i = 0..3) x {re, im} x {low, high tap digit} are the 16 rows of a banded Toeplitz
// matrix A over the BYTES of their common window (rotation signs and the re/im byte selection folded in, taps
// split as h = 128*hi + lo with |lo| <= 64, |hi| <= 16 so both digits are i8); 16 such output quads, 8*decim
// bytes apart, are the 16 columns of B, which is nothing but the channel's byte stream (xor 0x80 -> s8) read
// straight out of LDS: lane (col j, k-group q) of K-chunk k reads the 16 bytes at 8*decim*j + 64*k + 16*q.
// v_mfma_i32_16x16x64_i8 accumulates; lane (j, q) ends up with exactly (re_lo, re_hi, im_lo, im_hi) of output
// 4*j + q and combines them.  127 taps / decimate 8: 5 MFMAs per KiB of input per wave -- HBM-bound.
//
// Streaming state: the last T-1 samples (rounded up to an even count) of every channel live in HBM as raw
// bytes, double-buffered; the stream position is the same for all channels and is kept on the host.
#include "../../include/fmd.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "fmd_fir_common.h"
#include "fmd_host.h"

namespace {

#if defined(__HIP_DEVICE_COMPILE__)
#define FMD_AS_GLOBAL __attribute__((address_space(1)))
#define FMD_AS_CONSTANT __attribute__((address_space(4)))
#else
#define FMD_AS_GLOBAL
#define FMD_AS_CONSTANT
#endif

constexpr int kFirThreads = 256;
constexpr int kFirGroupsPerWave = 4;                      // MFMA form: a wave accumulates up to 4 column groups (64 outputs each)
typedef short fir_s2 __attribute__((ext_vector_type(2)));

// What a fresh block of fmd_fir_mfma_kernel needs before its LDS-DMAs can go out, in the FIRST 64 bytes of the kernel
// arguments (one s_load_dwordx16, one wait): a block holds its LDS from dispatch to exit, and hipcc waits for a block's scalar
// loads before every branch -- the round-3 prologue had seven such round trips in front of the first DMA
// (fmd_firdemod.hip FdHot, fmd_kernels.h FmdFastGeo).
struct FirHot {
    uint64_t iq;               // device address of the input
    uint64_t stride_w;         // dwords per channel in this call
    uint64_t amat;             // device address of the tap fragments
    uint32_t n_channels, n_out, out_tile;
    uint32_t Hw, NP, half_M, wd_first;
    uint32_t pad[3];
};
static_assert(sizeof(FirHot) == 64, "one s_load_dwordx16");

struct FirLaunch {
    FirHot hot;                // matrix-core kernel (must stay the first member)
    const uint32_t* iq;        // [C][stride_w] dwords
    uint64_t stride_w;         // dwords per channel in this call (= nbytes / 4)
    const uint32_t* hist_in;   // [C][Hw]
    uint32_t* hist_out;        // [C][Hw]
    uint32_t Hw;               // history dwords per channel
    const uint32_t* wre;       // [NP] packed (h[2i], -h[2i+1]) * (-1)^i
    const uint32_t* wim;       // [NP] packed (h[2i],  h[2i+1]) * (-1)^i
    int32_t cre[2], cim[2];    // additive constants by window parity
    uint32_t NP;               // tap pairs = ceil(T / 2)
    uint32_t half_M;           // decim / 2 (dwords between consecutive windows)
    uint32_t out_tile;         // outputs per block
    uint32_t n_out;            // outputs this call (per channel)
    uint32_t wd_first;         // virtual dword index of the first output's window
    uint32_t par_first;        // stream-dword parity of the first output's window (0/1)
    uint32_t par_step;         // parity step per output: (decim / 2) & 1
    uint32_t n_channels;
    int32_t* out;              // [C][out_cap][2]
    uint64_t out_cap;
    // MFMA form
    const uint32_t* amat;      // [n_pass * nku][64 lanes][4 dwords]: A fragments of the banded tap matrix
    uint32_t n_pass, nku;      // K-chunks of 64 bytes = n_pass * nku
    uint32_t groups;           // 16-column groups (64 outputs each) per tile
    uint32_t col_bytes;        // 8 * decim: byte distance between consecutive columns
    int32_t mre[2], mim[2];    // additive constants (s8 domain) by window parity
    uint32_t dbg;              // ablation bits, honoured by -DFMD_EXPERIMENT builds only
};

#ifdef FMD_EXPERIMENT
#define FIR_ABLATE(bit) ((L.dbg >> (bit)) & 1u)
#else
#define FIR_ABLATE(bit) false
#endif

typedef int fir_i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(fir_s2, a), __builtin_bit_cast(fir_s2, b), c, false);
}

// virtual stream of one channel: history dwords followed by this call's dwords
__device__ __forceinline__ uint32_t virt_dword(const FirLaunch& L, uint32_t c, uint32_t w)
{
    typedef const FMD_AS_GLOBAL uint32_t* gw;
    if (w < L.Hw) return ((gw)(uintptr_t)L.hist_in)[(uint64_t)c * L.Hw + w];
    uint64_t k = w - L.Hw;
    if (k >= L.stride_w) k = L.stride_w - 1;              // the zero-weighted pad sample of an odd tap count
    return ((gw)(uintptr_t)L.iq)[(uint64_t)c * L.stride_w + k];
}

__global__ void __launch_bounds__(kFirThreads) fmd_fir_kernel(const FirLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t c = blockIdx.y + 65535u * blockIdx.z;
    if (c >= L.n_channels) return;
    const uint32_t o0 = blockIdx.x * L.out_tile;
    if (o0 >= L.n_out) return;
    const uint32_t no = L.n_out - o0 < L.out_tile ? L.n_out - o0 : L.out_tile;
    const uint32_t w0 = L.wd_first + o0 * L.half_M;       // first virtual dword of the tile
    const uint32_t nd = (no - 1) * L.half_M + L.NP;       // dwords the tile's windows cover
    for (uint32_t i = tid; i < nd; i += kFirThreads) lds[i] = virt_dword(L, c, w0 + i);
    __syncthreads();

    typedef const FMD_AS_CONSTANT uint32_t* cw;
    const cw wre = (cw)(uintptr_t)L.wre, wim = (cw)(uintptr_t)L.wim;
    const bool aligned = (L.half_M & 3u) == 0u;           // windows start on 16-byte LDS boundaries: ds_read_b128
    for (uint32_t o = tid; o < no; o += kFirThreads) {
        const uint32_t* p = lds + o * L.half_M;
        int are = 0, aim = 0;
        // L.NP is a multiple of 4 (zero-padded taps): four tap pairs per step, taps by s_load_dwordx4
        for (uint32_t i = 0; i < L.NP; i += 4) {
            uint32_t w[4];
            if (aligned) {
                const uint4 q = *reinterpret_cast<const uint4*>(p + i);
                w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
            } else {
                w[0] = p[i]; w[1] = p[i + 1]; w[2] = p[i + 2]; w[3] = p[i + 3];
            }
            typedef uint32_t t4 __attribute__((ext_vector_type(4)));
            const t4 tr = *(const FMD_AS_CONSTANT t4*)(wre + i), ti = *(const FMD_AS_CONSTANT t4*)(wim + i);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t pre = __builtin_amdgcn_perm(w[u], w[u], 0x0C030C00u);   // (b0, b3) zero-extended to 16 bit
                const uint32_t pim = __builtin_amdgcn_perm(w[u], w[u], 0x0C020C01u);   // (b1, b2)
                are = sdot2(pre, tr[u], are);
                aim = sdot2(pim, ti[u], aim);
            }
        }
        const uint32_t par = (L.par_first + (o0 + o) * L.par_step) & 1u;
        const int re = (par ? -are : are) + L.cre[par];
        const int im = (par ? -aim : aim) + L.cim[par];
        int2* dst = reinterpret_cast<int2*>(L.out) + ((uint64_t)c * L.out_cap + o0 + o);
        *dst = make_int2(re, im);
    }
}


// ---- MFMA form ------------------------------------------------------------------------------------------------
// 16 raw bytes of the virtual stream starting at dword w; dwords past the end are never weighted
__device__ __forceinline__ fir_i4 virt_chunk(const FirLaunch& L, uint32_t c, uint32_t w, bool fast)
{
    fir_i4 v;
    if (fast && w >= L.Hw && (uint64_t)(w - L.Hw) + 4 <= L.stride_w) {
        typedef const FMD_AS_GLOBAL fir_i4* gq;
        v = *(gq)(uintptr_t)(L.iq + (uint64_t)c * L.stride_w + (w - L.Hw));
    } else {
        v.x = (int)virt_dword(L, c, w);     v.y = (int)virt_dword(L, c, w + 1);
        v.z = (int)virt_dword(L, c, w + 2); v.w = (int)virt_dword(L, c, w + 3);
    }
    return v;
}

// Cache policy of the staging loads (aux of global_load_lds): 2 = nt -- the stream is read once.  Measured on
// config 4: 0.1369 ms vs 0.1440 ms with the default policy (-5 %); nontemporal output stores on top of it: +1.4 %.
constexpr int kFirDmaAux = 2;

__device__ __forceinline__ void fir_dma16(const unsigned char* g, unsigned char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, kFirDmaAux);
}

// SWZ (decim == 8, columns 64 bytes apart): a fragment read puts lanes (j, q) at 64*j + 16*q, and the four
// 16-lane groups ds_read_b128 is serviced in ({0-3,12-15,20-27}, ...) then hit every 16-byte slot of the 256-byte
// bank row twice.  Swapping the 32-byte halves of each 64-byte chunk in every other bank row (address bit 5 ^=
// bit 8) makes all four groups conflict-free for every K-chunk; the LDS-DMA applies it for free by permuting
// which lane fetches which 16 bytes.  (PMC: SQ_LDS_BANK_CONFLICT 10.5 M cycles per launch without it.)
__device__ __forceinline__ uint32_t fir_swz_slot(uint32_t slot) { return slot ^ (((slot >> 4) & 1u) << 1); }

// Round 6 (the layout every matrix-core form of this kernel runs on, unless SWZ above is chosen): the tile image is stored with the
// four 32-byte pieces of every 128-byte column permuted by the column's index -- byte address a lives at a ^ ((a >> 3) & 0x60), i.e.
// bits 5, 6 (the piece) are xor-ed with bits 8, 9 (the column pair).  Operand columns are 64 or 128 bytes apart, so without it the
// lanes the LDS serves together (tools/lds_model.py) keep hitting the same 16-byte slots of the 256-byte bank row: measured with
// tools/ldsbench.py, a fragment read cost 3.1 x a conflict-free one in the sparse two-digit form (SQ_LDS_BANK_CONFLICT 75 % of the
// LDS-active cycles), 1.64 x in the dense two-digit form, and by the model 4 x in the dense one-digit form; with the
// permutation (and the sparse form's half swap, fmd_fir_kswap) they are conflict-free / alternate 1 x and 2 x / conflict-free.
// The LDS-DMA applies it for free by permuting which lane fetches which 16 bytes -- in PAIRS of lanes (32 contiguous bytes), within
// one 128-byte line.
__device__ __forceinline__ uint32_t fir_img(uint32_t byte_addr) { return byte_addr ^ ((byte_addr >> 3) & 0x60u); }
__device__ __forceinline__ uint32_t fir_img_slot(uint32_t slot) { return slot ^ (((slot >> 4) & 3u) << 1); }   // the same on 16-byte slot indices

// DIGITS: tap digits of the A fragments (fmd_fir_common.h).  2: a column is four outputs, lane (j, q) ends up with (re_lo, re_hi,
// im_lo, im_hi) of output 4j + q; 1 (every |tap| <= 127): a column is EIGHT outputs, lane (j, q) ends up with (re, im) of outputs
// 8j + 2q and 8j + 2q + 1 -- the same matrix instructions and operand reads cover twice the outputs.
template <int NKU, bool SWZ, int DIGITS>
// (8 blocks per CU = 64 VGPRs per lane hold up to six operand fragments beside the accumulators; the two longest filter shapes
//  -- NKU 7, 8: beyond ~190 taps at decimate 8 -- spilled 4 ... 32 registers to scratch memory under that bound, which
//  tests/test_isa_invariants.py now forbids for every kernel: they take 6 and 4 blocks per CU)
__global__ void __launch_bounds__(kFirThreads, NKU <= 6 ? 8 : (NKU == 7 ? 6 : 4)) fmd_fir_mfma_kernel(const FirLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t tid = threadIdx.x;
    // the permuted tile image (fir_img) under the sparse two-digit form -- config 4's.  The dense forms keep the linear image: their
    // reads would need a permuted address per (group, chunk) -- six more live registers than their 64-register budget has (the
    // first build spilled and ran the 8-bit filter 24 % slower: session r06b) -- and their matrix phase is hidden anyway.
    constexpr bool IMG = DIGITS == 3;
    __builtin_amdgcn_s_setprio(3);                                  // get the loads out first (see fmd_tile_body.h)
    // Grid (8, tiles, ceil(C / 8)) always: blockIdx.x IS the XCD, which works through its own contiguous eighth of the channels
    // tile after tile (about -1 % per call against the plain mapping).  No branch in front of the LDS-DMAs: a surplus block
    // of the grid (channel count not a multiple of 8) stages channel 0's first tile like any other and leaves behind its loads.
    const FirHot H = L.hot;
    const uint32_t gz = gridDim.z, ntiles = gridDim.y;
    __builtin_amdgcn_sched_barrier(0);                              // both scalar loads go out before anything waits
    const uint32_t c_grid = blockIdx.x * gz + blockIdx.z;
    const bool surplus = c_grid >= H.n_channels || blockIdx.y * H.out_tile >= H.n_out;
    const uint32_t c = surplus ? 0u : c_grid, tix = surplus ? 0u : blockIdx.y;
    const uint32_t o0 = tix * H.out_tile;
    const uint32_t no = H.n_out - o0 < H.out_tile ? H.n_out - o0 : H.out_tile;
    const uint32_t w0 = H.wd_first + o0 * H.half_M;                 // first virtual dword of the tile
    // 16-byte slots the valid windows cover, in whole 64-byte chunks (the swizzle permutes within a chunk)
    // (the permuted image moves 32-byte pieces within a 128-byte column: whole columns of 8 slots are staged)
    const uint32_t nq = IMG ? ((((no - 1) * H.half_M + H.NP + 3u) >> 2) + 7u) & ~7u : ((((no - 1) * H.half_M + H.NP + 3u) >> 2) + 3u) & ~3u;
    // 16-byte global loads when this tile's chunks are 16-byte aligned in the caller's buffer
    const bool fast = (((H.iq + ((uint64_t)c * H.stride_w + (uint64_t)w0 - H.Hw) * 4u)) & 15u) == 0u;
    const uint32_t lane = tid & 63u, wave = tid >> 6, j = lane & 15u, q = lane >> 4;

    // Interior tiles (whole tile inside this call's buffer, 16-byte aligned): global_load_lds_dwordx4, 1 KiB per
    // wave-instruction straight into LDS, no VGPR round trip.  Tiles that touch the history, the end of the
    // stream or an unaligned buffer go through registers.
    const bool whole = fast && w0 >= H.Hw && (uint64_t)(w0 - H.Hw) + 4ull * nq <= H.stride_w;
    fir_i4* lq = reinterpret_cast<fir_i4*>(lds);
    // the tap fragments FIRST: five small L2-resident loads per lane whose latency then hides under the tile's own; issued
    // behind the 16 KB of DMAs they queue behind them and every wave waits for them at the barrier (+2 % measured)
    typedef const FMD_AS_GLOBAL fir_i4* gq;
    const gq amat = (gq)(uintptr_t)H.amat + lane;
    // DIGITS == 3: two digits on the 4:2 sparse matrix instruction, re and im rows in fragment sets of their own ((lo, hi) of EIGHT
    // outputs per fragment: fmd_fir_common.h `split`) -- a lane then ends up with two complete adjacent outputs, like the one-digit
    // form, and stores them as 16 bytes.  NKU stays the dense chunk count of that shape; the form has (NKU + 1) / 2 128-byte chunks.
    constexpr int NKS = (NKU + 1) / 2;
    constexpr int NAF = DIGITS == 3 ? 2 * NKS : NKU;
    fir_i4 A[NAF];
#pragma unroll
    for (int k = 0; k < NAF; ++k) A[k] = FIR_ABLATE(4) ? fir_i4{(int)lane, k, 1, 2} : amat[k * 64];   // (probe: what the tap fragments' L2 traffic costs)
    if (FIR_ABLATE(1)) {
    } else if (whole) {
        // (lane tid of DMA l writes LDS slot tid + 256 l; the image permutation only involves slot bits 1, 2, 4, 5: the same for every l)
        const unsigned char* src = reinterpret_cast<const unsigned char*>(reinterpret_cast<const uint32_t*>((uintptr_t)H.iq) + (uint64_t)c * H.stride_w + (w0 - H.Hw)) + 16u * (SWZ ? fir_swz_slot(tid) : IMG ? fir_img_slot(tid) : tid);
        unsigned char* dst = reinterpret_cast<unsigned char*>(lds) + 1024u * wave;
        const uint32_t nfull = nq / kFirThreads, ntail = nq - nfull * kFirThreads;
        for (uint32_t l = 0; l < nfull; ++l) fir_dma16(src + (16u * kFirThreads) * l, dst + (16u * kFirThreads) * l);
        if (tid < ntail) fir_dma16(src + (16u * kFirThreads) * nfull, dst + (16u * kFirThreads) * nfull);
    } else {
        for (uint32_t base = 0; base < nq; base += 5u * kFirThreads) {  // one trip for tiles up to 20 KiB
            const uint32_t i0 = base + tid, i1 = i0 + kFirThreads, i2 = i1 + kFirThreads, i3 = i2 + kFirThreads,
                           i4 = i3 + kFirThreads;
            fir_i4 a = {0, 0, 0, 0}, b = a, d = a, e = a, h = a;        // five loads in flight per lane
            if (i0 < nq) a = virt_chunk(L, c, w0 + 4u * i0, fast);
            if (i1 < nq) b = virt_chunk(L, c, w0 + 4u * i1, fast);
            if (i2 < nq) d = virt_chunk(L, c, w0 + 4u * i2, fast);
            if (i3 < nq) e = virt_chunk(L, c, w0 + 4u * i3, fast);
            if (i4 < nq) h = virt_chunk(L, c, w0 + 4u * i4, fast);
            if (i0 < nq) lq[SWZ ? fir_swz_slot(i0) : IMG ? fir_img_slot(i0) : i0] = a;
            if (i1 < nq) lq[SWZ ? fir_swz_slot(i1) : IMG ? fir_img_slot(i1) : i1] = b;
            if (i2 < nq) lq[SWZ ? fir_swz_slot(i2) : IMG ? fir_img_slot(i2) : i2] = d;
            if (i3 < nq) lq[SWZ ? fir_swz_slot(i3) : IMG ? fir_img_slot(i3) : i3] = e;
            if (i4 < nq) lq[SWZ ? fir_swz_slot(i4) : IMG ? fir_img_slot(i4) : i4] = h;
        }
    }
    if (surplus) {                                                  // (rare) behind the staging loads: they write this block's LDS
        __builtin_amdgcn_s_waitcnt(0x0F70);
        return;
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): the LDS-DMAs (and the A fragments) have landed
    __syncthreads();

    if (FIR_ABLATE(8)) __builtin_amdgcn_s_sleep(16);                // pacing probes (experiment build): 1024 clocks behind the staging barrier ...
    const uint8_t* lb = reinterpret_cast<const uint8_t*>(lds);
    constexpr int GPW = DIGITS == 3 ? 2 : kFirGroupsPerWave;       // (sparse form: the host keeps the tile at <= 8 groups of 128 outputs)
    fir_i4 acc[DIGITS == 3 ? 2 * GPW : GPW];                        // (sparse form: acc[gi] re, acc[GPW + gi] im)
#pragma unroll
    for (int gi = 0; gi < (DIGITS == 3 ? 2 * GPW : GPW); ++gi) acc[gi] = fir_i4{0, 0, 0, 0};
    if constexpr (DIGITS == 3) {
        typedef int fir_i8 __attribute__((ext_vector_type(8)));
        // Where the lane's 32 bytes of chunk kc live in the permuted image.  Logical address (16 g + j) * col_bytes + 128 kc + 32 q + (0 | 16)
        // with g = wave + 4 gi: the group's share, 64 gi col_bytes, is a multiple of 1024 (col_bytes = 16 decim, decim even) and never
        // reaches the address bits fir_img reads or flips -- so the permuted address is ONE register per chunk and half, set up once per
        // wave, plus the group's (scalar) offset; the chunk's 128 kc rides in the instruction's immediate offset.  The lanes of the odd
        // K quarters read their second half first (fmd_fir_kswap).
        static_assert(NKS <= 4, "address registers of the sparse form");
        uint32_t pa[NKS], pb[NKS];
#pragma unroll
        for (int kc = 0; kc < NKS; ++kc) {
            const uint32_t a = (16u * wave + j) * L.col_bytes + 128u * kc + 32u * q + (FIR_ABLATE(6) ? 0u : 16u * (q & 1u));   // (probe, experiment build: round 5's half order -- wrong outputs, same work)
            pa[kc] = fir_img(a) - 128u * kc;
            pb[kc] = pa[kc] ^ 16u;
        }
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
            const uint32_t g = wave + 4u * gi;
            if (g < L.groups && 128u * g < no && !FIR_ABLATE(0)) {  // wave-uniform
#pragma unroll
                for (int kc = 0; kc < NKS; ++kc) {
                    const fir_i4 b0 = *reinterpret_cast<const fir_i4*>(lb + pa[kc] + 64u * gi * L.col_bytes + 128 * kc) ^ (int)0x80808080;          // u8 -> s8
                    const fir_i4 b1 = *reinterpret_cast<const fir_i4*>(lb + pb[kc] + 64u * gi * L.col_bytes + 128 * kc) ^ (int)0x80808080;
                    const fir_i8 B = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                    // every re row keeps bytes 0 and 3 of each stream dword (index pairs (0, 3)), every im row bytes 1 and 2
                    acc[gi] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(A[kc], B, acc[gi], (int)0xCCCCCCCCu, 0, 0);
                    acc[GPW + gi] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(A[NKS + kc], B, acc[GPW + gi], (int)0x99999999u, 0, 0);
                }
            }
        }
    }
    for (uint32_t pass = 0; pass < L.n_pass && !FIR_ABLATE(0) && DIGITS != 3; ++pass) {
        if (pass) {
#pragma unroll
            for (int k = 0; k < NKU; ++k) A[k] = amat[(pass * NKU + k) * 64u];
        }
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
            const uint32_t g = wave + 4u * gi;
            if (g < L.groups && (DIGITS == 1 ? 128u : 64u) * g < no) {   // wave-uniform
                const uint32_t col = (16u * g + j) * L.col_bytes + 64u * NKU * pass;
#pragma unroll
                for (int k = 0; k < NKU; ++k) {
                    // SWZ: col is a multiple of 64, so only the low part (16q) sees the bit-5 flip of chunk j + kk
                    const uint32_t low = SWZ ? (16u * q) ^ ((((j + pass * NKU + k) >> 2) & 1u) << 5) : 16u * q;
                    const fir_i4 B = *reinterpret_cast<const fir_i4*>(lb + col + 64 * k + low) ^ (int)0x80808080;   // u8 -> s8
                    acc[gi] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[k], B, acc[gi], 0, 0, 0);
                }
            }
        }
    }
    // the channel's last tile also writes the next call's history (saves the separate launch)
    if (tix == ntiles - 1u && !FIR_ABLATE(3)) {
        typedef const FMD_AS_GLOBAL uint32_t* gw;
        for (uint32_t k = tid; k < L.Hw; k += kFirThreads) {
            const uint64_t w = L.stride_w + k;                      // virtual dword (history ++ call), < Hw + stride_w
            L.hist_out[(uint64_t)c * L.Hw + k] = w < L.Hw ? ((gw)(uintptr_t)L.hist_in)[(uint64_t)c * L.Hw + w]
                                                         : ((gw)(uintptr_t)L.iq)[(uint64_t)c * L.stride_w + (w - L.Hw)];
        }
    }
    if (FIR_ABLATE(9)) __builtin_amdgcn_s_sleep(16);                // ... or in front of the output stores
    if (FIR_ABLATE(10)) __builtin_amdgcn_s_sleep(8);
    // Round 6: the outputs leave the wave in LANE ORDER.  The matrix instruction leaves lane (j, q) = j + 16 q with the outputs of
    // column j -- consecutive LANES then write addresses 64 bytes apart and a store instruction reaches the memory pipeline as 64
    // separate 16-byte pieces (sixteen 64-byte runs in four passes), which is what VERDICT r5 asked to be whole lines.  One
    // ds_bpermute_b32 per register (a crossbar move: no LDS memory, no bank conflicts, no barrier) hands lane l' = 4 j + q the values
    // of lane (j, q): lane l' then holds the group's outputs 2 l', 2 l' + 1 (two digits, dense: output l') and the 64 lanes of a
    // store instruction write 1024 (512) CONTIGUOUS bytes in lane order.  Every lane of the wave takes part in the moves (a
    // masked-off source lane would read as 0); only the stores are predicated.
    const int xsrc = (int)(((lane >> 2) + 16u * (lane & 3u)) << 2);     // byte address of the lane whose values lane l' stores
    if constexpr (DIGITS == 2) {
        // lane (j, q) holds rows 4q..4q+3 of column j: (re_lo, re_hi, im_lo, im_hi) of output 4j + q
        const uint32_t par = (L.par_first ^ (L.half_M * q)) & 1u;
        const int cre = L.mre[par], cim = L.mim[par];
#pragma unroll
        for (int gi = 0; gi < kFirGroupsPerWave; ++gi) {
            const uint32_t g = wave + 4u * gi;
            if (g < L.groups && 64u * g < no && !FIR_ABLATE(2)) {           // wave-uniform
                int re = acc[gi].x + (acc[gi].y << 7), im = acc[gi].z + (acc[gi].w << 7);
                if (L.par_first) { re = -re; im = -im; }
                re += cre; im += cim;
                uint32_t o = 64u * g + 4u * j + q;
                if (!FIR_ABLATE(5)) {                                       // (probe, experiment build: the round-5 store order)
                    re = __builtin_amdgcn_ds_bpermute(xsrc, re); im = __builtin_amdgcn_ds_bpermute(xsrc, im);
                    o = 64u * g + lane;
                }
                if (o < no) {
                    int2* dst = reinterpret_cast<int2*>(L.out) + ((uint64_t)c * L.out_cap + o0 + o);
                    *dst = make_int2(re, im);
                }
            }
        }
    } else {
        // lane (j, q) holds rows 4q..4q+3 of column j: (re, im) of output 8j + 2q and of output 8j + 2q + 1; the first one's window
        // has the column's parity (an even number of windows further on), the second one's is half_M dwords later
        const uint32_t par0 = L.par_first & 1u, par1 = (L.par_first ^ L.half_M) & 1u;
        const int cre0 = L.mre[par0], cim0 = L.mim[par0], cre1 = L.mre[par1], cim1 = L.mim[par1];
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
            const uint32_t g = wave + 4u * gi;
            if (g < L.groups && 128u * g < no && !FIR_ABLATE(2)) {          // wave-uniform
                int re0, im0, re1, im1;
                if constexpr (DIGITS == 3) {                 // (lo, hi) of the two outputs, re and im in accumulators of their own
                    re0 = acc[gi].x + (acc[gi].y << 7); re1 = acc[gi].z + (acc[gi].w << 7);
                    im0 = acc[GPW + gi].x + (acc[GPW + gi].y << 7); im1 = acc[GPW + gi].z + (acc[GPW + gi].w << 7);
                } else {
                    re0 = acc[gi].x; im0 = acc[gi].y; re1 = acc[gi].z; im1 = acc[gi].w;
                }
                if (L.par_first) { re0 = -re0; im0 = -im0; re1 = -re1; im1 = -im1; }
                re0 += cre0; im0 += cim0; re1 += cre1; im1 += cim1;          // (the same constants in every lane: before or behind the moves)
                uint32_t o = 128u * g + 8u * j + 2u * q;
                if (!FIR_ABLATE(5)) {                                       // (probe, experiment build: the round-5 store order)
                    re0 = __builtin_amdgcn_ds_bpermute(xsrc, re0); im0 = __builtin_amdgcn_ds_bpermute(xsrc, im0);
                    re1 = __builtin_amdgcn_ds_bpermute(xsrc, re1); im1 = __builtin_amdgcn_ds_bpermute(xsrc, im1);
                    o = 128u * g + 2u * lane;
                }
                if (o < no) {
                    // both outputs as ONE 16-byte store where the row allows it (an even out_cap: every channel's row 16-byte aligned) -- two
                    // 8-byte stores 16 bytes apart per lane touch every line twice
                    int2* dst = reinterpret_cast<int2*>(L.out) + ((uint64_t)c * L.out_cap + o0 + o);
                    if (o + 1u < no && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0u) {
                        *reinterpret_cast<int4*>(dst) = make_int4(re0, im0, re1, im1);
                    } else {
                        dst[0] = make_int2(re0, im0);
                        if (o + 1u < no) dst[1] = make_int2(re1, im1);
                    }
                }
            }
        }
    }
}

template <int NKU, int DIGITS>
void launch_mfma_d(const FirLaunch& L, dim3 g, size_t lds, hipStream_t stream, bool swz)
{
    // The conflict-free LDS layout (SWZ) removes every bank conflict of the fragment reads (PMC: 10.5 M -> 0 cycles
    // per launch) but the lane-permuted DMA that produces it costs more than the conflicts did: 0.1365 vs 0.1344 ms
    // per config-4 call, loads alone 0.0996 vs 0.0973 ms.  Off unless FMD_FIR_SWZ=1 (experiment build, read at creation).
    if (L.col_bytes == 64u && swz) hipLaunchKernelGGL((fmd_fir_mfma_kernel<NKU, true, DIGITS>), g, dim3(kFirThreads), lds, stream, L);
    else hipLaunchKernelGGL((fmd_fir_mfma_kernel<NKU, false, DIGITS>), g, dim3(kFirThreads), lds, stream, L);
}

template <int NKU>
void launch_mfma(const FirLaunch& L, dim3 g, size_t lds, hipStream_t stream, bool swz, uint32_t digits)
{
    if (digits == 1u) launch_mfma_d<NKU, 1>(L, g, lds, stream, swz);
    else if (digits == 3u) hipLaunchKernelGGL((fmd_fir_mfma_kernel<NKU, false, 3>), g, dim3(kFirThreads), lds, stream, L);
    else launch_mfma_d<NKU, 2>(L, g, lds, stream, swz);
}

// hist_out[c][k] = virtual dword (stride_w + k): the last Hw dwords of history ++ call
__global__ void __launch_bounds__(kFirThreads) fmd_fir_hist_kernel(const FirLaunch L)
{
    const uint64_t gid = (uint64_t)blockIdx.x * kFirThreads + threadIdx.x;
    if (gid >= (uint64_t)L.n_channels * L.Hw) return;
    const uint32_t c = (uint32_t)(gid / L.Hw), k = (uint32_t)(gid - (uint64_t)c * L.Hw);
    const uint64_t w = L.stride_w + k;                    // < Hw + stride_w
    typedef const FMD_AS_GLOBAL uint32_t* gw;
    uint32_t v;
    if (w < L.Hw) v = ((gw)(uintptr_t)L.hist_in)[(uint64_t)c * L.Hw + w];
    else v = ((gw)(uintptr_t)L.iq)[(uint64_t)c * L.stride_w + (w - L.Hw)];
    L.hist_out[(uint64_t)c * L.Hw + k] = v;
}

#define FIR_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            char m_[256];                                                                   \
            snprintf(m_, sizeof m_, "%s failed: %s", #expr, hipGetErrorString(e_));         \
            fmd_internal_set_err(m_);                                                       \
            return FMD_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

#define FIR_ON_DEVICE(dev)                                                                  \
    FmdDeviceGuard dev_guard_(dev);                                                         \
    if (dev_guard_.error() != hipSuccess) { fmd_internal_set_err("hipSetDevice failed"); return FMD_ERR_HIP; }

}  // namespace

struct fmd_fir {
    uint32_t T = 0, M = 0, C = 0, NP = 0, Hw = 0;
    int device = 0;
    uint64_t pos = 0;                                     // samples consumed per channel
    uint32_t* d_wre = nullptr; uint32_t* d_wim = nullptr;
    uint32_t* d_amat = nullptr;                           // MFMA form: banded tap matrix, fragment order
    uint32_t n_pass = 0, nku = 0, groups = 0;             // n_pass == 0: VALU kernel only
    uint32_t digits = 0;                                  // tap digits of the matrix-core form (1: every |tap| <= 127; 2); 0 without it
    uint32_t dbg = 0;                                     // ablation bits of the -DFMD_EXPERIMENT build (fmd_host.h), read at creation
    bool swz = false;
    int32_t mre[2] = {0, 0}, mim[2] = {0, 0};
    uint32_t* d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    int32_t cre[2] = {0, 0}, cim[2] = {0, 0};
    hipStream_t stream = nullptr;
    FmdStreamOrder order;                                 // consecutive launches on different streams (fmd_host.h)
    uint8_t* d_iq = nullptr; size_t d_iq_cap = 0;
    int32_t* d_out = nullptr; size_t d_out_cap = 0;
};

namespace {

void fir_counts(const fmd_fir* f, uint64_t ns, uint64_t* m0, uint64_t* m1)
{
    const uint64_t S = f->pos, T = f->T, M = f->M;
    *m0 = S >= T ? (S - T) / M + 1 : 0;
    *m1 = S + ns >= T ? (S + ns - T) / M + 1 : 0;
}

int fir_enqueue(fmd_fir* f, const void* d_iq, size_t nbytes, void* d_out, size_t out_cap, size_t* n_each, hipStream_t stream)
{
    if (nbytes % 8 != 0) { fmd_internal_set_err("nbytes % 8 != 0"); return FMD_ERR_BAD_LENGTH; }
    if (nbytes == 0 || nbytes > (1ull << 31)) { fmd_internal_set_err("nbytes out of range"); return FMD_ERR_UNSUPPORTED; }
    if (((uintptr_t)d_iq & 3u) != 0 || ((uintptr_t)d_out & 7u) != 0) { fmd_internal_set_err("misaligned device buffer"); return FMD_ERR_INVALID_ARG; }
    const uint64_t ns = nbytes / 2;
    uint64_t m0, m1;
    fir_counts(f, ns, &m0, &m1);
    const uint64_t n_out = m1 - m0;
    if (n_out > out_cap) { fmd_internal_set_err("out_cap too small"); return FMD_ERR_CAPACITY; }
    FIR_TRY(f->order.before(stream));                     // the history ping-pong orders launch n+1 behind launch n
    FirLaunch L{};
    L.iq = static_cast<const uint32_t*>(d_iq);
    L.stride_w = nbytes / 4;
    L.hist_in = f->d_hist[f->cur]; L.hist_out = f->d_hist[f->cur ^ 1];
    L.Hw = f->Hw;
    L.wre = f->d_wre; L.wim = f->d_wim;
    L.cre[0] = f->cre[0]; L.cre[1] = f->cre[1]; L.cim[0] = f->cim[0]; L.cim[1] = f->cim[1];
    L.NP = f->NP; L.half_M = f->M / 2;
    L.n_out = (uint32_t)n_out; L.n_channels = f->C;
    L.out = static_cast<int32_t*>(d_out); L.out_cap = out_cap;
    // first window: stream sample M*m0; virtual sample index = M*m0 - (S - 2*Hw); both even
    const uint64_t vs0 = f->M * m0 + 2ull * f->Hw - f->pos;
    L.wd_first = (uint32_t)(vs0 / 2);
    L.par_first = (uint32_t)((f->M * m0 / 2) & 1u);
    L.par_step = (f->M / 2) & 1u;
    // ~18 KB of LDS per tile
    uint64_t ot = (4500 > L.NP ? (4500 - L.NP) / (L.half_M ? L.half_M : 1) + 1 : 1);
    if (ot > 1024) ot = 1024;
    if (ot < 1) ot = 1;
    L.out_tile = (uint32_t)ot;
    L.dbg = f->dbg;
    if (n_out && f->n_pass) {
        const uint32_t opc = f->digits != 2u ? 8u : 4u;              // outputs per column (one digit, or two on the sparse instruction: eight)
        L.amat = f->d_amat; L.n_pass = f->n_pass; L.nku = f->nku; L.groups = f->groups; L.col_bytes = 2u * opc * f->M;
        L.mre[0] = f->mre[0]; L.mre[1] = f->mre[1]; L.mim[0] = f->mim[0]; L.mim[1] = f->mim[1];
        L.out_tile = 16u * opc * f->groups;
        // staged bytes of a full tile, and the furthest byte any fragment read touches
        const size_t staged = (((((size_t)(L.out_tile - 1) * L.half_M + L.NP + 3) / 4) + 7) & ~(size_t)7) * 16;     // whole 128-byte columns (fir_img)
        const size_t touched = ((size_t)16 * f->groups * L.col_bytes + (size_t)64 * f->n_pass * (f->nku + 1u) + 127) & ~(size_t)127;   // (+1: the sparse form's 128-byte chunks)
        const size_t lds = staged > touched ? staged : touched;
        const uint32_t tiles = (uint32_t)((n_out + L.out_tile - 1) / L.out_tile), per = (f->C + 7u) / 8u;
        if (tiles > 65535u || per > 65535u) { fmd_internal_set_err("call too large for the matrix-core FIR grid"); return FMD_ERR_UNSUPPORTED; }
        const dim3 g(8u, tiles, per);                               // XCD-aware grid: blockIdx.x is the XCD (surplus blocks exit)
        FirHot& H = L.hot;
        H.iq = (uint64_t)(uintptr_t)L.iq; H.stride_w = L.stride_w; H.amat = (uint64_t)(uintptr_t)L.amat;
        H.n_channels = L.n_channels; H.n_out = L.n_out; H.out_tile = L.out_tile;
        H.Hw = L.Hw; H.NP = L.NP; H.half_M = L.half_M; H.wd_first = L.wd_first;
        switch (f->nku) {
            case 1: launch_mfma<1>(L, g, lds, stream, f->swz, f->digits); break;
            case 2: launch_mfma<2>(L, g, lds, stream, f->swz, f->digits); break;
            case 3: launch_mfma<3>(L, g, lds, stream, f->swz, f->digits); break;
            case 4: launch_mfma<4>(L, g, lds, stream, f->swz, f->digits); break;
            case 5: launch_mfma<5>(L, g, lds, stream, f->swz, f->digits); break;
            case 6: launch_mfma<6>(L, g, lds, stream, f->swz, f->digits); break;
            case 7: launch_mfma<7>(L, g, lds, stream, f->swz, f->digits); break;
            default: launch_mfma<8>(L, g, lds, stream, f->swz, f->digits); break;
        }
        FIR_TRY(hipGetLastError());
    } else if (n_out) {
        const size_t lds = (((size_t)(L.out_tile - 1) * L.half_M + L.NP) * 4 + 15) & ~(size_t)15;
        const uint32_t gy = f->C < 65535u ? f->C : 65535u, gz = (f->C + 65534u) / 65535u;
        hipLaunchKernelGGL(fmd_fir_kernel, dim3((uint32_t)((n_out + ot - 1) / ot), gy, gz), dim3(kFirThreads), lds, stream, L);
        FIR_TRY(hipGetLastError());
    }
    if (f->Hw && !(n_out && f->n_pass)) {               // the MFMA kernel's last tiles wrote it already
        const uint64_t th = (uint64_t)f->C * f->Hw;
        hipLaunchKernelGGL(fmd_fir_hist_kernel, dim3((uint32_t)((th + kFirThreads - 1) / kFirThreads)), dim3(kFirThreads), 0, stream, L);
        FIR_TRY(hipGetLastError());
    }
    (void)f->order.after(stream);
    if (f->Hw) f->cur ^= 1;
    f->pos += ns;
    if (n_each) *n_each = (size_t)n_out;
    return FMD_OK;
}

}  // namespace

extern "C" {

size_t fmd_fir_out_cap(uint32_t n_taps, uint32_t decim, size_t nbytes)
{
    if (!decim) return 0;
    (void)n_taps;
    return nbytes / 2 / decim + 2;
}

int fmd_fir_new(const int16_t* taps, uint32_t n_taps, uint32_t decim, const fmd_device_config* dev, fmd_fir** out)
{
    if (!taps || !dev || !out || dev->n_channels == 0) { fmd_internal_set_err("null / empty argument"); return FMD_ERR_INVALID_ARG; }
    *out = nullptr;
    if (n_taps == 0 || n_taps > 1024 || decim == 0 || decim % 2 != 0 || decim > 4096) {
        fmd_internal_set_err("need 1 <= n_taps <= 1024 and an even 2 <= decim <= 4096");
        return FMD_ERR_UNSUPPORTED;
    }
    for (uint32_t t = 0; t < n_taps; ++t)
        if (taps[t] > 2047 || taps[t] < -2047) { fmd_internal_set_err("|tap| > 2047"); return FMD_ERR_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fmd_internal_set_err("no HIP device (this library has no CPU path)"); return FMD_ERR_NO_DEVICE; }
    int device = dev->device_id;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    hipDeviceProp_t prop;
    if (device >= ndev || hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fmd_internal_set_err("device is not a gfx950");
        return FMD_ERR_NO_DEVICE;
    }
    fmd_fir* f = new (std::nothrow) fmd_fir();
    if (!f) return FMD_ERR_NOMEM;
    f->T = n_taps; f->M = decim; f->C = dev->n_channels; f->device = device;
    f->NP = ((n_taps + 1) / 2 + 3u) & ~3u;                // tap pairs, zero-padded to a multiple of 4
    const uint32_t H = n_taps - 1, Hp = H + (H & 1u);     // history samples, rounded up to whole dwords
    f->Hw = Hp / 2;
    // tap pairs with the alternating dword sign folded in, and the additive constants of both window parities
    std::vector<uint32_t> wre(f->NP), wim(f->NP);
    int64_t cre[2] = {0, 0}, cim[2] = {0, 0};
    for (uint32_t i = 0; i < f->NP; ++i) {
        const int h0 = 2 * i < n_taps ? taps[2 * i] : 0, h1 = 2 * i + 1 < n_taps ? taps[2 * i + 1] : 0;
        const int s = (i & 1u) ? -1 : 1;
        wre[i] = (uint32_t)(uint16_t)(int16_t)(s * h0) | ((uint32_t)(uint16_t)(int16_t)(-s * h1) << 16);
        wim[i] = (uint32_t)(uint16_t)(int16_t)(s * h0) | ((uint32_t)(uint16_t)(int16_t)(s * h1) << 16);
        for (int par = 0; par < 2; ++par) {
            const bool even = ((i + par) & 1u) == 0;      // stream parity of this dword when the window starts at `par`
            if (even) { cre[par] += -127 * h0 + 128 * h1; cim[par] += -127 * (h0 + h1); }
            else      { cre[par] += 128 * h0 - 127 * h1;  cim[par] += 128 * (h0 + h1); }
        }
    }
    for (int par = 0; par < 2; ++par) { f->cre[par] = (int32_t)cre[par]; f->cim[par] = (int32_t)cim[par]; }
    // MFMA form: see the header comment.  Rows r = 4*i + reg (reg: re_lo, re_hi, im_lo, im_hi), K index = byte
    // offset v from the window start of output i = 0; fragment order [chunk][lane = row + 16*q][16 bytes].
    std::vector<uint32_t> amat;
    // knobs: -DFMD_EXPERIMENT builds only (fmd_host.h); constants in the shipped library
    const char* env_mfma = fmd_knob("FMD_FIR_MFMA");
    f->dbg = fmd_knob_u32("FMD_DBG", 0);
    f->swz = fmd_knob_u32("FMD_FIR_SWZ", 0) == 1u;
    FmdFirMfmaPlan plan;
    // an 8-bit filter (every |tap| <= 127) takes the one-digit form: twice the outputs per matrix instruction (FMD_FIR_DIGITS=2, experiment
    // build: the two-digit form for an A/B)
    uint32_t digits = 1u;
    for (uint32_t t = 0; t < n_taps; ++t) if (taps[t] > 127 || taps[t] < -127) digits = 2u;
    if (fmd_knob_u32("FMD_FIR_DIGITS", 0) == 2u) digits = 2u;
    // two digits at decim >= 8 with all K chunks in one pass: the sparse form with re / im split (fmd_fir_common.h) -- its lanes hold
    // two adjacent outputs each and store 16 bytes, which is what this store-bound operator is paid for (FMD_FIR_SPARSE=0, experiment
    // build: the dense interleaved form)
    bool split = false;
    if (digits == 2u && decim >= 8u && fmd_knob_u32("FMD_FIR_SPARSE", 1) != 0u && !(env_mfma && env_mfma[0] == '0')) {
        FmdFirMfmaPlan sp;
        if (fmd_fir_build_mfma(taps, n_taps, decim, sp, 2u, true) && sp.n_pass == 1u) { plan = sp; split = true; }
    }
    if (!(env_mfma && env_mfma[0] == '0') && (split || fmd_fir_build_mfma(taps, n_taps, decim, plan, digits))) {
        f->n_pass = plan.n_pass; f->nku = plan.nku; f->digits = split ? 3u : plan.digits;
        uint32_t groups = 16384u / ((f->digits != 2u ? 256u : 128u) * decim);   // ~16 KB of input per tile
        f->groups = groups < 1u ? 1u : (groups > 16u ? 16u : groups);
        if (const char* eg = fmd_knob("FMD_FIR_GROUPS")) { const uint32_t g = (uint32_t)atoi(eg); if (g >= 1 && g <= 4u * kFirGroupsPerWave) f->groups = g; }   // tuning
        if (f->digits == 3u && f->groups > 8u) f->groups = 8u;      // (sparse form: two accumulator pairs per wave)
        if (split) amat.swap(plan.amat_s); else amat.swap(plan.amat);
        for (int par = 0; par < 2; ++par) { f->mre[par] = plan.mre[par]; f->mim[par] = plan.mim[par]; }
    }
    auto fail = [&](const char* what) { fmd_internal_set_err(what); fmd_fir_free(f); return FMD_ERR_HIP; };
    FmdDeviceGuard guard(device);                         // the caller's current device comes back on return
    if (guard.error() != hipSuccess) return fail("hipSetDevice");
    if (!amat.empty()) {
        if (hipMalloc(&f->d_amat, amat.size() * 4) != hipSuccess) return fail("hipMalloc(tap matrix)");
        if (hipMemcpy(f->d_amat, amat.data(), amat.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy(tap matrix)");
    }
    if (hipMalloc(&f->d_wre, f->NP * 4) != hipSuccess || hipMalloc(&f->d_wim, f->NP * 4) != hipSuccess) return fail("hipMalloc(taps)");
    if (hipMemcpy(f->d_wre, wre.data(), f->NP * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(f->d_wim, wim.data(), f->NP * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy(taps)");
    const size_t hb = (size_t)f->C * (f->Hw ? f->Hw : 1) * 4;
    for (int i = 0; i < 2; ++i) {
        if (hipMalloc(&f->d_hist[i], hb) != hipSuccess) return fail("hipMalloc(history)");
        if (hipMemset(f->d_hist[i], 0, hb) != hipSuccess) return fail("hipMemset(history)");
    }
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate");
    if (hipDeviceSynchronize() != hipSuccess) return fail("hipDeviceSynchronize");
    *out = f;
    return FMD_OK;
}

void fmd_fir_free(fmd_fir* f)
{
    if (!f) return;
    FmdDeviceGuard guard(f->device);
    (void)hipDeviceSynchronize();
    f->order.destroy();
    if (f->d_wre) (void)hipFree(f->d_wre);
    if (f->d_wim) (void)hipFree(f->d_wim);
    if (f->d_amat) (void)hipFree(f->d_amat);
    for (int i = 0; i < 2; ++i) if (f->d_hist[i]) (void)hipFree(f->d_hist[i]);
    if (f->d_iq) (void)hipFree(f->d_iq);
    if (f->d_out) (void)hipFree(f->d_out);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
}

int fmd_fir_tap_digits(const fmd_fir* f) { return f ? (int)(f->digits == 3u ? 2u : f->digits) : FMD_ERR_INVALID_ARG; }

int fmd_fir_kernel_name(const fmd_fir* f, char* name, size_t cap)
{
    if (!f || !name || cap == 0) return FMD_ERR_INVALID_ARG;
    // (the kernels live in this file's anonymous namespace; the template arguments as launch_mfma / launch_mfma_d pick them)
    int n;
    if (!f->n_pass) n = snprintf(name, cap, "(anonymous namespace)::fmd_fir_kernel");
    else {
        const uint32_t nku = f->nku < 8u ? (f->nku ? f->nku : 1u) : 8u;
        const uint32_t col_bytes = 2u * (f->digits != 2u ? 8u : 4u) * f->M;
        const bool swz = f->digits != 3u && col_bytes == 64u && f->swz;
        n = snprintf(name, cap, "(anonymous namespace)::fmd_fir_mfma_kernel<%u, %s, %u>", nku, swz ? "true" : "false", f->digits);
    }
    return n < 0 || (size_t)n >= cap ? FMD_ERR_CAPACITY : FMD_OK;
}

int fmd_fir_reset(fmd_fir* f)
{
    if (!f) return FMD_ERR_INVALID_ARG;
    FIR_ON_DEVICE(f->device);
    FIR_TRY(hipDeviceSynchronize());
    f->order.reset();
    const size_t hb = (size_t)f->C * (f->Hw ? f->Hw : 1) * 4;
    FIR_TRY(hipMemset(f->d_hist[0], 0, hb));
    FIR_TRY(hipMemset(f->d_hist[1], 0, hb));
    FIR_TRY(hipDeviceSynchronize());
    f->pos = 0; f->cur = 0;
    return FMD_OK;
}

int fmd_fir_filter_device(fmd_fir* f, const void* d_iq, size_t nbytes, void* d_out, size_t out_cap, size_t* out_len_each,
                          void* stream)
{
    if (!f || !d_iq || !d_out) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    FIR_ON_DEVICE(f->device);
    return fir_enqueue(f, d_iq, nbytes, d_out, out_cap, out_len_each, static_cast<hipStream_t>(stream));
}

int fmd_fir_filter_batch(fmd_fir* f, const uint8_t* iq, size_t nbytes, int32_t* out, size_t out_cap, size_t* out_len)
{
    if (!f || !iq || !out || !out_len) { fmd_internal_set_err("null argument"); return FMD_ERR_INVALID_ARG; }
    FIR_ON_DEVICE(f->device);
    if (nbytes % 8 != 0) { fmd_internal_set_err("nbytes % 8 != 0"); return FMD_ERR_BAD_LENGTH; }
    uint64_t m0, m1;
    fir_counts(f, nbytes / 2, &m0, &m1);
    if (m1 - m0 > out_cap) { fmd_internal_set_err("out_cap too small"); return FMD_ERR_CAPACITY; }
    const size_t in_bytes = nbytes * (size_t)f->C, out_elems = out_cap * (size_t)f->C * 2;
    if (in_bytes > f->d_iq_cap) {
        if (f->d_iq) { FIR_TRY(hipFree(f->d_iq)); f->d_iq = nullptr; f->d_iq_cap = 0; }
        FIR_TRY(hipMalloc(&f->d_iq, in_bytes ? in_bytes : 1));
        f->d_iq_cap = in_bytes;
    }
    if (out_elems > f->d_out_cap) {
        if (f->d_out) { FIR_TRY(hipFree(f->d_out)); f->d_out = nullptr; f->d_out_cap = 0; }
        FIR_TRY(hipMalloc(&f->d_out, (out_elems ? out_elems : 1) * sizeof(int32_t)));
        f->d_out_cap = out_elems;
    }
    FIR_TRY(hipMemcpyAsync(f->d_iq, iq, in_bytes, hipMemcpyHostToDevice, f->stream));
    size_t n = 0;
    int rc = fir_enqueue(f, f->d_iq, nbytes, f->d_out, out_cap, &n, f->stream);
    if (rc) return rc;
    if (n && n * 2 >= out_cap)      // nearly full rows: one linear copy instead of a strided one
        FIR_TRY(hipMemcpyAsync(out, f->d_out, out_cap * 8 * (size_t)f->C, hipMemcpyDeviceToHost, f->stream));
    else if (n) FIR_TRY(hipMemcpy2DAsync(out, out_cap * 8, f->d_out, out_cap * 8, n * 8, f->C, hipMemcpyDeviceToHost, f->stream));
    FIR_TRY(hipStreamSynchronize(f->stream));
    for (uint32_t c = 0; c < f->C; ++c) out_len[c] = n;
    return FMD_OK;
}

}  // extern "C"
