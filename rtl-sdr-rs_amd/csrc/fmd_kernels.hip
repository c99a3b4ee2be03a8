// fmd_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the FM demodulation path.
//
// One kernel fuses every pass of Demod::demodulate (examples/simple_fm.rs:256-269):
//   rotate_90 (:276-299) + `as i16 - 127` (:258) + buf_to_complex (:441-450)
//     -> folded into signed-byte dot products (v_dot4_i32_i8) straight off the raw u8 stream
//   low_pass_complex (:337-352)  -> per-lane window sums out of an LDS-staged tile
//   fm_demod / fast_atan2 (:355-405) incl. the one f64 atan2 sample per call (:359,370-374)
//   low_pass_real (:408-426)     -> per-lane group sums over the tile's discriminator samples
// The intermediate vectors of the reference (512 KiB + 1 MiB + ... per 256 KiB call) never
// exist: HBM traffic is the u8 input once (+ a <1% tile halo) and the s16 output.
//
// Work decomposition: grid = channels x tiles; a tile is `kt` consecutive audio samples of
// one channel-call (fmd_index.h: fmd_tile).  Memory-bound streaming kernel: no MFMA.
#include "fmd_kernels.h"

namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;

__device__ __forceinline__ int sdot4(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);   // v_dot4_i32_i8
}

// Sum of the rotated + centred complex samples n in [n0, n1) of this channel-call, read from the
// LDS image of the raw bytes.  `wofs`: LDS dword index of the call's dword 0 (may be negative).
__device__ __forceinline__ void lds_window_sum(const uint32_t* __restrict__ raw_w, int wofs, int n0, int n1,
                                               int& re, int& im)
{
    int ar = 0, ai = 0;
    const int m0 = n0 >> 1, m1 = (n1 - 1) >> 1;
    for (int m = m0; m <= m1; ++m) {
        const uint32_t w = raw_w[wofs + m] ^ 0x80808080u;     // u8 -> s8 (b - 128)
        uint32_t mask = 0xFFFFFFFFu;
        if (2 * m < n0) mask = 0xFFFF0000u;                   // window starts at the dword's 2nd sample
        if (2 * m + 1 >= n1) mask &= 0x0000FFFFu;             // window ends after the dword's 1st sample
        const bool odd = m & 1;
        ar = sdot4(w, (odd ? FMD_W_RE_ODD : FMD_W_RE_EVEN) & mask, ar);
        ai = sdot4(w, (odd ? FMD_W_IM_ODD : FMD_W_IM_EVEN) & mask, ai);
    }
    if (n1 > n0) {
        ar += fmd_const_re(n1) - fmd_const_re(n0);
        ai += fmd_const_im(n1) - fmd_const_im(n0);
    }
    re = ar; im = ai;
}

// Demod::polar_discriminant (:370-374) on the already-formed product c = a * conj(b).
__device__ __noinline__ int polar_f64(int cr, int ci)
{
    const double angle = atan2((double)ci, (double)cr);
    return (int)(angle / kPi * 16384.0);
}

__device__ __forceinline__ uint32_t pack_lp(int re, int im) { return ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16); }
__device__ __forceinline__ int lp_re(uint32_t p) { return (int)(int16_t)(p & 0xFFFFu); }
__device__ __forceinline__ int lp_im(uint32_t p) { return (int)(int16_t)(p >> 16); }

__global__ void __launch_bounds__(FMD_BLOCK_THREADS) fmd_demod_generic_kernel(const FmdLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* const raw_w = reinterpret_cast<uint32_t*>(smem);
    uint32_t* const lp_pk = reinterpret_cast<uint32_t*>(smem + L.raw_cap);
    int16_t* const d16 = reinterpret_cast<int16_t*>(smem + L.raw_cap + 4u * L.lp_cap);

    const uint32_t tid = threadIdx.x;
    const uint32_t c = blockIdx.x / L.tiles;
    const uint32_t t = blockIdx.x - c * L.tiles;
    const FmdRates r = L.r;

    // ---- per-channel call geometry (wave-uniform) ---------------------------------------
    const FmdChanState st = L.st_in[c];
    const uint32_t p0 = st.prev_index, i0r = st.lpr_index_r;
    const uint32_t M = fmd_num_decimated(r.D, p0, L.ns);
    const uint32_t K = fmd_num_audio(r, i0r, M);
    const uint32_t nt = fmd_num_tiles(r, K);
    if (t >= nt) return;
    const FmdTile T = fmd_tile(r, p0, i0r, L.ns, M, K, nt, t);
    const int jfirst = T.jA - 1;                 // lp[jfirst .. jB] are needed; jfirst == -1 -> demod_pre
    const int cnt = T.jB - jfirst + 1;

    // ---- stage the tile's raw bytes into LDS: coalesced 16-byte loads --------------------
    const uint64_t gbase = (uint64_t)(uintptr_t)L.iq + (uint64_t)c * L.chan_stride;
    const uint64_t gLo = gbase + 2ull * (uint32_t)T.nLo;
    const uint64_t gHi = gbase + 2ull * (uint32_t)T.nHi;
    const uint64_t a0 = gLo & ~15ull;
    const uint32_t nchunks = (uint32_t)((gHi - a0 + 15) >> 4);
    if ((uint32_t)cnt > L.lp_cap || nchunks * 16u > L.raw_cap) {
        if (tid == 0) atomicOr(L.err, (uint32_t)cnt > L.lp_cap ? FMD_DEVERR_LP_CAP : FMD_DEVERR_RAW_CAP);
        return;
    }
    const uint64_t gend = (uint64_t)(uintptr_t)L.iq + L.total_bytes;
    for (uint32_t i = tid; i < nchunks; i += FMD_BLOCK_THREADS) {
        const uint64_t a = a0 + 16ull * i;
        uint4 v;
        if (a + 16 <= gend) {
            v = *reinterpret_cast<const uint4*>((uintptr_t)a);
        } else {   // the array ends in the middle of this chunk (sizes are multiples of 8)
            const uint2 h = *reinterpret_cast<const uint2*>((uintptr_t)a);
            v = make_uint4(h.x, h.y, 0u, 0u);
        }
        reinterpret_cast<uint4*>(smem)[i] = v;
    }
    const int wofs = (int)((int64_t)(gbase - a0) >> 2);      // LDS dword index of the call's dword 0
    __syncthreads();

    // ---- low_pass_complex: one decimated sample per lane per round ------------------------
    for (int i = tid; i < cnt; i += FMD_BLOCK_THREADS) {
        const int j = jfirst + i;
        int re, im;
        if (j < 0) {
            re = st.demod_pre_re; im = st.demod_pre_im;
        } else {
            lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, j), fmd_win_end(r.D, p0, j), re, im);
            if (j == 0) { re += st.lp_now_re; im += st.lp_now_im; }
        }
        lp_pk[i] = pack_lp(re, im);
    }
    __syncthreads();

    // ---- fm_demod: polar discriminator against the predecessor ----------------------------
    for (int i = tid + 1; i < cnt; i += FMD_BLOCK_THREADS) {
        const uint32_t a = lp_pk[i], b = lp_pk[i - 1];
        int cr, ci;
        fmd_mul_conj(lp_re(a), lp_im(a), lp_re(b), lp_im(b), cr, ci);
        int pcm;
        if (jfirst + i == 0) pcm = polar_f64(cr, ci);       // first sample of the call (:359)
        else                 pcm = fmd_fast_atan2(ci, cr);  // (:362)
        d16[i] = (int16_t)pcm;
    }
    __syncthreads();

    // ---- low_pass_real: one audio sample per lane ----------------------------------------
    const uint32_t nk = T.k1 - T.k0;
    int16_t* const outc = L.out + (uint64_t)c * L.out_stride;
    for (uint32_t q = tid; q < nk; q += FMD_BLOCK_THREADS) {
        const int e = (int)(T.eq + (T.er + q * r.fr) / r.sr);
        const int s = q == 0 ? T.jA : (int)(T.eq + (T.er + (q - 1) * r.fr) / r.sr) + 1;
        int sum = (T.k0 + q == 0) ? st.now_lpr : 0;
        for (int j = s; j <= e; ++j) sum += d16[j - jfirst];
        outc[T.k0 + q] = (int16_t)(sum / r.R);
    }

    // ---- Demod state after the call (last tile only) --------------------------------------
    if (T.last && tid == 0) {
        FmdChanState ns_;
        const int s = K == 0 ? 0 : (int)fmd_audio_end(r, i0r, K - 1) + 1;
        int sum = K == 0 ? st.now_lpr : 0;
        for (int j = s; j <= T.jB; ++j) sum += d16[j - jfirst];
        ns_.now_lpr = sum;
        ns_.lpr_index_r = fmd_next_lpr_index_r(r, i0r, M, K);
        ns_.prev_index = fmd_next_prev_index(r.D, p0, L.ns);
        int tr, ti;
        lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, (int)M), (int)L.ns, tr, ti);
        if (M == 0) { tr += st.lp_now_re; ti += st.lp_now_im; }
        ns_.lp_now_re = tr; ns_.lp_now_im = ti;
        if (M == 0) { ns_.demod_pre_re = st.demod_pre_re; ns_.demod_pre_im = st.demod_pre_im; }
        else { const uint32_t l = lp_pk[cnt - 1]; ns_.demod_pre_re = lp_re(l); ns_.demod_pre_im = lp_im(l); }
        ns_.reserved = 0;
        L.st_out[c] = ns_;
        if (L.out_len) L.out_len[c] = K;
    }
}

// =================================================================================================
// Tile kernel: the production path.  Differences from the generic kernel above:
//   * tile geometry comes from host-made per-phase-class plans (FmdClassPlan): no integer division
//     on the device except one exact f32-reciprocal divide per discriminator / audio sample;
//   * every 16-byte load of the tile is in flight before the first wait (up to 8 per lane);
//   * decimation windows of an even downsample are DH whole dwords: 3 VALU ops per dword
//     (xor, 2 x v_dot4_i32_i8), sign by dword parity applied once per window;
//   * the predecessor sample comes from the neighbouring lane (DPP wave_shr:1); each wave-round
//     covers 63 new windows + 1 overlap, so no LDS exchange and no barrier between the boxcar
//     and the discriminator; complex multiply by v_dot2_i32_i16 on packed (re, im).
// =================================================================================================
typedef short fmd_s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t wave_shr1(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(fmd_s2, a), __builtin_bit_cast(fmd_s2, b), 0, false);
}

// c = a * conj(b) for packed (re | im << 16) operands whose components fit i16.
__device__ __forceinline__ void mul_conj_pk(uint32_t a, uint32_t b, int& cr, int& ci)
{
    const uint32_t a_sw = __builtin_amdgcn_alignbit(a, a, 16);               // (im, re)
    const uint32_t b_cj = (b & 0xFFFFu) | ((0u - (b >> 16)) << 16);           // (re, -im)
    cr = sdot2(a, b);
    ci = sdot2(a_sw, b_cj);
}

// DH whole dwords starting at LDS dword index `w0`; `odd` = parity of the call-relative dword index.
template <int DH>
__device__ __forceinline__ void lds_window_fast(const uint32_t* __restrict__ raw_w, int w0, bool odd, int& re, int& im)
{
    int er = 0, ei = 0, orr = 0, oi = 0;
#pragma unroll
    for (int i = 0; i < DH; ++i) {
        const uint32_t w = raw_w[w0 + i] ^ 0x80808080u;
        if (i & 1) { orr = sdot4(w, FMD_W_RE_EVEN, orr); oi = sdot4(w, FMD_W_IM_EVEN, oi); }
        else       { er = sdot4(w, FMD_W_RE_EVEN, er);  ei = sdot4(w, FMD_W_IM_EVEN, ei); }
    }
    const int dr = er - orr, di = ei - oi;
    // additive constants: every dword +1 on re; every even-indexed (call-relative) dword +2 on im
    re = (odd ? -dr : dr) + DH;
    im = (odd ? -di : di) + 2 * (odd ? DH / 2 : (DH + 1) / 2);
}

template <int DH>
__global__ void __launch_bounds__(FMD_BLOCK_THREADS) fmd_demod_tile_kernel(const FmdLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* const raw_w = reinterpret_cast<uint32_t*>(smem);
    int16_t* const d16 = reinterpret_cast<int16_t*>(smem + L.raw_cap);
    uint32_t* const last_lp = reinterpret_cast<uint32_t*>(smem + L.raw_cap + ((2u * L.lp_cap + 15u) & ~15u));

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t c = blockIdx.x / L.tiles;
    const uint32_t t = blockIdx.x - c * L.tiles;
    const FmdRates r = L.r;
    const uint32_t ci = L.chan_class ? L.chan_class[c] : 0u;
    const FmdClassPlan P = L.cls[ci];
    if (t >= P.nt) return;
    const FmdTile T = fmd_tile_fast(r, P, L.Qt, L.ns, t);
    const uint32_t p0 = P.p0;
    const int jfirst = T.jA - 1;
    const int cnt = T.jB - jfirst + 1;

    // ---- stage: all loads in flight, then one pass of ds_write_b128 ------------------------
    const uint64_t gbase = (uint64_t)(uintptr_t)L.iq + (uint64_t)c * L.chan_stride;
    const uint64_t gLo = gbase + 2ull * (uint32_t)T.nLo;
    const uint64_t gHi = gbase + 2ull * (uint32_t)T.nHi;
    const uint64_t a0 = gLo & ~15ull;
    const uint32_t nchunks = (uint32_t)((gHi - a0 + 15) >> 4);
    if ((uint32_t)cnt > L.lp_cap || nchunks * 16u > L.raw_cap || nchunks > FMD_TILE_MAX_LOADS * FMD_BLOCK_THREADS) {
        if (tid == 0) atomicOr(L.err, (uint32_t)cnt > L.lp_cap ? FMD_DEVERR_LP_CAP : FMD_DEVERR_RAW_CAP);
        return;
    }
    const uint64_t gend = (uint64_t)(uintptr_t)L.iq + L.total_bytes;
    if (a0 + 16ull * nchunks <= gend) {
        // LDS-DMA (global_load_lds_dwordx4): 1 KiB per wave-instruction straight into the tile image,
        // destination = wave-uniform base (M0) + lane * 16; no VGPR round trip, no ds_write pass.
        const unsigned char* src = reinterpret_cast<const unsigned char*>((uintptr_t)a0);
#pragma unroll
        for (int l = 0; l < FMD_TILE_MAX_LOADS; ++l) {
            const uint32_t i = tid + l * FMD_BLOCK_THREADS;
            if (i < nchunks)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + 16ull * i),
                    (__attribute__((address_space(3))) void*)(smem + 16u * (i - lane)), 16, 0, 0);
        }
    } else {   // the whole array ends inside this tile's last chunk (sizes are multiples of 8)
        for (uint32_t i = tid; i < nchunks; i += FMD_BLOCK_THREADS) {
            const uint64_t a = a0 + 16ull * i;
            uint4 v;
            if (a + 16 <= gend) v = *reinterpret_cast<const uint4*>((uintptr_t)a);
            else { const uint2 h = *reinterpret_cast<const uint2*>((uintptr_t)a); v = make_uint4(h.x, h.y, 0u, 0u); }
            reinterpret_cast<uint4*>(smem)[i] = v;
        }
    }
    const int wofs = (int)((int64_t)(gbase - a0) >> 2);      // LDS dword index of the call's dword 0
    const bool fastwin = DH > 0 && (p0 & 1u) == 0u;          // windows are whole dwords
    const FmdChanState st = L.st_in[c];
    __syncthreads();

    // ---- boxcar + discriminator, 63 new decimated samples per wave-round --------------------
    constexpr int NW = FMD_BLOCK_THREADS / 64;
    for (int base = (int)wave * 63; base < cnt; base += NW * 63) {
        const int i = base + (int)lane;                       // lane 0 re-does the previous round's last window
        const bool act = i < cnt;
        const int j = jfirst + i;
        int re = 0, im = 0;
        if (fastwin) {
            if (act) {
                const int jj = j < 1 ? 1 : j;
                const int m0 = DH * jj - (int)(p0 >> 1);      // call-relative dword index of the window
                lds_window_fast<(DH > 0 ? DH : 1)>(raw_w, wofs + m0, (m0 & 1) != 0, re, im);
            }
            if (jfirst <= 0 && base == 0 && act && j <= 0) {   // call start only: demod_pre and the clipped first window
                if (j < 0) { re = st.demod_pre_re; im = st.demod_pre_im; }
                else {
                    lds_window_sum(raw_w, wofs, 0, fmd_win_end(r.D, p0, 0), re, im);
                    re += st.lp_now_re; im += st.lp_now_im;
                }
            }
        } else if (act) {
            if (j < 0) { re = st.demod_pre_re; im = st.demod_pre_im; }
            else {
                lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, j), fmd_win_end(r.D, p0, j), re, im);
                if (j == 0) { re += st.lp_now_re; im += st.lp_now_im; }
            }
        }
        const uint32_t pk = pack_lp(re, im);
        const uint32_t prev = wave_shr1(pk);
        if (act && lane > 0) {
            int cr, cim;
            mul_conj_pk(pk, prev, cr, cim);
            int pcm = fmd_fast_atan2_q(cim, cr);              // (:362)
            if (jfirst < 0 && base == 0 && j == 0) pcm = polar_f64(cr, cim);   // first sample of the call (:359)
            d16[i] = (int16_t)pcm;
        }
        if (T.last && act && j == T.jB) last_lp[0] = pk;
    }
    __syncthreads();

    // ---- low_pass_real: one audio sample per lane -----------------------------------------
    const uint32_t nk = T.k1 - T.k0;
    int16_t* const outc = L.out + (uint64_t)c * L.out_stride;
    for (uint32_t q = tid; q < nk; q += FMD_BLOCK_THREADS) {
        const int e = (int)(T.eq + q * L.fa + fmd_udiv_small(T.er + q * L.fb, r.sr, L.inv_sr));
        const int s = q == 0 ? T.jA
                             : (int)(T.eq + (q - 1) * L.fa + fmd_udiv_small(T.er + (q - 1) * L.fb, r.sr, L.inv_sr)) + 1;
        int sum = (T.k0 + q == 0) ? st.now_lpr : 0;
        for (int jj = s; jj <= e; ++jj) sum += d16[jj - jfirst];
        outc[T.k0 + q] = (int16_t)fmd_sdiv_small(sum, r.R, L.inv_R);
    }

    // ---- Demod state after the call (last tile only) ----------------------------------------
    if (T.last && tid == 0) {
        FmdChanState ns_;
        const int s = P.K == 0 ? 0 : (int)fmd_audio_end(r, P.i0r, P.K - 1) + 1;
        int sum = P.K == 0 ? st.now_lpr : 0;
        for (int jj = s; jj <= T.jB; ++jj) sum += d16[jj - jfirst];
        ns_.now_lpr = sum;
        ns_.lpr_index_r = fmd_next_lpr_index_r(r, P.i0r, P.M, P.K);
        ns_.prev_index = fmd_next_prev_index(r.D, p0, L.ns);
        int tr, ti;
        lds_window_sum(raw_w, wofs, fmd_win_begin(r.D, p0, (int)P.M), (int)L.ns, tr, ti);
        ns_.lp_now_re = tr; ns_.lp_now_im = ti;
        const uint32_t l = last_lp[0];                        // lp[M-1]; M >= 2 is guaranteed by the host
        ns_.demod_pre_re = lp_re(l); ns_.demod_pre_im = lp_im(l);
        ns_.reserved = 0;
        L.st_out[c] = ns_;
        if (L.out_len) L.out_len[c] = P.K;
    }
}

// ---- synthetic FM source (integer only; mirrored bit-for-bit by rtl-sdr-rs_amd/synth.py) ------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__device__ __forceinline__ int isin_q15(uint32_t phase)
{
    const int xs = (int)phase >> 16;
    const int ax = xs < 0 ? -xs : xs;
    int y = (xs * (32768 - ax)) >> 13;
    const int ay = y < 0 ? -y : y;
    const int y2 = (y * ay) >> 15;
    y = y + (((y2 - y) * 7373) >> 15);
    return y;
}

__global__ void __launch_bounds__(256) fmd_synth_kernel(const FmdSynthLaunch S)
{
    const uint64_t quads_per_chan = S.chan_stride >> 3;          // 4 complex samples = 8 bytes per thread
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= quads_per_chan * S.n_channels) return;
    const uint32_t c = (uint32_t)(gid / quads_per_chan);
    const uint64_t quad = gid - (uint64_t)c * quads_per_chan;
    const uint64_t hc = mix64(S.seed + c + 0x9E3779B97F4A7C15ull);
    const uint32_t chan_phase = (uint32_t)hc;
    const uint32_t mod_step = (uint32_t)(0x100000000ull / S.mod_period);
    const uint32_t mod_step_c = mod_step + (c % 61u) * (mod_step >> 6);
    const uint32_t beta_q16 = (uint32_t)(((uint64_t)S.dev_q32 * S.mod_period * 10430ull) >> 32);
    const int span = 2 * (int)S.noise + 1;
    uint32_t w[2];
    for (int h = 0; h < 2; ++h) {
        uint32_t word = 0;
        for (int s = 0; s < 2; ++s) {
            const uint64_t n = S.sample_offset + quad * 4 + (uint64_t)(h * 2 + s);
            const uint32_t am = (uint32_t)(n * mod_step_c) + (uint32_t)(hc >> 32);
            const uint32_t dphi = (uint32_t)((int64_t)beta_q16 * (int64_t)isin_q15(am) * 2);
            const uint32_t theta = (uint32_t)(0xC0000000ull * n) + chan_phase + dphi;
            const int I = ((int)S.amplitude * isin_q15(theta + 0x40000000u)) >> 15;
            const int Q = ((int)S.amplitude * isin_q15(theta)) >> 15;
            const uint64_t hz = mix64(hc ^ (n * 0x9E3779B97F4A7C15ull));
            const int nI = (int)(((hz & 0xFFFFull) * (uint64_t)span) >> 16) - (int)S.noise;
            const int nQ = (int)((((hz >> 16) & 0xFFFFull) * (uint64_t)span) >> 16) - (int)S.noise;
            int bI = 127 + I + nI, bQ = 127 + Q + nQ;
            bI = bI < 0 ? 0 : (bI > 255 ? 255 : bI);
            bQ = bQ < 0 ? 0 : (bQ > 255 ? 255 : bQ);
            word |= ((uint32_t)bI | ((uint32_t)bQ << 8)) << (16 * s);
        }
        w[h] = word;
    }
    *reinterpret_cast<uint2*>(S.iq + (uint64_t)c * S.chan_stride + quad * 8) = make_uint2(w[0], w[1]);
}

}  // namespace

size_t fmd_generic_lds_bytes(const FmdLaunch& L)
{
    return ((size_t)L.raw_cap + 4u * (size_t)L.lp_cap + 2u * (size_t)L.lp_cap + 15u) & ~(size_t)15u;
}

size_t fmd_tile_lds_bytes(const FmdLaunch& L)
{
    return (size_t)L.raw_cap + ((2u * (size_t)L.lp_cap + 15u) & ~(size_t)15u) + 16u;
}

bool fmd_tile_kernel_supports(const FmdRates& r, uint32_t raw_cap)
{
    if (r.kt % r.sr != 0) return false;                                   // plans need kt*fr = Qt*sr
    if (r.D > 64) return false;                                           // fmd_sdiv_q13: x + |y| < 2^30
    if ((uint64_t)r.sr * (r.kt + 2) >= (1u << 24)) return false;          // fmd_udiv_small operands
    if ((uint64_t)((r.fr + r.sr - 1) / r.sr + 2) * 32768ull >= (1u << 24)) return false;   // |group sum| < 2^24
    if ((uint32_t)r.R >= (1u << 24)) return false;
    if (raw_cap > 16u * FMD_TILE_MAX_LOADS * FMD_BLOCK_THREADS) return false;
    return true;
}

hipError_t fmd_launch_generic(const FmdLaunch& L, hipStream_t stream)
{
    const size_t lds = fmd_generic_lds_bytes(L);
    const uint64_t blocks = (uint64_t)L.n_channels * L.tiles;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fmd_demod_generic_kernel, dim3((uint32_t)blocks), dim3(FMD_BLOCK_THREADS), lds, stream, L);
    return hipGetLastError();
}

hipError_t fmd_launch_tile(const FmdLaunch& L, hipStream_t stream)
{
    const size_t lds = fmd_tile_lds_bytes(L);
    const uint64_t blocks = (uint64_t)L.n_channels * L.tiles;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 g((uint32_t)blocks), b(FMD_BLOCK_THREADS);
    const uint32_t dh = (L.r.D % 2 == 0) ? L.r.D / 2 : 0;
    switch (dh) {
        case 1: hipLaunchKernelGGL(fmd_demod_tile_kernel<1>, g, b, lds, stream, L); break;
        case 2: hipLaunchKernelGGL(fmd_demod_tile_kernel<2>, g, b, lds, stream, L); break;
        case 3: hipLaunchKernelGGL(fmd_demod_tile_kernel<3>, g, b, lds, stream, L); break;   // cfg-ref, D = 6
        case 4: hipLaunchKernelGGL(fmd_demod_tile_kernel<4>, g, b, lds, stream, L); break;
        case 5: hipLaunchKernelGGL(fmd_demod_tile_kernel<5>, g, b, lds, stream, L); break;   // 2.4 Msps, D = 10
        default: hipLaunchKernelGGL(fmd_demod_tile_kernel<0>, g, b, lds, stream, L); break;  // generic windows
    }
    return hipGetLastError();
}

hipError_t fmd_launch_synth(const FmdSynthLaunch& S, hipStream_t stream)
{
    const uint64_t threads = (S.chan_stride >> 3) * S.n_channels;
    const uint64_t blocks = (threads + 255) / 256;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fmd_synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, S);
    return hipGetLastError();
}
