// vorbis_walk.hip -- Vorbis inverse MDCT + window / overlap-add in AFG_NUMERIC_TOLERANCE (round 4; other sizes round 5).
//
// Same seam as vorbis_transform.hip (reference stb_vorbis2.d:2526-2527 inverse_mdct, :2606-2657 vorbis_finish_frame,
// :3927-3952 interleave), same records, same walk over 16-packet segments -- but NOT the reference's factorisation.
// stb_vorbis' inverse_mdct (:1941-2242) is an 8-step in-place algorithm whose steps hand the n/2 values to each other in
// six different orders; scheduling it bit for bit costs six pass -> LDS -> pass boundaries per transform
// (vorbis_transform.hip).  north_star asks for 1e-5 RMS, so this file computes the same transform,
//
//     y[m] = sum_{k < n/2} X[k] cos(pi/(2n) (2m + 1 + n/2)(2k + 1)),   n = blocksize_1 in {1024, 2048, 4096}   
//
// the textbook way: y is the odd/even extension of the DCT-IV u of X, and u comes from ONE complex FFT of N = n/4 points,
//
//     t[q] = (X[2q] + i X[n/2-1-2q]) w[q],   Z = FFT_N(t),   c[k] = Z[k] w[k],   w[k] = exp(-2 pi i (k + 1/8) / n)
//     u[2k] = Re c[k],   u[n/2-1-2k] = -Im c[k]
//
// held R = N/64 points per lane: register passes with transposes through LDS between them,
//
//     R = 4  (n = 1024):  4 x 4 x 4 x 4        R = 8  (n = 2048):  8 x 8 x 8        R = 16 (n = 4096): 16 x 4 x 16
//
// (tests/vorbis_walk_model.py restates the index algebra of every pass on [64 lanes][R] arrays and checks it against a
// library FFT, and every LDS access for bank conflicts, on the CPU.)
// Window and overlap never see y: output frames j and n/2-1-j of a long block that follows a long block are
//
//     out[j]        =  a w[j]        - b w[n/2-1-j]          a = u_cur[n/4+j],  b = u_prev[n/4-1-j]
//     out[n/2-1-j]  = -a w[n/2-1-j]  - b w[j]
//
// (time-domain alias cancellation written on u), and both a and b of frames j, j+1 come from the pair c[k], c[N-1-k] with
// k = N/2 + j/2.  The lane that ends the FFT with c[k] therefore keeps the carried state of exactly its own frames: R
// floats per channel instead of the n/128 of previous_window (:2641-2643), and the PCM leaves as 16-byte stores of two
// interleaved stereo frames (8-byte stores of two frames for a mono stream).  The mirror k <-> N-1-k -- needed once on the
// way in (X[2q] and X[n/2-1-2q] arrive in different lanes) and once on the way out -- is an exchange between lanes l and
// l ^ 32 (v_permlane32_swap), because lanes 32..63 walk their point groups in descending order.
//
// Packets that are not "long between two long blocks" (short blocks, the long blocks beside them) take the reference's
// own arithmetic for the parts that differ: short blocks run vorbis_core.h's inverse_mdct_lds, long blocks with a short
// neighbour scatter u to LDS and window through a y(m) accessor.  One wavefront walks all channels (1 or 2) of a segment;
// of a stream with more channels, one channel or one pair of channels (their column of the interleaved frames).
// Streams this walk does not take (other long block sizes, blocksize_0 > 512 unless it is blocksize_1 itself) stay on the
// bit-exact kernels, which are within any tolerance.
//
// Compiled with -ffp-contract=fast (Makefile): multiply-adds fuse.  Error against the oracle on the C3 workload (N(0,1)
// spectra, output RMS 6.8): 1e-6 RMS, i.e. 1.5e-7 of the signal (tests/test_vorbis_walk_gpu.py).
#include "afg_common.h"
#include "afg_pk.h"
#include "vorbis_core.h"
#include "vorbis_walk.h"

#include <mutex>
#include <type_traits>

#include <cmath>

namespace afg_vorbis {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));           // (f2: afg_pk.h)

// Geometry of one long block size: R points per lane.  Table block (walk_build_tables): W[N] | pass twiddles as float2,
// then the window of the long block (n/2 floats).
template <int R>
struct Geo {
    static constexpr int kN = 256 * R;                              // blocksize_1
    static constexpr int kPts = 64 * R;                             // N: complex points of the FFT
    // complex slots of one channel's transform area: the padded transposes (tests/vorbis_walk_model.py: highest slot 303 /
    // 567 / 1143), u in natural order (n/2 floats) and a short block with its scratch (1.5 blocksize_0 <= 768 floats)
    static constexpr int kChanF2 = R == 4 ? 384 : R == 8 ? 576 : 1152;
    static constexpr int kTw1 = kPts;                               // W_N^(j k), k = 1..R-1: [R-1][64]
    static constexpr int kTw2 = kTw1 + (R - 1) * 64;                // R = 8: W_64^(n0 k) [7][8]; else W_64^(m k) [3][16]
    static constexpr int kTw3 = kTw2 + (R == 8 ? 56 : 48);          // R = 4: W_16^(n4 k) [3][4]
    static constexpr int kTwEnd = kTw3 + (R == 4 ? 12 : 0);
    static constexpr int kTabFloats = 2 * kTwEnd + kN / 2;          // 1528 / 3056 / 6112
    static_assert(kTabFloats % 4 == 0, "staged in 16-byte pieces");
};

// complex product: a pair of packed instructions (round 6; four scalar ones before)
__device__ __forceinline__ f2 cmul(f2 a, f2 w) { return pk_cmul_fused(a, w); }

// 4-point DFT, forward, natural order in and out
__device__ __forceinline__ void dft4(f2 &a0, f2 &a1, f2 &a2, f2 &a3)
{
    const f2 s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a2 = s02 - s13;
    a1 = f2{ d02.x + d13.y, d02.y - d13.x };
    a3 = f2{ d02.x - d13.y, d02.y + d13.x };
}

// 8-point DFT, forward (exp(-2 pi i a b / 8)), natural order in and out
__device__ __forceinline__ void dft8(f2 (&a)[8])
{
    constexpr float s = 0.70710678118654752440f;
    const f2 b0 = a[0] + a[4], b4 = a[0] - a[4], b1 = a[1] + a[5], b5 = a[1] - a[5];
    const f2 b2 = a[2] + a[6], b6 = a[2] - a[6], b3 = a[3] + a[7], b7 = a[3] - a[7];
    const f2 c0 = b0 + b2, c2 = b0 - b2, c1 = b1 + b3, c3 = b1 - b3;
    a[0] = c0 + c1;
    a[4] = c0 - c1;
    a[2] = f2{ c2.x + c3.y, c2.y - c3.x };
    a[6] = f2{ c2.x - c3.y, c2.y + c3.x };
    const f2 d0 = f2{ b4.x + b6.y, b4.y - b6.x }, d2 = f2{ b4.x - b6.y, b4.y + b6.x };
    const f2 p5 = f2{ b5.x + b5.y, b5.y - b5.x }, p7 = f2{ b7.y - b7.x, -(b7.x + b7.y) };
    const f2 D1 = p5 + p7, D3 = p5 - p7;
    a[1] = f2{ d0.x + s * D1.x, d0.y + s * D1.y };
    a[5] = f2{ d0.x - s * D1.x, d0.y - s * D1.y };
    a[3] = f2{ d2.x + s * D3.y, d2.y - s * D3.x };
    a[7] = f2{ d2.x - s * D3.y, d2.y + s * D3.x };
}

// a * exp(-2 pi i M / 16) for the exponents a 4 x 4 split of 16 points meets (0, 1, 2, 3, 4, 6, 9)
template <int M>
__device__ __forceinline__ f2 mul_w16(f2 a)
{
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    if constexpr (M == 0) return a;
    else if constexpr (M == 1) return f2{ a.x * c1 + a.y * s1, a.y * c1 - a.x * s1 };
    else if constexpr (M == 2) return f2{ (a.x + a.y) * h, (a.y - a.x) * h };
    else if constexpr (M == 3) return f2{ a.x * s1 + a.y * c1, a.y * s1 - a.x * c1 };
    else if constexpr (M == 4) return f2{ a.y, -a.x };
    else if constexpr (M == 6) return f2{ (a.y - a.x) * h, -(a.x + a.y) * h };
    else {
        static_assert(M == 9, "exponent");
        return f2{ -(a.x * c1 + a.y * s1), a.x * s1 - a.y * c1 };
    }
}

// 16-point DFT, forward, natural order in and out: index n = nl + 4 nh, four 4-point DFTs over nh, W_16^(nl ka), four over nl
__device__ __forceinline__ void dft16(f2 (&a)[16])
{
#pragma unroll
    for (int nl = 0; nl < 4; nl++) dft4(a[nl], a[nl + 4], a[nl + 8], a[nl + 12]);       // a[nl + 4 ka]
    a[5] = mul_w16<1>(a[5]);
    a[9] = mul_w16<2>(a[9]);
    a[13] = mul_w16<3>(a[13]);
    a[6] = mul_w16<2>(a[6]);
    a[10] = mul_w16<4>(a[10]);
    a[14] = mul_w16<6>(a[14]);
    a[7] = mul_w16<3>(a[7]);
    a[11] = mul_w16<6>(a[11]);
    a[15] = mul_w16<9>(a[15]);
    // over nl for each ka: inputs a[nl + 4 ka], outputs k = ka + 4 kb
    f2 o[16];
#pragma unroll
    for (int ka = 0; ka < 4; ka++) {
        f2 x0 = a[4 * ka], x1 = a[4 * ka + 1], x2 = a[4 * ka + 2], x3 = a[4 * ka + 3];
        dft4(x0, x1, x2, x3);
        o[ka] = x0;
        o[ka + 4] = x1;
        o[ka + 8] = x2;
        o[ka + 12] = x3;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = o[k];
}

// ---- channel-packed arithmetic (round 6) ----------------------------------------------------------------------------------
// A stereo wavefront runs the same transform on both channels.  Held as c2 -- re = (ch0, ch1), im = (ch0, ch1), each a register
// pair -- every real operation of the transform is ONE packed instruction for the two channels (v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32: 2 x 64 lanes per issue slot), twiddles and window taps are broadcast through op_sel, multiplications by +-i
// are register renaming, and the windowed frames come out as (ch0, ch1) pairs: the order of the interleaved PCM.  Against
// the complex-pair form (re, im of one channel per register pair, above: what a mono stream runs) a stereo long block needs
// about 410 vector instructions instead of 725 -- the walk was issue-bound, not memory-bound (profiles/r06_pmc_vorbis_walk_kernel.json:
// VALU busy 0.61 of a SIMD's cycles at two wavefronts, fetch 1.07 x the bytes needed).
struct c2 { f2 re, im; };
__device__ __forceinline__ c2 operator+(c2 a, c2 b) { return c2{ a.re + b.re, a.im + b.im }; }
__device__ __forceinline__ c2 operator-(c2 a, c2 b) { return c2{ a.re - b.re, a.im - b.im }; }
__device__ __forceinline__ c2 add_mi(c2 a, c2 b) { return c2{ a.re + b.im, a.im - b.re }; }       // a + (-i) b
__device__ __forceinline__ c2 add_pi(c2 a, c2 b) { return c2{ a.re - b.im, a.im + b.re }; }       // a + i b
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{ v, v }; }
// both channels' values times one complex factor
__device__ __forceinline__ c2 cmulw(c2 a, f2 w)
{
    const f2 wr = w.xx, wi = w.yy;
    return c2{ fma2(-a.im, wi, a.re * wr), fma2(a.im, wr, a.re * wi) };
}

__device__ __forceinline__ void dft4(c2 &a0, c2 &a1, c2 &a2, c2 &a3)
{
    const c2 s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a2 = s02 - s13;
    a1 = add_mi(d02, d13);
    a3 = add_pi(d02, d13);
}

__device__ __forceinline__ void dft8(c2 (&a)[8])
{
    const f2 S = splat(0.70710678118654752440f);
    const c2 b0 = a[0] + a[4], b4 = a[0] - a[4], b1 = a[1] + a[5], b5 = a[1] - a[5];
    const c2 b2 = a[2] + a[6], b6 = a[2] - a[6], b3 = a[3] + a[7], b7 = a[3] - a[7];
    const c2 c0 = b0 + b2, c2_ = b0 - b2, c1 = b1 + b3, c3 = b1 - b3;
    a[0] = c0 + c1;
    a[4] = c0 - c1;
    a[2] = add_mi(c2_, c3);
    a[6] = add_pi(c2_, c3);
    const c2 d0 = add_mi(b4, b6), d2 = add_pi(b4, b6);
    // p5 = b5 (1 - i), p7 = -b7 (1 + i) (both without their factor 1/sqrt 2): p5 = (b5.re + b5.im, b5.im - b5.re),
    // p7 = (b7.im - b7.re, -(b7.re + b7.im)); q7 holds p7 with the sign of its imaginary part flipped
    const c2 p5 = c2{ b5.re + b5.im, b5.im - b5.re }, q7 = c2{ b7.im - b7.re, b7.re + b7.im };
    const c2 D1 = c2{ p5.re + q7.re, p5.im - q7.im }, D3 = c2{ p5.re - q7.re, p5.im + q7.im };
    a[1] = c2{ fma2(D1.re, S, d0.re), fma2(D1.im, S, d0.im) };
    a[5] = c2{ fma2(-D1.re, S, d0.re), fma2(-D1.im, S, d0.im) };
    a[3] = c2{ fma2(D3.im, S, d2.re), fma2(-D3.re, S, d2.im) };
    a[7] = c2{ fma2(-D3.im, S, d2.re), fma2(D3.re, S, d2.im) };
}

template <int M>
__device__ __forceinline__ c2 mul_w16(c2 a)
{
    const f2 c1 = splat(0.92387953251128675613f), s1 = splat(0.38268343236508977173f), h = splat(0.70710678118654752440f);
    if constexpr (M == 0) return a;
    else if constexpr (M == 1) return c2{ fma2(a.im, s1, a.re * c1), fma2(-a.re, s1, a.im * c1) };
    else if constexpr (M == 2) return c2{ (a.re + a.im) * h, (a.im - a.re) * h };
    else if constexpr (M == 3) return c2{ fma2(a.im, c1, a.re * s1), fma2(-a.re, c1, a.im * s1) };
    else if constexpr (M == 4) return c2{ a.im, -a.re };
    else if constexpr (M == 6) return c2{ (a.im - a.re) * h, -((a.re + a.im) * h) };
    else {
        static_assert(M == 9, "exponent");
        return c2{ -fma2(a.im, s1, a.re * c1), fma2(a.re, s1, -(a.im * c1)) };
    }
}

__device__ __forceinline__ void dft16(c2 (&a)[16])
{
#pragma unroll
    for (int nl = 0; nl < 4; nl++) dft4(a[nl], a[nl + 4], a[nl + 8], a[nl + 12]);
    a[5] = mul_w16<1>(a[5]);
    a[9] = mul_w16<2>(a[9]);
    a[13] = mul_w16<3>(a[13]);
    a[6] = mul_w16<2>(a[6]);
    a[10] = mul_w16<4>(a[10]);
    a[14] = mul_w16<6>(a[14]);
    a[7] = mul_w16<3>(a[7]);
    a[11] = mul_w16<6>(a[11]);
    a[15] = mul_w16<9>(a[15]);
    c2 o[16];
#pragma unroll
    for (int ka = 0; ka < 4; ka++) {
        c2 x0 = a[4 * ka], x1 = a[4 * ka + 1], x2 = a[4 * ka + 2], x3 = a[4 * ka + 3];
        dft4(x0, x1, x2, x3);
        o[ka] = x0;
        o[ka + 4] = x1;
        o[ka + 8] = x2;
        o[ka + 12] = x3;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = o[k];
}

// an ordering point for this wavefront's own LDS accesses: the hardware runs them in program order, the compiler is held by
// the fence (wave_barrier alone is declared to touch no memory); no instruction, no wait
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void lane_swap32(float &x, float &y)      // x of lanes 32..63 <-> y of lanes 0..31
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
}
// a <- b of lane ^ 32, b <- a of lane ^ 32
__device__ __forceinline__ void cross32(float &a, float &b)
{
    lane_swap32(a, b);
    lane_swap32(b, a);
}

template <int CH, int R>
__device__ __forceinline__ void settle(f2 (&x)[CH][R])
{
#pragma unroll
    for (int c = 0; c < CH; c++)
#pragma unroll
        for (int r = 0; r < R; r += 4)
            asm volatile("" : "+v"(x[c][r]), "+v"(x[c][r + 1]), "+v"(x[c][r + 2]), "+v"(x[c][r + 3]) : : "memory");
}

// Lane geometry (constant for the life of a wavefront).  Group index j: lanes 0..31 take j = lane, lanes 32..63 take
// j = 95 - lane (63..32), so that lane ^ 32 always holds group 63 - j.
// Every phase derives what it needs from an opaque copy of the lane id: addresses computed from a plain threadIdx.x are
// hoisted out of the packet loop and then sit in registers (or scratch) for the whole walk.
__device__ __forceinline__ int fresh_lane()
{
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    return l;
}
__device__ __forceinline__ int group_of(int lane) { return lane < 32 ? lane : 95 - lane; }

// The twiddles a lane multiplies by are the same for every packet it ever transforms (its point group, its butterfly):
// where the register budget has room -- the stereo 2048 kernel: 181 -> 239 registers at the two wavefronts per SIMD it runs
// at anyway -- they live in registers for the life of the wavefront: 30 of a packet's 88 LDS reads are not issued (C3
// 6.99 -> 6.91 ms on one box, the same bits).
template <int R>
struct LaneTw {
    f2 w[R], w1[R - 1], w2[7];
};
template <int R>
__device__ __forceinline__ void load_lane_tw(LaneTw<R> &tw, const f2 *T)
{
    using G = Geo<R>;
    const int lane = threadIdx.x & 63, j = group_of(lane);
#pragma unroll
    for (int r = 0; r < R; r++) tw.w[r] = T[j + 64 * r];
#pragma unroll
    for (int k = 1; k < R; k++) tw.w1[k - 1] = T[G::kTw1 + j + 64 * (k - 1)];
    if constexpr (R == 8) {
#pragma unroll
        for (int k = 1; k < 8; k++) tw.w2[k - 1] = T[G::kTw2 + (lane >> 3) + 8 * (k - 1)];
    }
}
// which of them: the pre- / post-twiddle w (R values), the pass-1 twiddles (R - 1), pass 2's (R = 8)
#ifndef AFG_WALK_TW
#define AFG_WALK_TW 3       // stereo 2048: which sets live in registers (bit 0: w, bit 1: pass 1's, bit 2: pass 2's)
#endif
template <int R, int CH> constexpr bool kTwRegs1 = R == 8 && CH == 2 && (AFG_WALK_TW & 2);
template <int R, int CH> constexpr bool kTwRegs2 = R == 8 && CH == 2 && (AFG_WALK_TW & 4);
template <int R, int CH> constexpr bool kTwRegsW = (R == 8 && CH == 2 && (AFG_WALK_TW & 1)) || (R == 16 && CH == 1);   // (mono 4096: 197 -> 245 registers, +0.6 %)
template <int R, int CH> constexpr bool kTwAny = kTwRegsW<R, CH> || kTwRegs1<R, CH> || kTwRegs2<R, CH>;

// The passes between the pre- and the post-twiddle: e[c][r] = point j + 64 r in, e[c][s] = bin j + 64 s out (not yet
// multiplied by w).  U: the wavefront's transform area (channel c at U + c kChanF2); T: the table block.
template <int R, int CH>
__device__ __forceinline__ void fft_passes(f2 (&e)[CH][R], f2 *U, const f2 *T, const LaneTw<R> &tw)
{
    using G = Geo<R>;
    constexpr int F = G::kChanF2;
    int j = group_of(fresh_lane());
    if constexpr (R == 8) {
        // pass 1: over r = q >> 6 -> k0; twiddle W512^(j k0)
#pragma unroll
        for (int c = 0; c < CH; c++) dft8(e[c]);
        {
            const f2 *W1 = T + G::kTw1 + j;
#pragma unroll
            for (int k = 1; k < 8; k++) {
                const f2 w = kTwRegs1<R, CH> ? tw.w1[k - 1] : W1[64 * (k - 1)];
#pragma unroll
                for (int c = 0; c < CH; c++) e[c][k] = cmul(e[c][k], w);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int k = 0; k < 8; k++) U[c * F + j + 68 * k] = e[c][k];
        wave_lds_sync();
        // pass 2: lane (n0, k0) = (lane >> 3, lane & 7) over n1 -> k1; reads U[n0 + 8 n1 + 68 k0], twiddle W64^(n0 k1),
        // writes V[k0 + 8 k1 + 72 n0] (the same area: all reads are issued before the first write)
        const int l2 = fresh_lane();
        const int n0 = l2 >> 3, r2 = n0 + 68 * (l2 & 7), w2 = (l2 & 7) + 72 * n0;
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int k = 0; k < 8; k++) e[c][k] = U[c * F + r2 + 8 * k];
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < CH; c++) dft8(e[c]);
        {
            const f2 *W2 = T + G::kTw2 + n0;
#pragma unroll
            for (int k = 1; k < 8; k++) {
                const f2 w = kTwRegs2<R, CH> ? tw.w2[k - 1] : W2[8 * (k - 1)];
#pragma unroll
                for (int c = 0; c < CH; c++) e[c][k] = cmul(e[c][k], w);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int k = 0; k < 8; k++) U[c * F + w2 + 8 * k] = e[c][k];
        wave_lds_sync();
        // pass 3: lane (k0 + 8 k1 = j) over n0 -> k2
        j = group_of(fresh_lane());
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int k = 0; k < 8; k++) e[c][k] = U[c * F + j + 72 * k];
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < CH; c++) dft8(e[c]);
    } else if constexpr (R == 16) {
        // 16 x 4 x 16: q = n3 + 16 n2 + 64 n1 -> k = k1 + 16 k2 + 64 k3.  pass 1: over r = n1 -> k1; twiddle W1024^(j k1)
#pragma unroll
        for (int c = 0; c < CH; c++) dft16(e[c]);
        {
            const f2 *W1 = T + G::kTw1 + j;
#pragma unroll
            for (int k = 1; k < 16; k++) {
                const f2 w = W1[64 * (k - 1)];
#pragma unroll
                for (int c = 0; c < CH; c++) e[c][k] = cmul(e[c][k], w);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int k = 0; k < 16; k++) U[c * F + j + 72 * k] = e[c][k];
        wave_lds_sync();
        // pass 2: lane (n3, g) = (lane >> 2, lane & 3) takes the items k1 = g + 4 i, each a 4-point DFT over n2 -> k2:
        // reads U[n3 + 16 n2 + 72 k1], twiddle W64^(n3 k2), writes V[k1 + 16 k2 + 68 n3]
        const int l2 = fresh_lane();
        const int n3 = l2 >> 2, r2 = n3 + 72 * (l2 & 3), w2 = (l2 & 3) + 68 * n3;
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int n2 = 0; n2 < 4; n2++) e[c][4 * i + n2] = U[c * F + r2 + 16 * n2 + 288 * i];
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) dft4(e[c][4 * i], e[c][4 * i + 1], e[c][4 * i + 2], e[c][4 * i + 3]);
        {
            const f2 *W2 = T + G::kTw2 + n3;
#pragma unroll
            for (int k = 1; k < 4; k++) {
                const f2 w = W2[16 * (k - 1)];
#pragma unroll
                for (int c = 0; c < CH; c++)
#pragma unroll
                    for (int i = 0; i < 4; i++) e[c][4 * i + k] = cmul(e[c][4 * i + k], w);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 0; k < 4; k++) U[c * F + w2 + 4 * i + 16 * k] = e[c][4 * i + k];
        wave_lds_sync();
        // pass 3: lane j = k1 + 16 k2 over n3 -> k3
        j = group_of(fresh_lane());
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int k = 0; k < 16; k++) e[c][k] = U[c * F + j + 68 * k];
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < CH; c++) dft16(e[c]);
    } else {
        static_assert(R == 4, "points per lane");
        // 4 x 4 x 4 x 4: q = n4 + 4 n3 + 16 n2 + 64 n1 -> k = k1 + 4 k2 + 16 k3 + 64 k4.  pass 1: over r = n1 -> k1
#pragma unroll
        for (int c = 0; c < CH; c++) dft4(e[c][0], e[c][1], e[c][2], e[c][3]);
        {
            const f2 *W1 = T + G::kTw1 + j;
#pragma unroll
            for (int k = 1; k < 4; k++) {
                const f2 w = W1[64 * (k - 1)];
#pragma unroll
                for (int c = 0; c < CH; c++) e[c][k] = cmul(e[c][k], w);
            }
        }
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int k = 0; k < 4; k++) U[c * F + j + 80 * k] = e[c][k];
        wave_lds_sync();
        // pass 2: lane (m, k1) = (lane & 15, lane >> 4), m = n4 + 4 n3, over n2 -> k2: reads U[m + 16 n2 + 80 k1], twiddle
        // W64^(m k2), writes slot m + 16 kk + 4 (kk >> 1) of kk = k1 + 4 k2
        {
            const int l2 = fresh_lane();
            const int m = l2 & 15, k1 = l2 >> 4, r2 = m + 80 * k1, w2 = m + 16 * k1 + 4 * (k1 >> 1);
#pragma unroll
            for (int c = 0; c < CH; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) e[c][k] = U[c * F + r2 + 16 * k];
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < CH; c++) dft4(e[c][0], e[c][1], e[c][2], e[c][3]);
            const f2 *W2 = T + G::kTw2 + m;
#pragma unroll
            for (int k = 1; k < 4; k++) {
                const f2 w = W2[16 * (k - 1)];
#pragma unroll
                for (int c = 0; c < CH; c++) e[c][k] = cmul(e[c][k], w);
            }
#pragma unroll
            for (int c = 0; c < CH; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) U[c * F + w2 + 72 * k] = e[c][k];
            wave_lds_sync();
        }
        // pass 3: lane (n4, kk) = (lane & 3, lane >> 2) over n3 -> k3: twiddle W16^(n4 k3), writes V[kk + 16 k3 + 72 n4]
        {
            const int l3 = fresh_lane();
            const int n4 = l3 & 3, kk = l3 >> 2, r3 = n4 + 16 * kk + 4 * (kk >> 1), w3 = kk + 72 * n4;
#pragma unroll
            for (int c = 0; c < CH; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) e[c][k] = U[c * F + r3 + 4 * k];
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < CH; c++) dft4(e[c][0], e[c][1], e[c][2], e[c][3]);
            const f2 *W3 = T + G::kTw3 + n4;
#pragma unroll
            for (int k = 1; k < 4; k++) {
                const f2 w = W3[4 * (k - 1)];
#pragma unroll
                for (int c = 0; c < CH; c++) e[c][k] = cmul(e[c][k], w);
            }
#pragma unroll
            for (int c = 0; c < CH; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) U[c * F + w3 + 16 * k] = e[c][k];
            wave_lds_sync();
        }
        // pass 4: lane j = kk + 16 k3 over n4 -> k4
        j = group_of(fresh_lane());
#pragma unroll
        for (int c = 0; c < CH; c++)
#pragma unroll
            for (int k = 0; k < 4; k++) e[c][k] = U[c * F + j + 72 * k];
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < CH; c++) dft4(e[c][0], e[c][1], e[c][2], e[c][3]);
    }
}

// The N-point FFT of all channels with the pre- and post-twiddle: xin[c][r] = (X[2q], X[2q+1]) at q = j + 64 r in,
// P[c][s] = c[j + 64 s] out.
template <int R, int CH, typename Next>
__device__ __forceinline__ void fft_lanes(f2 (&xin)[CH][R], f2 (&P)[CH][R], f2 *U, const f2 *T, const LaneTw<R> &tw, Next next)
{
    int j = group_of(fresh_lane());      // points j + 64 r
    // X[2q+1] is the imaginary part of point N-1 - q = (63 - j) + 64 (R-1 - r): the other half-wave's slot R-1 - r
#pragma unroll
    for (int c = 0; c < CH; c++)
#pragma unroll
        for (int r = 0; r < R / 2; r++) {
            float a = xin[c][r].y, b = xin[c][R - 1 - r].y;
            cross32(a, b);
            xin[c][r].y = a;
            xin[c][R - 1 - r].y = b;
        }
    f2 e[CH][R];
    {
        const f2 *W = T + j;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const f2 w = kTwRegsW<R, CH> ? tw.w[r] : W[64 * r];
#pragma unroll
            for (int c = 0; c < CH; c++) e[c][r] = cmul(xin[c][r], w);
        }
    }
    next();                                          // the spectrum registers are free: fetch the next packet's
    fft_passes<R, CH>(e, U, T, tw);
    j = group_of(fresh_lane());
    {
        const f2 *W = T + j;
#pragma unroll
        for (int k = 0; k < R; k++) {
            const f2 w = kTwRegsW<R, CH> ? tw.w[k] : W[64 * k];
#pragma unroll
            for (int c = 0; c < CH; c++) P[c][k] = cmul(e[c][k], w);
        }
    }
}

// The same passes for a stereo pair held channel-packed (c2): the transform area holds a plane of real parts and a plane of
// imaginary parts, each slot the two channels' values (8 bytes: the slot indices, and so the bank behaviour, are those of
// fft_passes, tests/vorbis_walk_model.py).
template <int R>
__device__ __forceinline__ void fft_passes2(c2 (&e)[R], f2 *U, const f2 *T, const LaneTw<R> &tw)
{
    using G = Geo<R>;
    constexpr int F = G::kChanF2;
    f2 *const Ur = U, *const Ui = U + F;
    int j = group_of(fresh_lane());
    if constexpr (R == 8) {
        dft8(e);
        {
            const f2 *W1 = T + G::kTw1 + j;
#pragma unroll
            for (int k = 1; k < 8; k++) e[k] = cmulw(e[k], kTwRegs1<R, 2> ? tw.w1[k - 1] : W1[64 * (k - 1)]);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { Ur[j + 68 * k] = e[k].re; Ui[j + 68 * k] = e[k].im; }
        wave_lds_sync();
        const int l2 = fresh_lane();
        const int n0 = l2 >> 3, r2 = n0 + 68 * (l2 & 7), w2 = (l2 & 7) + 72 * n0;
#pragma unroll
        for (int k = 0; k < 8; k++) e[k] = c2{ Ur[r2 + 8 * k], Ui[r2 + 8 * k] };
        wave_lds_sync();
        dft8(e);
        {
            const f2 *W2 = T + G::kTw2 + n0;
#pragma unroll
            for (int k = 1; k < 8; k++) e[k] = cmulw(e[k], kTwRegs2<R, 2> ? tw.w2[k - 1] : W2[8 * (k - 1)]);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) { Ur[w2 + 8 * k] = e[k].re; Ui[w2 + 8 * k] = e[k].im; }
        wave_lds_sync();
        j = group_of(fresh_lane());
#pragma unroll
        for (int k = 0; k < 8; k++) e[k] = c2{ Ur[j + 72 * k], Ui[j + 72 * k] };
        wave_lds_sync();
        dft8(e);
    } else if constexpr (R == 16) {
        dft16(e);
        {
            const f2 *W1 = T + G::kTw1 + j;
#pragma unroll
            for (int k = 1; k < 16; k++) e[k] = cmulw(e[k], W1[64 * (k - 1)]);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) { Ur[j + 72 * k] = e[k].re; Ui[j + 72 * k] = e[k].im; }
        wave_lds_sync();
        const int l2 = fresh_lane();
        const int n3 = l2 >> 2, r2 = n3 + 72 * (l2 & 3), w2 = (l2 & 3) + 68 * n3;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int n2 = 0; n2 < 4; n2++) e[4 * i + n2] = c2{ Ur[r2 + 16 * n2 + 288 * i], Ui[r2 + 16 * n2 + 288 * i] };
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 4; i++) dft4(e[4 * i], e[4 * i + 1], e[4 * i + 2], e[4 * i + 3]);
        {
            const f2 *W2 = T + G::kTw2 + n3;
#pragma unroll
            for (int k = 1; k < 4; k++) {
                const f2 w = W2[16 * (k - 1)];
#pragma unroll
                for (int i = 0; i < 4; i++) e[4 * i + k] = cmulw(e[4 * i + k], w);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int k = 0; k < 4; k++) { Ur[w2 + 4 * i + 16 * k] = e[4 * i + k].re; Ui[w2 + 4 * i + 16 * k] = e[4 * i + k].im; }
        wave_lds_sync();
        j = group_of(fresh_lane());
#pragma unroll
        for (int k = 0; k < 16; k++) e[k] = c2{ Ur[j + 68 * k], Ui[j + 68 * k] };
        wave_lds_sync();
        dft16(e);
    } else {
        static_assert(R == 4, "points per lane");
        dft4(e[0], e[1], e[2], e[3]);
        {
            const f2 *W1 = T + G::kTw1 + j;
#pragma unroll
            for (int k = 1; k < 4; k++) e[k] = cmulw(e[k], W1[64 * (k - 1)]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { Ur[j + 80 * k] = e[k].re; Ui[j + 80 * k] = e[k].im; }
        wave_lds_sync();
        {
            const int l2 = fresh_lane();
            const int m = l2 & 15, k1 = l2 >> 4, r2 = m + 80 * k1, w2 = m + 16 * k1 + 4 * (k1 >> 1);
#pragma unroll
            for (int k = 0; k < 4; k++) e[k] = c2{ Ur[r2 + 16 * k], Ui[r2 + 16 * k] };
            wave_lds_sync();
            dft4(e[0], e[1], e[2], e[3]);
            const f2 *W2 = T + G::kTw2 + m;
#pragma unroll
            for (int k = 1; k < 4; k++) e[k] = cmulw(e[k], W2[16 * (k - 1)]);
#pragma unroll
            for (int k = 0; k < 4; k++) { Ur[w2 + 72 * k] = e[k].re; Ui[w2 + 72 * k] = e[k].im; }
            wave_lds_sync();
        }
        {
            const int l3 = fresh_lane();
            const int n4 = l3 & 3, kk = l3 >> 2, r3 = n4 + 16 * kk + 4 * (kk >> 1), w3 = kk + 72 * n4;
#pragma unroll
            for (int k = 0; k < 4; k++) e[k] = c2{ Ur[r3 + 4 * k], Ui[r3 + 4 * k] };
            wave_lds_sync();
            dft4(e[0], e[1], e[2], e[3]);
            const f2 *W3 = T + G::kTw3 + n4;
#pragma unroll
            for (int k = 1; k < 4; k++) e[k] = cmulw(e[k], W3[4 * (k - 1)]);
#pragma unroll
            for (int k = 0; k < 4; k++) { Ur[w3 + 16 * k] = e[k].re; Ui[w3 + 16 * k] = e[k].im; }
            wave_lds_sync();
        }
        j = group_of(fresh_lane());
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = c2{ Ur[j + 72 * k], Ui[j + 72 * k] };
        wave_lds_sync();
        dft4(e[0], e[1], e[2], e[3]);
    }
}

// fft_lanes for a stereo pair: xin as loaded (one register pair per channel and point), P channel-packed
template <int R, typename Next>
__device__ __forceinline__ void fft_lanes2(f2 (&xin)[2][R], c2 (&P)[R], f2 *U, const f2 *T, const LaneTw<R> &tw, Next next)
{
    int j = group_of(fresh_lane());
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int r = 0; r < R / 2; r++) {
            float a = xin[c][r].y, b = xin[c][R - 1 - r].y;
            cross32(a, b);
            xin[c][r].y = a;
            xin[c][R - 1 - r].y = b;
        }
    c2 e[R];
    {
        const f2 *W = T + j;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const f2 w = kTwRegsW<R, 2> ? tw.w[r] : W[64 * r];
            e[r] = cmulw(c2{ f2{ xin[0][r].x, xin[1][r].x }, f2{ xin[0][r].y, xin[1][r].y } }, w);
        }
    }
    next();                                          // the spectrum registers are free: fetch the next packet's
#ifndef AFG_WALK_EXP_NOFFT                                // (experiment: the memory floor of the walk -- loads, window, stores, no passes)
    fft_passes2<R>(e, U, T, tw);
#endif
    j = group_of(fresh_lane());
    {
        const f2 *W = T + j;
#pragma unroll
        for (int k = 0; k < R; k++) P[k] = cmulw(e[k], kTwRegsW<R, 2> ? tw.w[k] : W[64 * k]);
    }
}

// y[m] of a long block of n samples from its u (floats at uf[0..n/2)): the odd / even extension of the DCT-IV
template <int N>
__device__ __forceinline__ float y_of_u(const float *uf, int m)
{
    if (m < N / 4) return uf[N / 4 + m];
    if (m < 3 * N / 4) return -uf[3 * N / 4 - 1 - m];
    return -uf[m - 3 * N / 4];
}

// two frames of all channels: 16 bytes of a stereo stream, 8 of a mono one; ST (CH channels of a stream with more than
// two): these channels' 4- or 8-byte column of the interleaved frames, `stride` floats apart, cached -- the other columns'
// wavefronts fill the lines in L2
template <int CH, bool ST>
__device__ __forceinline__ void store_pair(float *o, int pair, const float (&f0)[CH], const float (&f1)[CH], int stride)
{
    if constexpr (ST && CH == 2) {
        *(f2 *)(o + (2 * pair) * stride) = f2{ f0[0], f0[1] };
        *(f2 *)(o + (2 * pair + 1) * stride) = f2{ f1[0], f1[1] };
    } else if constexpr (ST) {
        o[(2 * pair) * stride] = f0[0];
        o[(2 * pair + 1) * stride] = f1[0];
    } else if constexpr (CH == 2) __builtin_nontemporal_store((f4{ f0[0], f0[1], f1[0], f1[1] }), (f4 *)o + pair);
    else __builtin_nontemporal_store((f2{ f0[0], f1[0] }), (f2 *)o + pair);
}
// the same from channel-packed values: f0, f1 = (ch0, ch1) of two consecutive frames
template <bool ST>
__device__ __forceinline__ void store_pair2(float *o, int pair, f2 f0, f2 f1, int stride)
{
    if constexpr (ST) {
        *(f2 *)(o + (2 * pair) * stride) = f0;
        *(f2 *)(o + (2 * pair + 1) * stride) = f1;
    } else __builtin_nontemporal_store((f4{ f0.x, f0.y, f1.x, f1.y }), (f4 *)o + pair);
}
// Streams with more than two channels (round 6): the wavefronts of a workgroup are the channels (or channel pairs) of ONE
// segment and walk it in step.  A chunk -- 128 consecutive frames and the 128 mirrored ones of a long block, all channels --
// is put together in LDS (two chunks alternate, so one barrier per chunk is enough) and leaves as whole interleaved frames,
// 16 bytes per lane: round 5 stored each wavefront's 4- or 8-byte column of the frames by itself (8-byte pieces at a 24-byte
// stride for six channels; WRITE_SIZE 1.8 x the bytes, 12-14 ms per C3-sized batch).
template <int N>
__device__ __forceinline__ void staged_store(const float *S, float *obase, int i, int nch)
{
    __syncthreads();
    const int half = 32 * nch;                                  // 16-byte pieces of 128 frames
    float *const lo = obase + (128 * i) * nch, *const hi = obase + (N / 2 - 128 * (i + 1)) * nch;
    for (int q = (int)threadIdx.x; q < 2 * half; q += (int)blockDim.x) {
        const f4 v = ((const f4 *)S)[q];
        __builtin_nontemporal_store(v, q < half ? (f4 *)lo + q : (f4 *)hi + (q - half));
    }
}
template <int CH, bool ST>
__device__ __forceinline__ void store_frame(float *o, int frame, const float (&v)[CH], int stride)
{
    if constexpr (ST && CH == 2) *(f2 *)(o + frame * stride) = f2{ v[0], v[1] };
    else if constexpr (ST) o[frame * stride] = v[0];
    else if constexpr (CH == 2) __builtin_nontemporal_store((f2{ v[0], v[1] }), (f2 *)o + frame);
    else __builtin_nontemporal_store(v[0], o + frame);
}

// One wavefront walks the CH channels of a segment whose long blocks have 256 R samples; ST: channels seg.pad .. seg.pad +
// CH - 1 of a stream with more than two (pairs when the count is even: 8-byte columns need even frame strides).
template <int R, int CH, bool ST>
__device__ __forceinline__ void walk_body(
    f2 *U, const f2 *T, const LaneTw<R> &tw, const VorbisSeg &seg, const VorbisStream &st, const uint8_t *__restrict__ pflags,
    const uint64_t *__restrict__ spec_off, const uint64_t *__restrict__ out_off, const float *tables,
    const float *__restrict__ spec, float *__restrict__ out, float *stage, int first_chan)
{
    using G = Geo<R>;
    constexpr int kN = G::kN, kPts = G::kPts, kChanF2 = G::kChanF2, H = R / 2;
    const int lane = threadIdx.x & 63;
    const int bs0 = (int)st.bs[0], bs1 = (int)st.bs[1];
    const uint32_t tab0 = st.tab[0];
    const float *const lwin = (const float *)(T + G::kTwEnd);    // window of the long block (LDS)
    const int stride = ST ? (int)st.nch : CH, chan = ST ? first_chan : 0;

    int previous_length = 0;
    const int p_first = seg.p0 > 0 ? (int)seg.p0 - 1 : 0;
    const int p_end = (int)(seg.p0 + seg.count);

    // Carried state, per channel.  After a long block whose right window is long: cb[c][2 i], cb[c][2 i + 1] =
    // u[n/2-1 - 2k], u[n/2-2 - 2k] for k = j + 64 (R/2 + i) -- the b of frames 2k - n/4 and 2k - n/4 + 1 of the next block.
    // Otherwise previous_window itself (:2641-2643): sample lane + 64 i in cb[c][i] (64 .. 256 samples).
    // (a stereo pair: channel-packed, cbs[i] = (ch0, ch1))
    using CB = std::conditional_t<CH == 2, f2, float>;
    CB cbs[R];
#pragma unroll
    for (int i = 0; i < R; i++) cbs[i] = CB(0.0f);
    auto cbget = [&](int c, int i) -> float {
        if constexpr (CH == 2) return c ? cbs[i].y : cbs[i].x;
        else return cbs[i];
    };
    auto cbset = [&](int c, int i, float v) {
        if constexpr (CH == 2) {
            if (c) cbs[i].y = v;
            else cbs[i].x = v;
        } else cbs[i] = v;
    };

    int fbase = 0;
    unsigned fl_reg = 0;
    uint64_t so_reg = 0, oo_reg = 0;
    auto refill = [&](int from) {
        fbase = from;
        const int q = from + lane;
        const bool in = q < p_end;
        fl_reg = in ? (unsigned)pflags[st.pkt_base + (uint64_t)q] : 0u;
        so_reg = in ? spec_off[st.pkt_base + (uint64_t)q] : 0;
        oo_reg = in ? out_off[st.pkt_base + (uint64_t)q] : 0;
        uint32_t s0 = (uint32_t)so_reg, s1 = (uint32_t)(so_reg >> 32), o0 = (uint32_t)oo_reg, o1 = (uint32_t)(oo_reg >> 32);
        asm volatile("" : "+v"(fl_reg), "+v"(s0), "+v"(s1), "+v"(o0), "+v"(o1) : : "memory");
        so_reg = ((uint64_t)s1 << 32) | s0;
        oo_reg = ((uint64_t)o1 << 32) | o0;
    };
    auto flags_of = [&](int p) -> unsigned { return (unsigned)__builtin_amdgcn_readlane((int)fl_reg, p - fbase); };
    auto lane64 = [&](uint64_t v, int p) -> uint64_t {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, p - fbase);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), p - fbase);
        return ((uint64_t)hi << 32) | lo;
    };
    refill(p_first);

    // One packet's spectra are in flight: the pre-twiddle of transform k empties xin, the loads of transform k+1 follow
    // at once and are waited for (settle) just before the PCM stores of transform k enter the queue -- loads and stores
    // share one in-order counter (HISTORY.md 8).
    f2 xin[CH][R];
    auto issue = [&](int p) {
        const unsigned flp = p < p_end ? flags_of(p) : 0u;
        if (flp & AFG_VORBIS_LONG) {
            const f2 *src = (const f2 *)(spec + lane64(so_reg, p)) + chan * kPts + group_of(fresh_lane());
            // AFG_VORBIS_NZ_EIGHTHS: load r of a channel covers bins 128 r .. 128 r + 127 of n/2 = 128 R, so a load whose
            // bins all lie in the declared-empty eighths is a whole instruction that is not issued (a scalar test)
            const int nz = (int)(flp >> 4) ? (int)(flp >> 4) - 1 : 8;
#if defined(AFG_WALK_EXP_NOFFT) && AFG_WALK_EXP_NOFFT == 2
            // (experiment, with the passes compiled out: the same bytes fetched 16 per lane -- is the 8-byte load what sets the floor?)
            {
                const f4 *src4 = (const f4 *)((const f2 *)(spec + lane64(so_reg, p)) + chan * kPts) + fresh_lane();
#pragma unroll
                for (int c = 0; c < CH; c++)
#pragma unroll
                    for (int r = 0; r < R; r += 2) {
                        f4 v = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
                        if (8 * r < nz * R) v = __builtin_nontemporal_load(src4 + c * (kPts / 2) + 64 * (r / 2));
                        xin[c][r] = f2{ v.x, v.y };
                        xin[c][r + 1] = f2{ v.z, v.w };
                    }
            }
#else
#pragma unroll
            for (int c = 0; c < CH; c++)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    xin[c][r] = f2{ 0.0f, 0.0f };
                    if (8 * r < nz * R) xin[c][r] = __builtin_nontemporal_load(src + c * kPts + 64 * r);
                }
#endif
        } else {
            // nothing reads xin before the next issue(): say so, or the old values are copied around to survive the branch
#pragma unroll
            for (int c = 0; c < CH; c++)
#pragma unroll
                for (int r = 0; r < R; r++) asm volatile("" : "=v"(xin[c][r]));
        }
    };
    issue(p_first);
    settle(xin);

    for (int p = p_first; p < p_end; p++) {
        if (p + 1 - fbase >= 64) refill(p);
        const unsigned fl = flags_of(p);
        int n, left, right, right_end;
        window_bounds(bs0, bs1, fl, n, left, right, right_end);
        const bool lng = (fl & AFG_VORBIS_LONG) != 0;
        const bool emit = (p >= (int)seg.p0) && previous_length > 0;
        const int pn = previous_length;
        const int nout = right - left, plen = right_end - right;
        float *const o = out + lane64(oo_reg, p) + chan;   // interleaved frames; 16-byte aligned (checked at launch) unless ST
        auto next = [&]() { issue(p + 1); };

        if (lng) {
            const bool wprev = (fl & AFG_VORBIS_PREV) != 0, wnext = (fl & AFG_VORBIS_NEXT) != 0;
            if constexpr (R == 4) {
                // a load covers two eighths of this size's spectrum (lanes 0..31 the first): an odd declaration leaves the
                // upper lanes of its last load holding bins that must not be looked at
                const int nz = (int)(fl >> 4) ? (int)(fl >> 4) - 1 : 8;
                if (nz & 1) {
                    const bool low = fresh_lane() < 32;
#pragma unroll
                    for (int c = 0; c < CH; c++)
#pragma unroll
                        for (int r = 0; r < R; r++)
                            if (r == (nz >> 1)) xin[c][r] = low ? xin[c][r] : f2{ 0.0f, 0.0f };
                }
            }
            // P: c[j + 64 s] of every channel (a stereo pair: channel-packed); px / py read one channel's parts
            using PT = std::conditional_t<CH == 2, c2, f2>;
            PT P[R];
            auto px = [&](int c, int k) -> float {
                if constexpr (CH == 2) return c ? P[k].re.y : P[k].re.x;
                else return P[k].x;
            };
            auto py = [&](int c, int k) -> float {
                if constexpr (CH == 2) return c ? P[k].im.y : P[k].im.x;
                else return P[k].y;
            };
            if constexpr (CH == 2) fft_lanes2<R>(xin, P, U, T, tw, next);
            else {
                f2 P1[1][R];
                fft_lanes<R, 1>(xin, P1, U, T, tw, next);
#pragma unroll
                for (int k = 0; k < R; k++) P[k] = P1[0][k];
            }
            if (!(wprev && wnext)) {
                // a short neighbour: u of every channel to LDS in natural order for the y(m) accessor below
                float *const uf = (float *)U;
#pragma unroll
                for (int c = 0; c < CH; c++)
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const int kk = group_of(lane) + 64 * k;
                        uf[c * (2 * kChanF2) + 2 * kk] = px(c, k);
                        uf[c * (2 * kChanF2) + kN / 2 - 1 - 2 * kk] = -py(c, k);
                    }
                wave_lds_sync();
            }
            // c[N-1 - k] for k = j + 64 s, s = R/2 .. R-1: slot R-1 - s of lane ^ 32
#pragma unroll
            for (int k = 0; k < H; k += 2) {
                if constexpr (CH == 2) {
                    float ax0 = P[k].re.x, bx0 = P[k + 1].re.x, ax1 = P[k].re.y, bx1 = P[k + 1].re.y;
                    float ay0 = P[k].im.x, by0 = P[k + 1].im.x, ay1 = P[k].im.y, by1 = P[k + 1].im.y;
                    cross32(ax0, bx0);
                    cross32(ax1, bx1);
                    cross32(ay0, by0);
                    cross32(ay1, by1);
                    P[k] = c2{ f2{ ax0, ax1 }, f2{ ay0, ay1 } };
                    P[k + 1] = c2{ f2{ bx0, bx1 }, f2{ by0, by1 } };
                } else {
                    float ax = P[k].x, bx = P[k + 1].x, ay = P[k].y, by = P[k + 1].y;
                    cross32(ax, bx);
                    cross32(ay, by);
                    P[k] = f2{ ax, ay };
                    P[k + 1] = f2{ bx, by };
                }
            }
            // after the crossing slot k holds lane ^ 32's slot k ^ 1: the mirror of s = R/2 + i, slot R/2-1 - i, is slot mir(i)
            auto mir = [](int i) constexpr { return (H - 1 - i) ^ 1; };
            settle(xin);
            if (emit && wprev) {
                // frames j0, j0+1 (j0 = 2 (j + 64 i)) and n/2-2-j0, n/2-1-j0 of the block, all channels: two stores
                const int j = group_of(fresh_lane());
                const f2 *const wlo = (const f2 *)lwin + j;
                const f2 *const whi = (const f2 *)lwin + (kPts - 1) - j;
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const f2 w0 = wlo[64 * i];                     // w[j0], w[j0+1]
                    const f2 w1 = whi[-64 * i];                    // w[n/2-2-j0], w[n/2-1-j0]
                    if constexpr (CH == 2) {
                        // both channels at once: a0 = P.re, a1 = -P.im of the mirror slot, b the carried pair
                        const f2 a0 = P[H + i].re, a1n = P[mir(i)].im, b0 = cbs[2 * i], b1 = cbs[2 * i + 1];
                        const f2 lo0 = fma2(-b0, w1.yy, a0 * w0.xx);
                        const f2 lo1 = fma2(-b1, w1.xx, -(a1n * w0.yy));
                        const f2 hi1 = fma2(-b0, w0.xx, -(a0 * w1.yy));
                        const f2 hi0 = fma2(-b1, w0.yy, a1n * w1.xx);
                        if constexpr (ST) {
                            // the workgroup's wavefronts -- the channel pairs of this segment -- meet in the staging area:
                            // frames [128 i, 128 i + 128) and their mirror, whole, then stored 16 bytes per lane
                            float *const S = stage + (i & 1) * (256 * stride) + chan;
                            *(f2 *)(S + (2 * j) * stride) = lo0;
                            *(f2 *)(S + (2 * j + 1) * stride) = lo1;
                            *(f2 *)(S + (128 + 126 - 2 * j) * stride) = hi0;
                            *(f2 *)(S + (128 + 127 - 2 * j) * stride) = hi1;
                            staged_store<kN>(stage + (i & 1) * (256 * stride), o - chan, i, stride);
                        } else {
                            store_pair2<ST>(o, j + 64 * i, lo0, lo1, stride);
                            store_pair2<ST>(o, (kPts - 1) - j - 64 * i, hi0, hi1, stride);
                        }
                    } else {
                        float lo0[1], lo1[1], hi0[1], hi1[1];
                        const float a0 = P[H + i].x, a1 = -P[mir(i)].y, b0 = cbs[2 * i], b1 = cbs[2 * i + 1];
                        lo0[0] = a0 * w0.x - b0 * w1.y;
                        lo1[0] = a1 * w0.y - b1 * w1.x;
                        hi1[0] = -a0 * w1.y - b0 * w0.x;
                        hi0[0] = -a1 * w1.x - b1 * w0.y;
                        if constexpr (ST) {
                            float *const S = stage + (i & 1) * (256 * stride) + chan;
                            S[(2 * j) * stride] = lo0[0];
                            S[(2 * j + 1) * stride] = lo1[0];
                            S[(128 + 126 - 2 * j) * stride] = hi0[0];
                            S[(128 + 127 - 2 * j) * stride] = hi1[0];
                            staged_store<kN>(stage + (i & 1) * (256 * stride), o - chan, i, stride);
                        } else {
                            store_pair<1, ST>(o, j + 64 * i, lo0, lo1, stride);
                            store_pair<1, ST>(o, (kPts - 1) - j - 64 * i, hi0, hi1, stride);
                        }
                    }
                }
            }
            if (!(wprev && wnext)) {
                const float *const uf = (const float *)U;
                if (emit) {
                    // :2606-2657 on y(m): frames the u-form path above did not write
                    const float *wt = tables + tab0 + bs0 + bs0 / 4;      // window of blocksize_0 (:2245-2251)
                    // (the frames that meet previous_window -- the first blocksize_0 / 2 <= 256 -- in four spelled-out
                    // steps: the carried value of step `it` is a register known at compile time)
                    int jfrom = wprev ? kN / 2 : 256;
                    if (!wprev) {
#pragma unroll
                        for (int it = 0; it < 4; it++) {
                            const int jj = lane + 64 * it;
                            if (jj < nout) {
                                float v[CH];
#pragma unroll
                                for (int c = 0; c < CH; c++) v[c] = y_of_u<kN>(uf + c * (2 * kChanF2), left + jj);
                                if (jj < pn) {                             // pn = blocksize_0 / 2 here
                                    const float wa = wt[jj], wb = wt[pn - 1 - jj];
#pragma unroll
                                    for (int c = 0; c < CH; c++) v[c] = v[c] * wa + cbget(c, it) * wb;
                                }
                                store_frame<CH, ST>(o, jj, v, stride);
                            }
                        }
                    }
                    for (int jj = jfrom + lane; jj < nout; jj += 64) {
                        float v[CH];
#pragma unroll
                        for (int c = 0; c < CH; c++) v[c] = y_of_u<kN>(uf + c * (2 * kChanF2), left + jj);
                        store_frame<CH, ST>(o, jj, v, stride);
                    }
                }
                if (!wnext) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int m = right + lane + 64 * i;
                        const bool in = lane + 64 * i < plen;
#pragma unroll
                        for (int c = 0; c < CH; c++) cbset(c, i, in ? y_of_u<kN>(uf + c * (2 * kChanF2), m < kN ? m : 0) : 0.0f);
                    }
                }
                wave_lds_sync();
            }
            if (wnext) {
#pragma unroll
                for (int i = 0; i < H; i++) {
                    if constexpr (CH == 2) {
                        cbs[2 * i] = -P[H + i].im;                   // u[n/2-1 - 2k], both channels
                        cbs[2 * i + 1] = P[mir(i)].re;               // u[2 (N-1 - k)] = u[n/2-2 - 2k]
                    } else {
                        cbs[2 * i] = -P[H + i].y;
                        cbs[2 * i + 1] = P[mir(i)].x;
                    }
                }
            }
        } else {
            // short block: the reference's own transform (vorbis_core.h) over LDS, previous_window in cb
            const float *Tn = tables + tab0;
            const float *A = Tn, *B = Tn + n / 2, *Ct = Tn + n;
            const int n2 = n >> 1;
            const float *src = spec + lane64(so_reg, p) + chan * n2;
            float *const sm = (float *)U;
            for (int k = lane; k < n2; k += 64) {
#pragma unroll
                for (int c = 0; c < CH; c++) sm[c * (2 * kChanF2) + k] = src[c * n2 + k];
            }
            wave_lds_sync();
#pragma nounroll
            for (int c = 0; c < CH; c++) {
                float *const s = sm + c * (2 * kChanF2);
                inverse_mdct_lds<64>(s, s + n, n, 31 - __clz(n), A, B, Ct);
            }
            next();
            settle(xin);
            if (emit) {
                const float *wt = Tn + n + n / 4;
                // a short block emits blocksize_0 / 2 <= 256 frames: four spelled-out steps (see above)
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int jj = lane + 64 * it;
                    if (jj < nout) {
                        float v[CH];
#pragma unroll
                        for (int c = 0; c < CH; c++) v[c] = sm[c * (2 * kChanF2) + left + jj];
                        if (jj < pn) {
                            const float wa = wt[jj], wb = wt[pn - 1 - jj];
#pragma unroll
                            for (int c = 0; c < CH; c++) v[c] = v[c] * wa + cbget(c, it) * wb;
                        }
                        store_frame<CH, ST>(o, jj, v, stride);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool in = lane + 64 * i < plen;
#pragma unroll
                for (int c = 0; c < CH; c++) cbset(c, i, in ? sm[c * (2 * kChanF2) + right + lane + 64 * i] : 0.0f);
            }
            wave_lds_sync();
        }
        previous_length = plen;
    }
}


// Wavefronts per workgroup of each shape and the wavefronts per SIMD it is compiled for: what the LDS of a CU holds
// (tables once per workgroup + one transform area per wavefront) at the register budget that pays.  The small shapes
// have little work per packet beside the fixed cost of a packet (flags, bounds, three or four LDS round trips): four
// wavefronts per SIMD hide it (mono 1024: 11.5 -> 8.4 ms per C3-sized batch; a handful of spilled registers).
#ifndef AFG_WALK_W82
#define AFG_WALK_W82 8      // stereo 2048: wavefronts per workgroup (8: two per SIMD; 12: three, at <= 168 registers)
#endif
template <int R, int CH> struct Shape { static constexpr int kWaves = 8, kPerSimd = 2; };
template <> struct Shape<8, 2> { static constexpr int kWaves = AFG_WALK_W82, kPerSimd = AFG_WALK_W82 / 4; };
template <> struct Shape<4, 1> { static constexpr int kWaves = 16, kPerSimd = 4; };
template <> struct Shape<4, 2> { static constexpr int kWaves = 16, kPerSimd = 4; };
template <> struct Shape<8, 1> { static constexpr int kWaves = 16, kPerSimd = 4; };
template <> struct Shape<16, 2> { static constexpr int kWaves = 7, kPerSimd = 2; };      // 7 x 18432 + 24448 bytes of 160 KB

template <int R, int CH, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (Shape<R, CH>::kPerSimd)) void vorbis_walk_kernel(
    const VorbisSeg *__restrict__ segs, uint32_t n_segs, const VorbisStream *__restrict__ streams,
    const uint8_t *__restrict__ pflags, const uint64_t *__restrict__ spec_off, const uint64_t *__restrict__ out_off,
    const float *tables, const float *__restrict__ walk_tables, const float *__restrict__ spec, float *__restrict__ out,
    uint32_t *__restrict__ next_seg)
{
    using G = Geo<R>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        constexpr int kQuads = G::kTabFloats / 4, kPer = (kQuads + 64 * WAVES - 1) / (64 * WAVES);
        f4 t[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const int i = (int)threadIdx.x + k * 64 * WAVES;
            t[k] = ((const f4 *)walk_tables)[i < kQuads ? i : kQuads - 1];
        }
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const int i = (int)threadIdx.x + k * 64 * WAVES;
            if (i < kQuads) ((f4 *)lds)[i] = t[k];
        }
    }
    __syncthreads();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    f2 *const U = (f2 *)(lds + G::kTabFloats) + wave * (CH * G::kChanF2);
    const f2 *const T = (const f2 *)lds;
    LaneTw<R> tw;
    if constexpr (kTwAny<R, CH>) load_lane_tw<R>(tw, T);
    for (;;) {
        uint32_t sidx = 0;
        if ((threadIdx.x & 63) == 0) sidx = atomicAdd(next_seg, 1u);
        sidx = (uint32_t)__builtin_amdgcn_readfirstlane((int)sidx);
        if (sidx >= n_segs) return;
        const VorbisSeg seg = segs[sidx];
        const VorbisStream st = streams[seg.stream];
        walk_body<R, CH, false>(U, T, tw, seg, st, pflags, spec_off, out_off, tables, spec, out, nullptr, 0);
    }
}

// wavefronts per SIMD the multi-channel kernels are compiled for (the register budget): as the mono / stereo kernel of the shape unless
// overridden -- workgroups of 3-7 wavefronts pack a CU differently than the 16-wavefront groups of the small mono / stereo shapes
#ifndef AFG_MC_PERSIMD_SMALL
#define AFG_MC_PERSIMD_SMALL 4
#endif
template <int R, int CH> struct McPerSimd { static constexpr int k = Shape<R, CH>::kPerSimd == 4 ? AFG_MC_PERSIMD_SMALL : Shape<R, CH>::kPerSimd; };

// Streams with more than two channels: a workgroup is the nch / CH wavefronts of ONE segment (wavefront w: channels w CH ..),
// every stream of the launch has the same channel count.  LDS: tables | one transform area per wavefront | two staging chunks
// of 256 frames (staged_store).  At most 8 wavefronts (512 threads): the register budget of the stereo kernels.
template <int R, int CH>
__global__ __launch_bounds__(512, (McPerSimd<R, CH>::k)) void vorbis_walk_mc_kernel(
    const VorbisSeg *__restrict__ segs, uint32_t n_segs, const VorbisStream *__restrict__ streams,
    const uint8_t *__restrict__ pflags, const uint64_t *__restrict__ spec_off, const uint64_t *__restrict__ out_off,
    const float *tables, const float *__restrict__ walk_tables, const float *__restrict__ spec, float *__restrict__ out,
    uint32_t *__restrict__ next_seg)
{
    using G = Geo<R>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = (int)threadIdx.x; i < G::kTabFloats / 4; i += (int)blockDim.x) ((f4 *)lds)[i] = ((const f4 *)walk_tables)[i];
    __syncthreads();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t waves = blockDim.x >> 6;
    f2 *const U = (f2 *)(lds + G::kTabFloats) + wave * (CH * G::kChanF2);
    float *const stage = lds + G::kTabFloats + waves * (CH * 2 * G::kChanF2);
    const f2 *const T = (const f2 *)lds;
    LaneTw<R> tw;
    if constexpr (kTwAny<R, CH>) load_lane_tw<R>(tw, T);
    __shared__ uint32_t drawn;
    for (;;) {
        if (threadIdx.x == 0) drawn = atomicAdd(next_seg, 1u);
        __syncthreads();
        const uint32_t sidx = drawn;
        __syncthreads();
        if (sidx >= n_segs) return;
        const VorbisSeg seg = segs[sidx];
        const VorbisStream st = streams[seg.stream];
        walk_body<R, CH, true>(U, T, tw, seg, st, pflags, spec_off, out_off, tables, spec, out, stage, (int)wave * CH);
    }
}

template <int R, int CH>
constexpr size_t shape_lds() { return sizeof(float) * (Geo<R>::kTabFloats + (size_t)Shape<R, CH>::kWaves * CH * 2 * Geo<R>::kChanF2); }
// (more than two channels: nch / CH transform areas and two chunks of 256 frames)
template <int R, int CH>
constexpr size_t mc_lds(int nch) { return sizeof(float) * (Geo<R>::kTabFloats + (size_t)(nch / CH) * CH * 2 * Geo<R>::kChanF2 + 2 * 256 * (size_t)nch); }

// what one device holds of a kernel, found once per device (whichever host thread gets here first: afg.h allows concurrent
// launches of one plan) and checked on every call
struct DevShape {
    std::once_flag once;
    hipError_t rc = hipSuccess;
    int per_cu[17] = {}, cus = 256;
};

template <int R, int CH, bool ST>
int launch_shape(const VorbisSeg *segs, uint32_t n_segs, int nch, const VorbisStream *streams, const uint8_t *pflags,
                 const uint64_t *spec_off, const uint64_t *out_off, const float *tables, const float *walk_tables,
                 const float *spec, float *out, uint32_t *counter, hipStream_t stream)
{
    static DevShape state[AFG_MAX_DEVICES];
    int dev = 0;
    if (int rc = afg::device_slot(&dev, "afg_vorbis_transform_hip")) return rc;
    DevShape &ds = state[dev];
    if constexpr (ST) {
        // (walk_shape admits what fits: 4096-sample blocks of 14 or 16 channels would need more than the 160 KB of LDS)
        constexpr int kMcMax = CH == 2 ? 16 : R == 16 ? 8 : 7;
        static_assert(mc_lds<R, CH>(kMcMax) <= 160 * 1024, "LDS budget");
        if (nch < 3 || nch > kMcMax || nch % CH) {
            afg::set_error("afg_vorbis_transform_hip: %d channels on the %d-channel walk", nch, CH);
            return AFG_ERR_INVALID;
        }
        const void *fn = (const void *)vorbis_walk_mc_kernel<R, CH>;
        std::call_once(ds.once, [&] {
            ds.rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mc_lds<R, CH>(kMcMax));
            int n_cu = 256, cur = 0;
            if (hipGetDevice(&cur) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, cur);
            ds.cus = n_cu;
            for (int c = CH == 2 ? 4 : 3; ds.rc == hipSuccess && c <= kMcMax; c += CH) {
                int nb = 1;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * (c / CH), mc_lds<R, CH>(c)) != hipSuccess) nb = 1;
                ds.per_cu[c] = nb < 1 ? 1 : nb;
            }
        });
        AFG_HIP_CHECK(ds.rc);
        const uint32_t room = (uint32_t)(ds.per_cu[nch] * ds.cus);
        const size_t lds_bytes = mc_lds<R, CH>(nch);
        hipLaunchKernelGGL((vorbis_walk_mc_kernel<R, CH>), dim3(n_segs < room ? n_segs : room), dim3(64 * (nch / CH)), lds_bytes, stream,
                           segs, n_segs, streams, pflags, spec_off, out_off, tables, walk_tables, spec, out, counter);
    } else {
        constexpr int kWaves = Shape<R, CH>::kWaves;
        constexpr size_t kLds = shape_lds<R, CH>();
        const void *fn = (const void *)vorbis_walk_kernel<R, CH, kWaves>;
        std::call_once(ds.once, [&] {
            ds.rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
            int nb = 1, n_cu = 256, cur = 0;
            if (ds.rc == hipSuccess && hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * kWaves, kLds) != hipSuccess) nb = 1;
            if (hipGetDevice(&cur) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, cur);
            ds.per_cu[0] = nb < 1 ? 1 : nb;
            ds.cus = n_cu;
        });
        AFG_HIP_CHECK(ds.rc);
        // persistent wavefronts: as many workgroups as the device holds at once, each drawing segments from the counter
        const uint32_t need = (n_segs + kWaves - 1) / kWaves, room = (uint32_t)(ds.per_cu[0] * ds.cus);
        hipLaunchKernelGGL((vorbis_walk_kernel<R, CH, kWaves>), dim3(need < room ? need : room), dim3(64 * kWaves), kLds, stream, segs,
                           n_segs, streams, pflags, spec_off, out_off, tables, walk_tables, spec, out, counter);
    }
    return AFG_OK;
}

template <int R>
void build_tables_r(float *dst, const float *window)
{
    using G = Geo<R>;
    const double two_pi = 6.283185307179586476925286766559;
    auto put = [&](int slot, double angle) {
        dst[2 * slot] = (float)std::cos(angle);
        dst[2 * slot + 1] = (float)-std::sin(angle);
    };
    for (int k = 0; k < G::kPts; k++) put(k, two_pi * (k + 0.125) / (double)G::kN);
    for (int k = 1; k < R; k++)
        for (int j = 0; j < 64; j++) put(G::kTw1 + (k - 1) * 64 + j, two_pi * (double)(j * k) / (double)G::kPts);
    if (R == 8) {
        for (int k1 = 1; k1 < 8; k1++)
            for (int n0 = 0; n0 < 8; n0++) put(G::kTw2 + (k1 - 1) * 8 + n0, two_pi * (double)(n0 * k1) / 64.0);
    } else {
        for (int k2 = 1; k2 < 4; k2++)
            for (int m = 0; m < 16; m++) put(G::kTw2 + (k2 - 1) * 16 + m, two_pi * (double)(m * k2) / 64.0);
    }
    if (R == 4)
        for (int k3 = 1; k3 < 4; k3++)
            for (int n4 = 0; n4 < 4; n4++) put(G::kTw3 + (k3 - 1) * 4 + n4, two_pi * (double)(n4 * k3) / 16.0);
    std::memcpy(dst + 2 * G::kTwEnd, window, sizeof(float) * (G::kN / 2));
}

}  // namespace

int walk_shape(int channels, int blocksize0, int blocksize1)
{
    // (equal block sizes: every packet is a long block between long blocks to the walk -- the plan rewrites its flags)
    if (channels < 1 || (blocksize0 > 512 && blocksize0 != blocksize1) || blocksize0 > blocksize1) return -1;
    const int size = blocksize1 == 1024 ? 0 : blocksize1 == 2048 ? 1 : blocksize1 == 4096 ? 2 : -1;
    if (size < 0) return -1;
    if (channels <= 2) return 2 * size + (channels - 1);
    // More than two channels: one workgroup of at most 8 wavefronts per segment.  4096-sample blocks: one wavefront per channel
    // whatever the count (a pair's transform areas, 18 KB, would leave room for three wavefronts per CU: 15.9 ms per C3-sized
    // batch of 6 channels, against 13.9 for round 5's column stores), so at most 8 channels; the other sizes 3, 5, 7 or an
    // even number up to 16.  Everything else stays on the bit-exact kernels.
    if (size == 2) return channels <= 8 ? 8 : -1;
    if (channels & 1) return channels <= 7 ? 6 + size : -1;
    return channels <= 16 ? 9 + size : -1;
}

int walk_shape_channels(int shape) { return shape < 6 ? (shape & 1) + 1 : shape < 9 ? 1 : 2; }

int walk_shape_blocksize(int shape) { return 1024 << (shape < 6 ? shape >> 1 : (shape - 6) % 3); }

size_t walk_table_floats(int shape)
{
    const int n = walk_shape_blocksize(shape);
    return n == 1024 ? Geo<4>::kTabFloats : n == 2048 ? Geo<8>::kTabFloats : Geo<16>::kTabFloats;
}

void walk_build_tables(int shape, float *dst, const float *window)
{
    const int n = walk_shape_blocksize(shape);
    if (n == 1024) build_tables_r<4>(dst, window);
    else if (n == 2048) build_tables_r<8>(dst, window);
    else build_tables_r<16>(dst, window);
}

int walk_launch(int shape, const VorbisSeg *segs, uint32_t n_segs, int nch, const VorbisStream *streams, const uint8_t *pflags,
                const uint64_t *spec_off, const uint64_t *out_off, const float *tables, const float *walk_tables,
                const float *spec, float *out, uint32_t *counter, hipStream_t stream)
{
#define AFG_WALK_SHAPE(S, R, CH, ST) \
    case S: return launch_shape<R, CH, ST>(segs, n_segs, nch, streams, pflags, spec_off, out_off, tables, walk_tables, spec, out, counter, stream)
    switch (shape) {
        AFG_WALK_SHAPE(0, 4, 1, false);
        AFG_WALK_SHAPE(1, 4, 2, false);
        AFG_WALK_SHAPE(2, 8, 1, false);
        AFG_WALK_SHAPE(3, 8, 2, false);
        AFG_WALK_SHAPE(4, 16, 1, false);
        AFG_WALK_SHAPE(5, 16, 2, false);
        AFG_WALK_SHAPE(6, 4, 1, true);
        AFG_WALK_SHAPE(7, 8, 1, true);
        AFG_WALK_SHAPE(8, 16, 1, true);
        AFG_WALK_SHAPE(9, 4, 2, true);
        AFG_WALK_SHAPE(10, 8, 2, true);
    }
#undef AFG_WALK_SHAPE
    afg::set_error("afg_vorbis_transform_hip: walk shape %d", shape);
    return AFG_ERR_INVALID;
}

}  // namespace afg_vorbis
