// celt_walk.hip -- Opus/CELT transform stage on gfx950, tolerance mode (AFG_NUMERIC_TOLERANCE).
//
// The same seam as celt_transform.hip (the per-channel tail of ff_celt_decode_frame, dopus.d:3680-3702: imdct15_half
// :1611-1637 + vector_fmul_window :230-243, celt_postfilter :3281-3378, de-emphasis / scaling :3695-3701), for callers
// that accept north_star's tolerance (1e-5 RMS on the API scale) instead of bit-identity with the D expression trees.
// Two things become possible, and both remove a serial chain:
//
//   * De-emphasis (tmp = x + m; m = tmp * 0.8500061; out = tmp / 32768, :3695-3701) is a one-pole IIR whose float
//     rounding order the exact path must follow sample by sample -- a second pass over the PCM plane with one lane per
//     channel sequence.  Re-associated it is a weighted prefix sum: a lane takes 15 consecutive samples of a frame
//     (the recurrence from a zero memory, in registers), the memories the lanes' runs start from come out of one
//     six-step DPP scan per frame and channel, *inside* the frame walk: the PCM plane is written once and never read.
//   * The comb post-filter (:3281-3378) feeds on its own output, but only while a filter is live: a frame whose old,
//     current and new gains are all zero is left untouched (:3294-3296, :3333).  Where the records show enough such
//     frames in a row (the filter reaches back at most 1024 samples) everything after them is independent of
//     everything before them, *exactly*.  A wavefront therefore walks a *segment* of a stream: it starts a few frames
//     early (the warm-up: transform only, nothing stored) at such a cut and stops at the first cut behind its nominal
//     range.  Cuts are found on the device from the records themselves, so the entry point keeps the device-only
//     signature of afg_celt_transform_hip; a stream with a live filter throughout stays one serial walk.
//     The de-emphasis memory a segment starts from is rebuilt by the warm-up (0.85^1026 < 1e-72).
//
// Launch shape: persistent workgroups (8 wavefronts sharing one LDS copy of the tables, one per CU) draw items from
// an atomic counter; an item is a range of `seg_recs` records of the flat record array (or a whole channel pair when
// seg_recs is 0), mapped to the channel pairs it overlaps by a search in rec_base.
//
// This file is compiled with -ffp-contract=fast (Makefile): multiply-adds of the transform fuse; the 15-point base
// transform runs in prime-factor form (fft15_pfa) instead of the reference's radix-3 x 5 form.
#define AFG_CELT_NT 0       // plain coefficient loads in this walk (measured 7.38 -> 7.30 ms against nontemporal ones)
#include "celt_core.h"

#include <algorithm>
#include <mutex>
#include <type_traits>

#ifndef AFG_WALK_FFT_SIGN
#define AFG_WALK_FFT_SIGN 1
#endif
#ifndef AFG_WALK_PFA
#define AFG_WALK_PFA 1      // the 15-point base transform in prime-factor form (0: the reference's radix-3 x 5 form)
#endif
#ifndef AFG_WALK_ST16
#define AFG_WALK_ST16 1     // 16-byte PCM stores (two samples of both channels per lane); 0: 8-byte stores
#endif
#ifndef AFG_WALK_PFU
#define AFG_WALK_PFU 8      // samples a lane takes per step of the steady-state comb filter (4 or 8)
#endif

namespace {

constexpr int kWWaves = 8;                                   // wavefronts per workgroup, sharing the tables
constexpr int kWLdsFloats = kTabFloatsMax + 240 + kWWaves * 2 * 2048;
constexpr float kDeemph = 0.85000610f;                       // CELT_DEEMPH_COEFF, dopus.d:1964

struct PfW {
    int period, period_old;
    float g[3], g_old[3];
};

// ---- 15-point transform, prime-factor form ---------------------------------------------------------------------------------
// imdct15's base transform (dopus.d:1552-1581) is a radix-3 step over three 5-point transforms, with a twiddle on two
// thirds of the points: ~440 operations per lane.  15 = 3 x 5 with coprime factors, so the Good-Thomas index maps
// n = (5 n1 + 3 n2) mod 15, k = (10 k1 + 6 k2) mod 15 turn it into three 5-point and five 3-point transforms with NO
// twiddles in between (W15^(nk) = W3^(n1 k1) W5^(n2 k2)), each in its real-constant form: 178 operations.  Same sums,
// another association: tolerance mode only.
constexpr int kFftSign = AFG_WALK_FFT_SIGN;                // exponent sign of the reference's tables: e^(+2 pi i j / 15)

__device__ __forceinline__ cpx mul_i(cpx a) { return kFftSign > 0 ? cpx{ -a.im, a.re } : cpx{ a.im, -a.re }; }   // a * (s i)

__device__ __forceinline__ void fft5_pfa(cpx &X0, cpx &X1, cpx &X2, cpx &X3, cpx &X4, cpx x0, cpx x1, cpx x2, cpx x3, cpx x4)
{
    constexpr float c1 = 0.30901699437494745f, c2 = -0.80901699437494734f;     // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float s1 = 0.95105651629515353f, s2 = 0.58778525229247325f;      // sin(2 pi / 5), sin(4 pi / 5)
    const cpx t1 = { x1.re + x4.re, x1.im + x4.im }, t2 = { x2.re + x3.re, x2.im + x3.im };
    const cpx t3 = { x1.re - x4.re, x1.im - x4.im }, t4 = { x2.re - x3.re, x2.im - x3.im };
    X0 = cpx{ x0.re + t1.re + t2.re, x0.im + t1.im + t2.im };
    const cpx m1 = { x0.re + c1 * t1.re + c2 * t2.re, x0.im + c1 * t1.im + c2 * t2.im };
    const cpx m2 = { x0.re + c2 * t1.re + c1 * t2.re, x0.im + c2 * t1.im + c1 * t2.im };
    const cpx n1 = mul_i(cpx{ s1 * t3.re + s2 * t4.re, s1 * t3.im + s2 * t4.im });
    const cpx n2 = mul_i(cpx{ s2 * t3.re - s1 * t4.re, s2 * t3.im - s1 * t4.im });
    X1 = cpx{ m1.re + n1.re, m1.im + n1.im };
    X4 = cpx{ m1.re - n1.re, m1.im - n1.im };
    X2 = cpx{ m2.re + n2.re, m2.im + n2.im };
    X3 = cpx{ m2.re - n2.re, m2.im - n2.im };
}

__device__ __forceinline__ void fft3_pfa(cpx &X0, cpx &X1, cpx &X2, cpx x0, cpx x1, cpx x2)
{
    constexpr float sn = 0.86602540378443865f;                                  // sin(2 pi / 3)
    const cpx t = { x1.re + x2.re, x1.im + x2.im }, d = { x1.re - x2.re, x1.im - x2.im };
    X0 = cpx{ x0.re + t.re, x0.im + t.im };
    const cpx m = { x0.re - 0.5f * t.re, x0.im - 0.5f * t.im };
    const cpx n = mul_i(cpx{ sn * d.re, sn * d.im });
    X1 = cpx{ m.re + n.re, m.im + n.im };
    X2 = cpx{ m.re - n.re, m.im - n.im };
}

__device__ __forceinline__ void fft15_pfa(cpx (&y)[15], const cpx (&x)[15])
{
    cpx A[3][5];
#pragma unroll
    for (int n1 = 0; n1 < 3; n1++)
        fft5_pfa(A[n1][0], A[n1][1], A[n1][2], A[n1][3], A[n1][4], x[(5 * n1) % 15], x[(5 * n1 + 3) % 15], x[(5 * n1 + 6) % 15],
                 x[(5 * n1 + 9) % 15], x[(5 * n1 + 12) % 15]);
#pragma unroll
    for (int k2 = 0; k2 < 5; k2++)
        fft3_pfa(y[(6 * k2) % 15], y[(10 + 6 * k2) % 15], y[(20 + 6 * k2) % 15], A[0][k2], A[1][k2], A[2][k2]);
}

// frame_fft of celt_core.h with the base transform above
__device__ __forceinline__ void frame_fft_pfa(cpx *z, const float (&xa)[15], const float (&xb)[15], const Geo &g, const float *ltab,
                                              int l, bool act)
{
    const cpx *tw = (const cpx *)(ltab + tw_off(g.N));
    if (act && l < g.nb15) {
        const int an = l & (g.nblk - 1);
        const int a = (int)(__brev((unsigned)an) >> (32 - g.fft_n));
        cpx x[15];
#pragma unroll
        for (int k = 0; k < 15; k++) x[k] = cmul(cpx{ xa[k], xb[k] }, tw[a + g.nblk * k]);
        cpx y[15];
        fft15_pfa(y, x);
#pragma unroll
        for (int m = 0; m < 15; m++) z[15 * l + m] = y[m];
    }
    __builtin_amdgcn_wave_barrier();
}

// The hot geometry (960 samples, one block): levels 4-5 of fft_calc (dopus.d:1596-1606) and the post-rotation of imdct15_half
// (:1629-1636) in ONE pass over z.  Outputs n0 + 120 q (q = 0..3) of the radix-4 butterfly n0 are the post-rotation partners
// of outputs (119 - n0) + 120 (3 - q) -- the partners' indices add up to 479 -- so a lane that runs the butterflies p and
// 119 - p holds both ends of every pair: the rotation happens in registers, z is read once and written once (frame_rest: two
// passes, 30 more LDS reads and 15 more writes per lane and channel).  60 butterfly pairs per channel on 32 lanes.
// The window of block 0 (vector_fmul_window, :230-243: d[k], d[119 - k] for k < 60, d = buf + 1024) rides along: d[60 + i] is
// output i of this pass, so the lane that produces z[p], p < 30, holds s1 of k = 59 - 2p and 58 - 2p and only fetches the two
// values of the previous frame's tail that go with them (w4 = win[k], win[119 - k] for those two k).
__device__ __forceinline__ void last_pass_960(cpx *z, const float *ltab, float scale, int l, float *d, const float (&w4)[4])
{
    const cpx *ex4 = (const cpx *)(ltab + ex_off(4)), *ex5 = (const cpx *)(ltab + ex_off(5));
    const cpx *tw = (const cpx *)(ltab + tw_off(6));
    auto bfly = [&](cpx (&v)[4], int n0) {
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = z[n0 + 120 * q];
        const cpx e4 = ex4[n0], e5a = ex5[n0], e5b = ex5[n0 + 120];
        cpx t, l0;
        t = cmul(v[1], e4); l0 = v[0]; v[1] = cpx{ l0.re - t.re, l0.im - t.im }; v[0] = cpx{ l0.re + t.re, l0.im + t.im };
        t = cmul(v[3], e4); l0 = v[2]; v[3] = cpx{ l0.re - t.re, l0.im - t.im }; v[2] = cpx{ l0.re + t.re, l0.im + t.im };
        t = cmul(v[2], e5a); l0 = v[0]; v[2] = cpx{ l0.re - t.re, l0.im - t.im }; v[0] = cpx{ l0.re + t.re, l0.im + t.im };
        t = cmul(v[3], e5b); l0 = v[1]; v[3] = cpx{ l0.re - t.re, l0.im - t.im }; v[1] = cpx{ l0.re + t.re, l0.im + t.im };
    };
    auto rot = [&](cpx za, cpx zb, int a, int b) {             // a < 240 <= b, a + b = 479
        const cpx ta = tw[a], tc = tw[b];
        const float r0 = za.im * ta.im - za.re * ta.re;
        const float i1 = za.im * ta.re + za.re * ta.im;
        const float r1 = zb.im * tc.im - zb.re * tc.re;
        const float i0 = zb.im * tc.re + zb.re * tc.im;
        z[a] = cpx{ scale * r0, scale * i0 };
        z[b] = cpx{ scale * r1, scale * i1 };
    };
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int p = l + 32 * u;
        if (p < 60) {
            cpx A[4], B[4];
            bfly(A, p);
            bfly(B, 119 - p);
            if (u == 0 && p < 30) {
                const cpx ta = tw[p], tc = tw[479 - p];
                const cpx za = A[0], zb = B[3];
                const float r0 = za.im * ta.im - za.re * ta.re;
                const float i1 = za.im * ta.re + za.re * ta.im;
                const float r1 = zb.im * tc.im - zb.re * tc.re;
                const float i0 = zb.im * tc.re + zb.re * tc.im;
                z[479 - p] = cpx{ scale * r1, scale * i1 };
                const float y0 = scale * r0, y1 = scale * i0;               // d[60 + 2p], d[61 + 2p]
                const f32x2 t = *(const f32x2 *)(d + 58 - 2 * p);           // d[k2], d[k1]: the previous frame's tail
                *(f32x2 *)(d + 58 - 2 * p) = f32x2{ t.x * w4[3] - y1 * w4[2], t.y * w4[1] - y0 * w4[0] };
                z[p] = cpx{ t.y * w4[0] + y0 * w4[1], t.x * w4[2] + y1 * w4[3] };
            } else {
                rot(A[0], B[3], p, 479 - p);
            }
            rot(A[1], B[2], p + 120, 359 - p);
            rot(B[1], A[2], 239 - p, p + 240);
            rot(B[0], A[3], 119 - p, p + 360);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// load_inputs of celt_core.h for the hot geometry (960 samples, one block): two base pointers per lane and compile-time
// element offsets, so that the 30 loads carry their offsets as immediates instead of a 64-bit address each
__device__ __forceinline__ void load_inputs_960(float (&xa)[15], float (&xb)[15], const float *__restrict__ coeffs,
                                                const afg_celt_frame &fr, int l)
{
    const int a = (int)(__brev((unsigned)(l & 31)) >> 27);
    const float *pb = coeffs + fr.coef_off + 2 * a;                  // x[2 i],        i = a + 32 k
    const float *pa = coeffs + fr.coef_off + (959 - 2 * a);          // x[959 - 2 i]
#pragma unroll
    for (int k = 0; k < 15; k++) {
        xa[k] = AFG_CELT_LD(pa - 64 * k);
        xb[k] = AFG_CELT_LD(pb + 64 * k);
    }
}

// ---- post-filter on the linear frame buffer (data = buf + 1024: CeltFrame.buf + 1024 of the reference) ------------------
// celt_postfilter_apply_transition (dopus.d:3281-3324) on data[n0 .. n0 + 120)
// Everything a step reads is at least min(T0, T1) - 2 samples old, so that many samples are independent: a lane (32 per
// channel) takes U = 4, 2 or 1 of them per step -- all reads of a step are issued before its first write.
template <int U>
__device__ __forceinline__ void wpf_transition_steps(float *d, const float *win2, const PfW &pf, int l, bool go, bool live0,
                                                     bool live1, int step)
{
    const int T0 = pf.period_old, T1 = pf.period;
    for (int i0 = 0; i0 < 120; i0 += step) {
        float v[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + l + 32 * u;
            on[u] = go && l + 32 * u < step && i < 120;
            v[u] = 0.0f;
            if (on[u]) {
                const float w = win2[i];
                const float x0 = live1 ? d[i - T1 + 2] : 0.0f, x1 = live1 ? d[i - T1 + 1] : 0.0f, x2 = live1 ? d[i - T1] : 0.0f,
                            x3 = live1 ? d[i - T1 - 1] : 0.0f, x4 = live1 ? d[i - T1 - 2] : 0.0f;
                const float y0 = live0 ? d[i - T0 + 2] : 0.0f, y1 = live0 ? d[i - T0 + 1] : 0.0f, y2 = live0 ? d[i - T0] : 0.0f,
                            y3 = live0 ? d[i - T0 - 1] : 0.0f, y4 = live0 ? d[i - T0 - 2] : 0.0f;
                const float uw = 1.0f - w;
                const float acc = uw * (pf.g_old[0] * y2 + pf.g_old[1] * (y3 + y1) + pf.g_old[2] * (y4 + y0)) +
                                  w * (pf.g[0] * x2 + pf.g[1] * (x1 + x3) + pf.g[2] * (x0 + x4));
                v[u] = d[i] + acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < U; u++)
            if (on[u]) d[i0 + l + 32 * u] = v[u];
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ void wpf_transition(float *data, const float *win2, int n0, const PfW &pf, int l, bool lane_on)
{
    const bool go = lane_on && !(pf.g[0] == 0.0f && pf.g_old[0] == 0.0f);
    if (!__any(go)) return;
    const int T0 = pf.period_old, T1 = pf.period;
    // a filter whose gains are all zero contributes exact zeros whatever it reads (its period may still be 0 on a
    // fresh decoder): only live filters bound the parallel step, and a dead filter's taps are not read
    const bool live0 = pf.g_old[0] != 0.0f || pf.g_old[1] != 0.0f || pf.g_old[2] != 0.0f;
    const bool live1 = pf.g[0] != 0.0f || pf.g[1] != 0.0f || pf.g[2] != 0.0f;
    int step = 128;
    if (go && live0) step = min(step, T0 - 2);
    if (go && live1) step = min(step, T1 - 2);
    step = max(step, 1);
    step = min(__shfl(step, 0), __shfl(step, 32));
    float *d = data + n0;
    if (step >= 120) wpf_transition_steps<4>(d, win2, pf, l, go, live0, live1, 128);
    else if (step > 32) wpf_transition_steps<2>(d, win2, pf, l, go, live0, live1, min(step, 64));
    else wpf_transition_steps<1>(d, win2, pf, l, go, live0, live1, step);
}

// celt_postfilter_apply (dopus.d:3326-3355) on data[n0 .. n0 + len): a lane takes up to U samples of a step of up to
// T - 2 samples (everything a step reads is at least that old)
template <int U>
__device__ __forceinline__ void wpf_apply_steps(float *d, int len, const PfW &pf, int l, bool go, int step)
{
    const int T = pf.period;
    for (int i0 = 0; i0 < len; i0 += step) {
        float v[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + l + 32 * u;
            on[u] = go && l + 32 * u < step && i < len;
            v[u] = 0.0f;
            if (on[u]) {
                const float x0 = d[i - T + 2], x1 = d[i - T + 1], x2 = d[i - T], x3 = d[i - T - 1], x4 = d[i - T - 2];
                v[u] = d[i] + (pf.g[0] * x2 + pf.g[1] * (x1 + x3) + pf.g[2] * (x0 + x4));
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < U; u++)
            if (on[u]) d[i0 + l + 32 * u] = v[u];
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ void wpf_apply(float *data, int n0, int len, const PfW &pf, int l, bool lane_on)
{
    const bool go = lane_on && pf.g[0] != 0.0f && len > 0;
    if (!__any(go)) return;
    int step = go ? max(min(pf.period - 2, 32 * AFG_WALK_PFU), 1) : 32 * AFG_WALK_PFU;
    step = min(__shfl(step, 0), __shfl(step, 32));
    float *d = data + n0;
    if (AFG_WALK_PFU > 4 && step > 128) wpf_apply_steps<AFG_WALK_PFU>(d, len, pf, l, go, step);
    else if (step > 64) wpf_apply_steps<4>(d, len, pf, l, go, step);
    else if (step > 32) wpf_apply_steps<2>(d, len, pf, l, go, step);
    else wpf_apply_steps<1>(d, len, pf, l, go, step);
}

// ---- de-emphasis (dopus.d:3695-3701), re-associated ---------------------------------------------------------------------
//   tmp = x[j] + m;  m = tmp * c;  out[j] = tmp / 32768          (c = 0.8500061)
// A lane takes 15 consecutive samples of a channel (64 lanes x 15 = a 20 ms frame): the recurrence from a zero memory
// in registers (t), then the memory each lane's run really starts from -- B[l+1] = c * t14[l] + c^15 * B[l], B[0] = m, a
// weighted prefix sum over the lanes: doubling steps inside the rows of 16 (DPP row_shr), the two cross-row steps
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- and out[k] = (t[k] + c^k * B) / 32768.
struct DeW {
    float w16, w32, wl;                                      // r^((lane & 15) + 1), r^((lane & 31) + 1), r^lane; r = c^15
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}

constexpr float cpow(float c, int n) { float p = 1.0f; for (int i = 0; i < n; i++) p *= c; return p; }
constexpr float kR15 = cpow(kDeemph, 15);

__device__ __forceinline__ float lane_value(float v, int lane_uniform)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}

// x[k]: sample 15 * lane + k of the frame (zeros on lanes >= n_act); returns the scaled outputs in x and the new memory
__device__ __forceinline__ void deemph_chunk(float (&x)[15], float &m, const DeW &w, int n_act)
{
    constexpr float r1 = kR15, r2 = r1 * r1, r4 = r2 * r2, r8 = r4 * r4;
#pragma unroll
    for (int k = 1; k < 15; k++) x[k] = __builtin_fmaf(kDeemph, x[k - 1], x[k]);
    float S = kDeemph * x[14];
    S = __builtin_fmaf(r1, dpp0<0x111, 0xf>(S), S);          // row_shr:1
    S = __builtin_fmaf(r2, dpp0<0x112, 0xf>(S), S);          // row_shr:2
    S = __builtin_fmaf(r4, dpp0<0x114, 0xf>(S), S);          // row_shr:4
    S = __builtin_fmaf(r8, dpp0<0x118, 0xf>(S), S);          // row_shr:8
    S = __builtin_fmaf(w.w16, dpp0<0x142, 0xa>(S), S);       // row_bcast:15 -> rows 1, 3
    S = __builtin_fmaf(w.w32, dpp0<0x143, 0xc>(S), S);       // row_bcast:31 -> rows 2, 3
    const float B = __builtin_fmaf(w.wl, m, dpp0<0x138, 0xf>(S));             // wave_shr:1: lane 0 reads 0
    m = lane_value(__builtin_fmaf(w.wl * r1, m, S), n_act - 1);
#pragma unroll
    for (int k = 0; k < 15; k++) x[k] = __builtin_fmaf(cpow(kDeemph, k), B, x[k]) * (1.0f / 32768.0f);
}

// ---- records ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool rec_dead(const afg_celt_frame *r)     // the frame installs a filter that does nothing
{
    return r->pf_gains_new[0] == 0.0f && r->pf_gains_new[1] == 0.0f && r->pf_gains_new[2] == 0.0f;
}

// Is frame q of the sequence at `seq` (cnt frames) a cut?  Returns the warm-up length W > 0 (walk frames q - W .. q - 1
// without storing, from a zeroed buffer, then q onwards is exact), else 0.  Conditions: the records q - W - 1 .. q - 1 all
// install dead filters -- so frames q - W .. q - 1 are left untouched by celt_postfilter apart from the first 120 samples
// of frame q - W, and the filter state entering frame q - W + 1 .. q is dead -- and the warm-up frames hold at least
// 120 + 1026 samples: the first 120 of a walk that starts from nothing lack the previous frame's overlap, the filter of
// frame q reaches back 1022 + 2 samples (period <= 1022, dopus.d:3399-3407).
__device__ __forceinline__ int cut_warmup(const afg_celt_frame *seq, uint64_t cnt, uint64_t q)
{
    if (q < 2 || q >= cnt) return 0;
    int sum = 0;
    for (int w = 1; w <= 12 && (uint64_t)w < q; w++) {
        const afg_celt_frame *r = seq + (q - w);
        if (!rec_dead(r)) return 0;
        sum += r->frame_size;
        if (sum >= 120 + 1026) return rec_dead(r - 1) ? w : 0;
    }
    return 0;
}

// first cut in [lo, hi) of a channel pair (seq1 == nullptr: of one sequence); W of each channel through w0 / w1
__device__ __forceinline__ uint64_t first_cut(const afg_celt_frame *seq0, const afg_celt_frame *seq1, uint64_t cnt,
                                              uint64_t lo, uint64_t hi, int lane, int &w0, int &w1)
{
    for (uint64_t q0 = lo; q0 < hi; q0 += 64) {
        const uint64_t q = q0 + lane;
        int a = 0, b = 0;
        if (q < hi) {
            a = cut_warmup(seq0, cnt, q);
            b = (seq1 && a) ? cut_warmup(seq1, cnt, q) : a;
        }
        const unsigned long long hit = __ballot(a > 0 && b > 0);
        if (hit) {
            const int first = __ffsll(hit) - 1;
            w0 = __shfl(a, first);
            w1 = __shfl(b, first);
            return q0 + first;
        }
    }
    w0 = w1 = 0;
    return hi;
}

// largest c in [0, n) with rec_base[c] <= x (rec_base is non-decreasing, rec_base[0] <= x): 64 probes per round
__device__ __forceinline__ uint32_t find_seq(const uint64_t *__restrict__ rec_base, uint32_t n, uint64_t x, int lane)
{
    uint32_t lo = 0, hi = n;                                  // answer in [lo, hi)
    while (hi - lo > 1) {
        const uint32_t span = hi - lo;
        const uint32_t stepw = (span + 63) / 64;
        const uint32_t c = lo + (uint32_t)lane * stepw;
        const bool le = c < hi && rec_base[c] <= x;
        const unsigned long long m = __ballot(le);            // lanes are ordered: a prefix of ones
        const int k = m ? 63 - __clzll(m) : 0;                // (lane 0 holds whenever rec_base[lo] <= x)
        const uint32_t nlo = lo + (uint32_t)k * stepw;
        const uint32_t nhi = min(hi, nlo + stepw);
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

__global__ __launch_bounds__(64 * kWWaves) void celt_walk_kernel(
    const uint64_t *__restrict__ rec_base, const afg_celt_frame *__restrict__ recs,
    const float *__restrict__ coeffs, float *__restrict__ out, float *__restrict__ states,
    const float *__restrict__ tables, CeltTables tb, uint32_t tab_floats, uint32_t n_chan, uint32_t seg_recs,
    uint32_t whole_frames, uint32_t *__restrict__ counter)
{
    extern __shared__ __attribute__((aligned(16))) float slds[];
    float *ltab = slds, *lwin = slds + kTabFloatsMax, *win2 = lwin + 120;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, l = lane & 31;
    const uint32_t pairs = (n_chan + 1) / 2;
    const uint64_t total = rec_base[n_chan];
    const uint64_t n_items = seg_recs ? (total + seg_recs - 1) / seg_recs : pairs;

    auto draw = [&]() -> uint64_t {
        uint32_t it = 0;
        if (lane == 0) it = atomicAdd(counter, 1u);
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)it);
    };
    uint64_t item = draw();
    // a workgroup none of whose wavefronts got an item leaves before it stages anything
    if (!__syncthreads_or(item < n_items)) return;
    for (uint32_t i = threadIdx.x; i < tab_floats; i += 64 * kWWaves) ltab[i] = tables[i];
    if (threadIdx.x < 240) lwin[threadIdx.x] = tables[kCeltWinAt + threadIdx.x];
    __syncthreads();

    float *bufs = win2 + 120 + (size_t)wv * 2 * 2048;
    DeW dw;
    {
        float p = 1.0f, c16 = 0.0f, c32 = 0.0f, cl = 0.0f;   // r^lane by repeated products (once per wavefront)
        for (int k = 0; k < 64; k++) {
            if (k == lane) cl = p;
            p *= kR15;
            if (k == (lane & 15)) c16 = p;
            if (k == (lane & 31)) c32 = p;
        }
        dw.w16 = c16; dw.w32 = c32; dw.wl = cl;
    }

    // Walks frames [a, end) of one sequence (lanes 0..31) or of both sequences of a pair (lane half = channel);
    // frames before `start` are the warm-up: transformed, nothing stored.
    auto walk = [&](bool both_rt, uint32_t chan_lo, uint32_t chan_hi, uint64_t base_lo, uint64_t base_hi, uint64_t cnt,
                    uint64_t a, uint64_t start, uint64_t end) __attribute__((always_inline)) {
        const bool act_rt = both_rt || h == 0;
        const bool both = both_rt, act = act_rt;
        const uint32_t my_chan = (both && h) ? chan_hi : chan_lo;
        const uint64_t my_base = (both && h) ? base_hi : base_lo;
        float *st = states ? states + (size_t)my_chan * AFG_CELT_STATE_FLOATS : nullptr;
        float *buf = bufs + (both ? h : 0) * 2048;
        cpx *z = (cpx *)(buf + 1024 + 60);
        const float *b0 = bufs, *b1 = bufs + 2048;

        PfW pf;
        pf.period = pf.period_old = 0;
        pf.g[0] = pf.g[1] = pf.g[2] = pf.g_old[0] = pf.g_old[1] = pf.g_old[2] = 0.0f;
        float m = 0.0f;                                      // de-emphasis memory of this lane's channel (uniform per half)
        const bool from_state = a == 0 && st != nullptr;
        if (act) {
            for (int i = l; i < 2048; i += 32) buf[i] = from_state ? st[i] : 0.0f;
            if (from_state) {
                pf.period = __float_as_int(st[2048]);
                pf.g[0] = st[2049]; pf.g[1] = st[2050]; pf.g[2] = st[2051];
                pf.period_old = __float_as_int(st[2052]);
                pf.g_old[0] = st[2053]; pf.g_old[1] = st[2054]; pf.g_old[2] = st[2055];
                m = st[2056];
            } else if (a > 0) {
                pf.period = pf.period_old = recs[my_base + a - 1].pf_period_new;   // dead filters: gains stay zero
            }
        }
        float wi[2], wj[2];                                  // block-0 window taps of this lane
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = min(l + 32 * u, 59);
            wi[u] = lwin[k];
            wj[u] = lwin[119 - k];
        }
        float w960[4];                                       // ... and in last_pass_960's mapping (k = 59 - 2l, 58 - 2l)
        {
            const int k1 = 59 - 2 * min(l, 29), k2 = k1 - 1;
            w960[0] = lwin[k1]; w960[1] = lwin[119 - k1]; w960[2] = lwin[k2]; w960[3] = lwin[119 - k2];
        }
        __builtin_amdgcn_wave_barrier();

        afg_celt_frame fr = recs[my_base + a], fr_next = fr;
        if (a + 1 < end) fr_next = recs[my_base + a + 1];
        float xa[15], xb[15];
        {
            const Geo g0 = geo_of(fr);
            if (is_960(g0)) load_inputs_960(xa, xb, coeffs, fr, l);
            else load_inputs(xa, xb, coeffs, fr, g0, l);
        }

        // one frame; K960: the hot case -- a stereo pair's dominant record (20 ms in one block) -- with the geometry and
        // the lane roles as compile-time constants (no lane predicates, fixed trip counts, immediate offsets)
        auto frame_step = [&](auto k960, bool storing, const afg_celt_frame &fr_next2) __attribute__((always_inline)) {
            constexpr bool K960 = decltype(k960)::value;
            const bool both = K960 ? true : both_rt;
            const bool act = K960 ? true : act_rt;
            const Geo g = K960 ? geo_960() : geo_of(fr);
            const int F = K960 ? 960 : g.F;
            // iMDCT and overlap-add, dopus.d:3684-3690
#if AFG_WALK_PFA
            frame_fft_pfa(z, xa, xb, g, ltab, l, act);
#else
            frame_fft(z, xa, xb, fr, g, ltab, lwin, tb, l, act);
#endif
            {                                                // (the last frame's again at the end)
                const Geo gn = geo_of(fr_next);
                if (is_960(gn)) {
                    load_inputs_960(xa, xb, coeffs, fr_next, l);
                    asm volatile("");                          // keeps the two branches' loads apart: merged into one block
                } else {                                       // they take selected 64-bit addresses instead of immediates
                    load_inputs(xa, xb, coeffs, fr_next, gn, l);
                }
            }
            if (K960) {
                radix_pass<3>(z, l, 32, 1, ltab, tb, true);
                last_pass_960(z, ltab, fr.imdct_scale, l, buf + 1024, w960);
            } else {
                frame_rest(z, fr, g, ltab, lwin, tb, l, act);
            }
            if (!K960 && act) {                              // vector_fmul_window of block 0 (K960: inside last_pass_960)
                float *d = buf + 1024;
                float va[2], vb[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int k = l + 32 * u;
                    if (k < 60) {
                        const float s0 = d[k], s1 = d[119 - k];
                        va[u] = s0 * wj[u] - s1 * wi[u];
                        vb[u] = s0 * wi[u] + s1 * wj[u];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int k = l + 32 * u;
                    if (k < 60) { d[k] = va[u]; d[119 - k] = vb[u]; }
                }
            }
            __builtin_amdgcn_wave_barrier();

            // celt_postfilter, dopus.d:3357-3378 (a no-op on warm-up frames: their filters are dead by construction)
            wpf_transition(buf + 1024, win2, 0, pf, l, act);
            pf.period_old = pf.period;
            pf.g_old[0] = pf.g[0]; pf.g_old[1] = pf.g[1]; pf.g_old[2] = pf.g[2];
            pf.period = fr.pf_period_new;
            pf.g[0] = fr.pf_gains_new[0]; pf.g[1] = fr.pf_gains_new[1]; pf.g[2] = fr.pf_gains_new[2];
            if (F > 120) {
                wpf_transition(buf + 1024, win2, 120, pf, l, act);
                wpf_apply(buf + 1024, 240, F - 240, pf, l, act);
                pf.period_old = pf.period;
                pf.g_old[0] = pf.g[0]; pf.g_old[1] = pf.g[1]; pf.g_old[2] = pf.g[2];
            }

            // make the prefetched inputs resident before the stores enter the queue (one in-order memory counter)
#pragma unroll
            for (int i = 0; i < 15; i++) asm volatile("" : "+v"(xa[i]), "+v"(xb[i]) : : "memory");
            {
                uint32_t *w = (uint32_t *)&fr_next2;
#pragma unroll
                for (int i = 0; i < 12; i++) asm volatile("" : : "v"(w[i]) : "memory");
            }
            // De-emphasis + scaling (dopus.d:3695-3701, re-associated: deemph_chunk), then coalesced stores.  The samples
            // pass through buf[0, F) of their channel on the way: history the next frame's filter no longer reaches.
            const int n_act = K960 ? 64 : F / 15;
            {
                float m0 = lane_value(m, 0), m1 = lane_value(m, 32);
                float xs[15];
                const int at = K960 ? 15 * lane : 15 * min(lane, n_act - 1);      // (idle lanes re-read the last run)
                const bool in = K960 || lane < n_act;
#pragma unroll
                for (int k = 0; k < 15; k++) { const float t = b0[1024 + at + k]; xs[k] = in ? t : 0.0f; }
                deemph_chunk(xs, m0, dw, n_act);
                if (in) {
#pragma unroll
                    for (int k = 0; k < 15; k++) bufs[at + k] = xs[k];
                }
                if (both) {
#pragma unroll
                    for (int k = 0; k < 15; k++) { const float t = b1[1024 + at + k]; xs[k] = in ? t : 0.0f; }
                    deemph_chunk(xs, m1, dw, n_act);
                    if (in) {
#pragma unroll
                        for (int k = 0; k < 15; k++) bufs[2048 + at + k] = xs[k];
                    }
                }
                m = (both && h) ? m1 : m0;
            }
            __builtin_amdgcn_wave_barrier();
            constexpr int so = 0;
            if (both) {
                f32x2 *o = (f32x2 *)(out + __shfl(fr.out_off, 0));
                // (celt_pair_ok guarantees an even out_off, i.e. 8-byte alignment; a pair whose frames start 2 mod 4 floats
                // into the plane -- an odd number of stereo frames in front of it -- takes the 8-byte stores below)
                if (K960 && ((uintptr_t)o & 15) == 0) {
#if AFG_WALK_ST16
                    // 16-byte stores: a lane takes two consecutive samples of both channels (8 x 1 KB rows per frame)
                    f32x4 v[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int j = min(lane + 64 * i, 479);                   // sample pair (the last row is half full)
                        const f32x2 a = *(const f32x2 *)(b0 + so + 2 * j), b = *(const f32x2 *)(b1 + so + 2 * j);
                        v[i] = f32x4{ a.x, b.x, a.y, b.y };
                    }
                    if (storing) {
#pragma unroll
                        for (int i = 0; i < 8; i++)
                            if (i < 7 || lane < 32) __builtin_nontemporal_store(v[i], (f32x4 *)(o + 2 * (lane + 64 * i)));
                    }
#else
                    f32x2 v[15];
#pragma unroll
                    for (int i = 0; i < 15; i++) v[i] = f32x2{ b0[so + lane + 64 * i], b1[so + lane + 64 * i] };
                    if (storing) {
#pragma unroll
                        for (int i = 0; i < 15; i++) o[lane + 64 * i] = v[i];
                    }
#endif
                } else if (storing) {
                    for (int p = lane; p < F; p += 64) o[p] = f32x2{ b0[so + p], b1[so + p] };
                }
            } else if (storing) {
                float *o = out + __shfl(fr.out_off, 0);
                const size_t stride = __shfl(fr.out_stride, 0);
                for (int p = lane; p < F; p += 64) o[(size_t)p * stride] = b0[so + p];
            }
            __builtin_amdgcn_wave_barrier();
            // memmove(buf, buf + F, 1084 floats) (:3370): every read is issued before the first write
            {
                f32x4 mv[9];
#pragma unroll
                for (int u = 0; u < 9; u++)
                    if (act && l + 32 * u < 271) mv[u] = *(const f32x4 *)(buf + F + 4 * (l + 32 * u));
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 9; u++)
                    if (act && l + 32 * u < 271) *(f32x4 *)(buf + 4 * (l + 32 * u)) = mv[u];
                __builtin_amdgcn_wave_barrier();
            }
        };

        for (uint64_t q = a; q < end; q++) {
            afg_celt_frame fr_next2 = recs[my_base + (q + 2 < end ? q + 2 : end - 1)];
            const Geo gq = geo_of(fr);
            if (both && is_960(gq)) frame_step(std::true_type{}, q >= start, fr_next2);
            else frame_step(std::false_type{}, q >= start, fr_next2);
            fr = fr_next;
            fr_next = fr_next2;
        }

        if (st && act && end == cnt) {
            for (int i = l; i < 2048; i += 32) st[i] = buf[i];
            if (l == 0) {
                st[2048] = __int_as_float(pf.period);
                st[2049] = pf.g[0]; st[2050] = pf.g[1]; st[2051] = pf.g[2];
                st[2052] = __int_as_float(pf.period_old);
                st[2053] = pf.g_old[0]; st[2054] = pf.g_old[1]; st[2055] = pf.g_old[2];
                st[2056] = m;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // One call site for the walk (so that it is inlined and its LDS pointers keep their address space): an item is
    // broken into the channel pairs it overlaps, a pair into one run (two sequences of equal length, walked together when
    // their records pair up) or two (sequences of different lengths), a run into one or two walks.
    while (item < n_items) {
        uint32_t p_lo = (uint32_t)item, p_hi = p_lo + 1;
        uint64_t x0 = 0, x1 = 0;
        if (seg_recs) {
            x0 = item * seg_recs;
            x1 = x0 + seg_recs < total ? x0 + seg_recs : total;
            p_lo = find_seq(rec_base, n_chan, x0, lane) >> 1;
            p_hi = pairs;
        }
        for (uint32_t p = p_lo; p < p_hi; p++) {
            const uint32_t c0 = 2 * p, c1 = c0 + 1;
            const uint64_t base0 = rec_base[c0], cnt0 = rec_base[c0 + 1] - base0;
            if (seg_recs && base0 >= x1) break;
            uint64_t base1 = base0 + cnt0, cnt1 = 0;
            if (c1 < n_chan) { base1 = rec_base[c1]; cnt1 = rec_base[c1 + 1] - base1; }
            const uint64_t span = cnt0 + cnt1, cmax = cnt0 > cnt1 ? cnt0 : cnt1;
            if (span == 0) continue;
            uint64_t qa = 0, qb = cmax;
            if (seg_recs) {
                const uint64_t lo = x0 > base0 ? x0 - base0 : 0, hi = (x1 < base0 + span ? x1 : base0 + span) - base0;
                if (lo >= hi) continue;
                // a monotone map of flat offsets onto frames of the longer sequence: consecutive items partition [0, cmax)
                qa = (lo * cmax + span - 1) / span;
                qb = (hi * cmax + span - 1) / span;
            }
            const bool equal = cnt0 == cnt1;
            for (int r = 0; r < (equal ? 1 : 2); r++) {
                // run r: frames [ra, rb) of one sequence, or of both sequences when they are equally long
                const bool two = equal;
                const uint32_t rc = r ? c1 : c0;
                const uint64_t rbase = r ? base1 : base0, rcnt = r ? cnt1 : cnt0;
                const uint64_t ra = qa < rcnt ? qa : rcnt, rb = qb < rcnt ? qb : rcnt;
                if (ra >= rb) continue;
                const afg_celt_frame *s0 = recs + rbase, *s1 = two ? recs + base1 : nullptr;
                uint64_t start = 0, end = rcnt;
                int w0 = 0, w1 = 0;
                if (rcnt <= whole_frames) {
                    // a short sequence is not worth a warm-up (two frames of transform per cut): the item that holds its
                    // first frame walks all of it, the others pass
                    if (ra > 0) continue;
                } else
                if (ra > 0) {
                    start = first_cut(s0, s1, rcnt, ra, rb, lane, w0, w1);
                    if (start >= rb) continue;               // no cut in the range: an earlier item walks through it
                }
                if (rb < rcnt && rcnt > whole_frames) {
                    int e0, e1;
                    end = first_cut(s0, s1, rcnt, rb, rcnt, lane, e0, e1);
                }
                bool paired = two && w0 == w1;
                if (paired) {
                    bool bad = false;
                    for (uint64_t q = start - w0 + lane; q < end; q += 64) bad = bad || !celt_pair_ok(s0[q], s1[q]);
                    paired = !__any(bad);
                }
                for (int j = 0; j < ((two && !paired) ? 2 : 1); j++) {
                    const uint32_t jc = j ? c1 : rc;
                    const uint64_t jbase = j ? base1 : rbase;
                    walk(paired, jc, jc + 1, jbase, base1, rcnt, start - (j ? w1 : w0), start, end);
                }
            }
        }
        item = draw();
    }
}

std::mutex g_mu;
uint32_t *g_counters[AFG_MAX_DEVICES] = {};
uint32_t g_launches[AFG_MAX_DEVICES] = {};
int g_cus[AFG_MAX_DEVICES] = {};

}  // namespace

// Records per item from the number of channel sequences.  Long streams are cut into 64-frame pieces (the warm-up of a
// piece costs two frames of transform: 3 %); with thousands of streams the pieces only have to even out the tail.
// afg_dev_option("celt_seg_recs") overrides (0: whole channel pairs).
static uint32_t seg_recs_for(uint32_t n_chan)
{
    {
        const long v = afg::dev_option(afg::kDevCeltSegRecs);
        if (v >= 0 && v <= (1 << 24)) return (uint32_t)v;
    }
    return 128;
}

int afg::celt_walk_launch(uint32_t n_chan, const uint64_t *d_rec_base, const afg_celt_frame *d_recs, const float *d_coeffs,
                          float *d_out, float *d_states, hipStream_t stream)
{
    const float *d_tables = nullptr;
    CeltTables tb;
    uint32_t tab_floats = 0;
    if (int rc = afg::celt_tables_for_device(&d_tables, &tb, &tab_floats)) return rc;
    int dev = 0;
    if (int rc = afg::device_slot(&dev, "afg_celt_transform_hip")) return rc;
    uint32_t *counter = nullptr;
    int cus = 0;
    {
        // work counters of the persistent kernel: launch k uses (and first clears, on its stream) counter k % 64, so up to
        // 64 launches in flight on different streams never share one (a 65th would: the ring is the documented limit)
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_counters[dev]) {
            // published only when every step succeeded: a half-initialised slot would fail every later launch
            uint32_t *ring = nullptr;
            int n_cus = 0;
            static_assert(kWLdsFloats * sizeof(float) <= 160 * 1024, "LDS budget");
            hipError_t e = hipMalloc(&ring, 64 * sizeof(uint32_t));
            if (e == hipSuccess) e = hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void *)celt_walk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)(kWLdsFloats * sizeof(float)));
            if (e != hipSuccess) {
                if (ring) (void)hipFree(ring);
                afg::set_error("afg_celt_transform_hip: device set-up failed: %s", hipGetErrorString(e));
                return AFG_ERR_HIP;
            }
            g_cus[dev] = n_cus;
            g_counters[dev] = ring;
        }
        counter = g_counters[dev] + (g_launches[dev]++ & 63u);
        cus = g_cus[dev] > 0 ? g_cus[dev] : 256;
    }
    AFG_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(uint32_t), stream));
    // A carry state is read by the walk that starts at a sequence's first frame and rewritten, in place, by the one that
    // ends at its last: with states every channel pair is one item, so that this is one wavefront, in that order.
    const uint32_t seg_recs = d_states ? 0 : seg_recs_for(n_chan);
    // With at least four channel pairs per wavefront slot (8 wavefronts x 256 CUs) sequences of up to 512 frames (10 s of 20 ms
    // frames: a walk of 4.6 ms) are walked whole -- a cut costs two frames of warm-up, and there are enough sequences to even
    // out the tail; with fewer every sequence is cut wherever it can be (the mixed corpus: 2 460 streams per wave, 22 ms against
    // 29 ms with its short ones whole).  afg_dev_option("celt_whole_frames") overrides.
    uint32_t whole_frames = (n_chan + 1) / 2 >= 4u * kWWaves * 256u ? 512 : 0;
    {
        const long v = afg::dev_option(afg::kDevCeltWholeFrames);
        if (v >= 0 && v <= (1 << 24)) whole_frames = (uint32_t)v;
    }
    const uint32_t pairs = (n_chan + 1) / 2;
    // one workgroup per CU (147 KB of LDS); with whole pairs as items never more wavefronts than pairs
    uint32_t groups = (uint32_t)cus;
    if (seg_recs == 0) groups = std::min<uint32_t>(groups, (pairs + kWWaves - 1) / kWWaves);
    hipLaunchKernelGGL(celt_walk_kernel, dim3(groups), dim3(64 * kWWaves), kWLdsFloats * sizeof(float), stream, d_rec_base,
                       d_recs, d_coeffs, d_out, d_states, d_tables, tb, tab_floats, n_chan, seg_recs, whole_frames, counter);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}
