// celt_transform.hip -- Opus/CELT transform stage on gfx950.
//
// Replaces the per-channel tail of ff_celt_decode_frame (reference dopus.d:3680-3702):
// imdct15_half (:1611-1637) + vector_fmul_window (:230-243) per block, celt_postfilter
// (:3281-3378) and de-emphasis / scaling (:3695-3701).  Float results follow the reference's
// expression trees (float, with the double sub-expressions of :3313-3318); tables are built on
// the host in x87 long double like the D source's `real` (:1489-1499).
//
// The iMDCT of a frame depends on no other frame; the comb post-filter feeds on its own output (lag >= 13
// samples) and the de-emphasis is a one-pole IIR across the whole stream.  Three device paths share the code:
//   * celt_stream_kernel (batches that fill the device): one wavefront walks a stereo stream with CeltFrame.buf
//     of both channels in LDS -- transform, windows, post-filter, interleaved stores -- reading the coefficients
//     once and writing the post-filtered PCM once;
//   * celt_imdct_kernel + celt_postfilter_kernel (few streams): the transform of all records in parallel, then the
//     sequential part in place on the output plane;
//   * celt_deemph_kernel: the de-emphasis recurrence (rounding order cannot be re-associated), one lane per
//     channel sequence, over the output plane.
#include "celt_core.h"
#include "celt_tables.h"


#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

namespace {

__device__ __constant__ float d_celt_window[120];
__device__ __constant__ float d_celt_window2[120];


constexpr int kAWaves = 4;               // wavefronts per workgroup of kernel A (they share the LDS copy of the tables)

__global__ __launch_bounds__(64 * kAWaves) void celt_imdct_kernel(
    const uint64_t *__restrict__ rec_base, const afg_celt_frame *__restrict__ recs,
    const float *__restrict__ coeffs, float *__restrict__ out, float *__restrict__ states,
    const float *__restrict__ tables, CeltTables tb, uint32_t tab_floats, uint32_t n_chan, uint32_t per_pair)
{
    __shared__ __attribute__((aligned(16))) float ltab[kTabFloatsMax];
    __shared__ __attribute__((aligned(16))) float lwin[120];
    __shared__ __attribute__((aligned(16))) cpx zbuf[kAWaves][2][480];
    for (uint32_t i = threadIdx.x; i < tab_floats; i += 64 * kAWaves) ltab[i] = tables[i];
    if (threadIdx.x < 120) lwin[threadIdx.x] = d_celt_window[threadIdx.x];
    __syncthreads();

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, l = lane & 31;
    // channel pair (2c, 2c+1); its records are split into per_pair contiguous chunks, one per wavefront
    const uint32_t wid = blockIdx.x * kAWaves + wv;
    const uint32_t pair = wid / per_pair, k0 = wid % per_pair;
    const uint32_t c0 = 2 * pair, c1 = c0 + 1;
    if (c0 >= n_chan) return;
    const uint64_t base0 = rec_base[c0], cnt0 = rec_base[c0 + 1] - base0;
    uint64_t base1 = 0, cnt1 = 0;
    if (c1 < n_chan) { base1 = rec_base[c1]; cnt1 = rec_base[c1 + 1] - base1; }
    const uint64_t cmax = cnt0 > cnt1 ? cnt0 : cnt1;
    const uint64_t chunk = (cmax + per_pair - 1) / per_pair;
    const uint64_t q_lo = (uint64_t)k0 * chunk, q_hi = q_lo + chunk < cmax ? q_lo + chunk : cmax;
    cpx *z = zbuf[wv][h];
    float *Y = (float *)z;                                   // block outputs, frame position p = 60 + index
    const float *Y0 = (const float *)zbuf[wv][0], *Y1 = (const float *)zbuf[wv][1];

    // one pass of the transform over the record this half-wave holds in `fr` (act = lane has a record)
    auto transform = [&](const afg_celt_frame &fr, bool act, bool paired, uint64_t out_even, uint32_t my_chan,
                         uint64_t my_base, uint64_t my_cnt, uint64_t q) {
        const Geo g = geo_of(fr);
        const int F = g.F;
        float xa[15], xb[15];
        if (is_960(g)) {
            load_inputs(xa, xb, coeffs, fr, geo_960(), l);
            frame_fft(z, xa, xb, fr, geo_960(), ltab, lwin, tb, l, act);
            frame_rest(z, fr, geo_960(), ltab, lwin, tb, l, act);
        } else {
            load_inputs(xa, xb, coeffs, fr, g, l);
            frame_fft(z, xa, xb, fr, g, ltab, lwin, tb, l, act);
            frame_rest(z, fr, g, ltab, lwin, tb, l, act);
        }
        // frame positions [60, F) -> this frame's slots
        if (paired) {
            f32x2 *o = (f32x2 *)(out + out_even);
            for (int p = 60 + lane; p < F; p += 64) o[p] = f32x2{ Y0[p - 60], Y1[p - 60] };
        } else if (act) {
            float *o = out + fr.out_off;
            for (int p = 60 + l; p < F; p += 32) o[(size_t)p * fr.out_stride] = Y[p - 60];
        }
        // frame positions [F, F + 60) -> the next frame's slots [0, 60), or the state blob
        if (act) {
            float *o = nullptr;
            size_t stride = 1;
            if (q + 1 < my_cnt) {
                const afg_celt_frame *nx = recs + my_base + q + 1;
                o = out + nx->out_off;
                stride = nx->out_stride;
            } else if (states) {
                o = states + (size_t)my_chan * AFG_CELT_STATE_FLOATS + kTailSlot;
            }
            if (o)
                for (int k = l; k < 60; k += 32) o[(size_t)k * stride] = Y[F - 60 + k];
        }
        __builtin_amdgcn_wave_barrier();
    };

    for (uint64_t q = q_lo; q < q_hi; q++) {
        const bool have0 = q < cnt0, have1 = q < cnt1;
        afg_celt_frame f0 = {}, f1 = {};
        if (have0) f0 = recs[base0 + q];
        if (have1) f1 = recs[base1 + q];
        const bool paired = have0 && have1 && celt_pair_ok(f0, f1);
        if (paired) {
            transform(h ? f1 : f0, true, true, f0.out_off, h ? c1 : c0, h ? base1 : base0, h ? cnt1 : cnt0, q);
        } else {
            if (have0) transform(f0, h == 0, false, 0, c0, base0, cnt0, q);
            if (have1) transform(f1, h == 0, false, 0, c1, base1, cnt1, q);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Kernel B: first overlap window of every frame + celt_postfilter (dopus.d:3281-3378), in place on the output
// plane.  The comb filter feeds on its own output, so a channel sequence is walked in order; one wavefront takes
// the two channels of a stereo stream (32 lanes each) when all their records pair up, else one sequence.  History
// lives in a 2048-sample LDS ring per channel (the reference's CeltFrame.buf without the memmove, :3370).
// ---------------------------------------------------------------------------------------------------------------
struct PfState {
    int period, period_old;
    float g[3], g_old[3];
};

constexpr int kRing = 2048;

// celt_postfilter_apply_transition (dopus.d:3281-3324) on ring positions [n0, n0 + 120)
__device__ __forceinline__ void pf_transition(float *ring, const float *win2, int n0, const PfState &pf, int l, bool lane_on)
{
    const bool go = lane_on && !(pf.g[0] == 0.0f && pf.g_old[0] == 0.0f);
    if (!__any(go)) return;
    const int T0 = pf.period_old, T1 = pf.period;
    // a filter whose gains are all zero contributes exact zeros whatever it reads (its period may
    // still be 0 on a fresh decoder): only live filters bound the parallel step
    const bool live0 = pf.g_old[0] != 0.0f || pf.g_old[1] != 0.0f || pf.g_old[2] != 0.0f;
    const bool live1 = pf.g[0] != 0.0f || pf.g[1] != 0.0f || pf.g[2] != 0.0f;
    int step = 32;
    if (go && live0) step = min(step, T0 - 2);
    if (go && live1) step = min(step, T1 - 2);
    step = max(step, 1);
    step = min(__shfl(step, 0), __shfl(step, 32));
    for (int i0 = 0; i0 < 120; i0 += step) {
        const int i = i0 + l;
        float v = 0.0f;
        const bool on = go && l < step && i < 120;
        if (on) {
            const float w = win2[i];
#define RD(x) ring[(n0 + (x)) & (kRing - 1)]
            // A dead filter (all gains zero) multiplies whatever it reads by zero -- and on a fresh decoder its period is
            // still 0, so it "reads" i - 2 .. i + 2: for the last two samples of a 120-sample frame that is past the
            // frame.  The reference finds the frame's finite overlap tail there (buf + 1024 + F); this kernel's ring holds
            // whatever LDS held (the tail lives in the next frame's slots), and 0 * NaN is NaN.  A dead filter's taps
            // are therefore taken as +0.0 without touching memory: the same sum whenever the reference's data are finite
            // (up to the sign of an all-zero sum).
            const float x0 = live1 ? RD(i - T1 + 2) : 0.0f, x1 = live1 ? RD(i - T1 + 1) : 0.0f, x2 = live1 ? RD(i - T1) : 0.0f,
                        x3 = live1 ? RD(i - T1 - 1) : 0.0f, x4 = live1 ? RD(i - T1 - 2) : 0.0f;
            const float y0 = live0 ? RD(i - T0 + 2) : 0.0f, y1 = live0 ? RD(i - T0 + 1) : 0.0f, y2 = live0 ? RD(i - T0) : 0.0f,
                        y3 = live0 ? RD(i - T0 - 1) : 0.0f, y4 = live0 ? RD(i - T0 - 2) : 0.0f;
            const double acc = (1.0 - w) * pf.g_old[0] * y2 +
                               (1.0 - w) * pf.g_old[1] * (y3 + y1) +
                               (1.0 - w) * pf.g_old[2] * (y4 + y0) +
                               w * pf.g[0] * x2 +
                               w * pf.g[1] * (x1 + x3) +
                               w * pf.g[2] * (x0 + x4);
            v = (float)(RD(i) + acc);
        }
        __builtin_amdgcn_wave_barrier();
        if (on) RD(i) = v;
        __builtin_amdgcn_wave_barrier();
    }
}

// celt_postfilter_apply (dopus.d:3326-3355) on ring positions [n0, n0 + len)
__device__ __forceinline__ void pf_apply(float *ring, int n0, int len, const PfState &pf, int l, bool lane_on)
{
    const bool go = lane_on && pf.g[0] != 0.0f && len > 0;
    if (!__any(go)) return;
    const int T = pf.period;
    // everything a step reads is at least T - 2 samples old: a lane takes samples l and l + 32 of a step when T allows
    int step = go ? max(min(T - 2, 64), 1) : 64;
    step = min(__shfl(step, 0), __shfl(step, 32));
    if (step > 32) {
        for (int i0 = 0; i0 < len; i0 += step) {
            float v[2] = { 0.0f, 0.0f };
            bool on[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + l + 32 * u;
                on[u] = go && l + 32 * u < step && i < len;
                if (on[u]) {
                    const float x0 = RD(i - T + 2), x1 = RD(i - T + 1), x2 = RD(i - T), x3 = RD(i - T - 1), x4 = RD(i - T - 2);
                    v[u] = RD(i) + (pf.g[0] * x2 + pf.g[1] * (x1 + x3) + pf.g[2] * (x0 + x4));
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (on[u]) RD(i0 + l + 32 * u) = v[u];
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    for (int i0 = 0; i0 < len; i0 += step) {
        const int i = i0 + l;
        float v = 0.0f;
        const bool on = go && l < step && i < len;
        if (on) {
            const float x0 = RD(i - T + 2), x1 = RD(i - T + 1), x2 = RD(i - T), x3 = RD(i - T - 1), x4 = RD(i - T - 2);
            v = RD(i) + (pf.g[0] * x2 + pf.g[1] * (x1 + x3) + pf.g[2] * (x0 + x4));
        }
        __builtin_amdgcn_wave_barrier();
        if (on) RD(i) = v;
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(64) void celt_postfilter_kernel(
    const uint64_t *__restrict__ rec_base, const afg_celt_frame *__restrict__ recs, float *__restrict__ out,
    float *__restrict__ states, uint32_t n_chan)
{
    __shared__ __attribute__((aligned(16))) float rings[2][kRing];
    __shared__ float win2[120];
    // a serial walk whose length sets the kernel's (and, beside other kernels on a second stream, the batch's) duration:
    // its instructions go first whenever they are ready
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x, h = lane >> 5, l = lane & 31;
    const uint32_t chan = blockIdx.x, pchan = chan ^ 1u;
    const uint64_t base = rec_base[chan], cnt = rec_base[chan + 1] - base;
    if (cnt == 0) return;

    // do all records of this sequence and its neighbour pair up?
    bool paired = false;
    uint64_t pbase = 0;
    if (pchan < n_chan) {
        pbase = rec_base[pchan];
        if (rec_base[pchan + 1] - pbase == cnt) {
            bool bad = false;
            for (uint64_t q = lane; q < cnt; q += 64) {
                const afg_celt_frame a = recs[base + q], b = recs[pbase + q];
                bad = bad || !((chan & 1u) ? celt_pair_ok(b, a) : celt_pair_ok(a, b));
            }
            paired = !__any(bad);
        }
    }
    if (paired && (chan & 1u)) return;
    const bool lane_on = paired || h == 0;
    const uint32_t my_chan = (paired && h == 1) ? pchan : chan;
    const uint64_t my_base = (paired && h == 1) ? pbase : base;
    float *st = states ? states + (size_t)my_chan * AFG_CELT_STATE_FLOATS : nullptr;
    float *ring = rings[h];

    PfState pf;
    pf.period = pf.period_old = 0;
    pf.g[0] = pf.g[1] = pf.g[2] = pf.g_old[0] = pf.g_old[1] = pf.g_old[2] = 0.0f;
    if (lane_on) {
        for (int i = l; i < 1024; i += 32) ring[i] = st ? st[i] : 0.0f;
        if (st) {
            pf.period = __float_as_int(st[2048]);
            pf.g[0] = st[2049]; pf.g[1] = st[2050]; pf.g[2] = st[2051];
            pf.period_old = __float_as_int(st[2052]);
            pf.g_old[0] = st[2053]; pf.g_old[1] = st[2054]; pf.g_old[2] = st[2055];
        }
    }
    int n0 = 1024;
    for (int i = lane; i < 120; i += 64) win2[i] = d_celt_window2[i];
    float wi[2], wj[2];                                      // block-0 window taps of this lane
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int k = min(l + 32 * u, 59);
        wi[u] = d_celt_window[k];
        wj[u] = d_celt_window[119 - k];
    }
    __builtin_amdgcn_wave_barrier();

    // frames are fetched one ahead into registers (15 x 64 positions cover the largest frame), records two ahead
    f32x2 nxt[15];
    auto fetch = [&](const afg_celt_frame &fr) {
        const int F = __shfl((int)fr.frame_size, 0);
        const uint64_t off = __shfl(fr.out_off, 0);
        if (paired) {
            const f32x2 *o = (const f32x2 *)(out + off);
#pragma unroll
            for (int i = 0; i < 15; i++) nxt[i] = __builtin_nontemporal_load(o + min(lane + 64 * i, F - 1));
        } else {
            const float *o = out + off;
            const size_t stride = __shfl(fr.out_stride, 0);
#pragma unroll
            for (int i = 0; i < 15; i++) nxt[i].x = __builtin_nontemporal_load(o + (size_t)min(lane + 64 * i, F - 1) * stride);
        }
    };
    afg_celt_frame fr = recs[my_base], fr_next = fr;
    if (cnt > 1) fr_next = recs[my_base + 1];
    fetch(fr);

    for (uint64_t q = 0; q < cnt; q++) {
        const int F = __shfl((int)fr.frame_size, 0);
        afg_celt_frame fr_next2 = recs[my_base + (q + 2 < cnt ? q + 2 : cnt - 1)];
        // the frame as kernel A left it: [0, 60) previous overlap, [60, F) this frame's iMDCT
#pragma unroll
        for (int i = 0; i < 15; i++) {
            const int p = lane + 64 * i;
            if (p < F) {
                rings[0][(n0 + p) & (kRing - 1)] = nxt[i].x;
                if (paired) rings[1][(n0 + p) & (kRing - 1)] = nxt[i].y;
            }
        }
        fetch(fr_next);                                      // (the last frame's again at the end)
        __builtin_amdgcn_wave_barrier();
        if (q == 0 && lane_on)                               // the overlap a fresh call starts from: state or silence
            for (int k = l; k < 60; k += 32) RD(k) = st ? st[1024 + k] : 0.0f;
        __builtin_amdgcn_wave_barrier();
        // vector_fmul_window of block 0 (dopus.d:3688, :230-243)
        if (lane_on) {
            float a[2], b[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int k = l + 32 * u;
                if (k < 60) {
                    const float s0 = RD(k), s1 = RD(119 - k);
                    a[u] = s0 * wj[u] - s1 * wi[u];
                    b[u] = s0 * wi[u] + s1 * wj[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int k = l + 32 * u;
                if (k < 60) { RD(k) = a[u]; RD(119 - k) = b[u]; }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // celt_postfilter, dopus.d:3357-3378
        pf_transition(ring, win2, n0, pf, l, lane_on);
        pf.period_old = pf.period;
        pf.g_old[0] = pf.g[0]; pf.g_old[1] = pf.g[1]; pf.g_old[2] = pf.g[2];
        pf.period = fr.pf_period_new;
        pf.g[0] = fr.pf_gains_new[0]; pf.g[1] = fr.pf_gains_new[1]; pf.g[2] = fr.pf_gains_new[2];
        if (F > 120) {
            pf_transition(ring, win2, n0 + 120, pf, l, lane_on);
            pf_apply(ring, n0 + 240, F - 240, pf, l, lane_on);
            pf.period_old = pf.period;
            pf.g_old[0] = pf.g[0]; pf.g_old[1] = pf.g[1]; pf.g_old[2] = pf.g[2];
        }

        // Make the prefetched frame resident *here*: loads and stores share one in-order counter on this hardware,
        // so a wait placed after the stores below would also wait for them to drain.
#pragma unroll
        for (int i = 0; i < 15; i++) asm volatile("" : "+v"(nxt[i].x), "+v"(nxt[i].y) : : "memory");
        settle_rec(fr_next2);
        // the post-filtered frame goes back; the de-emphasis recurrence runs over it in celt_deemph_kernel
        if (paired) {
            f32x2 *o = (f32x2 *)(out + __shfl(fr.out_off, 0));
#pragma unroll
            for (int i = 0; i < 15; i++) {
                const int p = lane + 64 * i;
                if (p < F) o[p] = f32x2{ rings[0][(n0 + p) & (kRing - 1)], rings[1][(n0 + p) & (kRing - 1)] };
            }
        } else {
            float *o = out + __shfl(fr.out_off, 0);
            const size_t stride = __shfl(fr.out_stride, 0);
            for (int p = lane; p < F; p += 64) o[(size_t)p * stride] = rings[0][(n0 + p) & (kRing - 1)];
        }
        __builtin_amdgcn_wave_barrier();
        n0 = (n0 + F) & (kRing - 1);
        fr = fr_next;
        fr_next = fr_next2;
    }

    if (st && lane_on) {
        for (int i = l; i < 1024; i += 32) st[i] = ring[(n0 - 1024 + i) & (kRing - 1)];
        for (int k = l; k < 60; k += 32) st[1024 + k] = st[kTailSlot + k];
        if (l == 0) {
            st[2048] = __int_as_float(pf.period);
            st[2049] = pf.g[0]; st[2050] = pf.g[1]; st[2051] = pf.g[2];
            st[2052] = __int_as_float(pf.period_old);
            st[2053] = pf.g_old[0]; st[2054] = pf.g_old[1]; st[2055] = pf.g_old[2];
        }
    }
}
#undef RD

// ---------------------------------------------------------------------------------------------------------------
// Fused path for batches with enough streams to fill the device: one wavefront walks a stereo stream (or a single
// channel sequence) frame by frame with CeltFrame.buf of both channels in LDS -- iMDCT (the code of kernel A) straight
// into buf + 1024 + 60, every overlap window, post-filter, interleaved store, memmove (dopus.d:3370) -- so the
// coefficients are read once and the post-filtered PCM written once.  The inputs of frame k+1 are fetched while
// frame k is filtered and are made resident just before frame k's stores (one in-order memory counter).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSWaves = 8;                                   // wavefronts per workgroup, sharing the tables
constexpr int kSLdsFloats = kTabFloatsMax + 240 + kSWaves * 2 * 2048;

__global__ __launch_bounds__(64 * kSWaves) void celt_stream_kernel(
    const uint64_t *__restrict__ rec_base, const afg_celt_frame *__restrict__ recs,
    const float *__restrict__ coeffs, float *__restrict__ out, float *__restrict__ states,
    const float *__restrict__ tables, CeltTables tb, uint32_t tab_floats, uint32_t n_chan)
{
    extern __shared__ __attribute__((aligned(16))) float slds[];
    float *ltab = slds, *lwin = slds + kTabFloatsMax, *win2 = lwin + 120;
    for (uint32_t i = threadIdx.x; i < tab_floats; i += 64 * kSWaves) ltab[i] = tables[i];
    if (threadIdx.x < 120) { lwin[threadIdx.x] = d_celt_window[threadIdx.x]; win2[threadIdx.x] = d_celt_window2[threadIdx.x]; }
    __syncthreads();

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, l = lane & 31;
    const uint32_t pair = blockIdx.x * kSWaves + wv;
    const uint32_t c0 = 2 * pair, c1 = c0 + 1;
    if (c0 >= n_chan) return;
    float *bufs = win2 + 120 + (size_t)wv * 2 * 2048;
    const uint64_t base0 = rec_base[c0], cnt0 = rec_base[c0 + 1] - base0;
    uint64_t base1 = 0, cnt1 = 0;
    if (c1 < n_chan) { base1 = rec_base[c1]; cnt1 = rec_base[c1 + 1] - base1; }

    bool paired = false;                                     // do all records of the two sequences pair up?
    if (cnt1 == cnt0 && cnt0 > 0) {
        bool bad = false;
        for (uint64_t q = lane; q < cnt0; q += 64) bad = bad || !celt_pair_ok(recs[base0 + q], recs[base1 + q]);
        paired = !__any(bad);
    }

    // walks one sequence (lanes 0..31) or both sequences of a pair (lane half = channel)
    auto walk = [&](bool both, uint32_t chan_lo, uint64_t base_lo, uint64_t cnt) {
        const bool act = both || h == 0;
        const uint32_t my_chan = (both && h) ? c1 : chan_lo;
        const uint64_t my_base = (both && h) ? base1 : base_lo;
        float *st = states ? states + (size_t)my_chan * AFG_CELT_STATE_FLOATS : nullptr;
        float *buf = bufs + (both ? h : 0) * 2048;
        cpx *z = (cpx *)(buf + 1024 + 60);
        const float *b0 = bufs, *b1 = bufs + 2048;

        PfState pf;
        pf.period = pf.period_old = 0;
        pf.g[0] = pf.g[1] = pf.g[2] = pf.g_old[0] = pf.g_old[1] = pf.g_old[2] = 0.0f;
        if (act) {
            for (int i = l; i < 2048; i += 32) buf[i] = st ? st[i] : 0.0f;
            if (st) {
                pf.period = __float_as_int(st[2048]);
                pf.g[0] = st[2049]; pf.g[1] = st[2050]; pf.g[2] = st[2051];
                pf.period_old = __float_as_int(st[2052]);
                pf.g_old[0] = st[2053]; pf.g_old[1] = st[2054]; pf.g_old[2] = st[2055];
            }
        }
        float wi[2], wj[2];                                  // block-0 window taps of this lane
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = min(l + 32 * u, 59);
            wi[u] = lwin[k];
            wj[u] = lwin[119 - k];
        }
        __builtin_amdgcn_wave_barrier();

        afg_celt_frame fr = recs[my_base], fr_next = fr;
        if (cnt > 1) fr_next = recs[my_base + 1];
        float xa[15], xb[15];
        {
            const Geo g0 = geo_of(fr);
            if (is_960(g0)) load_inputs(xa, xb, coeffs, fr, geo_960(), l);
            else load_inputs(xa, xb, coeffs, fr, g0, l);
        }

        for (uint64_t q = 0; q < cnt; q++) {
            const Geo g = geo_of(fr);
            const int F = g.F;
            afg_celt_frame fr_next2 = recs[my_base + (q + 2 < cnt ? q + 2 : cnt - 1)];
            // iMDCT and overlap-add, dopus.d:3684-3690
            if (is_960(g)) frame_fft(z, xa, xb, fr, geo_960(), ltab, lwin, tb, l, act);
            else frame_fft(z, xa, xb, fr, g, ltab, lwin, tb, l, act);
            {                                                // (the last frame's again at the end)
                const Geo gn = geo_of(fr_next);
                if (is_960(gn)) load_inputs(xa, xb, coeffs, fr_next, geo_960(), l);
                else load_inputs(xa, xb, coeffs, fr_next, gn, l);
            }
            if (is_960(g)) frame_rest(z, fr, geo_960(), ltab, lwin, tb, l, act);
            else frame_rest(z, fr, g, ltab, lwin, tb, l, act);
            if (act) {                                       // vector_fmul_window of block 0
                float *d = buf + 1024;
                float a[2], b[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int k = l + 32 * u;
                    if (k < 60) {
                        const float s0 = d[k], s1 = d[119 - k];
                        a[u] = s0 * wj[u] - s1 * wi[u];
                        b[u] = s0 * wi[u] + s1 * wj[u];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int k = l + 32 * u;
                    if (k < 60) { d[k] = a[u]; d[119 - k] = b[u]; }
                }
            }
            __builtin_amdgcn_wave_barrier();

            // celt_postfilter, dopus.d:3357-3378
            pf_transition(buf, win2, 1024, pf, l, act);
            pf.period_old = pf.period;
            pf.g_old[0] = pf.g[0]; pf.g_old[1] = pf.g[1]; pf.g_old[2] = pf.g[2];
            pf.period = fr.pf_period_new;
            pf.g[0] = fr.pf_gains_new[0]; pf.g[1] = fr.pf_gains_new[1]; pf.g[2] = fr.pf_gains_new[2];
            if (F > 120) {
                pf_transition(buf, win2, 1024 + 120, pf, l, act);
                pf_apply(buf, 1024 + 240, F - 240, pf, l, act);
                pf.period_old = pf.period;
                pf.g_old[0] = pf.g[0]; pf.g_old[1] = pf.g[1]; pf.g_old[2] = pf.g[2];
            }

            // make the prefetched inputs resident before the stores enter the queue
#pragma unroll
            for (int i = 0; i < 15; i++) asm volatile("" : "+v"(xa[i]), "+v"(xb[i]) : : "memory");
            settle_rec(fr_next2);
            // the post-filtered frame leaves; the de-emphasis recurrence runs over it in celt_deemph_kernel
            if (both && F == 960) {
                f32x2 *o = (f32x2 *)(out + __shfl(fr.out_off, 0));
                f32x2 v[15];
#pragma unroll
                for (int i = 0; i < 15; i++) v[i] = f32x2{ b0[1024 + lane + 64 * i], b1[1024 + lane + 64 * i] };
#pragma unroll
                for (int i = 0; i < 15; i++) o[lane + 64 * i] = v[i];
            } else if (both) {
                f32x2 *o = (f32x2 *)(out + __shfl(fr.out_off, 0));
                for (int p = lane; p < F; p += 64) o[p] = f32x2{ b0[1024 + p], b1[1024 + p] };
            } else {
                float *o = out + __shfl(fr.out_off, 0);
                const size_t stride = __shfl(fr.out_stride, 0);
                for (int p = lane; p < F; p += 64) o[(size_t)p * stride] = b0[1024 + p];
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
            fr = fr_next;
            fr_next = fr_next2;
        }

        if (st && act) {
            for (int i = l; i < 2048; i += 32) st[i] = buf[i];
            if (l == 0) {
                st[2048] = __int_as_float(pf.period);
                st[2049] = pf.g[0]; st[2050] = pf.g[1]; st[2051] = pf.g[2];
                st[2052] = __int_as_float(pf.period_old);
                st[2053] = pf.g_old[0]; st[2054] = pf.g_old[1]; st[2055] = pf.g_old[2];
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    if (paired) {
        walk(true, c0, base0, cnt0);
    } else {
        if (cnt0) walk(false, c0, base0, cnt0);
        if (cnt1) walk(false, c1, base1, cnt1);
    }
}

// De-emphasis and output scaling (dopus.d:3695-3701) over the planes the transform kernels wrote:
//   tmp = x[j] + m;  m = tmp * 0.85000610f;  out[j] = tmp / 32768
// a one-pole IIR across the whole channel sequence whose float rounding order cannot be re-associated, so the
// time axis is serial and the parallel axis is the channel sequence: one lane per sequence, SEQ sequences per
// wavefront.  Memory is touched in whole rows: a step takes GROUP samples of every sequence (GROUP divides every
// CELT frame size) as 16-byte loads along the interleaved rows, transposes them through LDS (de-interleaving stereo
// rows), runs the SEQ chains, and goes back the same way; kDeDepth steps are kept in flight in registers.
// SEQ x GROUP is about constant (the bytes one step moves), SEQ is chosen by the host from the number of sequences:
// 32 lanes of chains per wavefront when there are tens of thousands of sequences (few issue slots per sample: the
// pass runs at its memory rate), down to 2 when there are few, long ones -- the mixed corpus has ~1600 sequences
// of up to 1.4 M samples per wave: with 32 per wavefront 52 wavefronts crawled through 36 000 latency-bound steps
// each (100 ms); with 2 per wavefront 820 wavefronts each run their chain at ~10 cycles per sample behind loads
// issued four steps (8 us) ahead.  Layouts the row scheme does not cover (stride > 2, unaligned or ragged rows) take
// the per-lane strided path.
// Pipelining: the ring of D steps runs across frame boundaries.  A frame's record fields (out_off, stride, size) are
// fetched two frames ahead, and while the last D steps of a frame are chained the first D steps of the next frame are
// already being loaded (same slots) when that frame has the same geometry -- so neither the record fetch nor the
// first row fetch of a frame is ever waited for at full memory latency.  All prefetch loads are unconditional
// (address-selected), which lets the compiler count them instead of draining the queue at every step.
template <int STRIDE, int SEQ, int GROUP, int D>
__device__ __forceinline__ void deemph_rows(float *xs, float *__restrict__ out, bool have, uint64_t off, uint64_t off_next,
                                            int n, float &m, f32x4 (&ring)[D][(SEQ * GROUP / 4 + 63) / 64], bool primed,
                                            bool chain_next)
{
    constexpr int PITCH = GROUP + 4;                          // floats; rows stay 16-byte aligned
    constexpr int QUADS = SEQ * GROUP / 4;                    // float4 per step
    constexpr int LOADS = (QUADS + 63) / 64;                  // float4 per lane per step
    constexpr int F = GROUP / 4 * STRIDE;                     // float4 per row per step
    constexpr int GS = GROUP * STRIDE;                        // floats a row advances per step
    constexpr int CH = (GROUP / 4) % 10 == 0 ? 10 : GROUP / 4;   // float4 per chain chunk held in registers
    static_assert((GROUP / 4) % CH == 0, "chunking");
    const int lane = threadIdx.x;
    float *ptr[LOADS];                                        // where this lane's float4 of a step lives (stores)
    const float *lptr[LOADS], *lptr_n[LOADS];                 // load addresses: this frame / the next one (always valid memory)
    int lds_at[LOADS];
    bool valid[LOADS];
    const int lead0 = __ffsll((unsigned long long)__ballot(have)) - 1;
    const uint64_t safe = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(off >> 32), lead0) << 32) | (uint32_t)__shfl((int)(uint32_t)off, lead0);
#pragma unroll
    for (int i = 0; i < LOADS; i++) {
        const int idx = lane + 64 * i;
        const int row = idx / F, q = idx - row * F;
        const int lead = (row * STRIDE) & 63;                // first chain lane of the row
        const uint32_t lo = __shfl((uint32_t)off, lead), hi = __shfl((uint32_t)(off >> 32), lead);
        const uint32_t nlo = __shfl((uint32_t)off_next, lead), nhi = __shfl((uint32_t)(off_next >> 32), lead);
        valid[i] = idx < QUADS && __shfl((int)have, lead) != 0;
        ptr[i] = out + (((uint64_t)hi << 32) | lo) + 4 * q;
        lptr[i] = valid[i] ? ptr[i] : out + safe;
        lptr_n[i] = (valid[i] && chain_next) ? out + (((uint64_t)nhi << 32) | nlo) + 4 * q : lptr[i];
        lds_at[i] = STRIDE == 1 ? row * PITCH + 4 * q : (2 * row) * PITCH + 2 * q;
    }
    const int groups = n / GROUP;
    if (!primed) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int g = d < groups ? d : groups - 1;
#pragma unroll
            for (int i = 0; i < LOADS; i++) ring[d][i] = __builtin_nontemporal_load((const f32x4 *)(lptr[i] + (size_t)g * GS));
        }
    }
    auto step = [&](f32x4 (&b)[LOADS], int g) {
#pragma unroll
        for (int i = 0; i < LOADS; i++) {
            if (QUADS % 64 != 0 && lane + 64 * i >= QUADS) continue;
            if (STRIDE == 1) {
                *(f32x4 *)(xs + lds_at[i]) = b[i];
            } else {                                         // (L,R,L,R) -> two samples of each channel's row
                *(f32x2 *)(xs + lds_at[i]) = f32x2{ b[i].x, b[i].z };
                *(f32x2 *)(xs + lds_at[i] + PITCH) = f32x2{ b[i].y, b[i].w };
            }
        }
        {   // refill the slot: D steps ahead in this frame, else the next frame's step (same slot: groups % D == 0 there),
            // else a harmless re-read of this step (end of the sequence)
            const int gn = g + D;
            const bool here = gn < groups;
            const size_t at = (size_t)(here ? gn : (chain_next ? gn - groups : g)) * GS;
#pragma unroll
            for (int i = 0; i < LOADS; i++)
                b[i] = __builtin_nontemporal_load((const f32x4 *)(((here || !chain_next) ? lptr[i] : lptr_n[i]) + at));
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < SEQ && have) {
            f32x4 *row = (f32x4 *)(xs + lane * PITCH);
            // two register chunks, ping-pong: while one is chained (14 cycles per sample, measured: tools/ubench_chain.hip)
            // the LDS reads of the other are in flight; no register copies on the chain's path
            auto chain = [&](f32x4 (&c)[CH], int k0) {
#pragma unroll
                for (int k = 0; k < CH; k++) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float t = c[k][e] + m;                               // the chain: two dependent operations per
                        m = t * 0.85000610f;                                       // sample on SEQ lanes, nothing else
                        c[k][e] = t;
                    }
                    row[k0 + k] = c[k];
                }
            };
            auto fill = [&](f32x4 (&c)[CH], int k0) {
#pragma unroll
                for (int k = 0; k < CH; k++) c[k] = row[k0 + k];
            };
            constexpr int NCH = GROUP / 4 / CH;               // chunks per step: 1 (GROUP 40, 60), 3, 6 or 12
            f32x4 ca[CH], cb[CH];
            fill(ca, 0);
            if (NCH == 1) {
                chain(ca, 0);
            } else {
#pragma unroll 1
                for (int k0 = 0; k0 < GROUP / 4; k0 += 2 * CH) {
                    const bool two = k0 + CH < GROUP / 4;     // (an odd chunk count ends on `ca`)
                    if (two) fill(cb, k0 + CH);
                    chain(ca, k0);
                    if (k0 + 2 * CH < GROUP / 4) fill(ca, k0 + 2 * CH);
                    if (two) chain(cb, k0 + CH);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < LOADS; i++) {
            f32x4 o;
            if (QUADS % 64 != 0 && lane + 64 * i >= QUADS) continue;
            if (STRIDE == 1) {
                o = *(const f32x4 *)(xs + lds_at[i]);
            } else {
                const f32x2 l = *(const f32x2 *)(xs + lds_at[i]), r = *(const f32x2 *)(xs + lds_at[i] + PITCH);
                o = f32x4{ l.x, r.x, l.y, r.y };
            }
            o *= (1.0f / 32768.0f);                                                // tmp / 32768. (exact), on all 64 lanes
            if (valid[i]) *(f32x4 *)(ptr[i] + (size_t)g * GS) = o;
        }
        __builtin_amdgcn_wave_barrier();
    };
    for (int g = 0; g < groups; g += D) {
#pragma unroll
        for (int d = 0; d < D; d++)
            if (g + d < groups) step(ring[d], g + d);
    }
}

// The same pass as a pair of wavefronts (few, long sequences: the chain's latency is the whole run time).  In the
// single-wavefront form above a step is write-to-LDS, chain, read-from-LDS-and-store, one after the other: the chain lanes
// sit out most of it (measured: 35 cycles per sample against the chain's own 14).  Here wavefront 0 only moves (row
// loads D steps ahead, LDS writes of step g+1, LDS reads + scale + stores of step g-1) and wavefront 1 only chains
// (step g), on three LDS step buffers, one workgroup barrier per step: the chain never waits for a transpose.
template <int STRIDE, int SEQ, int GROUP, int D>
__device__ __forceinline__ void deemph_rows_duo(float *xs2, float *__restrict__ out, bool have, uint64_t off, uint64_t off_next,
                                                int n, float &m, f32x4 (&ring)[D][(SEQ * GROUP / 4 + 63) / 64], bool primed,
                                                bool chain_next, int role)
{
    constexpr int PITCH = GROUP + 4;                          // floats; rows stay 16-byte aligned
    constexpr int BUF = SEQ * PITCH;                          // floats per step buffer
    constexpr int QUADS = SEQ * GROUP / 4;                    // float4 per step
    constexpr int LOADS = (QUADS + 63) / 64;                  // float4 per lane per step
    constexpr int F = GROUP / 4 * STRIDE;                     // float4 per row per step
    constexpr int GS = GROUP * STRIDE;                        // floats a row advances per step
    constexpr int CH = (GROUP / 4) % 10 == 0 ? 10 : GROUP / 4;   // float4 per chain chunk held in registers
    static_assert((GROUP / 4) % CH == 0, "chunking");
    const int lane = threadIdx.x & 63;
    float *ptr[LOADS];
    const float *lptr[LOADS], *lptr_n[LOADS];
    int lds_at[LOADS];
    bool valid[LOADS];
    if (role == 0) {                                          // (the chainer needs none of the addresses)
    const int lead0 = __ffsll((unsigned long long)__ballot(have)) - 1;
    const uint64_t safe = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(off >> 32), lead0) << 32) | (uint32_t)__shfl((int)(uint32_t)off, lead0);
#pragma unroll
    for (int i = 0; i < LOADS; i++) {
        const int idx = lane + 64 * i;
        const int row = idx / F, q = idx - row * F;
        const int lead = (row * STRIDE) & 63;                // first chain lane of the row
        const uint32_t lo = __shfl((uint32_t)off, lead), hi = __shfl((uint32_t)(off >> 32), lead);
        const uint32_t nlo = __shfl((uint32_t)off_next, lead), nhi = __shfl((uint32_t)(off_next >> 32), lead);
        valid[i] = idx < QUADS && __shfl((int)have, lead) != 0;
        ptr[i] = out + (((uint64_t)hi << 32) | lo) + 4 * q;
        lptr[i] = valid[i] ? ptr[i] : out + safe;
        lptr_n[i] = (valid[i] && chain_next) ? out + (((uint64_t)nhi << 32) | nlo) + 4 * q : lptr[i];
        lds_at[i] = STRIDE == 1 ? row * PITCH + 4 * q : (2 * row) * PITCH + 2 * q;
    }
    } else {
#pragma unroll
        for (int i = 0; i < LOADS; i++) { ptr[i] = out; lptr[i] = lptr_n[i] = out; lds_at[i] = 0; valid[i] = false; }
    }
    const int groups = n / GROUP;
    if (role == 0 && !primed) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int g = d < groups ? d : groups - 1;
#pragma unroll
            for (int i = 0; i < LOADS; i++) ring[d][i] = __builtin_nontemporal_load((const f32x4 *)(lptr[i] + (size_t)g * GS));
        }
    }
    // Three LDS step buffers (step g in buffer g % 3): while the chainer works on step g the mover first fills step g+1's
    // (rows that were loaded D steps ago) and then empties step g-1's.  In that order, and with every load issued
    // unconditionally (the slot keeps its value by a select when there is nothing to fetch), the wait in front of the
    // LDS writes counts exactly the operations issued since those loads -- with the stores first, or a branch around
    // the loads, the compiler has to drain the stores of the same phase every time (one in-order memory counter).
    auto feed = [&](f32x4 (&b)[LOADS], int g, bool more) {
        float *xs = xs2 + (g % 3) * BUF;
        if (more) {
#pragma unroll
            for (int i = 0; i < LOADS; i++) {
                if (QUADS % 64 != 0 && lane + 64 * i >= QUADS) continue;
                if (STRIDE == 1) {
                    *(f32x4 *)(xs + lds_at[i]) = b[i];
                } else {
                    *(f32x2 *)(xs + lds_at[i]) = f32x2{ b[i].x, b[i].z };
                    *(f32x2 *)(xs + lds_at[i] + PITCH) = f32x2{ b[i].y, b[i].w };
                }
            }
        }
        const int gn = g + D;
        const bool here = more && gn < groups;
        const bool ahead = more && !here && chain_next;      // the next frame's first steps (same slots)
        const size_t at = (size_t)(here ? gn : (ahead ? gn - groups : (g < groups ? g : groups - 1))) * GS;   // else: a harmless re-read
#pragma unroll
        for (int i = 0; i < LOADS; i++) {
            const f32x4 t = __builtin_nontemporal_load((const f32x4 *)((ahead ? lptr_n[i] : lptr[i]) + at));
            b[i] = more ? t : b[i];
        }
    };
    // mover: step g's chained rows out of buffer g % 3, scaled, to memory (g = -1: nothing is stored)
    auto drain = [&](int g) {
        const float *xs = xs2 + ((g + 3) % 3) * BUF;
#pragma unroll
        for (int i = 0; i < LOADS; i++) {
            f32x4 o;
            if (QUADS % 64 != 0 && lane + 64 * i >= QUADS) continue;
            if (STRIDE == 1) {
                o = *(const f32x4 *)(xs + lds_at[i]);
            } else {
                const f32x2 l = *(const f32x2 *)(xs + lds_at[i]), r = *(const f32x2 *)(xs + lds_at[i] + PITCH);
                o = f32x4{ l.x, r.x, l.y, r.y };
            }
            o *= (1.0f / 32768.0f);                                                // tmp / 32768. (exact)
            if (valid[i] && g >= 0) *(f32x4 *)(ptr[i] + (size_t)(g >= 0 ? g : 0) * GS) = o;
        }
    };
    // chainer: step g in buffer g % 3
    auto chain_step = [&](int g) {
        if (lane < SEQ && have) {
            f32x4 *row = (f32x4 *)(xs2 + (g % 3) * BUF + lane * PITCH);
            auto chain = [&](f32x4 (&c)[CH], int k0) {
#pragma unroll
                for (int k = 0; k < CH; k++) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float t = c[k][e] + m;
                        m = t * 0.85000610f;
                        c[k][e] = t;
                    }
                    row[k0 + k] = c[k];
                }
            };
            auto fill = [&](f32x4 (&c)[CH], int k0) {
#pragma unroll
                for (int k = 0; k < CH; k++) c[k] = row[k0 + k];
            };
            constexpr int NCH = GROUP / 4 / CH;
            f32x4 ca[CH], cb[CH];
            fill(ca, 0);
            if (NCH == 1) {
                chain(ca, 0);
            } else {
#pragma unroll 1
                for (int k0 = 0; k0 < GROUP / 4; k0 += 2 * CH) {
                    const bool two = k0 + CH < GROUP / 4;
                    if (two) fill(cb, k0 + CH);
                    chain(ca, k0);
                    if (k0 + 2 * CH < GROUP / 4) fill(ca, k0 + 2 * CH);
                    if (two) chain(cb, k0 + CH);
                }
            }
        }
    };
    // The ring slot of step s is s % D (a frame that is chained to the next one has groups % D == 0, so the next frame's
    // step 0 finds its rows in slot 0 again).
    if (role == 0) feed(ring[0], 0, true);
    __syncthreads();
    for (int g = 0; g < groups; g += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int gg = g + d;
            if (gg < groups) {
                if (role == 0) {
                    feed(ring[(d + 1) % D], gg + 1, gg + 1 < groups);
                    drain(gg - 1);
                } else {
                    chain_step(gg);
                }
                __syncthreads();
            }
        }
    }
    if (role == 0) drain(groups - 1);
}

template <int SEQ, int GROUP>
__global__ __launch_bounds__(64) void celt_deemph_kernel(
    const uint64_t *__restrict__ rec_base, const afg_celt_frame *__restrict__ recs, float *__restrict__ out,
    float *__restrict__ states, uint32_t n_chan)
{
    constexpr int D = (960 / GROUP) >= 4 ? 4 : 2;            // steps in flight (divides the steps of a 20 ms frame)
    constexpr int LOADS = (SEQ * GROUP / 4 + 63) / 64;
    __shared__ __attribute__((aligned(16))) float xs[SEQ * (GROUP + 4)];
    __builtin_amdgcn_s_setprio(3);                            // a serial chain: its instructions go first whenever ready
    const int lane = threadIdx.x;
    const uint32_t chan = blockIdx.x * (uint32_t)SEQ + (uint32_t)lane;
    const bool mine = lane < SEQ && chan < n_chan;
    const uint32_t chan_c = chan < n_chan ? chan : n_chan - 1;
    float *st = (states && mine) ? states + (size_t)chan * AFG_CELT_STATE_FLOATS : nullptr;
    float m = st ? st[2056] : 0.0f;
    const uint64_t total = rec_base[n_chan];                  // records in the batch
    if (total == 0) return;
    uint64_t r = rec_base[chan_c];
    const uint64_t r_end = mine ? rec_base[chan_c + 1] : r;
    struct Rec { uint32_t n, stride; uint64_t off; };
    static_assert(offsetof(afg_celt_frame, out_off) == 8 && offsetof(afg_celt_frame, out_stride) == 16 &&
                  offsetof(afg_celt_frame, frame_size) == 20, "record layout");
    auto fetch = [&](uint64_t idx) -> Rec {                   // always a valid record: the fields of one past the end are unused
        const uint64_t *p = (const uint64_t *)(recs + (idx < total ? idx : total - 1));
        const uint64_t a = p[1], b = p[2];
        return Rec{ (uint32_t)(b >> 32) & 0xffffu, (uint32_t)b, a };
    };
    // can a step of the SEQ sequences be walked as rows?  Geometry of the first lane that still has a record
    // (sequences of a wavefront may end at different frames: the finished ones sit the step out)
    auto rows_ok = [&](const Rec &rc, bool hv, int &n0, int &s0) -> bool {
        const unsigned long long bal = __ballot(hv);
        n0 = s0 = 0;
        if (!bal) return false;
        const int first = __ffsll(bal) - 1;
        n0 = __shfl((int)rc.n, first);
        s0 = __shfl((int)rc.stride, first);
        bool bad = false;
        const bool p_have = __shfl_xor((int)hv, 1) != 0;
        const uint32_t plo = __shfl_xor((uint32_t)rc.off, 1), phi = __shfl_xor((uint32_t)(rc.off >> 32), 1);
        const uint64_t p_off = ((uint64_t)phi << 32) | plo;
        if (lane < SEQ) {
            if (hv) bad = (int)rc.n != n0 || (int)rc.stride != s0;
            if (s0 == 2) {
                if (hv != p_have) bad = true;
                else if (hv) bad = bad || ((lane & 1) ? rc.off != p_off + 1 : (rc.off & 3) != 0);
            } else if (hv) {
                bad = bad || (rc.off & 3) != 0;
            }
        }
        return (s0 == 1 || s0 == 2) && n0 > 0 && n0 % GROUP == 0 && !__any(bad);
    };
    Rec cur = fetch(r), nxt = fetch(r + 1);
    bool have = r < r_end;
    bool primed = false;
    f32x4 ring[D][LOADS];
    while (__any(have)) {
        const Rec nn = fetch(r + 2);                          // lands while this frame is chained
        const bool have_n = r + 1 < r_end;
        int n0, s0, n1, s1;
        const bool rows = rows_ok(cur, have, n0, s0);
        const bool rows_n = rows_ok(nxt, have_n, n1, s1);
        const bool chain_next = rows && rows_n && n1 == n0 && s1 == s0 && __ballot(have_n) == __ballot(have) && (n0 / GROUP) % D == 0;
        if (rows) {
            if (s0 == 2) deemph_rows<2, SEQ, GROUP, D>(xs, out, have, cur.off, nxt.off, n0, m, ring, primed, chain_next);
            else deemph_rows<1, SEQ, GROUP, D>(xs, out, have, cur.off, nxt.off, n0, m, ring, primed, chain_next);
            primed = chain_next;
        } else {
            primed = false;
            if (have) {
                const int n = (int)cur.n, stride = (int)cur.stride;
                float *o = out + cur.off;
                for (int j0 = 0; j0 < n; j0 += 8) {
                    float x[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) x[k] = j0 + k < n ? o[(size_t)(j0 + k) * stride] : 0.0f;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (j0 + k < n) {
                            const float t = x[k] + m;
                            m = t * 0.85000610f;
                            o[(size_t)(j0 + k) * stride] = t * (1.0f / 32768.0f);
                        }
                    }
                }
            }
        }
        cur = nxt;
        nxt = nn;
        r++;
        have = r < r_end;
    }
    if (st) st[2056] = m;
}

// two wavefronts per SEQ sequences: see deemph_rows_duo
template <int SEQ, int GROUP>
__global__ __launch_bounds__(128) void celt_deemph_duo_kernel(
    const uint64_t *__restrict__ rec_base, const afg_celt_frame *__restrict__ recs, float *__restrict__ out,
    float *__restrict__ states, uint32_t n_chan)
{
    constexpr int D = (960 / GROUP) >= 4 ? 4 : 2;            // steps in flight (divides the steps of a 20 ms frame)
    constexpr int LOADS = (SEQ * GROUP / 4 + 63) / 64;
    __shared__ __attribute__((aligned(16))) float xs[3 * SEQ * (GROUP + 4)];
    const int role = threadIdx.x >> 6;                        // 0 moves, 1 chains
    __builtin_amdgcn_s_setprio(3);                            // a serial chain: its instructions go first whenever ready
    const int lane = threadIdx.x & 63;
    const uint32_t chan = blockIdx.x * (uint32_t)SEQ + (uint32_t)lane;
    const bool mine = lane < SEQ && chan < n_chan;
    const uint32_t chan_c = chan < n_chan ? chan : n_chan - 1;
    float *st = (states && mine) ? states + (size_t)chan * AFG_CELT_STATE_FLOATS : nullptr;
    float m = st ? st[2056] : 0.0f;
    const uint64_t total = rec_base[n_chan];                  // records in the batch
    if (total == 0) return;
    uint64_t r = rec_base[chan_c];
    const uint64_t r_end = mine ? rec_base[chan_c + 1] : r;
    struct Rec { uint32_t n, stride; uint64_t off; };
    static_assert(offsetof(afg_celt_frame, out_off) == 8 && offsetof(afg_celt_frame, out_stride) == 16 &&
                  offsetof(afg_celt_frame, frame_size) == 20, "record layout");
    auto fetch = [&](uint64_t idx) -> Rec {                   // always a valid record: the fields of one past the end are unused
        const uint64_t *p = (const uint64_t *)(recs + (idx < total ? idx : total - 1));
        const uint64_t a = p[1], b = p[2];
        return Rec{ (uint32_t)(b >> 32) & 0xffffu, (uint32_t)b, a };
    };
    // can a step of the SEQ sequences be walked as rows?  Geometry of the first lane that still has a record
    // (sequences of a wavefront may end at different frames: the finished ones sit the step out)
    auto rows_ok = [&](const Rec &rc, bool hv, int &n0, int &s0) -> bool {
        const unsigned long long bal = __ballot(hv);
        n0 = s0 = 0;
        if (!bal) return false;
        const int first = __ffsll(bal) - 1;
        n0 = __shfl((int)rc.n, first);
        s0 = __shfl((int)rc.stride, first);
        bool bad = false;
        const bool p_have = __shfl_xor((int)hv, 1) != 0;
        const uint32_t plo = __shfl_xor((uint32_t)rc.off, 1), phi = __shfl_xor((uint32_t)(rc.off >> 32), 1);
        const uint64_t p_off = ((uint64_t)phi << 32) | plo;
        if (lane < SEQ) {
            if (hv) bad = (int)rc.n != n0 || (int)rc.stride != s0;
            if (s0 == 2) {
                if (hv != p_have) bad = true;
                else if (hv) bad = bad || ((lane & 1) ? rc.off != p_off + 1 : (rc.off & 3) != 0);
            } else if (hv) {
                bad = bad || (rc.off & 3) != 0;
            }
        }
        return (s0 == 1 || s0 == 2) && n0 > 0 && n0 % GROUP == 0 && !__any(bad);
    };
    Rec cur = fetch(r), nxt = fetch(r + 1);
    bool have = r < r_end;
    bool primed = false;
    f32x4 ring[D][LOADS];
    while (__any(have)) {
        const Rec nn = fetch(r + 2);                          // lands while this frame is chained
        const bool have_n = r + 1 < r_end;
        int n0, s0, n1, s1;
        const bool rows = rows_ok(cur, have, n0, s0);
        const bool rows_n = rows_ok(nxt, have_n, n1, s1);
        const bool chain_next = rows && rows_n && n1 == n0 && s1 == s0 && __ballot(have_n) == __ballot(have) && (n0 / GROUP) % D == 0;
        if (rows) {
            if (s0 == 2) deemph_rows_duo<2, SEQ, GROUP, D>(xs, out, have, cur.off, nxt.off, n0, m, ring, primed, chain_next, role);
            else deemph_rows_duo<1, SEQ, GROUP, D>(xs, out, have, cur.off, nxt.off, n0, m, ring, primed, chain_next, role);
            primed = chain_next;
        } else {
            primed = false;
            if (have && role == 1) {
                const int n = (int)cur.n, stride = (int)cur.stride;
                float *o = out + cur.off;
                for (int j0 = 0; j0 < n; j0 += 8) {
                    float x[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) x[k] = j0 + k < n ? o[(size_t)(j0 + k) * stride] : 0.0f;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (j0 + k < n) {
                            const float t = x[k] + m;
                            m = t * 0.85000610f;
                            o[(size_t)(j0 + k) * stride] = t * (1.0f / 32768.0f);
                        }
                    }
                }
            }
        }
        cur = nxt;
        nxt = nn;
        r++;
        have = r < r_end;
    }
    if (st && role == 1) st[2056] = m;
}

// Sequences per wavefront from the number of sequences: the most chain lanes per wavefront that still leaves about a
// wavefront per two SIMDs (512 wavefronts; measured: 16384 sequences run at the pass's memory rate with 32 per
// wavefront).  afg_dev_option("celt_de_seq") overrides (tests run every instantiation).
int deemph_seq_for(uint32_t n_chan)
{
    {
        const long v = afg::dev_option(afg::kDevCeltDeSeq);
        if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32) return (int)v;
    }
    for (int seq = 32; seq > 2; seq >>= 1)
        if (n_chan / (uint32_t)seq >= 512u) return seq;
    return 2;
}

void launch_deemph(const uint64_t *d_rec_base, const afg_celt_frame *d_recs, float *d_out, float *d_states, uint32_t n_chan,
                   hipStream_t stream)
{
    int seq = deemph_seq_for(n_chan);
    // Few, long sequences (fewer than 16384: not enough for 512 wavefronts of 32 chains, the form that runs at memory rate): the
    // pass is as long as its longest chain, so the chain gets a wavefront of its own (celt_deemph_duo_kernel).
    // afg_dev_option("celt_de_duo", 0 / 1) overrides (tests run both forms of every instantiation).
    bool duo = seq < 32;
    if (afg::dev_option(afg::kDevCeltDeDuo) >= 0) duo = afg::dev_option(afg::kDevCeltDeDuo) != 0;
    if (duo) {
        if (afg::dev_option(afg::kDevCeltDeSeq) < 0) seq = 8; // 120-sample steps: every CELT frame size walks as rows
        if (seq > 8) seq = 8;
        const dim3 grid((n_chan + (uint32_t)seq - 1) / (uint32_t)seq), block(128);
        switch (seq) {
        case 8:  hipLaunchKernelGGL((celt_deemph_duo_kernel<8, 120>), grid, block, 0, stream, d_rec_base, d_recs, d_out, d_states, n_chan); break;
        case 4:  hipLaunchKernelGGL((celt_deemph_duo_kernel<4, 240>), grid, block, 0, stream, d_rec_base, d_recs, d_out, d_states, n_chan); break;
        default: hipLaunchKernelGGL((celt_deemph_duo_kernel<2, 480>), grid, block, 0, stream, d_rec_base, d_recs, d_out, d_states, n_chan); break;
        }
        return;
    }
    const dim3 grid((n_chan + (uint32_t)seq - 1) / (uint32_t)seq), block(64);
    switch (seq) {
    case 32: hipLaunchKernelGGL((celt_deemph_kernel<32, 40>), grid, block, 0, stream, d_rec_base, d_recs, d_out, d_states, n_chan); break;
    case 16: hipLaunchKernelGGL((celt_deemph_kernel<16, 60>), grid, block, 0, stream, d_rec_base, d_recs, d_out, d_states, n_chan); break;
    case 8:  hipLaunchKernelGGL((celt_deemph_kernel<8, 120>), grid, block, 0, stream, d_rec_base, d_recs, d_out, d_states, n_chan); break;
    case 4:  hipLaunchKernelGGL((celt_deemph_kernel<4, 240>), grid, block, 0, stream, d_rec_base, d_recs, d_out, d_states, n_chan); break;
    default: hipLaunchKernelGGL((celt_deemph_kernel<2, 480>), grid, block, 0, stream, d_rec_base, d_recs, d_out, d_states, n_chan); break;
    }
}

// ---- host: tables exactly as ff_imdct15_init (dopus.d:1489-1499), in x87 long double like D's real ----
std::mutex g_mu;
float *g_tables[AFG_MAX_DEVICES] = {};
CeltTables g_tb;
uint32_t g_tab_floats = 0;
bool g_tb_ready = false;

// afg_dev_option("celt_path", 1 stream | 2 split) overrides the choice (tests exercise both paths on small batches)
bool use_stream_path(uint32_t pairs)
{
    const long e = afg::dev_option(afg::kDevCeltPath);       // 1 stream, 2 split, 3 walk
    if (e == 1) return true;
    if (e == 2) return false;
    // The stream walk runs one wavefront per stereo stream for the stream's whole length (~30-80 us per frame of
    // wavefront latency: its throughput comes from thousands of resident wavefronts).  Below about two wavefronts per
    // SIMD the record-parallel transform + per-sequence filter passes finish sooner.
    return pairs >= 2048;
}

int ensure_tables(const float **d_tables, CeltTables *tb)
{
    int dev = 0;
    if (int rc = afg::device_slot(&dev, "afg_celt_transform_hip")) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_tb_ready || !g_tables[dev]) {
        const long double pi = 3.14159265358979323846264338327950288L;      // std.math.PI
        std::vector<float> t;
        CeltTables c;
        for (int N = 3; N <= 6; N++) {
            const int len2 = 15 << N, len = 2 * len2, len4 = len2 / 2;
            c.twiddle[N - 3] = (uint32_t)t.size();
            for (int i = 0; i < len4; i++) {
                t.push_back((float)cosl(2 * pi * (i + 0.125 + len4) / len));
                t.push_back((float)sinl(2 * pi * (i + 0.125 + len4) / len));
            }
        }
        for (int i = 0; i < 6; i++) {
            const int NN = 15 << i;
            c.exptab[i] = (uint32_t)t.size();
            for (int j = 0; j < NN; j++) {
                t.push_back((float)cosl(2 * pi * j / NN));
                t.push_back((float)sinl(2 * pi * j / NN));
            }
            if (i == 0)
                for (int j = 15; j < 19; j++) {                             // wrap around to simplify fft15
                    t.push_back(t[c.exptab[0] + 2 * (j - 15)]);
                    t.push_back(t[c.exptab[0] + 2 * (j - 15) + 1]);
                }
        }
        for (int N = 3; N <= 6; N++)                                        // the device code computes these offsets
            if (c.twiddle[N - 3] != 120u * ((1u << (N - 3)) - 1)) return AFG_ERR_INVALID;
        for (int i = 0; i < 6; i++)
            if (c.exptab[i] != (i == 0 ? 1800u : 1838u + 30u * ((1u << i) - 2))) return AFG_ERR_INVALID;
        if (t.size() > (size_t)kTabFloatsMax) return AFG_ERR_INVALID;
        const size_t n_tab = t.size();
        t.resize(kCeltWinAt, 0.0f);                                         // the windows ride behind the tables (celt_walk.hip)
        t.insert(t.end(), k_celt_window, k_celt_window + 120);
        t.insert(t.end(), k_celt_window2, k_celt_window2 + 120);
        float *d = nullptr;
        AFG_HIP_CHECK(hipMalloc(&d, t.size() * sizeof(float)));
        AFG_HIP_CHECK(hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
        AFG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_celt_window), k_celt_window, sizeof(k_celt_window)));
        AFG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_celt_window2), k_celt_window2, sizeof(k_celt_window2)));
        g_tables[dev] = d;
        g_tb = c;
        g_tab_floats = (uint32_t)n_tab;
        g_tb_ready = true;
    }
    *d_tables = g_tables[dev];
    *tb = g_tb;
    return AFG_OK;
}

}  // namespace

int afg::celt_tables_for_device(const float **d_tables, void *tb_out, uint32_t *tab_floats)
{
    CeltTables tb;
    if (int rc = ensure_tables(d_tables, &tb)) return rc;
    memcpy(tb_out, &tb, sizeof(tb));
    *tab_floats = g_tab_floats;
    return AFG_OK;
}

// ---- OpusFile.readFrame's conversion (dopus.d:7923-7926, :8098-8105) and stream.d:480 --------------------------------
// Float2IntScaled: temp.f = x + (1.5f*(1<<8) + 0.5f/(1<<15)); d = temp.i - (((150-15)<<23) + (1<<22)); saturate.
// The constant is float-typed: 384 + 2^-16 is exactly half an ulp above 384 and rounds (to even) to 384.0f, so the
// float addition leaves round-to-nearest-even of x * 32768 in the mantissa -- one v_add_f32 and an integer subtract.
// GAIN: opus_decode_packet's vector_fmul_scalar by the header / R128 gain first (dopus.d:6688-6691), a float multiply of its own.
// Four samples per lane (the element-wise pass is pure HBM traffic); the tail of a count that is not a multiple of four
// goes through the scalar path of the last lane.
template <bool GAIN>
__global__ __launch_bounds__(256) void opus_output_kernel(const float *__restrict__ in, int16_t *__restrict__ out_i16,
                                                          float *__restrict__ out_f32, uint64_t n, float gain)
{
    const uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    auto one = [&](float x) -> int {
        if (GAIN) x = x * gain;
        const float t = x + 384.0f;
        // D's int arithmetic wraps (for -768 < x < -384 the difference passes INT_MIN and the sample saturates to +32767):
        // unsigned arithmetic keeps that behaviour defined here
        int d = (int)(__float_as_uint(t) - (unsigned)(((150 - 15) << 23) + (1 << 22)));
        if ((unsigned)d + 32768u > 65535u) d = d < 0 ? -32768 : 32767;
        return d;
    };
    const bool aligned = ((((uintptr_t)in) | ((uintptr_t)out_f32)) & 15) == 0 && (((uintptr_t)out_i16) & 7) == 0;
    if (i + 4 <= n && aligned) {
        const float4 v = *reinterpret_cast<const float4 *>(in + i);
        const int d0 = one(v.x), d1 = one(v.y), d2 = one(v.z), d3 = one(v.w);
        if (out_i16) {
            short4 o;
            o.x = (short)d0; o.y = (short)d1; o.z = (short)d2; o.w = (short)d3;
            *reinterpret_cast<short4 *>(out_i16 + i) = o;
        }
        if (out_f32) {
            float4 o;
            o.x = (float)(int16_t)d0 / 32767.0f;                            // stream.d:480
            o.y = (float)(int16_t)d1 / 32767.0f;
            o.z = (float)(int16_t)d2 / 32767.0f;
            o.w = (float)(int16_t)d3 / 32767.0f;
            *reinterpret_cast<float4 *>(out_f32 + i) = o;
        }
        return;
    }
    for (uint64_t k = i; k < n && k < i + 4; k++) {
        const int d = one(in[k]);
        if (out_i16) out_i16[k] = (int16_t)d;
        if (out_f32) out_f32[k] = (float)(int16_t)d / 32767.0f;
    }
}

static int opus_output_launch(const char *who, uint64_t n_samples, const float *d_in, bool apply_gain, float gain, int16_t *d_out_i16,
                              float *d_out_f32, void *hip_stream)
{
    if (n_samples == 0) return AFG_OK;
    if (!d_in || (!d_out_i16 && !d_out_f32)) {
        afg::set_error("%s: NULL device pointer", who);
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    const uint64_t blocks = (n_samples + 1023) / 1024;
    if (blocks > 0x7fffffffull) {
        afg::set_error("%s: at most 2^41 samples per call", who);
        return AFG_ERR_INVALID;
    }
    if (apply_gain)
        hipLaunchKernelGGL(opus_output_kernel<true>, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)hip_stream, d_in, d_out_i16, d_out_f32,
                           n_samples, gain);
    else
        hipLaunchKernelGGL(opus_output_kernel<false>, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)hip_stream, d_in, d_out_i16, d_out_f32,
                           n_samples, 1.0f);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}

extern "C" int afg_opus_output_hip(uint64_t n_samples, const float *d_in, int16_t *d_out_i16, float *d_out_f32, void *hip_stream)
{
    return opus_output_launch("afg_opus_output_hip", n_samples, d_in, false, 1.0f, d_out_i16, d_out_f32, hip_stream);
}

extern "C" int afg_opus_output_gain_hip(uint64_t n_samples, const float *d_in, float gain, int16_t *d_out_i16, float *d_out_f32,
                                        void *hip_stream)
{
    return opus_output_launch("afg_opus_output_gain_hip", n_samples, d_in, true, gain, d_out_i16, d_out_f32, hip_stream);
}

static int celt_transform_impl(uint32_t n_chan, const uint64_t *d_rec_base, const afg_celt_frame *d_recs, const float *d_coeffs,
                               float *d_out, float *d_states, hipStream_t hip_stream, hipStream_t tail_stream);

extern "C" int afg_celt_transform_hip(uint32_t n_chan, const uint64_t *d_rec_base, const afg_celt_frame *d_recs,
                                      const float *d_coeffs, float *d_out, float *d_states, void *hip_stream)
{
    return celt_transform_impl(n_chan, d_rec_base, d_recs, d_coeffs, d_out, d_states, (hipStream_t)hip_stream, nullptr);
}

extern "C" int afg_celt_transform_streams_hip(uint32_t n_chan, const uint64_t *d_rec_base, const afg_celt_frame *d_recs,
                                              const float *d_coeffs, float *d_out, float *d_states, void *hip_stream,
                                              void *hip_tail_stream)
{
    if (!hip_tail_stream || hip_tail_stream == hip_stream)
        return celt_transform_impl(n_chan, d_rec_base, d_recs, d_coeffs, d_out, d_states, (hipStream_t)hip_stream, nullptr);
    return celt_transform_impl(n_chan, d_rec_base, d_recs, d_coeffs, d_out, d_states, (hipStream_t)hip_stream, (hipStream_t)hip_tail_stream);
}

static int celt_transform_impl(uint32_t n_chan, const uint64_t *d_rec_base, const afg_celt_frame *d_recs, const float *d_coeffs,
                               float *d_out, float *d_states, hipStream_t hip_stream, hipStream_t tail_stream)
{
    if (n_chan == 0) return AFG_OK;
    if (!d_rec_base || !d_recs || !d_coeffs || !d_out) {
        afg::set_error("afg_celt_transform_hip: NULL device pointer");
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    const float *d_tables = nullptr;
    CeltTables tb;
    if (int rc = ensure_tables(&d_tables, &tb)) return rc;
    // afg_dev_option("celt_path"): 1 stream | 2 split name one of the bit-exact paths, 3 the tolerance-mode walk; else the numeric mode decides
    const long path_opt = afg::dev_option(afg::kDevCeltPath);
    const bool forced_exact = path_opt == 1 || path_opt == 2;
    const bool forced_walk = path_opt == 3;
    if (forced_walk || (!forced_exact && afg::numeric_mode() == AFG_NUMERIC_TOLERANCE)) {
        if (tail_stream) {                                   // one kernel, all of it beside what the caller queues next
            hipEvent_t done = nullptr;
            AFG_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
            hipError_t e = hipEventRecord(done, hip_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(tail_stream, done, 0);
            (void)hipEventDestroy(done);
            if (e != hipSuccess) { afg::set_error("afg_celt_transform_streams_hip: %s", hipGetErrorString(e)); return AFG_ERR_HIP; }
            hip_stream = tail_stream;
        }
        return afg::celt_walk_launch(n_chan, d_rec_base, d_recs, d_coeffs, d_out, d_states, hip_stream);
    }
    // wavefronts per channel sequence in the record-parallel kernel: enough to fill the device whatever n_chan is
    const uint32_t pairs = (n_chan + 1) / 2;
    if (use_stream_path(pairs)) {
        if (tail_stream) {                                   // the whole walk is the serial part: all of it behind the event
            hipEvent_t done = nullptr;
            AFG_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
            hipError_t e = hipEventRecord(done, hip_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(tail_stream, done, 0);
            (void)hipEventDestroy(done);
            if (e != hipSuccess) { afg::set_error("afg_celt_transform_streams_hip: %s", hipGetErrorString(e)); return AFG_ERR_HIP; }
            hip_stream = tail_stream;
        }
        // enough streams to fill the device: one pass over the coefficients, one over the PCM
        static_assert(kSLdsFloats * sizeof(float) <= 160 * 1024, "LDS budget");
        AFG_HIP_CHECK(hipFuncSetAttribute((const void *)celt_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(kSLdsFloats * sizeof(float))));
        hipLaunchKernelGGL(celt_stream_kernel, dim3((pairs + kSWaves - 1) / kSWaves), dim3(64 * kSWaves),
                           kSLdsFloats * sizeof(float), (hipStream_t)hip_stream, d_rec_base, d_recs, d_coeffs, d_out,
                           d_states, d_tables, tb, g_tab_floats, n_chan);
        AFG_HIP_CHECK(hipGetLastError());
        launch_deemph(d_rec_base, d_recs, d_out, d_states, n_chan, (hipStream_t)hip_stream);
        AFG_HIP_CHECK(hipGetLastError());
        return AFG_OK;
    }
    // few streams: the iMDCT of all frames in parallel, then the sequential part.
    // wavefronts per channel pair in the record-parallel kernel: enough to fill the device whatever n_chan is
    const uint32_t per_pair = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, 32768 / pairs));
    const uint64_t waves = (uint64_t)pairs * per_pair;
    hipLaunchKernelGGL(celt_imdct_kernel, dim3((uint32_t)((waves + kAWaves - 1) / kAWaves)), dim3(64 * kAWaves), 0,
                       (hipStream_t)hip_stream, d_rec_base, d_recs, d_coeffs, d_out, d_states, d_tables, tb,
                       g_tab_floats, n_chan, per_pair);
    AFG_HIP_CHECK(hipGetLastError());
    // The per-sequence passes are serial chains that occupy a fraction of the device for their whole length: with a tail
    // stream they run there, behind an event, beside whatever the caller queues on hip_stream next.
    hipStream_t serial = (hipStream_t)hip_stream;
    if (tail_stream) {
        hipEvent_t done = nullptr;
        AFG_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        hipError_t e = hipEventRecord(done, (hipStream_t)hip_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(tail_stream, done, 0);
        (void)hipEventDestroy(done);                                        // released once the record has completed
        if (e != hipSuccess) { afg::set_error("afg_celt_transform_streams_hip: %s", hipGetErrorString(e)); return AFG_ERR_HIP; }
        serial = tail_stream;
    }
    hipLaunchKernelGGL(celt_postfilter_kernel, dim3(n_chan), dim3(64), 0, serial, d_rec_base, d_recs, d_out, d_states, n_chan);
    AFG_HIP_CHECK(hipGetLastError());
    launch_deemph(d_rec_base, d_recs, d_out, d_states, n_chan, serial);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}
