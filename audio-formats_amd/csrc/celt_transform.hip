// celt_transform.hip -- Opus/CELT transform stage on gfx950.
//
// Replaces the per-channel tail of ff_celt_decode_frame (reference dopus.d:3680-3702):
// imdct15_half (:1611-1637) + vector_fmul_window (:230-243) per block, celt_postfilter
// (:3281-3378) and de-emphasis / scaling (:3695-3701).  Float results follow the reference's
// expression trees (float, with the double sub-expressions of :3313-3318); tables are built on
// the host in x87 long double like the D source's `real` (:1489-1499).
//
// The comb post-filter feeds on its own output (lag >= 13 samples) and the de-emphasis is a
// one-pole IIR across the whole stream, so a channel sequence is processed in order by ONE
// wavefront with the 2048-sample CeltFrame.buf resident in LDS:
//   * pre-rotation, the 15*2^n FFT (one lane per 15-point base transform, then radix-2 levels
//     spread over the lanes) and the post-rotation run wave-parallel;
//   * the post-filter advances min(T-2, 64) samples per step (everything a step reads is
//     older than the step);
//   * the de-emphasis recurrence (rounding order cannot be re-associated) is a second kernel
//     over the output plane, one lane per channel sequence (celt_deemph_kernel).
// Parallelism therefore comes from the number of channel sequences in the batch.
#include "afg_common.h"
#include "celt_tables.h"

#ifndef AFG_CELT_ABL
#define AFG_CELT_ABL 0      // development ablations (tools/build_variant.sh): 1 no de-emphasis, 2 no post-filter, 3 no iMDCT
#endif

#include <cmath>
#include <mutex>
#include <vector>

namespace {

struct alignas(8) cpx { float re, im; };

struct CeltTables {          // float offsets into one device table
    uint32_t twiddle[4];     // twiddle_exptab of N = 3..6 (len4 complex each)
    uint32_t exptab[6];      // exptab[i], 15 * 2^i complex (exptab[0] padded to 19)
};

__device__ __constant__ float d_celt_window[120];
__device__ __constant__ float d_celt_window2[120];

__device__ __forceinline__ cpx cmul(cpx a, cpx b)                       // CMUL, dopus.d:1430-1434
{
    cpx c;
    c.re = a.re * b.re - a.im * b.im;
    c.im = a.re * b.im + a.im * b.re;
    return c;
}

__device__ __forceinline__ void cmul2(cpx &c, cpx &d, cpx a, cpx b)     // CMUL2, dopus.d:1438-1455
{
    const float rr = a.re * b.re, ri = a.re * b.im, ir = a.im * b.re, ii = a.im * b.im;
    c.re = rr - ii;
    c.im = ri + ir;
    d.re = rr + ii;
    d.im = -ri + ir;
}

__device__ __forceinline__ void fft5(cpx (&out)[5], const cpx *in, int stride)   // dopus.d:1520-1550
{
    const cpx f0 = { (float)0.30901699437494745, (float)0.95105651629515353 };
    const cpx f1 = { (float)-0.80901699437494734, (float)0.58778525229247325 };
    cpx z[4][4];
    const cpx i0 = in[0], i1 = in[stride], i2 = in[2 * stride], i3 = in[3 * stride], i4 = in[4 * stride];
    cmul2(z[0][0], z[0][3], i1, f0); cmul2(z[0][1], z[0][2], i1, f1);
    cmul2(z[1][0], z[1][3], i2, f0); cmul2(z[1][1], z[1][2], i2, f1);
    cmul2(z[2][0], z[2][3], i3, f0); cmul2(z[2][1], z[2][2], i3, f1);
    cmul2(z[3][0], z[3][3], i4, f0); cmul2(z[3][1], z[3][2], i4, f1);
    out[0].re = i0.re + i1.re + i2.re + i3.re + i4.re;
    out[0].im = i0.im + i1.im + i2.im + i3.im + i4.im;
    out[1].re = i0.re + z[0][0].re + z[1][1].re + z[2][2].re + z[3][3].re;
    out[1].im = i0.im + z[0][0].im + z[1][1].im + z[2][2].im + z[3][3].im;
    out[2].re = i0.re + z[0][1].re + z[1][3].re + z[2][0].re + z[3][2].re;
    out[2].im = i0.im + z[0][1].im + z[1][3].im + z[2][0].im + z[3][2].im;
    out[3].re = i0.re + z[0][2].re + z[1][0].re + z[2][3].re + z[3][1].re;
    out[3].im = i0.im + z[0][2].im + z[1][0].im + z[2][3].im + z[3][1].im;
    out[4].re = i0.re + z[0][3].re + z[1][2].re + z[2][1].re + z[3][0].re;
    out[4].im = i0.im + z[0][3].im + z[1][2].im + z[2][1].im + z[3][0].im;
}

// 15-point transform of in[0], in[stride], ... -> out[0..15); dopus.d:1552-1581
__device__ __forceinline__ void fft15(cpx *out, const cpx *in, int stride, const cpx *__restrict__ exptab)
{
    cpx t0[5], t1[5], t2[5];
    fft5(t0, in, stride * 3);
    fft5(t1, in + stride, stride * 3);
    fft5(t2, in + 2 * stride, stride * 3);
#pragma unroll
    for (int k = 0; k < 5; k++) {
        cpx a, b;
        a = cmul(t1[k], exptab[k]);
        b = cmul(t2[k], exptab[2 * k]);
        out[k].re = t0[k].re + a.re + b.re;
        out[k].im = t0[k].im + a.im + b.im;
        a = cmul(t1[k], exptab[k + 5]);
        b = cmul(t2[k], exptab[2 * (k + 5)]);
        out[k + 5].re = t0[k].re + a.re + b.re;
        out[k + 5].im = t0[k].im + a.im + b.im;
        a = cmul(t1[k], exptab[k + 10]);
        b = cmul(t2[k], exptab[2 * k + 5]);
        out[k + 10].re = t0[k].re + a.re + b.re;
        out[k + 10].im = t0[k].im + a.im + b.im;
    }
}

// imdct15_half (dopus.d:1611-1637) of one block; dst = buf + 1024 + j*blocksize + 60
__device__ void imdct_half_wave(float *dst, cpx *tmp, const float *__restrict__ src, int stride, float scale,
                                int N, const float *__restrict__ tables, const CeltTables &tb)
{
    const int lane = threadIdx.x;
    const int len2 = 15 << N, len4 = len2 >> 1, len8 = len4 >> 1, fft_n = N - 1;
    const cpx *tw = (const cpx *)(tables + tb.twiddle[N - 3]);
    cpx *z = (cpx *)dst;

    for (int i = lane; i < len4; i += 64) {                              // pre-rotation
        const cpx t = { src[(size_t)(len2 - 1 - 2 * i) * stride], src[(size_t)(2 * i) * stride] };
        tmp[i] = cmul(t, tw[i]);
    }
    __syncthreads();
    {                                                                    // base transforms: lane = 15-point block
        const int nblk = 1 << fft_n;
        if (lane < nblk) {
            const int a = (int)(__brev((unsigned)lane) >> (32 - fft_n)) & (nblk - 1);   // fft_n == 0 -> shift 32: masked
            fft15(z + 15 * lane, tmp + (fft_n ? a : 0), nblk, (const cpx *)(tables + tb.exptab[0]));
        }
    }
    __syncthreads();
    for (int L = 1; L <= fft_n; L++) {                                   // radix-2 levels, dopus.d:1596-1606
        const cpx *ex = (const cpx *)(tables + tb.exptab[L]);
        const int h = 15 << (L - 1);
        for (int idx = lane; idx < len4 / 2; idx += 64) {
            const int q15 = idx / 15, r15 = idx - 15 * q15;
            const int blk = q15 >> (L - 1);
            const int k = (q15 & ((1 << (L - 1)) - 1)) * 15 + r15;
            cpx *lo = z + blk * 2 * h + k, *hi = lo + h;
            const cpx t = cmul(*hi, ex[k]);
            const cpx l0 = *lo;
            hi->re = l0.re - t.re;
            hi->im = l0.im - t.im;
            lo->re = l0.re + t.re;
            lo->im = l0.im + t.im;
        }
        __syncthreads();
    }
    for (int i = lane; i < len8; i += 64) {                              // post-rotation
        const cpx za = z[len8 - i - 1], zb = z[len8 + i];
        const cpx ta = tw[len8 - i - 1], tb2 = tw[len8 + i];
        // CMUL3(r0, i1, za.im, za.re, ta.im, ta.re);  CMUL3(r1, i0, zb.im, zb.re, tb.im, tb.re)
        const float r0 = za.im * ta.im - za.re * ta.re;
        const float i1 = za.im * ta.re + za.re * ta.im;
        const float r1 = zb.im * tb2.im - zb.re * tb2.re;
        const float i0 = zb.im * tb2.re + zb.re * tb2.im;
        z[len8 - i - 1].re = scale * r0;
        z[len8 - i - 1].im = scale * i0;
        z[len8 + i].re = scale * r1;
        z[len8 + i].im = scale * i1;
    }
    __syncthreads();
}

struct PfState {
    int period, period_old;
    float g[3], g_old[3];
};

// celt_postfilter_apply_transition (dopus.d:3281-3324) on data[0..120)
__device__ void pf_transition(float *data, const PfState &pf)
{
    if (pf.g[0] == 0.0f && pf.g_old[0] == 0.0f) return;
    const int lane = threadIdx.x;
    const int T0 = pf.period_old, T1 = pf.period;
    // a filter whose gains are all zero contributes exact zeros whatever it reads (its period may
    // still be 0 on a fresh decoder): only live filters bound the parallel step
    const bool live0 = pf.g_old[0] != 0.0f || pf.g_old[1] != 0.0f || pf.g_old[2] != 0.0f;
    const bool live1 = pf.g[0] != 0.0f || pf.g[1] != 0.0f || pf.g[2] != 0.0f;
    int step = 64;
    if (live0) step = min(step, T0 - 2);
    if (live1) step = min(step, T1 - 2);
    step = max(step, 1);
    for (int i0 = 0; i0 < 120; i0 += step) {
        const int i = i0 + lane;
        float v = 0.0f;
        const bool on = lane < step && i < 120;
        if (on) {
            const float w = d_celt_window2[i];
            const float x0 = data[i - T1 + 2], x1 = data[i - T1 + 1], x2 = data[i - T1], x3 = data[i - T1 - 1],
                        x4 = data[i - T1 - 2];
            const double acc = (1.0 - w) * pf.g_old[0] * data[i - T0] +
                               (1.0 - w) * pf.g_old[1] * (data[i - T0 - 1] + data[i - T0 + 1]) +
                               (1.0 - w) * pf.g_old[2] * (data[i - T0 - 2] + data[i - T0 + 2]) +
                               w * pf.g[0] * x2 +
                               w * pf.g[1] * (x1 + x3) +
                               w * pf.g[2] * (x0 + x4);
            v = (float)(data[i] + acc);
        }
        __syncthreads();
        if (on) data[i] = v;
        __syncthreads();
    }
}

// celt_postfilter_apply (dopus.d:3326-3355)
__device__ void pf_apply(float *data, int len, const PfState &pf)
{
    if (pf.g[0] == 0.0f || len <= 0) return;
    const int lane = threadIdx.x;
    const int T = pf.period;
    const int step = max(min(T - 2, 64), 1);
    for (int i0 = 0; i0 < len; i0 += step) {
        const int i = i0 + lane;
        float v = 0.0f;
        const bool on = lane < step && i < len;
        if (on) {
            const float x0 = data[i - T + 2], x1 = data[i - T + 1], x2 = data[i - T], x3 = data[i - T - 1],
                        x4 = data[i - T - 2];
            v = data[i] + (pf.g[0] * x2 + pf.g[1] * (x1 + x3) + pf.g[2] * (x0 + x4));
        }
        __syncthreads();
        if (on) data[i] = v;
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void celt_transform_kernel(
    const uint64_t *__restrict__ rec_base, const afg_celt_frame *__restrict__ recs,
    const float *__restrict__ coeffs, float *__restrict__ out, float *__restrict__ states,
    const float *__restrict__ tables, CeltTables tb)
{
    __shared__ __attribute__((aligned(16))) float buf[2048];
    __shared__ __attribute__((aligned(16))) cpx tmp[480];
    const int lane = threadIdx.x;
    const uint32_t chan = blockIdx.x;
    float *st = states ? states + (size_t)chan * AFG_CELT_STATE_FLOATS : nullptr;

    PfState pf;
    pf.period = pf.period_old = 0;
    pf.g[0] = pf.g[1] = pf.g[2] = pf.g_old[0] = pf.g_old[1] = pf.g_old[2] = 0.0f;
    for (int i = lane; i < 2048; i += 64) buf[i] = st ? st[i] : 0.0f;
    if (st) {
        pf.period = __float_as_int(st[2048]);
        pf.g[0] = st[2049]; pf.g[1] = st[2050]; pf.g[2] = st[2051];
        pf.period_old = __float_as_int(st[2052]);
        pf.g_old[0] = st[2053]; pf.g_old[1] = st[2054]; pf.g_old[2] = st[2055];
    }
    __syncthreads();

    for (uint64_t r = rec_base[chan]; r < rec_base[chan + 1]; r++) {
        const afg_celt_frame fr = recs[r];
        const int frame_size = fr.frame_size, blocks = fr.blocks, blocksize = frame_size / blocks;
        const int N = 31 - __clz(blocksize / 15);
        const float *src = coeffs + fr.coef_off;

        // iMDCT and overlap-add, dopus.d:3684-3690
        for (int j = 0; j < blocks; j++) {
            float *dst = buf + 1024 + j * blocksize;
#if AFG_CELT_ABL == 3
            for (int i = lane; i < blocksize; i += 64) dst[60 + i] = src[i * blocks + j];
            __syncthreads();
            continue;
#endif
            imdct_half_wave(dst + 60, tmp, src + j, blocks, fr.imdct_scale, N, tables, tb);
            if (lane < 60) {                                            // vector_fmul_window, dopus.d:230-243
                const int k = lane;
                const float s0 = dst[k], s1 = dst[119 - k];
                const float wi = d_celt_window[k], wj = d_celt_window[119 - k];
                dst[k] = s0 * wj - s1 * wi;
                dst[119 - k] = s0 * wi + s1 * wj;
            }
            __syncthreads();
        }

        // celt_postfilter, dopus.d:3357-3378
        {
            const int len = frame_size;
#if AFG_CELT_ABL != 2
            pf_transition(buf + 1024, pf);
#endif
            pf.period_old = pf.period;
            pf.g_old[0] = pf.g[0]; pf.g_old[1] = pf.g[1]; pf.g_old[2] = pf.g[2];
            pf.period = fr.pf_period_new;
            pf.g[0] = fr.pf_gains_new[0]; pf.g[1] = fr.pf_gains_new[1]; pf.g[2] = fr.pf_gains_new[2];
            if (len > 120) {
#if AFG_CELT_ABL != 2
                pf_transition(buf + 1024 + 120, pf);
                pf_apply(buf + 1024 + 240, len - 240, pf);
#endif
                pf.period_old = pf.period;
                pf.g_old[0] = pf.g[0]; pf.g_old[1] = pf.g[1]; pf.g_old[2] = pf.g[2];
            }
            // memmove(buf, buf + len, 1084 floats): ascending 64-wide chunks never overlap (len >= 120)
            for (int i0 = 0; i0 < 1024 + 60; i0 += 64) {
                const int i = i0 + lane;
                float v = 0.0f;
                if (i < 1024 + 60) v = buf[i + len];
                __syncthreads();
                if (i < 1024 + 60) buf[i] = v;
                __syncthreads();
            }
        }

        // the post-filtered frame leaves for the output plane as is; the de-emphasis recurrence runs over it in
        // place in celt_deemph_kernel (one lane per channel sequence instead of 64 redundant lanes)
        {
            const float *x = buf + 1024 - frame_size;
            float *o = out + fr.out_off;
            for (int j = lane; j < frame_size; j += 64) o[(size_t)j * fr.out_stride] = x[j];
        }
        __syncthreads();
    }

    if (st) {
        __syncthreads();
        for (int i = lane; i < 2048; i += 64) st[i] = buf[i];
        if (lane == 0) {
            st[2048] = __int_as_float(pf.period);
            st[2049] = pf.g[0]; st[2050] = pf.g[1]; st[2051] = pf.g[2];
            st[2052] = __int_as_float(pf.period_old);
            st[2053] = pf.g_old[0]; st[2054] = pf.g_old[1]; st[2055] = pf.g_old[2];
        }
    }
}

// De-emphasis and output scaling (dopus.d:3695-3701) over the planes celt_transform_kernel wrote:
//   tmp = x[j] + m;  m = tmp * 0.85000610f;  out[j] = tmp / 32768
// a one-pole IIR across the whole channel sequence whose float rounding order cannot be re-associated, so the
// time axis is serial and the parallel axis is the channel sequence: one lane per sequence, 32 sequences per
// wavefront.  Memory is touched in whole rows: a step takes 40 samples of every sequence (40 divides every CELT
// frame size) as 16-byte loads along the interleaved rows, transposes them through LDS (de-interleaving stereo
// rows), runs the 32 chains, and goes back the same way; two steps are kept in flight in registers.  Layouts
// the row scheme does not cover (stride > 2, unaligned or ragged rows) take the per-lane strided path.
#ifndef AFG_CELT_DE_SEQ
#define AFG_CELT_DE_SEQ 32
#endif
#ifndef AFG_CELT_DE_DEPTH
#define AFG_CELT_DE_DEPTH 4
#endif
constexpr int kDeSeq = AFG_CELT_DE_SEQ;                      // channel sequences per wavefront
constexpr int kDeDepth = AFG_CELT_DE_DEPTH;                  // steps kept in flight
constexpr int kDeGroup = 40;
constexpr int kDePitch = kDeGroup + 4;                       // floats; rows stay 16-byte aligned
constexpr int kDeQuads = kDeSeq * kDeGroup / 4;              // float4 per step
constexpr int kDeLoads = (kDeQuads + 63) / 64;               // float4 per lane per step
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int STRIDE>
__device__ __forceinline__ void deemph_rows(float *xs, float *__restrict__ out, bool have, uint64_t off, int n, float &m)
{
    constexpr int F = 10 * STRIDE;                           // float4 per row per step
    const int lane = threadIdx.x;
    float *ptr[kDeLoads];
    int lds_at[kDeLoads];
    bool valid[kDeLoads];
#pragma unroll
    for (int i = 0; i < kDeLoads; i++) {
        const int idx = lane + 64 * i;
        const int row = (idx * (STRIDE == 1 ? 6554 : 3277)) >> 16, q = idx - row * F;     // idx / F, idx % F
        const int lead = row * STRIDE;                       // first chain lane of the row
        const uint32_t lo = __shfl((uint32_t)off, lead), hi = __shfl((uint32_t)(off >> 32), lead);
        valid[i] = idx < kDeQuads && __shfl((int)have, lead) != 0;
        ptr[i] = out + (((uint64_t)hi << 32) | lo) + 4 * q;
        lds_at[i] = STRIDE == 1 ? row * kDePitch + 4 * q : (2 * row) * kDePitch + 2 * q;
    }
    f32x4 ring[kDeDepth][kDeLoads];
    const int groups = n / kDeGroup;
    auto load = [&](f32x4 (&b)[kDeLoads], int g) {
#pragma unroll
        for (int i = 0; i < kDeLoads; i++)
            if (valid[i]) b[i] = __builtin_nontemporal_load((const f32x4 *)(ptr[i] + (size_t)g * (kDeGroup * STRIDE)));
    };
    auto step = [&](f32x4 (&b)[kDeLoads], int g, int g_next) {
#pragma unroll
        for (int i = 0; i < kDeLoads; i++) {
            if (kDeQuads % 64 != 0 && lane + 64 * i >= kDeQuads) continue;
            if (STRIDE == 1) {
                *(f32x4 *)(xs + lds_at[i]) = b[i];
            } else {                                         // (L,R,L,R) -> two samples of each channel's row
                *(f32x2 *)(xs + lds_at[i]) = f32x2{ b[i].x, b[i].z };
                *(f32x2 *)(xs + lds_at[i] + kDePitch) = f32x2{ b[i].y, b[i].w };
            }
        }
        if (g_next < groups) load(b, g_next);
        __builtin_amdgcn_wave_barrier();
        if (lane < kDeSeq && have) {
            f32x4 *row = (f32x4 *)(xs + lane * kDePitch);
            f32x4 v[kDeGroup / 4];
#pragma unroll
            for (int k = 0; k < kDeGroup / 4; k++) v[k] = row[k];
#pragma unroll
            for (int k = 0; k < kDeGroup / 4; k++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float t = v[k][e] + m;
                    m = t * 0.85000610f;
                    v[k][e] = t * (1.0f / 32768.0f);                                // tmp / 32768. (exact)
                }
                row[k] = v[k];
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < kDeLoads; i++) {
            f32x4 o;
            if (kDeQuads % 64 != 0 && lane + 64 * i >= kDeQuads) continue;
            if (STRIDE == 1) {
                o = *(const f32x4 *)(xs + lds_at[i]);
            } else {
                const f32x2 l = *(const f32x2 *)(xs + lds_at[i]), r = *(const f32x2 *)(xs + lds_at[i] + kDePitch);
                o = f32x4{ l.x, r.x, l.y, r.y };
            }
            if (valid[i]) *(f32x4 *)(ptr[i] + (size_t)g * (kDeGroup * STRIDE)) = o;
        }
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll
    for (int d = 0; d < kDeDepth; d++)
        if (d < groups) load(ring[d], d);
    for (int g = 0; g < groups; g += kDeDepth) {
#pragma unroll
        for (int d = 0; d < kDeDepth; d++)
            if (g + d < groups) step(ring[d], g + d, g + d + kDeDepth);
    }
}

__global__ __launch_bounds__(64) void celt_deemph_kernel(
    const uint64_t *__restrict__ rec_base, const afg_celt_frame *__restrict__ recs, float *__restrict__ out,
    float *__restrict__ states, uint32_t n_chan)
{
    __shared__ __attribute__((aligned(16))) float xs[kDeSeq * kDePitch];
    const int lane = threadIdx.x;
    const uint32_t chan = blockIdx.x * (uint32_t)kDeSeq + (uint32_t)lane;
    const bool mine = lane < kDeSeq && chan < n_chan;
    float *st = (states && mine) ? states + (size_t)chan * AFG_CELT_STATE_FLOATS : nullptr;
    float m = st ? st[2056] : 0.0f;
    uint64_t r = mine ? rec_base[chan] : 0;
    const uint64_t r_end = mine ? rec_base[chan + 1] : 0;
    while (__any(r < r_end)) {
        const bool have = r < r_end;
        int n = 0, stride = 0;
        uint64_t off = 0;
        if (have) {
            const afg_celt_frame *fr = recs + r;
            n = fr->frame_size; stride = (int)fr->out_stride; off = fr->out_off;
            r++;
        }
        // can this step of the 32 sequences be walked as rows?
        const int n0 = __builtin_amdgcn_readfirstlane(n), s0 = __builtin_amdgcn_readfirstlane(stride);
        bool bad = false;
        if (lane < kDeSeq) {
            if (have) bad = n != n0 || stride != s0;
            if (s0 == 2) {
                const bool p_have = __shfl_xor((int)have, 1) != 0;
                const uint32_t plo = __shfl_xor((uint32_t)off, 1), phi = __shfl_xor((uint32_t)(off >> 32), 1);
                const uint64_t p_off = ((uint64_t)phi << 32) | plo;
                if (have != p_have) bad = true;
                else if (have) bad = bad || ((lane & 1) ? off != p_off + 1 : (off & 3) != 0);
            } else if (have) {
                bad = bad || (off & 3) != 0;
            }
        }
        const bool rows = (s0 == 1 || s0 == 2) && n0 > 0 && n0 % kDeGroup == 0 && !__any(bad);
        if (rows) {
            if (s0 == 2) deemph_rows<2>(xs, out, have, off, n0, m);
            else deemph_rows<1>(xs, out, have, off, n0, m);
        } else if (have) {
            float *o = out + off;
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
    if (st) st[2056] = m;
}

// ---- host: tables exactly as ff_imdct15_init (dopus.d:1489-1499), in x87 long double like D's real ----
std::mutex g_mu;
float *g_tables[16] = {};
CeltTables g_tb;
bool g_tb_ready = false;

int ensure_tables(const float **d_tables, CeltTables *tb)
{
    int dev = 0;
    AFG_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    if (dev < 0 || dev >= 16) return AFG_ERR_INVALID;
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
        float *d = nullptr;
        AFG_HIP_CHECK(hipMalloc(&d, t.size() * sizeof(float)));
        AFG_HIP_CHECK(hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
        AFG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_celt_window), k_celt_window, sizeof(k_celt_window)));
        AFG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_celt_window2), k_celt_window2, sizeof(k_celt_window2)));
        g_tables[dev] = d;
        g_tb = c;
        g_tb_ready = true;
    }
    *d_tables = g_tables[dev];
    *tb = g_tb;
    return AFG_OK;
}

}  // namespace

extern "C" int afg_celt_transform_hip(uint32_t n_chan, const uint64_t *d_rec_base, const afg_celt_frame *d_recs,
                                      const float *d_coeffs, float *d_out, float *d_states, void *hip_stream)
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
    hipLaunchKernelGGL(celt_transform_kernel, dim3(n_chan), dim3(64), 0, (hipStream_t)hip_stream,
                       d_rec_base, d_recs, d_coeffs, d_out, d_states, d_tables, tb);
    AFG_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(celt_deemph_kernel, dim3((n_chan + kDeSeq - 1) / kDeSeq), dim3(64), 0, (hipStream_t)hip_stream,
                       d_rec_base, d_recs, d_out, d_states, n_chan);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}
