// celt_core.h -- device code shared by the CELT kernels (celt_transform.hip: the bit-exact paths; celt_walk.hip:
// the tolerance-mode segment walk): complex helpers, the 15-point base transform (dopus.d:1520-1581), the fused
// radix passes of fft_calc (:1596-1606), pre-/post-rotation and in-frame windows of imdct15_half (:1611-1637, :230-243).
// A translation unit's -ffp-contract setting decides whether these expression trees are kept as the reference
// writes them (off: celt_transform.hip) or may fuse multiply-adds (fast: celt_walk.hip).
#pragma once
#include "afg_common.h"

#ifndef AFG_CELT_NT
#define AFG_CELT_NT 1
#endif
#if AFG_CELT_NT
#define AFG_CELT_LD(p) __builtin_nontemporal_load(p)
#else
#define AFG_CELT_LD(p) (*(p))
#endif

namespace {

struct alignas(8) cpx { float re, im; };
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct CeltTables {          // float offsets into one device table
    uint32_t twiddle[4];     // twiddle_exptab of N = 3..6 (len4 complex each)
    uint32_t exptab[6];      // exptab[i], 15 * 2^i complex (exptab[0] padded to 19)
};


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

// ---------------------------------------------------------------------------------------------------------------
// Kernel A: iMDCT + the in-frame overlap windows, fully parallel over frame records.
//
// A 32-lane half of a wavefront owns one record (all its blocks at once); the two halves take the two channels of a
// stereo frame when the records pair up (same geometry, interleaved output), so the 32 15-point base transforms
// of a 960-sample frame fill the wavefront and the PCM leaves as whole interleaved rows.
//   * lane (block j, base transform a): loads its 15 strided inputs straight from HBM, pre-rotates them
//     (dopus.d:1619-1625) and runs fft15 (:1552-1581) in registers;
//   * the radix-2 levels of fft_calc (:1596-1606) run as fused radix-8 / radix-4 register passes over LDS;
//   * post-rotation (:1629-1636) in place, then vector_fmul_window (:230-243) of blocks j >= 1.
// A frame's iMDCT output covers frame positions [60, F + 60): [60, F) goes to the frame's own output slots, the
// last 60 values -- the overlap the NEXT frame's first window consumes (dst[0..60) of :3688) -- to slots [0, 60)
// of the next record of the sequence (or to the state blob after the last one).  The first window of every frame
// (block 0) is applied by kernel B, which walks the sequence in order.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kTailSlot = 1084;          // state words [1084, 1144): kernel A's hand-over of the last frame's overlap

// Forces the wait for a prefetched record to this point (see the note on the in-order memory counter below).
__device__ __forceinline__ void settle_rec(afg_celt_frame &f)
{
    uint32_t *w = (uint32_t *)&f;
    static_assert(sizeof(afg_celt_frame) == 48, "record layout");
#pragma unroll
    for (int i = 0; i < 12; i++) asm volatile("" : "+v"(w[i]) : : "memory");
}

__device__ __forceinline__ bool celt_pair_ok(const afg_celt_frame &even, const afg_celt_frame &odd)
{
    return even.out_stride == 2 && odd.out_stride == 2 && odd.out_off == even.out_off + 1 && (even.out_off & 1) == 0 &&
           even.frame_size == odd.frame_size && even.blocks == odd.blocks;
}

// Offsets into the table ensure_tables() builds (floats): twiddle_exptab of N = 3..6 back to back, then exptab[0]
// (19 entries), exptab[1], ... -- computed, so that no lookup goes through memory.
__device__ __forceinline__ int tw_off(int N) { return 120 * ((1 << (N - 3)) - 1); }
__device__ __forceinline__ int ex_off(int i) { return i == 0 ? 1800 : 1838 + 30 * ((1 << i) - 2); }

template <int G>
__device__ __forceinline__ void radix_pass(cpx *z, int l, int nb15, int L0, const float *__restrict__ tables,
                                           const CeltTables &tb, bool act)
{
    const int ngrp = nb15 >> G;                              // groups of 2^G base transforms (1..8)
    const int lg = 31 - __clz(ngrp);
    const int o = l & (ngrp - 1), rstep = 32 >> lg;
    const int low = o & ((1 << (L0 - 1)) - 1);
    const int base15 = ((o >> (L0 - 1)) << (L0 - 1 + G)) | low;
    for (int r = l >> lg; r < 15; r += rstep) {
        if (!act) continue;
        cpx v[1 << G];
#pragma unroll
        for (int q = 0; q < (1 << G); q++) v[q] = z[15 * (base15 | (q << (L0 - 1))) + r];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const cpx *ex = (const cpx *)(tables + ex_off(L0 + g));
#pragma unroll
            for (int q = 0; q < (1 << G); q++) {
                if (q & (1 << g)) continue;
                const int k = (low | ((q & ((1 << g) - 1)) << (L0 - 1))) * 15 + r;
                const cpx t = cmul(v[q | (1 << g)], ex[k]);
                const cpx l0 = v[q];
                v[q | (1 << g)].re = l0.re - t.re;
                v[q | (1 << g)].im = l0.im - t.im;
                v[q].re = l0.re + t.re;
                v[q].im = l0.im + t.im;
            }
        }
#pragma unroll
        for (int q = 0; q < (1 << G); q++) z[15 * (base15 | (q << (L0 - 1))) + r] = v[q];
    }
    __builtin_amdgcn_wave_barrier();
}

struct Geo {                             // geometry of a frame record
    int F, B, bs, N, fft_n, nblk, len4, len8, nb15;
};

__device__ __forceinline__ Geo geo_of(const afg_celt_frame &fr)
{
    Geo g;
    // wave-uniform: the halves of a paired wavefront hold records of equal geometry, an unpaired one uses lanes 0..31
    g.F = __builtin_amdgcn_readfirstlane((int)fr.frame_size); g.B = __builtin_amdgcn_readfirstlane((int)fr.blocks);
    g.bs = g.F / g.B;
    g.N = 31 - __clz(g.bs / 15); g.fft_n = g.N - 1; g.nblk = 1 << g.fft_n;
    g.len4 = g.bs >> 1; g.len8 = g.len4 >> 1; g.nb15 = g.F / 30;
    return g;
}

// The dominant record: a 20 ms frame in one block.  With this geometry as a compile-time constant the index
// arithmetic of the transform folds away (immediate load offsets, fixed trip counts).
__device__ __forceinline__ Geo geo_960() { return Geo{ 960, 1, 960, 6, 5, 32, 480, 240, 32 }; }
__device__ __forceinline__ bool is_960(const Geo &g) { return g.F == 960 && g.B == 1; }

// lane (block j, base transform a) fetches the 15 strided input pairs of its 15-point transform (dopus.d:1619-1625)
__device__ __forceinline__ void load_inputs(float (&xa)[15], float (&xb)[15], const float *__restrict__ coeffs,
                                            const afg_celt_frame &fr, const Geo &g, int l)
{
    // unconditional on purpose (idle lanes repeat a neighbour's addresses): straight-line loads let the compiler
    // count what is in flight instead of draining the queue at every merge point
    const float *src = coeffs + fr.coef_off;
    const int lc = l & (g.nb15 - 1);
    const int j = lc >> g.fft_n, an = lc & (g.nblk - 1);
    const int a = (int)(__brev((unsigned)an) >> (32 - g.fft_n));
#pragma unroll
    for (int k = 0; k < 15; k++) {
        const int i = a + g.nblk * k;
        xa[k] = AFG_CELT_LD(src + (size_t)(g.bs - 1 - 2 * i) * g.B + j);
        xb[k] = AFG_CELT_LD(src + (size_t)(2 * i) * g.B + j);
    }
}

// pre-rotation + fft15 in registers, then everything up to the in-frame windows in LDS: afterwards
// Y[i] = ((float *)z)[i] is frame position 60 + i of the frame's iMDCT output (blocks 1..B-1 windowed)
__device__ __forceinline__ void frame_fft(cpx *z, const float (&xa)[15], const float (&xb)[15], const afg_celt_frame &fr,
                                          const Geo &g, const float *ltab, const float *lwin, const CeltTables &tb,
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
        fft15(y, x, 1, (const cpx *)(ltab + ex_off(0)));
#pragma unroll
        for (int m = 0; m < 15; m++) z[15 * l + m] = y[m];
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void frame_rest(cpx *z, const afg_celt_frame &fr, const Geo &g, const float *ltab,
                                           const float *lwin, const CeltTables &tb, int l, bool act)
{
    const cpx *tw = (const cpx *)(ltab + tw_off(g.N));
    float *Y = (float *)z;
    // radix-2 levels 1..fft_n
    if (g.fft_n == 5) { radix_pass<3>(z, l, g.nb15, 1, ltab, tb, act); radix_pass<2>(z, l, g.nb15, 4, ltab, tb, act); }
    else if (g.fft_n == 4) { radix_pass<2>(z, l, g.nb15, 1, ltab, tb, act); radix_pass<2>(z, l, g.nb15, 3, ltab, tb, act); }
    else if (g.fft_n == 3) radix_pass<3>(z, l, g.nb15, 1, ltab, tb, act);
    else radix_pass<2>(z, l, g.nb15, 1, ltab, tb, act);
    // post-rotation, in place: block j's bs floats are frame positions 60 + j*bs + [0, bs)
    if (act) {
        const float scale = fr.imdct_scale;
        for (int t0 = l; t0 < g.F / 4; t0 += 128) {
            cpx za[4], zb[4], ta[4], tc[4];
            cpx *pa[4], *pb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = min(t0 + 32 * u, g.F / 4 - 1);
                const int j = t / g.len8, i = t - j * g.len8;
                cpx *zj = z + j * g.len4;
                pa[u] = zj + g.len8 - i - 1; pb[u] = zj + g.len8 + i;
                za[u] = *pa[u]; zb[u] = *pb[u];
                ta[u] = tw[g.len8 - i - 1]; tc[u] = tw[g.len8 + i];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (t0 + 32 * u >= g.F / 4) continue;
                const float r0 = za[u].im * ta[u].im - za[u].re * ta[u].re;
                const float i1 = za[u].im * ta[u].re + za[u].re * ta[u].im;
                const float r1 = zb[u].im * tc[u].im - zb[u].re * tc[u].re;
                const float i0 = zb[u].im * tc[u].re + zb[u].re * tc[u].im;
                *pa[u] = cpx{ scale * r0, scale * i0 };
                *pb[u] = cpx{ scale * r1, scale * i1 };
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // vector_fmul_window of blocks 1..B-1 (frame positions [j*bs, j*bs + 120) = Y[j*bs - 60 ...])
    if (act) {
        for (int t = l; t < (g.B - 1) * 60; t += 32) {
            const int j = 1 + t / 60, k = t - (j - 1) * 60;
            float *d = Y + j * g.bs - 60;
            const float s0 = d[k], s1 = d[119 - k];
            const float wi = lwin[k], wj = lwin[119 - k];
            d[k] = s0 * wj - s1 * wi;
            d[119 - k] = s0 * wi + s1 * wj;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

constexpr int kTabFloatsMax = 3712;      // >= the whole table (celt_tables_for_device): 3698 floats
constexpr int kCeltWinAt = 3712;         // the device table carries window[120] and window2[120] behind the transform tables

}  // namespace

namespace afg {
// The device copy of the transform tables (built once per device, celt_transform.hip): *d_tables holds tab_floats
// floats of twiddles / exptabs, then (at kCeltWinAt) ff_celt_window and ff_celt_window2.
int celt_tables_for_device(const float **d_tables, void *tb_out, uint32_t *tab_floats);
// Tolerance-mode transform stage (celt_walk.hip): one persistent kernel on `stream`.
int celt_walk_launch(uint32_t n_chan, const uint64_t *d_rec_base, const afg_celt_frame *d_recs, const float *d_coeffs,
                     float *d_out, float *d_states, hipStream_t stream);
}
