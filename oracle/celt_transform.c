/*
 * oracle/celt_transform.c -- CPU restatement of the Opus/CELT transform stage.
 * TEST INFRASTRUCTURE ONLY (see afg_oracle.h).  PARITY UNPINNED by reference vectors (the
 * reference has none); pinned by tests/test_oracle_celt.py (float64 direct IMDCT, TDAC, filter
 * definitions).
 *
 * Follows source/audioformats/dopus.d of the reference (a D translation of FFmpeg's Opus decoder):
 *   vector_fmul_window            :230-243       IMDCT15 tables          :1465-1517
 *   fft5 / fft15 / fft_calc       :1520-1609     imdct15_half            :1611-1637
 *   post-filter                   :3281-3378     frame tail + de-emphasis :3680-3702
 * Twiddles are evaluated in x87 `real` (std.math.PI is a real, :29) and rounded to float; the
 * transition filter computes in double (:3313-3318); the output scaling is a double divide (:3699).
 */
#include "afg_oracle.h"
#include "celt_tables.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float re, im; } cpx;

typedef struct {
    int fft_n, len2, len4;
    cpx *tmp;
    cpx *twiddle_exptab;
    cpx *exptab[6];
} imdct15_ctx;

static const long double k_pi_real = 3.14159265358979323846264338327950288L;    /* std.math.PI */

static int g_celt_table_mode = 0;
void afgo_celt_set_table_mode(int mode) { g_celt_table_mode = mode == 1; }

static int imdct15_init(imdct15_ctx *s, int N)                 /* :1465-1517 */
{
    int len2 = 15 * (1 << N), len = 2 * len2;
    memset(s, 0, sizeof(*s));
    s->fft_n = N - 1;
    s->len4 = len2 / 2;
    s->len2 = len2;
    s->tmp = (cpx *)malloc(sizeof(cpx) * (size_t)len);
    s->twiddle_exptab = (cpx *)malloc(sizeof(cpx) * (size_t)s->len4);
    if (!s->tmp || !s->twiddle_exptab) return -1;
    /* std.math.PI is a `real`: 80-bit x87 on the x86 targets the reference is built for (mode 0, the default and what the
     * product builds), plain double where real == double (mode 1, e.g. AArch64).  tests/test_oracle_numeric_readings.py
     * measures that both give the same PCM within the 1e-5 RMS tolerance. */
    for (int i = 0; i < s->len4; i++) {
        if (g_celt_table_mode == 0) {
            s->twiddle_exptab[i].re = (float)cosl(2 * k_pi_real * (i + 0.125 + s->len4) / len);
            s->twiddle_exptab[i].im = (float)sinl(2 * k_pi_real * (i + 0.125 + s->len4) / len);
        } else {
            s->twiddle_exptab[i].re = (float)cos(2 * (double)k_pi_real * (i + 0.125 + s->len4) / len);
            s->twiddle_exptab[i].im = (float)sin(2 * (double)k_pi_real * (i + 0.125 + s->len4) / len);
        }
    }
    for (int i = 0; i < 6; i++) {
        int NN = 15 * (1 << i);
        s->exptab[i] = (cpx *)malloc(sizeof(cpx) * (size_t)(NN > 19 ? NN : 19));
        if (!s->exptab[i]) return -1;
        for (int j = 0; j < NN; j++) {
            if (g_celt_table_mode == 0) {
                s->exptab[i][j].re = (float)cosl(2 * k_pi_real * j / NN);
                s->exptab[i][j].im = (float)sinl(2 * k_pi_real * j / NN);
            } else {
                s->exptab[i][j].re = (float)cos(2 * (double)k_pi_real * j / NN);
                s->exptab[i][j].im = (float)sin(2 * (double)k_pi_real * j / NN);
            }
        }
    }
    for (int j = 15; j < 19; j++) s->exptab[0][j] = s->exptab[0][j - 15];
    return 0;
}

static void imdct15_free(imdct15_ctx *s)
{
    for (int i = 0; i < 6; i++) free(s->exptab[i]);
    free(s->twiddle_exptab);
    free(s->tmp);
    memset(s, 0, sizeof(*s));
}

#define CMUL3(cre, cim, are, aim, bre, bim) do { \
        (cre) = (are) * (bre) - (aim) * (bim);   \
        (cim) = (are) * (bim) + (aim) * (bre);   \
    } while (0)
#define CMUL(c, a, b) CMUL3((c).re, (c).im, (a).re, (a).im, (b).re, (b).im)

/* c = a * b, d = a * conj(b)  (:1440-1455) */
static void cmul2(cpx *c, cpx *d, cpx a, cpx b)
{
    float rr = a.re * b.re, ri = a.re * b.im, ir = a.im * b.re, ii = a.im * b.im;
    c->re = rr - ii;
    c->im = ri + ir;
    d->re = rr + ii;
    d->im = -ri + ir;
}

static void fft5(cpx *out, const cpx *in, ptrdiff_t stride)     /* :1520-1550 */
{
    /* double literals converted to float, as the D initialiser does */
    static const cpx fact[2] = { { (float)0.30901699437494745, (float)0.95105651629515353 },
                                 { (float)-0.80901699437494734, (float)0.58778525229247325 } };
    cpx z[4][4];
    for (int r = 0; r < 4; r++) {
        cmul2(&z[r][0], &z[r][3], in[(r + 1) * stride], fact[0]);
        cmul2(&z[r][1], &z[r][2], in[(r + 1) * stride], fact[1]);
    }
    out[0].re = in[0].re + in[stride].re + in[2 * stride].re + in[3 * stride].re + in[4 * stride].re;
    out[0].im = in[0].im + in[stride].im + in[2 * stride].im + in[3 * stride].im + in[4 * stride].im;

    out[1].re = in[0].re + z[0][0].re + z[1][1].re + z[2][2].re + z[3][3].re;
    out[1].im = in[0].im + z[0][0].im + z[1][1].im + z[2][2].im + z[3][3].im;

    out[2].re = in[0].re + z[0][1].re + z[1][3].re + z[2][0].re + z[3][2].re;
    out[2].im = in[0].im + z[0][1].im + z[1][3].im + z[2][0].im + z[3][2].im;

    out[3].re = in[0].re + z[0][2].re + z[1][0].re + z[2][3].re + z[3][1].re;
    out[3].im = in[0].im + z[0][2].im + z[1][0].im + z[2][3].im + z[3][1].im;

    out[4].re = in[0].re + z[0][3].re + z[1][2].re + z[2][1].re + z[3][0].re;
    out[4].im = in[0].im + z[0][3].im + z[1][2].im + z[2][1].im + z[3][0].im;
}

static void fft15(const imdct15_ctx *s, cpx *out, const cpx *in, ptrdiff_t stride)   /* :1552-1581 */
{
    const cpx *exptab = s->exptab[0];
    cpx tmp[5], tmp1[5], tmp2[5];
    fft5(tmp, in, stride * 3);
    fft5(tmp1, in + stride, stride * 3);
    fft5(tmp2, in + 2 * stride, stride * 3);
    for (int k = 0; k < 5; k++) {
        cpx t1, t2;
        CMUL(t1, tmp1[k], exptab[k]);
        CMUL(t2, tmp2[k], exptab[2 * k]);
        out[k].re = tmp[k].re + t1.re + t2.re;
        out[k].im = tmp[k].im + t1.im + t2.im;

        CMUL(t1, tmp1[k], exptab[k + 5]);
        CMUL(t2, tmp2[k], exptab[2 * (k + 5)]);
        out[k + 5].re = tmp[k].re + t1.re + t2.re;
        out[k + 5].im = tmp[k].im + t1.im + t2.im;

        CMUL(t1, tmp1[k], exptab[k + 10]);
        CMUL(t2, tmp2[k], exptab[2 * k + 5]);
        out[k + 10].re = tmp[k].re + t1.re + t2.re;
        out[k + 10].im = tmp[k].im + t1.im + t2.im;
    }
}

static void fft_calc(const imdct15_ctx *s, cpx *out, const cpx *in, int N, ptrdiff_t stride)   /* :1586-1609 */
{
    if (N) {
        const cpx *exptab = s->exptab[N];
        const int len2 = 15 * (1 << (N - 1));
        fft_calc(s, out, in, N - 1, stride * 2);
        fft_calc(s, out + len2, in + stride, N - 1, stride * 2);
        for (int k = 0; k < len2; k++) {
            cpx t;
            CMUL(t, out[len2 + k], exptab[k]);
            out[len2 + k].re = out[k].re - t.re;
            out[len2 + k].im = out[k].im - t.im;
            out[k].re += t.re;
            out[k].im += t.im;
        }
    } else {
        fft15(s, out, in, stride);
    }
}

static void imdct15_half(const imdct15_ctx *s, float *dst, const float *src, ptrdiff_t stride, float scale)   /* :1611-1637 */
{
    cpx *z = (cpx *)dst;
    const int len8 = s->len4 / 2;
    const float *in1 = src;
    const float *in2 = src + (s->len2 - 1) * stride;
    for (int i = 0; i < s->len4; i++) {
        cpx tmp = { *in2, *in1 };
        CMUL(s->tmp[i], tmp, s->twiddle_exptab[i]);
        in1 += 2 * stride;
        in2 -= 2 * stride;
    }
    fft_calc(s, z, s->tmp, s->fft_n, 1);
    for (int i = 0; i < len8; i++) {
        float r0, i0, r1, i1;
        CMUL3(r0, i1, z[len8 - i - 1].im, z[len8 - i - 1].re, s->twiddle_exptab[len8 - i - 1].im, s->twiddle_exptab[len8 - i - 1].re);
        CMUL3(r1, i0, z[len8 + i].im, z[len8 + i].re, s->twiddle_exptab[len8 + i].im, s->twiddle_exptab[len8 + i].re);
        z[len8 - i - 1].re = scale * r0;
        z[len8 - i - 1].im = scale * i0;
        z[len8 + i].re = scale * r1;
        z[len8 + i].im = scale * i1;
    }
}

void afgo_celt_imdct_half(int N, float *dst, const float *src, int stride, float scale)
{
    imdct15_ctx c;
    if (imdct15_init(&c, N) == 0) imdct15_half(&c, dst, src, stride, scale);
    imdct15_free(&c);
}

static void fmul_window(float *dst, const float *src0, const float *src1, const float *win, int len)   /* :230-243 */
{
    dst += len; win += len; src0 += len;
    for (int i = -len, j = len - 1; i < 0; ++i, --j) {
        float s0 = src0[i], s1 = src1[j], wi = win[i], wj = win[j];
        dst[i] = s0 * wj - s1 * wi;
        dst[j] = s0 * wi + s1 * wj;
    }
}

static void postfilter_transition(afgo_celt_state *f, float *data)      /* :3281-3324 */
{
    const int T0 = f->pf_period_old, T1 = f->pf_period;
    if (f->pf_gains[0] == 0.0 && f->pf_gains_old[0] == 0.0) return;
    float g00 = f->pf_gains_old[0], g01 = f->pf_gains_old[1], g02 = f->pf_gains_old[2];
    float g10 = f->pf_gains[0], g11 = f->pf_gains[1], g12 = f->pf_gains[2];
    float x0, x1 = data[-T1 + 1], x2 = data[-T1], x3 = data[-T1 - 1], x4 = data[-T1 - 2];
    for (int i = 0; i < 120; i++) {
        float w = k_celt_window2[i];
        x0 = data[i - T1 + 2];
        data[i] += (1.0 - w) * g00 * data[i - T0] +
                   (1.0 - w) * g01 * (data[i - T0 - 1] + data[i - T0 + 1]) +
                   (1.0 - w) * g02 * (data[i - T0 - 2] + data[i - T0 + 2]) +
                   w * g10 * x2 +
                   w * g11 * (x1 + x3) +
                   w * g12 * (x0 + x4);
        x4 = x3; x3 = x2; x2 = x1; x1 = x0;
    }
}

static void postfilter_apply(afgo_celt_state *f, float *data, int len)  /* :3326-3355 */
{
    const int T = f->pf_period;
    if (f->pf_gains[0] == 0.0 || len <= 0) return;
    float g0 = f->pf_gains[0], g1 = f->pf_gains[1], g2 = f->pf_gains[2];
    float x0, x4 = data[-T - 2], x3 = data[-T - 1], x2 = data[-T], x1 = data[-T + 1];
    for (int i = 0; i < len; i++) {
        x0 = data[i - T + 2];
        data[i] += g0 * x2 + g1 * (x1 + x3) + g2 * (x0 + x4);
        x4 = x3; x3 = x2; x2 = x1; x1 = x0;
    }
}

/* One frame of one channel: transform loop (:3684-3690), celt_postfilter (:3357-3378),
 * de-emphasis + scaling (:3695-3701).  coeffs: frame_size floats (blocks interleaved when
 * transient); out[j * out_stride], j < frame_size. */
void afgo_celt_frame_channel(afgo_celt_state *f, const afgo_celt_frame *fr, const float *coeffs,
                             float *out, int out_stride)
{
    const int frame_size = fr->frame_size, blocks = fr->blocks, blocksize = frame_size / blocks;
    int N = 0;
    while ((15 << N) < blocksize) N++;
    /* the reference builds its four contexts once per decoder (dopus.d:3792-3796); here once per thread and table
     * reading, so that timing this restatement does not time 2000 cosl/sinl calls per frame */
    static __thread imdct15_ctx cache[8];
    static __thread int cache_key[8];
    if (cache_key[N] != 1 + g_celt_table_mode) {
        if (cache_key[N]) imdct15_free(&cache[N]);
        cache_key[N] = 0;
        if (imdct15_init(&cache[N], N)) return;
        cache_key[N] = 1 + g_celt_table_mode;
    }
    const imdct15_ctx ctx = cache[N];
    for (int j = 0; j < blocks; j++) {
        float *dst = f->buf + 1024 + j * blocksize;
        imdct15_half(&ctx, dst + 60, coeffs + j, blocks, fr->imdct_scale);
        fmul_window(dst, dst, dst + 60, k_celt_window, 60);
    }

    {   /* celt_postfilter */
        int len = frame_size;
        postfilter_transition(f, f->buf + 1024);
        f->pf_period_old = f->pf_period;
        memcpy(f->pf_gains_old, f->pf_gains, sizeof(f->pf_gains));
        f->pf_period = fr->pf_period_new;
        memcpy(f->pf_gains, fr->pf_gains_new, sizeof(f->pf_gains));
        if (len > 120) {
            postfilter_transition(f, f->buf + 1024 + 120);
            postfilter_apply(f, f->buf + 1024 + 2 * 120, len - 2 * 120);
            f->pf_period_old = f->pf_period;
            memcpy(f->pf_gains_old, f->pf_gains, sizeof(f->pf_gains));
        }
        memmove(f->buf, f->buf + len, (1024 + 120 / 2) * sizeof(float));
    }
    float m = f->deemph_coeff;
    for (int j = 0; j < frame_size; j++) {
        float tmp = f->buf[1024 - frame_size + j] + m;
        m = tmp * 0.85000610f;
        out[(size_t)j * (size_t)out_stride] = (float)(tmp / 32768.);
    }
    f->deemph_coeff = m;
}

/* Batch: n_chan independent channel sequences; sequence k owns records
 * [rec_base[k], rec_base[k+1]) processed in order from zero state (or states[k] if given,
 * which also receives the final state). */
void afgo_celt_transform(uint32_t n_chan, const uint64_t *rec_base, const afgo_celt_frame *recs,
                         const float *coeffs, float *out, afgo_celt_state *states)
{
    for (uint32_t k = 0; k < n_chan; k++) {
        afgo_celt_state st;
        if (states) st = states[k]; else memset(&st, 0, sizeof(st));
        for (uint64_t r = rec_base[k]; r < rec_base[k + 1]; r++)
            afgo_celt_frame_channel(&st, &recs[r], coeffs + recs[r].coef_off, out + recs[r].out_off,
                                    (int)recs[r].out_stride);
        if (states) states[k] = st;
    }
}

/* OpusFile.readFrame's Float2IntScaled + saturation (dopus.d:7923-7926, :8098-8105) and AudioStream's
 * int16 / 32767.0f (stream.d:480).  The magic constant 1.5f*(1<<8) + 0.5f/(1<<15) is a float: 384 + 2^-16 lies
 * exactly between 384 and the next float and rounds to even, i.e. to 384.0f. */
void afgo_opus_output(uint64_t n, const float *in, int16_t *out_i16, float *out_f32)
{
    const volatile float magic = (float)(1.5f * (float)(1 << (23 - 15)) + 0.5f / (float)(1 << 15));
    for (uint64_t i = 0; i < n; i++) {
        union { float f; int32_t i; } temp;
        temp.f = in[i] + magic;
        /* D's int arithmetic wraps: for -768 < x < -384 (the sum is negative with a magnitude below 384) the subtraction passes
         * INT_MIN and the sample comes out as +32767.  Unsigned arithmetic states that in C. */
        int32_t d = (int32_t)((uint32_t)temp.i - (uint32_t)(((150 - 15) << 23) + (1 << 22)));
        if ((uint32_t)d + 32768u > 65535u) d = d < 0 ? -32768 : 32767;
        if (out_i16) out_i16[i] = (int16_t)d;
        if (out_f32) out_f32[i] = (float)(int16_t)d / 32767.0f;
    }
}
