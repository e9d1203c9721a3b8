/*
 * oracle/vorbis_transform.c -- CPU restatement of the Vorbis transform stage.
 * TEST INFRASTRUCTURE ONLY (see afg_oracle.h).  PARITY UNPINNED by reference
 * vectors (the reference has none); pinned by tests/test_oracle_vorbis.py.
 *
 * Follows source/audioformats/stb_vorbis2.d of the reference:
 *   bit_reverse / ilog           :617-650       M_PI (float!)       :652
 *   twiddles / window / bitrev   :851-898       inverse_mdct        :1720-2242
 *   window bounds                :2333-2349     finish_frame        :2606-2657
 *   interleave                   :3927-3952
 *
 * Numeric choices where the D source leaves room (documented in HISTORY.md 4):
 *   - M_PI is a float enum, so twiddle angles such as 4*k*M_PI/n are evaluated
 *     in float32; cos/sin are then taken in double and rounded to float.
 *   - the window (:872) is evaluated in double with the float-rounded pi and
 *     the inner sin rounded to float before squaring (square(float), :626).
 */
#include "afg_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const float k_pi_f = 3.14159265358979323846264f;   /* :652 */

static unsigned bit_reverse32(unsigned n)                   /* :617-624 */
{
    n = ((n & 0xAAAAAAAAu) >> 1) | ((n & 0x55555555u) << 1);
    n = ((n & 0xCCCCCCCCu) >> 2) | ((n & 0x33333333u) << 2);
    n = ((n & 0xF0F0F0F0u) >> 4) | ((n & 0x0F0F0F0Fu) << 4);
    n = ((n & 0xFF00FF00u) >> 8) | ((n & 0x00FF00FFu) << 8);
    return (n >> 16) | (n << 16);
}

static int ilog_vorbis(int32_t n)                           /* :634-650 (log2(1)=1, log2(2)=2, log2(4)=3) */
{
    int r = 0;
    if (n < 0) return 0;
    while (n) { r++; n >>= 1; }
    return r;
}

static float squaref(float x) { return x * x; }             /* :626-629 */

static int g_table_mode = 0;
void afgo_vorbis_set_table_mode(int mode) { g_table_mode = (mode >= 0 && mode <= 3) ? mode : 0; }

int afgo_vorbis_tables_init(afgo_vorbis_tables *t, int n)
{
    int n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
    memset(t, 0, sizeof(*t));
    t->n = n;
    t->A = (float *)malloc(sizeof(float) * (size_t)n2);
    t->B = (float *)malloc(sizeof(float) * (size_t)n2);
    t->C = (float *)malloc(sizeof(float) * (size_t)n4);
    t->window = (float *)malloc(sizeof(float) * (size_t)n2);
    t->bitrev = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)n8);
    if (!t->A || !t->B || !t->C || !t->window || !t->bitrev) { afgo_vorbis_tables_free(t); return -1; }

    /* compute_twiddle_factors, :851-866 / compute_window, :868-873.  M_PI is a float enum (:652), so the D
     * expression 4*k*M_PI/n has float type -- but D lets an implementation keep float intermediates in higher
     * precision and std.math.cos(float) has been a float, a double and a real function in different Phobos releases.
     * g_table_mode selects the reading (tests/test_oracle_numeric_readings.py measures that every one of them gives
     * the same PCM within the 1e-5 RMS tolerance):
     *   0  float angle, cos/sin in double                 (default; what the product builds)
     *   1  float angle, cosf/sinf
     *   2  angle kept in x87 real, cosl/sinl
     *   3  float angle, cosl/sinl                                                         */
    const int mode = g_table_mode;
#define AFGO_ANG(num, den)  (mode == 2 ? (long double)(num) * (long double)k_pi_f / (long double)(den) \
                                       : (long double)((float)(num) * k_pi_f / (float)(den)))
#define AFGO_COS(a) (mode == 1 ? (float)cosf((float)(a)) : mode == 0 ? (float)cos((double)(a)) : (float)cosl(a))
#define AFGO_SIN(a) (mode == 1 ? (float)sinf((float)(a)) : mode == 0 ? (float)sin((double)(a)) : (float)sinl(a))
    for (int k = 0, k2 = 0; k < n4; ++k, k2 += 2) {
        long double a0 = AFGO_ANG(4 * k, n);
        long double a1 = mode == 2 ? (long double)(k2 + 1) * (long double)k_pi_f / (long double)n / 2.0L
                                   : (long double)((float)(k2 + 1) * k_pi_f / (float)n / (float)2);
        t->A[k2]     =  AFGO_COS(a0);
        t->A[k2 + 1] = -AFGO_SIN(a0);
        t->B[k2]     =  AFGO_COS(a1) * 0.5f;
        t->B[k2 + 1] =  AFGO_SIN(a1) * 0.5f;
    }
    for (int k = 0, k2 = 0; k < n8; ++k, k2 += 2) {
        long double a2 = AFGO_ANG(2 * (k2 + 1), n);
        t->C[k2]     =  AFGO_COS(a2);
        t->C[k2 + 1] = -AFGO_SIN(a2);
    }
    /* compute_window, :868-873: double arithmetic with the float-rounded pi; the inner sine is rounded to float by the
     * cast before square() (:626).  Modes 2/3 evaluate the sines in real, mode 1 in float. */
    for (int i = 0; i < n2; ++i) {
        if (mode == 0) {
            double inner = sin((i - 0 + 0.5) / n2 * 0.5 * (double)k_pi_f);
            t->window[i] = (float) sin(0.5 * (double)k_pi_f * (double)squaref((float)inner));
        } else if (mode == 1) {
            float inner = sinf((float)((i - 0 + 0.5) / n2 * 0.5 * (double)k_pi_f));
            t->window[i] = sinf((float)(0.5 * (double)k_pi_f * (double)squaref(inner)));
        } else {
            long double inner = sinl((i - 0 + 0.5L) / n2 * 0.5L * (long double)k_pi_f);
            t->window[i] = (float) sinl(0.5L * (long double)k_pi_f * (long double)squaref((float)inner));
        }
    }
#undef AFGO_ANG
#undef AFGO_COS
#undef AFGO_SIN
    /* compute_bitreverse, :875-881 */
    int ld = ilog_vorbis(n) - 1;
    for (int i = 0; i < n8; ++i)
        t->bitrev[i] = (uint16_t)((bit_reverse32((unsigned)i) >> (32 - ld + 3)) << 2);
    return 0;
}

void afgo_vorbis_tables_free(afgo_vorbis_tables *t)
{
    free(t->A); free(t->B); free(t->C); free(t->window); free(t->bitrev);
    memset(t, 0, sizeof(*t));
}

/* ---- step-3 butterflies, :1720-1939 --------------------------------------- */

/* One radix-2 butterfly on the complex pair stored at (p[0], p[-1]) / (q[0], q[-1]). */
#define BFLY(p, q, c0, c1) do {                      \
        float d0_ = (p)[0] - (q)[0];                 \
        float d1_ = (p)[-1] - (q)[-1];               \
        (p)[0]  = (p)[0] + (q)[0];                   \
        (p)[-1] = (p)[-1] + (q)[-1];                 \
        (q)[0]  = d0_ * (c0) - d1_ * (c1);           \
        (q)[-1] = d1_ * (c0) + d0_ * (c1);           \
    } while (0)

static void step3_iter0(int n, float *e, int i_off, int k_off, const float *A)     /* :1720-1763 */
{
    float *ee0 = e + i_off;
    float *ee2 = ee0 + k_off;
    for (int i = n >> 2; i > 0; --i) {
        BFLY(ee0,     ee2,     A[0], A[1]); A += 8;
        BFLY(ee0 - 2, ee2 - 2, A[0], A[1]); A += 8;
        BFLY(ee0 - 4, ee2 - 4, A[0], A[1]); A += 8;
        BFLY(ee0 - 6, ee2 - 6, A[0], A[1]); A += 8;
        ee0 -= 8;
        ee2 -= 8;
    }
}

static void step3_inner_r(int lim, float *e, int d0, int k_off, const float *A, int k1)   /* :1765-1813 */
{
    float *e0 = e + d0;
    float *e2 = e0 + k_off;
    for (int i = lim >> 2; i > 0; --i) {
        BFLY(e0,     e2,     A[0], A[1]); A += k1;
        BFLY(e0 - 2, e2 - 2, A[0], A[1]); A += k1;
        BFLY(e0 - 4, e2 - 4, A[0], A[1]); A += k1;
        BFLY(e0 - 6, e2 - 6, A[0], A[1]); A += k1;
        e0 -= 8;
        e2 -= 8;
    }
}

static void step3_inner_s(int n, float *e, int i_off, int k_off, const float *A, int a_off, int k0)  /* :1815-1864 */
{
    float A0 = A[0],             A1 = A[1];
    float A2 = A[a_off],         A3 = A[a_off + 1];
    float A4 = A[a_off * 2],     A5 = A[a_off * 2 + 1];
    float A6 = A[a_off * 3],     A7 = A[a_off * 3 + 1];
    float *ee0 = e + i_off;
    float *ee2 = ee0 + k_off;
    for (int i = n; i > 0; --i) {
        BFLY(ee0,     ee2,     A0, A1);
        BFLY(ee0 - 2, ee2 - 2, A2, A3);
        BFLY(ee0 - 4, ee2 - 4, A4, A5);
        BFLY(ee0 - 6, ee2 - 6, A6, A7);
        ee0 -= k0;
        ee2 -= k0;
    }
}

static void iter_54(float *z)                                                     /* :1866-1896 */
{
    float k00 = z[0] - z[-4];
    float y0  = z[0] + z[-4];
    float y2  = z[-2] + z[-6];
    float k22 = z[-2] - z[-6];

    z[0]  = y0 + y2;
    z[-2] = y0 - y2;

    float k33 = z[-3] - z[-7];

    z[-4] = k00 + k33;
    z[-6] = k00 - k33;

    float k11 = z[-1] - z[-5];
    float y1  = z[-1] + z[-5];
    float y3  = z[-3] + z[-7];

    z[-1] = y1 + y3;
    z[-3] = y1 - y3;
    z[-5] = k11 - k22;
    z[-7] = k11 + k22;
}

static void step3_ld654(int n, float *e, int i_off, const float *A, int base_n)   /* :1898-1939 */
{
    int a_off = base_n >> 3;
    float A2 = A[a_off];
    float *z = e + i_off;
    float *base = z - 16 * n;

    while (z > base) {
        float k00, k11, l00, l11;

        k00 = z[0] - z[-8];
        k11 = z[-1] - z[-9];
        l00 = z[-2] - z[-10];
        l11 = z[-3] - z[-11];
        z[0]   = z[0] + z[-8];
        z[-1]  = z[-1] + z[-9];
        z[-2]  = z[-2] + z[-10];
        z[-3]  = z[-3] + z[-11];
        z[-8]  = k00;
        z[-9]  = k11;
        z[-10] = (l00 + l11) * A2;
        z[-11] = (l11 - l00) * A2;

        k00 = z[-4] - z[-12];
        k11 = z[-5] - z[-13];
        l00 = z[-6] - z[-14];
        l11 = z[-7] - z[-15];
        z[-4]  = z[-4] + z[-12];
        z[-5]  = z[-5] + z[-13];
        z[-6]  = z[-6] + z[-14];
        z[-7]  = z[-7] + z[-15];
        z[-12] = k11;
        z[-13] = -k00;
        z[-14] = (l11 - l00) * A2;
        z[-15] = (l00 + l11) * -A2;

        iter_54(z);
        iter_54(z - 8);
        z -= 16;
    }
}

/* ---- inverse_mdct, :1941-2242 --------------------------------------------- */
void afgo_vorbis_inverse_mdct(float *buffer, int n, const afgo_vorbis_tables *t, float *buf2)
{
    int n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
    const float *A = t->A;
    float *u, *v;

    /* copy-and-reflect + step 0, :1972-1994 */
    {
        float *d = &buf2[n2 - 2];
        const float *AA = A;
        const float *e = &buffer[0];
        const float *e_stop = &buffer[n2];
        while (e != e_stop) {
            d[1] = (e[0] * AA[0] - e[2] * AA[1]);
            d[0] = (e[0] * AA[1] + e[2] * AA[0]);
            d -= 2;
            AA += 2;
            e += 4;
        }
        e = &buffer[n2 - 3];
        while (d >= buf2) {
            d[1] = (-e[2] * AA[0] - -e[0] * AA[1]);
            d[0] = (-e[2] * AA[1] + -e[0] * AA[0]);
            d -= 2;
            AA += 2;
            e -= 4;
        }
    }

    u = buffer;
    v = buf2;

    /* step 2, :2006-2040 */
    {
        const float *AA = &A[n2 - 8];
        const float *e0 = &v[n4], *e1 = &v[0];
        float *d0 = &u[n4], *d1 = &u[0];
        while (AA >= A) {
            float v40_20, v41_21;

            v41_21 = e0[1] - e1[1];
            v40_20 = e0[0] - e1[0];
            d0[1] = e0[1] + e1[1];
            d0[0] = e0[0] + e1[0];
            d1[1] = v41_21 * AA[4] - v40_20 * AA[5];
            d1[0] = v40_20 * AA[4] + v41_21 * AA[5];

            v41_21 = e0[3] - e1[3];
            v40_20 = e0[2] - e1[2];
            d0[3] = e0[3] + e1[3];
            d0[2] = e0[2] + e1[2];
            d1[3] = v41_21 * AA[0] - v40_20 * AA[1];
            d1[2] = v40_20 * AA[0] + v41_21 * AA[1];

            AA -= 8;
            d0 += 4; d1 += 4; e0 += 4; e1 += 4;
        }
    }

    /* step 3, :2043-2090 */
    int ld = ilog_vorbis(n) - 1;

    step3_iter0(n >> 4, u, n2 - 1 - n4 * 0, -(n >> 3), A);
    step3_iter0(n >> 4, u, n2 - 1 - n4 * 1, -(n >> 3), A);

    step3_inner_r(n >> 5, u, n2 - 1 - n8 * 0, -(n >> 4), A, 16);
    step3_inner_r(n >> 5, u, n2 - 1 - n8 * 1, -(n >> 4), A, 16);
    step3_inner_r(n >> 5, u, n2 - 1 - n8 * 2, -(n >> 4), A, 16);
    step3_inner_r(n >> 5, u, n2 - 1 - n8 * 3, -(n >> 4), A, 16);

    int l = 2;
    for (; l < (ld - 3) >> 1; ++l) {
        int k0 = n >> (l + 2), k0_2 = k0 >> 1;
        int lim = 1 << (l + 1);
        for (int i = 0; i < lim; ++i)
            step3_inner_r(n >> (l + 4), u, n2 - 1 - k0 * i, -k0_2, A, 1 << (l + 3));
    }
    for (; l < ld - 6; ++l) {
        int k0 = n >> (l + 2), k1 = 1 << (l + 3), k0_2 = k0 >> 1;
        int rlim = n >> (l + 6);
        int lim = 1 << (l + 1);
        const float *A0 = A;
        int i_off = n2 - 1;
        for (int r = rlim; r > 0; --r) {
            step3_inner_s(lim, u, i_off, -k0_2, A0, k1, k0);
            A0 += k1 * 4;
            i_off -= 8;
        }
    }
    step3_ld654(n >> 5, u, n2 - 1, A, n);

    /* steps 4-6 (bit reverse), :2096-2124 */
    {
        const uint16_t *bitrev = t->bitrev;
        float *d0 = &v[n4 - 4];
        float *d1 = &v[n2 - 4];
        while (d0 >= v) {
            int k4;
            k4 = bitrev[0];
            d1[3] = u[k4 + 0];
            d1[2] = u[k4 + 1];
            d0[3] = u[k4 + 2];
            d0[2] = u[k4 + 3];
            k4 = bitrev[1];
            d1[1] = u[k4 + 0];
            d1[0] = u[k4 + 1];
            d0[1] = u[k4 + 2];
            d0[0] = u[k4 + 3];
            d0 -= 4;
            d1 -= 4;
            bitrev += 2;
        }
    }

    /* step 7, :2133-2175 */
    {
        const float *C = t->C;
        float *d = v;
        float *e = v + n2 - 4;
        while (d < e) {
            float a02, a11, b0, b1, b2, b3;

            a02 = d[0] - e[2];
            a11 = d[1] + e[3];
            b0 = C[1] * a02 + C[0] * a11;
            b1 = C[1] * a11 - C[0] * a02;
            b2 = d[0] + e[2];
            b3 = d[1] - e[3];
            d[0] = b2 + b0;
            d[1] = b3 + b1;
            e[2] = b2 - b0;
            e[3] = b1 - b3;

            a02 = d[2] - e[0];
            a11 = d[3] + e[1];
            b0 = C[3] * a02 + C[2] * a11;
            b1 = C[3] * a11 - C[2] * a02;
            b2 = d[2] + e[0];
            b3 = d[3] - e[1];
            d[2] = b2 + b0;
            d[3] = b3 + b1;
            e[0] = b2 - b0;
            e[1] = b1 - b3;

            C += 4;
            d += 4;
            e -= 4;
        }
    }

    /* step 8 + decode, :2187-2238 */
    {
        const float *B = t->B + n2 - 8;
        const float *e = buf2 + n2 - 8;
        float *d0 = &buffer[0];
        float *d1 = &buffer[n2 - 4];
        float *d2 = &buffer[n2];
        float *d3 = &buffer[n - 4];
        while (e >= v) {
            float p0, p1, p2, p3;

            p3 =  e[6] * B[7] - e[7] * B[6];
            p2 = -e[6] * B[6] - e[7] * B[7];
            d0[0] = p3; d1[3] = -p3; d2[0] = p2; d3[3] = p2;

            p1 =  e[4] * B[5] - e[5] * B[4];
            p0 = -e[4] * B[4] - e[5] * B[5];
            d0[1] = p1; d1[2] = -p1; d2[1] = p0; d3[2] = p0;

            p3 =  e[2] * B[3] - e[3] * B[2];
            p2 = -e[2] * B[2] - e[3] * B[3];
            d0[2] = p3; d1[1] = -p3; d2[2] = p2; d3[1] = p2;

            p1 =  e[0] * B[1] - e[1] * B[0];
            p0 = -e[0] * B[0] - e[1] * B[1];
            d0[3] = p1; d1[0] = -p1; d2[3] = p0; d3[0] = p0;

            B -= 8;
            e -= 8;
            d0 += 4; d2 += 4;
            d1 -= 4; d3 -= 4;
        }
    }
}

/* ---- window bounds, :2333-2349 -------------------------------------------- */
void afgo_vorbis_window_bounds(int blocksize0, int blocksize1, unsigned pflags,
                               int *pn, int *left_start, int *left_end,
                               int *right_start, int *right_end)
{
    int blockflag = (pflags & AFGO_VORBIS_LONG) != 0;
    int prev = blockflag ? ((pflags & AFGO_VORBIS_PREV) != 0) : 0;
    int next = blockflag ? ((pflags & AFGO_VORBIS_NEXT) != 0) : 0;
    int n = blockflag ? blocksize1 : blocksize0;
    int window_center = n >> 1;
    if (blockflag && !prev) {
        *left_start = (n - blocksize0) >> 2;
        *left_end   = (n + blocksize0) >> 2;
    } else {
        *left_start = 0;
        *left_end   = window_center;
    }
    if (blockflag && !next) {
        *right_start = (n * 3 - blocksize0) >> 2;
        *right_end   = (n * 3 + blocksize0) >> 2;
    } else {
        *right_start = window_center;
        *right_end   = n;
    }
    *pn = n;
}

uint64_t afgo_vorbis_layout(uint32_t npkt, int nch, int blocksize0, int blocksize1,
                            const uint8_t *pflags, uint64_t spec_base, uint64_t out_base,
                            uint64_t *spec_off, uint64_t *out_off, uint64_t *spec_total)
{
    uint64_t so = spec_base, oo = out_base;
    int prev_len = 0;
    for (uint32_t p = 0; p < npkt; p++) {
        int n, ls, le, rs, re;
        afgo_vorbis_window_bounds(blocksize0, blocksize1, pflags[p], &n, &ls, &le, &rs, &re);
        spec_off[p] = so;
        out_off[p] = oo;
        so += (uint64_t)(n / 2) * (uint64_t)nch;
        if (prev_len) oo += (uint64_t)(rs - ls) * (uint64_t)nch;   /* :2645-2656 */
        prev_len = re - rs;                                      /* len = right_end, :2594, :2633 */
    }
    if (spec_total) *spec_total = so;
    return oo;
}

/* ---- whole batch ----------------------------------------------------------- */
int afgo_vorbis_transform(uint32_t n_streams, const uint32_t *npkt, const uint8_t *nch,
                          const uint16_t *blocksize0, const uint16_t *blocksize1,
                          const uint8_t *pflags, const uint64_t *spec_off,
                          const uint64_t *out_off, const float *spec, float *out)
{
    uint64_t pkt = 0;
    int rc = 0;
    for (uint32_t s = 0; s < n_streams; s++) {
        int C = nch[s];
        int bs0 = blocksize0[s], bs1 = blocksize1[s];
        afgo_vorbis_tables tab[2];
        if (afgo_vorbis_tables_init(&tab[0], bs0) || afgo_vorbis_tables_init(&tab[1], bs1)) return -2;
        float *chan = (float *)malloc(sizeof(float) * (size_t)bs1 * (size_t)C);        /* channel_buffers */
        float *prevw = (float *)calloc((size_t)(bs1 / 2) * (size_t)C, sizeof(float));   /* previous_window */
        float *scratch = (float *)malloc(sizeof(float) * (size_t)(bs1 / 2));
        int previous_length = 0;

        for (uint32_t p = 0; p < npkt[s]; p++, pkt++) {
            int n, left, left_end, right, right_end;
            unsigned fl = pflags[pkt];
            afgo_vorbis_window_bounds(bs0, bs1, fl, &n, &left, &left_end, &right, &right_end);
            const afgo_vorbis_tables *t = &tab[(fl & AFGO_VORBIS_LONG) ? 1 : 0];
            int len = right_end;                                              /* :2594 */

            for (int c = 0; c < C; c++) {
                float *buf = chan + (size_t)c * (size_t)bs1;
                memcpy(buf, spec + spec_off[pkt] + (size_t)c * (size_t)(n / 2), sizeof(float) * (size_t)(n / 2));
                afgo_vorbis_inverse_mdct(buf, n, t, scratch);                 /* :2526-2527 */
            }

            /* vorbis_finish_frame, :2606-2657 */
            if (previous_length) {
                int pn = previous_length;
                const float *w = NULL;                                        /* get_window, :2245-2251 */
                if (pn * 2 == bs0) w = tab[0].window;
                else if (pn * 2 == bs1) w = tab[1].window;
                if (!w) { rc = -1; continue; }                                /* :2621 returns before any state update */
                for (int c = 0; c < C; c++) {
                    float *buf = chan + (size_t)c * (size_t)bs1;
                    const float *pw = prevw + (size_t)c * (size_t)(bs1 / 2);
                    for (int j = 0; j < pn; j++)
                        buf[left + j] = buf[left + j] * w[j] + pw[j] * w[pn - 1 - j];
                }
            }
            int prev = previous_length;
            previous_length = len - right;
            for (int c = 0; c < C; c++) {
                const float *buf = chan + (size_t)c * (size_t)bs1;
                float *pw = prevw + (size_t)c * (size_t)(bs1 / 2);
                for (int j = 0; right + j < len; j++)
                    pw[j] = buf[right + j];
            }
            if (!prev) continue;                                              /* :2645-2649 */

            /* interleave, :3927-3952 (no clipping, no scaling) */
            float *o = out + out_off[pkt];
            for (int j = 0; j < right - left; j++)
                for (int c = 0; c < C; c++)
                    o[(size_t)j * (size_t)C + (size_t)c] = chan[(size_t)c * (size_t)bs1 + (size_t)(left + j)];
        }
        free(chan); free(prevw); free(scratch);
        afgo_vorbis_tables_free(&tab[0]);
        afgo_vorbis_tables_free(&tab[1]);
    }
    return rc;
}
