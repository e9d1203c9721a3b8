/*
 * oracle/mp3_transform.c -- CPU restatement of the MP3 Layer III transform stage.
 * TEST INFRASTRUCTURE ONLY (see afg_oracle.h).  PARITY UNPINNED by reference
 * vectors (the reference has none); pinned by tests/test_oracle_mp3.py.
 *
 * Follows source/audioformats/minimp3.d of the reference:
 *   antialias            :1002-1020      imdct36 / dct3_9   :1022-1100
 *   imdct12 / short      :1102-1142      change_sign        :1144-1150
 *   imdct_gr dispatch    :1152-1168      L3_decode tail     :1215-1229
 *   DCT-II (32 point)    :1232-1298      synth_pair/synth   :1300-1406
 *   synth_granule        :1408-1434      granule loop       :1549-1554
 * All arithmetic is float32 with the operation order of those lines; compile
 * with -ffp-contract=off.
 */
#include "afg_oracle.h"
#include <string.h>
#include <stdlib.h>

/* ---- alias reduction, minimp3.d:1002-1020 -------------------------------- */
static const float k_aa_cs[8] = {
    0.85749293f, 0.88174200f, 0.94962865f, 0.98331459f, 0.99551782f, 0.99916056f, 0.99989920f, 0.99999316f };
static const float k_aa_ca[8] = {
    0.51449576f, 0.47173197f, 0.31337745f, 0.18191320f, 0.09457419f, 0.04096558f, 0.01419856f, 0.00369997f };

void afgo_mp3_antialias(float *grbuf, int nbands)
{
    for (int b = 0; b < nbands; b++) {
        float *edge = grbuf + 18 * (b + 1);      /* first line of band b+1 */
        for (int i = 0; i < 8; i++) {
            float up = edge[i];
            float dn = edge[-1 - i];
            edge[i]      = up * k_aa_cs[i] - dn * k_aa_ca[i];
            edge[-1 - i] = up * k_aa_ca[i] + dn * k_aa_cs[i];
        }
    }
}

/* ---- 9-point DCT-III, minimp3.d:1022-1060 -------------------------------- */
static void dct3_9(float *y)
{
    /* even-indexed inputs */
    float e0 = y[0], e2 = y[2], e4 = y[4], e6 = y[6], e8 = y[8];
    float m0 = e0 + e6 * 0.5f;
    e0 = e0 - e6;
    float m4 = (e4 + e2) * 0.93969262f;
    float m2 = (e8 + e2) * 0.76604444f;
    e6 = (e4 - e8) * 0.17364818f;
    e4 = e4 + (e8 - e2);

    e2 = e0 - e4 * 0.5f;
    y[4] = e4 + e0;
    e8 = m0 - m2 + e6;
    e0 = m0 - m4 + m2;
    e4 = m0 + m4 - e6;

    /* odd-indexed inputs */
    float o1 = y[1], o3 = y[3], o5 = y[5], o7 = y[7];
    o3 = o3 * 0.86602540f;
    m0 = (o5 + o1) * 0.98480775f;
    m4 = (o5 - o7) * 0.34202014f;
    m2 = (o1 + o7) * 0.64278761f;
    o1 = (o1 - o5 - o7) * 0.86602540f;

    o5 = m0 - o3 - m2;
    o7 = m4 - o3 - m0;
    o3 = m4 + o3 - m2;

    y[0] = e4 - o7;
    y[1] = e2 + o1;
    y[2] = e0 - o3;
    y[3] = e8 + o5;
    y[5] = e8 - o5;
    y[6] = e0 + o3;
    y[7] = e2 - o1;
    y[8] = e4 + o7;
}

/* ---- long-block IMDCT (36), minimp3.d:1062-1100 -------------------------- */
static const float k_twid9[18] = {
    0.73727734f, 0.79335334f, 0.84339145f, 0.88701083f, 0.92387953f, 0.95371695f, 0.97629601f, 0.99144486f, 0.99904822f,
    0.67559021f, 0.60876143f, 0.53729961f, 0.46174861f, 0.38268343f, 0.30070580f, 0.21643961f, 0.13052619f, 0.04361938f };

static void imdct36(float *grbuf, float *overlap, const float *window, int nbands)
{
    for (int band = 0; band < nbands; band++, grbuf += 18, overlap += 9) {
        float co[9], si[9];
        co[0] = -grbuf[0];
        si[0] = grbuf[17];
        for (int i = 0; i < 4; i++) {
            si[8 - 2 * i] = grbuf[4 * i + 1] - grbuf[4 * i + 2];
            co[1 + 2 * i] = grbuf[4 * i + 1] + grbuf[4 * i + 2];
            si[7 - 2 * i] = grbuf[4 * i + 4] - grbuf[4 * i + 3];
            co[2 + 2 * i] = -(grbuf[4 * i + 3] + grbuf[4 * i + 4]);
        }
        dct3_9(co);
        dct3_9(si);
        si[1] = -si[1];
        si[3] = -si[3];
        si[5] = -si[5];
        si[7] = -si[7];

        for (int i = 0; i < 9; i++) {
            float ovl = overlap[i];
            float sum = co[i] * k_twid9[9 + i] + si[i] * k_twid9[i];
            overlap[i] = co[i] * k_twid9[i] - si[i] * k_twid9[9 + i];
            grbuf[i]      = ovl * window[i] - sum * window[9 + i];
            grbuf[17 - i] = ovl * window[9 + i] + sum * window[i];
        }
    }
}

/* ---- short-block IMDCT (3 x 12), minimp3.d:1102-1142 --------------------- */
static void idct3(float x0, float x1, float x2, float *dst)
{
    float m1 = x1 * 0.86602540f;
    float a1 = x0 - x2 * 0.5f;
    dst[1] = x0 + x2;
    dst[0] = a1 + m1;
    dst[2] = a1 - m1;
}

static const float k_twid3[6] = {
    0.79335334f, 0.92387953f, 0.99144486f, 0.60876143f, 0.38268343f, 0.13052619f };

static void imdct12(const float *x, float *dst, float *overlap)
{
    float co[3], si[3];
    idct3(-x[0], x[6] + x[3], x[12] + x[9], co);
    idct3(x[15], x[12] - x[9], x[6] - x[3], si);
    si[1] = -si[1];
    for (int i = 0; i < 3; i++) {
        float ovl = overlap[i];
        float sum = co[i] * k_twid3[3 + i] + si[i] * k_twid3[i];
        overlap[i] = co[i] * k_twid3[i] - si[i] * k_twid3[3 + i];
        dst[i]     = ovl * k_twid3[2 - i] - sum * k_twid3[5 - i];
        dst[5 - i] = ovl * k_twid3[5 - i] + sum * k_twid3[2 - i];
    }
}

static void imdct_short(float *grbuf, float *overlap, int nbands)
{
    for (; nbands > 0; nbands--, overlap += 9, grbuf += 18) {
        float tmp[18];
        memcpy(tmp, grbuf, sizeof(tmp));
        memcpy(grbuf, overlap, 6 * sizeof(float));
        imdct12(tmp, grbuf + 6, overlap + 6);
        imdct12(tmp + 1, grbuf + 12, overlap + 6);
        imdct12(tmp + 2, overlap, overlap + 6);
    }
}

/* ---- frequency inversion, minimp3.d:1144-1150 ---------------------------- */
void afgo_mp3_change_sign(float *grbuf)
{
    for (int band = 1; band < 32; band += 2)
        for (int i = 1; i < 18; i += 2)
            grbuf[band * 18 + i] = -grbuf[band * 18 + i];
}

/* ---- per-granule IMDCT dispatch, minimp3.d:1152-1168 --------------------- */
static const float k_mdct_window[2][18] = {
    { 0.99904822f, 0.99144486f, 0.97629601f, 0.95371695f, 0.92387953f, 0.88701083f, 0.84339145f, 0.79335334f, 0.73727734f,
      0.04361938f, 0.13052619f, 0.21643961f, 0.30070580f, 0.38268343f, 0.46174861f, 0.53729961f, 0.60876143f, 0.67559021f },
    { 1, 1, 1, 1, 1, 1, 0.99144486f, 0.92387953f, 0.79335334f,
      0, 0, 0, 0, 0, 0, 0.13052619f, 0.38268343f, 0.60876143f } };

void afgo_mp3_imdct_gr(float *grbuf, float *overlap, unsigned block_type, unsigned n_long_bands)
{
    if (n_long_bands) {
        imdct36(grbuf, overlap, k_mdct_window[0], (int)n_long_bands);
        grbuf += 18 * n_long_bands;
        overlap += 9 * n_long_bands;
    }
    if (block_type == 2)                                   /* SHORT_BLOCK_TYPE */
        imdct_short(grbuf, overlap, 32 - (int)n_long_bands);
    else
        imdct36(grbuf, overlap, k_mdct_window[block_type == 3], 32 - (int)n_long_bands); /* STOP_BLOCK_TYPE */
}

/* ---- 32-point DCT-II down the subband axis, minimp3.d:1232-1298 ---------- */
static const float k_sec[24] = {
    10.19000816f, 0.50060302f, 0.50241929f, 3.40760851f, 0.50547093f, 0.52249861f, 2.05778098f, 0.51544732f,
    0.56694406f, 1.48416460f, 0.53104258f, 0.64682180f, 1.16943991f, 0.55310392f, 0.78815460f, 0.97256821f,
    0.58293498f, 1.06067765f, 0.83934963f, 0.62250412f, 1.72244716f, 0.74453628f, 0.67480832f, 5.10114861f };

void afgo_mp3_dct2(float *grbuf, int n)
{
    for (int k = 0; k < n; k++) {
        float t[4][8];
        float *y = grbuf + k;

        for (int i = 0; i < 8; i++) {
            float x0 = y[i * 18];
            float x1 = y[(15 - i) * 18];
            float x2 = y[(16 + i) * 18];
            float x3 = y[(31 - i) * 18];
            float t0 = x0 + x3;
            float t1 = x1 + x2;
            float t2 = (x1 - x2) * k_sec[3 * i + 0];
            float t3 = (x0 - x3) * k_sec[3 * i + 1];
            t[0][i] = t0 + t1;
            t[1][i] = (t0 - t1) * k_sec[3 * i + 2];
            t[2][i] = t3 + t2;
            t[3][i] = (t3 - t2) * k_sec[3 * i + 2];
        }
        for (int r = 0; r < 4; r++) {
            float *x = t[r];
            float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], x4 = x[4], x5 = x[5], x6 = x[6], x7 = x[7], xt;
            xt = x0 - x7; x0 = x0 + x7;
            x7 = x1 - x6; x1 = x1 + x6;
            x6 = x2 - x5; x2 = x2 + x5;
            x5 = x3 - x4; x3 = x3 + x4;
            x4 = x0 - x3; x0 = x0 + x3;
            x3 = x1 - x2; x1 = x1 + x2;
            x[0] = x0 + x1;
            x[4] = (x0 - x1) * 0.70710677f;
            x5 = x5 + x6;
            x6 = (x6 + x7) * 0.70710677f;
            x7 = x7 + xt;
            x3 = (x3 + x4) * 0.70710677f;
            x5 = x5 - x7 * 0.198912367f;       /* rotate by PI/8 */
            x7 = x7 + x5 * 0.382683432f;
            x5 = x5 - x7 * 0.198912367f;
            x0 = xt - x6; xt = xt + x6;
            x[1] = (xt + x7) * 0.50979561f;
            x[2] = (x4 + x3) * 0.54119611f;
            x[3] = (x0 - x5) * 0.60134488f;
            x[5] = (x0 + x5) * 0.89997619f;
            x[6] = (x4 - x3) * 1.30656302f;
            x[7] = (xt - x7) * 2.56291556f;
        }
        for (int i = 0; i < 7; i++, y += 4 * 18) {
            y[0 * 18] = t[0][i];
            y[1 * 18] = t[2][i] + t[3][i] + t[3][i + 1];
            y[2 * 18] = t[1][i] + t[1][i + 1];
            y[3 * 18] = t[2][i + 1] + t[3][i] + t[3][i + 1];
        }
        y[0 * 18] = t[0][7];
        y[1 * 18] = t[2][7] + t[3][7];
        y[2 * 18] = t[1][7];
        y[3 * 18] = t[3][7];
    }
}

/* ---- polyphase synthesis, minimp3.d:1300-1406 ---------------------------- */
static float scale_pcm(float sample)                                /* :1300-1303 */
{
    return sample * (1.0f / 32768.0f);
}

static void synth_pair(float *pcm, int nch, const float *z)          /* :1305-1328 */
{
    float a;
    a  = (z[14 * 64] - z[0]) * 29;
    a += (z[1 * 64] + z[13 * 64]) * 213;
    a += (z[12 * 64] - z[2 * 64]) * 459;
    a += (z[3 * 64] + z[11 * 64]) * 2037;
    a += (z[10 * 64] - z[4 * 64]) * 5153;
    a += (z[5 * 64] + z[9 * 64]) * 6574;
    a += (z[8 * 64] - z[6 * 64]) * 37489;
    a +=  z[7 * 64] * 75038;
    pcm[0] = scale_pcm(a);

    z += 2;
    a  = z[14 * 64] * 104;
    a += z[12 * 64] * 1567;
    a += z[10 * 64] * 9727;
    a += z[8 * 64] * 64019;
    a += z[6 * 64] * -9975;
    a += z[4 * 64] * -45;
    a += z[2 * 64] * 146;
    a += z[0 * 64] * -5;
    pcm[16 * nch] = scale_pcm(a);
}

static const float k_win[15 * 16] = {
    -1, 26, -31, 208, 218, 401, -519, 2063, 2000, 4788, -5517, 7134, 5959, 35640, -39336, 74992,
    -1, 24, -35, 202, 222, 347, -581, 2080, 1952, 4425, -5879, 7640, 5288, 33791, -41176, 74856,
    -1, 21, -38, 196, 225, 294, -645, 2087, 1893, 4063, -6237, 8092, 4561, 31947, -43006, 74630,
    -1, 19, -41, 190, 227, 244, -711, 2085, 1822, 3705, -6589, 8492, 3776, 30112, -44821, 74313,
    -1, 17, -45, 183, 228, 197, -779, 2075, 1739, 3351, -6935, 8840, 2935, 28289, -46617, 73908,
    -1, 16, -49, 176, 228, 153, -848, 2057, 1644, 3004, -7271, 9139, 2037, 26482, -48390, 73415,
    -2, 14, -53, 169, 227, 111, -919, 2032, 1535, 2663, -7597, 9389, 1082, 24694, -50137, 72835,
    -2, 13, -58, 161, 224, 72, -991, 2001, 1414, 2330, -7910, 9592, 70, 22929, -51853, 72169,
    -2, 11, -63, 154, 221, 36, -1064, 1962, 1280, 2006, -8209, 9750, -998, 21189, -53534, 71420,
    -2, 10, -68, 147, 215, 2, -1137, 1919, 1131, 1692, -8491, 9863, -2122, 19478, -55178, 70590,
    -3, 9, -73, 139, 208, -29, -1210, 1870, 970, 1388, -8755, 9935, -3300, 17799, -56778, 69679,
    -3, 8, -79, 132, 200, -57, -1283, 1817, 794, 1095, -8998, 9966, -4533, 16155, -58333, 68692,
    -4, 7, -85, 125, 189, -83, -1356, 1759, 605, 814, -9219, 9959, -5818, 14548, -59838, 67629,
    -4, 7, -91, 117, 177, -106, -1428, 1698, 402, 545, -9416, 9916, -7154, 12980, -61289, 66494,
    -5, 6, -97, 111, 163, -127, -1498, 1634, 185, 288, -9585, 9838, -8540, 11455, -62684, 65290 };

/* One call = two time slots of both channels, :1330-1406.  `lins` is the
 * 64-float-per-row line buffer; rows 0..14 are history, row 15 and 16 are
 * (partly) filled here. */
static void synth(float *xl, float *dstl, int nch, float *lins)
{
    float *xr = xl + 576 * (nch - 1);
    float *dstr = dstl + (nch - 1);
    float *zlin = lins + 15 * 64;
    const float *w = k_win;

    zlin[4 * 15]     = xl[18 * 16];
    zlin[4 * 15 + 1] = xr[18 * 16];
    zlin[4 * 15 + 2] = xl[0];
    zlin[4 * 15 + 3] = xr[0];

    zlin[4 * 31]     = xl[1 + 18 * 16];
    zlin[4 * 31 + 1] = xr[1 + 18 * 16];
    zlin[4 * 31 + 2] = xl[1];
    zlin[4 * 31 + 3] = xr[1];

    synth_pair(dstr, nch, lins + 4 * 15 + 1);
    synth_pair(dstr + 32 * nch, nch, lins + 4 * 15 + 64 + 1);
    synth_pair(dstl, nch, lins + 4 * 15);
    synth_pair(dstl + 32 * nch, nch, lins + 4 * 15 + 64);

    for (int i = 14; i >= 0; i--) {
        float a[4], b[4];

        zlin[4 * i]     = xl[18 * (31 - i)];
        zlin[4 * i + 1] = xr[18 * (31 - i)];
        zlin[4 * i + 2] = xl[1 + 18 * (31 - i)];
        zlin[4 * i + 3] = xr[1 + 18 * (31 - i)];
        zlin[4 * (i + 16)]     = xl[1 + 18 * (1 + i)];
        zlin[4 * (i + 16) + 1] = xr[1 + 18 * (1 + i)];
        zlin[4 * (i - 16) + 2] = xl[18 * (1 + i)];
        zlin[4 * (i - 16) + 3] = xr[18 * (1 + i)];

        /* the S0(0) S2(1) S1(2) S2(3) S1(4) S2(5) S1(6) S2(7) ladder of :1388-1395 */
        for (int k = 0; k < 8; k++) {
            float w0 = *w++;
            float w1 = *w++;
            const float *vz = &zlin[4 * i - k * 64];
            const float *vy = &zlin[4 * i - (15 - k) * 64];
            for (int j = 0; j < 4; j++) {
                if (k == 0) {
                    b[j] = vz[j] * w1 + vy[j] * w0;
                    a[j] = vz[j] * w0 - vy[j] * w1;
                } else if (k & 1) {
                    b[j] += vz[j] * w1 + vy[j] * w0;
                    a[j] += vy[j] * w1 - vz[j] * w0;
                } else {
                    b[j] += vz[j] * w1 + vy[j] * w0;
                    a[j] += vz[j] * w0 - vy[j] * w1;
                }
            }
        }

        dstr[(15 - i) * nch] = scale_pcm(a[1]);
        dstr[(17 + i) * nch] = scale_pcm(b[1]);
        dstl[(15 - i) * nch] = scale_pcm(a[0]);
        dstl[(17 + i) * nch] = scale_pcm(b[0]);
        dstr[(47 - i) * nch] = scale_pcm(a[3]);
        dstr[(49 + i) * nch] = scale_pcm(b[3]);
        dstl[(47 - i) * nch] = scale_pcm(a[2]);
        dstl[(49 + i) * nch] = scale_pcm(b[2]);
    }
}

/* :1408-1434.  lins must hold (15 + nbands) * 64 floats. */
void afgo_mp3_synth_granule(float *qmf_state, float *grbuf, int nbands, int nch, float *pcm, float *lins)
{
    for (int c = 0; c < nch; c++)
        afgo_mp3_dct2(grbuf + 576 * c, nbands);

    memcpy(lins, qmf_state, sizeof(float) * 15 * 64);

    for (int i = 0; i < nbands; i += 2)
        synth(grbuf + i, pcm + 32 * nch * i, nch, lins + i * 64);

    if (nch == 1) {
        for (int i = 0; i < 15 * 64; i += 2)
            qmf_state[i] = lins[nbands * 64 + i];
    } else {
        memcpy(qmf_state, lins + nbands * 64, sizeof(float) * 15 * 64);
    }
}

/* ---- one granule: L3_decode tail (:1215-1229) + synth (:1553) ------------- */
void afgo_mp3_granule(afgo_mp3_state *st, float *coef, const uint32_t *flags, int nch, float *pcm)
{
    /* The reference's line buffer is an uninitialised stack local (mp3dec_scratch_t, minimp3.d:1497);
     * a few of its slots (lanes 2,3 of the last row, and odd lanes for mono) are copied into
     * qmf_state before ever being written and are overwritten before ever being read.  Zeroed here
     * so that the saved state is deterministic. */
    float lins[(18 + 15) * 64];
    memset(lins, 0, sizeof(lins));
    for (int c = 0; c < nch; c++) {
        unsigned block_type = flags[c] & 3u;
        unsigned n_long = (flags[c] >> 8) & 0xffu;
        int aa_bands = (int)((flags[c] >> 16) & 0xffu) - 1;
        float *g = coef + 576 * c;
        if (flags[c] & 0x80000000u) continue;             /* AFG_MP3_SUBBAND: Layer I/II samples go to the synthesis as they are */
        afgo_mp3_antialias(g, aa_bands);
        afgo_mp3_imdct_gr(g, st->mdct_overlap[c], block_type, n_long);
        afgo_mp3_change_sign(g);
    }
    afgo_mp3_synth_granule(st->qmf_state, coef, 18, nch, pcm, lins);
}

/* ---- whole batch ---------------------------------------------------------- */
void afgo_mp3_transform(uint32_t n_streams, const uint32_t *ngr, const uint8_t *nch,
                        const float *coef, const uint32_t *flags, float *pcm,
                        afgo_mp3_state *states)
{
    uint64_t blk = 0;
    for (uint32_t s = 0; s < n_streams; s++) {
        afgo_mp3_state st;
        memset(&st, 0, sizeof(st));                       /* minimp3.d:1509 */
        int c = nch[s];
        for (uint32_t g = 0; g < ngr[s]; g++) {
            float grbuf[2 * 576];
            memset(grbuf, 0, sizeof(grbuf));              /* minimp3.d:1551 */
            memcpy(grbuf, coef + blk * 576, sizeof(float) * 576 * (size_t)c);
            afgo_mp3_granule(&st, grbuf, flags + blk, c, pcm + blk * 576);
            blk += (uint64_t)c;
        }
        if (states) states[s] = st;
    }
}
