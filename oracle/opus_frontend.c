/*
 * opus_frontend.c -- CPU restatement of the reference's Opus front-end, CELT-only path.
 * TEST INFRASTRUCTURE ONLY (see afg_oracle.h).
 *
 * Everything the reference does to an Ogg Opus file ahead of the transform seam (dopus.d:3680), for whole files in
 * memory: Ogg pages / lacing and the OpusHead / OpusTags packets (dopus.d:7793-7829, :8120-8193; gain from the header
 * and R128_TRACK_GAIN :8011-8059, :1311-1316), packet framing (ff_opus_parse_packet :1081-1258), the range decoder
 * (:809-1034, :6254-6272), and the CELT frame decoder up to the denormalised coefficients: coarse / fine / final
 * energy (:2128-2216), tf changes (:2218-2243), bit allocation (:2245-2575), PVQ decoding with spreading rotation,
 * band splitting and folding (:2577-3266), post-filter parameters (:3380-3418), anti-collapse (:3420-3470), the band
 * loop (:3472-3566) and the head of ff_celt_decode_frame (:3568-3678).  What comes out are the transform-stage
 * records of afgo_celt_transform (one afgo_celt_frame + frame_size coefficients per frame and output channel).
 *
 * Not restated: SILK and hybrid packets (:3815-6228) -- a file that holds one is reported as unsupported; multistream
 * mappings (the reference's opusOpen refuses them too, :8164-8169); seeking.  The Ogg layer is this project's own
 * minimal reader (capture pattern, version 0, flag bits, page CRC as parsePageHeader checks them, :7055-7097; lacing; a
 * bad page ends the stream; no resynchronisation, one logical stream), not the reference's OggStream (:6955-7785).
 *
 * Float readings where the D source leaves room (std.math on float arguments): cos / sin / exp2 are evaluated in
 * double and rounded to float; M_SQRT1_2 / M_SQRT2 products in double (they are double enums, :636-637).
 */
#include "afg_oracle.h"
#include "opus_tables.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CELT_SHORT_BLOCKSIZE 120
#define CELT_MAX_LOG_BLOCKS 3
#define CELT_MAX_FRAME_SIZE 960
#define CELT_MAX_BANDS 21
#define CELT_VECTORS 11
#define CELT_ALLOC_STEPS 6
#define CELT_FINE_OFFSET 21
#define CELT_MAX_FINE_BITS 8
#define CELT_QTHETA_OFFSET 4
#define CELT_QTHETA_OFFSET_TWOPHASE 16
#define CELT_POSTFILTER_MINPERIOD 15
#define CELT_ENERGY_SILENCE (-28.0f)
#define SPREAD_NONE 0
#define SPREAD_NORMAL 2
#define SPREAD_AGGRESSIVE 3
#define MAX_FRAME_SIZE 1275
#define MAX_FRAMES 48
#define MAX_PACKET_DUR 5760

#define FFMIN(a, b) ((a) < (b) ? (a) : (b))
#define FFMAX(a, b) ((a) > (b) ? (a) : (b))

static float tabf(const uint32_t *bits, int i) { float f; memcpy(&f, bits + i, 4); return f; }

/* ---- small integer helpers (dopus.d:107, :161-205, :664-669) ---------------------------------------------------- */
static int av_log2(uint32_t v) { int n = 0; while (v >>= 1) n++; return n; }
static int opus_ilog(uint32_t i) { return av_log2(i) + !!i; }
static uint32_t av_mod_uintp2(uint32_t a, uint32_t p) { return a & ((1u << p) - 1); }
static int av_clip(int a, int amin, int amax) { return a < amin ? amin : a > amax ? amax : a; }
static uint32_t av_clip_uintp2(int a, int p) { return (a & ~((1 << p) - 1)) ? (uint32_t)((-a >> 31) & ((1 << p) - 1)) : (uint32_t)a; }
static int ROUND_MUL16(int a, int b) { return (a * b + 16384) >> 15; }

static uint32_t isqrt32(uint32_t a)          /* ff_sqrt (:161-180): floor(sqrt(a)) */
{
    uint32_t r = (uint32_t)sqrt((double)a);
    while ((uint64_t)r * r > a) r--;
    while ((uint64_t)(r + 1) * (r + 1) <= a) r++;
    return r;
}

/* ---- range decoder (dopus.d:571-604, :809-1034, :6254-6272) ---------------------------------------------------- */
typedef struct {
    const uint8_t *buffer;
    uint32_t pos, bytestotal;
    uint8_t curv, bleft;
} gb_t;

typedef struct {
    const uint8_t *position;
    uint32_t bytes, cachelen, cacheval;
} rawbits_t;

typedef struct {
    gb_t gb;
    rawbits_t rb;
    uint32_t range, value, total_read_bits;
} rc_t;

static uint32_t get_bits(gb_t *g, unsigned n)
{
    uint32_t res = 0;
    for (int shift = (int)n - 1; shift >= 0; shift--) {
        if (g->bleft == 0) {
            g->curv = g->pos < g->bytestotal ? g->buffer[g->pos++] : 0;
            g->bleft = 8;
        }
        if (g->curv & 0x80) res |= 1u << shift;
        g->curv = (uint8_t)(g->curv << 1);
        g->bleft--;
    }
    return res;
}

static void rc_normalize(rc_t *rc)
{
    while (rc->range <= 1u << 23) {
        const uint8_t b = (uint8_t)(get_bits(&rc->gb, 8) ^ 0xFF);
        rc->value = ((rc->value << 8) | b) & ((1u << 31) - 1);
        rc->range <<= 8;
        rc->total_read_bits += 8;
    }
}

static void rc_init(rc_t *rc, const uint8_t *data, int size)
{
    memset(rc, 0, sizeof(*rc));
    rc->gb.buffer = data;
    rc->gb.bytestotal = (uint32_t)size;
    rc->range = 128;
    rc->value = 127 - get_bits(&rc->gb, 7);
    rc->total_read_bits = 9;
    rc_normalize(rc);
}

static void raw_init(rc_t *rc, const uint8_t *rightend, uint32_t bytes)
{
    rc->rb.position = rightend;
    rc->rb.bytes = bytes;
    rc->rb.cachelen = 0;
    rc->rb.cacheval = 0;
}

static void rc_update(rc_t *rc, uint32_t scale, uint32_t low, uint32_t high, uint32_t total)
{
    rc->value -= scale * (total - high);
    rc->range = low ? scale * (high - low) : rc->range - scale * (total - high);
    rc_normalize(rc);
}

static uint32_t rc_getsymbol(rc_t *rc, const uint16_t *cdf)
{
    uint32_t k, scale, total, symbol, low, high;
    total = *cdf++;
    scale = rc->range / total;
    symbol = rc->value / scale + 1;
    symbol = total - FFMIN(symbol, total);
    for (k = 0; cdf[k] <= symbol; k++) {}
    high = cdf[k];
    low = k ? cdf[k - 1] : 0;
    rc_update(rc, scale, low, high, total);
    return k;
}

static uint32_t rc_p2model(rc_t *rc, uint32_t bits)
{
    uint32_t k, scale = rc->range >> bits;
    if (rc->value >= scale) {
        rc->value -= scale;
        rc->range -= scale;
        k = 0;
    } else {
        rc->range = scale;
        k = 1;
    }
    rc_normalize(rc);
    return k;
}

static uint32_t rc_tell(const rc_t *rc) { return rc->total_read_bits - (uint32_t)av_log2(rc->range) - 1; }

static uint32_t rc_tell_frac(const rc_t *rc)
{
    uint32_t total_bits = rc->total_read_bits << 3;
    uint32_t rcbuffer = (uint32_t)av_log2(rc->range) + 1;
    uint32_t range = rc->range >> (rcbuffer - 16);
    for (int i = 0; i < 3; i++) {
        range = range * range >> 15;
        const int bit = (int)(range >> 16);
        rcbuffer = rcbuffer << 1 | (uint32_t)bit;
        range >>= bit;
    }
    return total_bits - rcbuffer;
}

static uint32_t getrawbits(rc_t *rc, uint32_t count)
{
    while (rc->rb.bytes && rc->rb.cachelen < count) {
        rc->rb.cacheval |= (uint32_t)*--rc->rb.position << rc->rb.cachelen;
        rc->rb.cachelen += 8;
        rc->rb.bytes--;
    }
    const uint32_t value = av_mod_uintp2(rc->rb.cacheval, count);
    rc->rb.cacheval >>= count;
    rc->rb.cachelen -= count;
    rc->total_read_bits += count;
    return value;
}

static uint32_t rc_unimodel(rc_t *rc, uint32_t size)
{
    uint32_t bits, k, scale, total;
    bits = (uint32_t)opus_ilog(size - 1);
    total = (bits > 8) ? ((size - 1) >> (bits - 8)) + 1 : size;
    scale = rc->range / total;
    k = rc->value / scale + 1;
    k = total - FFMIN(k, total);
    rc_update(rc, scale, k, k + 1, total);
    if (bits > 8) {
        k = k << (bits - 8) | getrawbits(rc, bits - 8);
        return FFMIN(k, size - 1);
    }
    return k;
}

static int rc_laplace(rc_t *rc, uint32_t symbol, int decay)
{
    int value = 0;
    uint32_t scale, low = 0, center;
    scale = rc->range >> 15;
    center = rc->value / scale + 1;
    center = (1u << 15) - FFMIN(center, 1u << 15);
    if (center >= symbol) {
        value++;
        low = symbol;
        symbol = 1 + ((32768 - 32 - symbol) * (uint32_t)(16384 - decay) >> 15);
        while (symbol > 1 && center >= low + 2 * symbol) {
            value++;
            symbol *= 2;
            low += symbol;
            symbol = (((symbol - 2) * (uint32_t)decay) >> 15) + 1;
        }
        if (symbol <= 1) {
            const int distance = (int)((center - low) >> 1);
            value += distance;
            low += 2u * (uint32_t)distance;
        }
        if (center < low + symbol) value *= -1;
        else low += symbol;
    }
    rc_update(rc, scale, low, FFMIN(low + symbol, 32768u), 32768);
    return value;
}

static uint32_t rc_stepmodel(rc_t *rc, int k0)
{
    uint32_t k, scale, symbol, total = (uint32_t)((k0 + 1) * 3 + k0);
    scale = rc->range / total;
    symbol = rc->value / scale + 1;
    symbol = total - FFMIN(symbol, total);
    k = (symbol < (uint32_t)(k0 + 1) * 3) ? symbol / 3 : symbol - (uint32_t)(k0 + 1) * 2;
    rc_update(rc, scale, (k <= (uint32_t)k0) ? 3 * (k + 0) : (k - 1 - (uint32_t)k0) + 3 * (uint32_t)(k0 + 1),
              (k <= (uint32_t)k0) ? 3 * (k + 1) : (k - 0 - (uint32_t)k0) + 3 * (uint32_t)(k0 + 1), total);
    return k;
}

static uint32_t rc_trimodel(rc_t *rc, int qn)
{
    uint32_t k, scale, symbol, total, low, center;
    total = (uint32_t)(((qn >> 1) + 1) * ((qn >> 1) + 1));
    scale = rc->range / total;
    center = rc->value / scale + 1;
    center = total - FFMIN(center, total);
    if (center < total >> 1) {
        k = (isqrt32(8 * center + 1) - 1) >> 1;
        low = k * (k + 1) >> 1;
        symbol = k + 1;
    } else {
        k = (2 * (uint32_t)(qn + 1) - isqrt32(8 * (total - center - 1) + 1)) >> 1;
        low = total - (((uint32_t)qn + 1 - k) * ((uint32_t)qn + 2 - k) >> 1);
        symbol = (uint32_t)qn + 1 - k;
    }
    rc_update(rc, scale, low, low + symbol, total);
    return k;
}

/* ---- CELT frame decoder state (dopus.d:1647-1711) --------------------------------------------------------------- */
typedef struct {
    float energy[CELT_MAX_BANDS];
    float prev_energy[2][CELT_MAX_BANDS];
    uint8_t collapse_masks[CELT_MAX_BANDS];
    int pf_period_new;
    float pf_gains_new[3];
} celt_chan;

typedef struct {
    int output_channels;
    celt_chan frame[2];
    uint32_t seed;
    int coded_channels, framebits, duration, blocks, blocksize, startband, endband, codedbands, anticollapse_bit;
    int intensitystereo, dualstereo, spread, remaining, remaining2;
    int fine_bits[CELT_MAX_BANDS], fine_priority[CELT_MAX_BANDS], pulses[CELT_MAX_BANDS], tf_change[CELT_MAX_BANDS];
    float coeffs[2][CELT_MAX_FRAME_SIZE];
    float scratch[22 * 8];
} celt_ctx;

static int16_t celt_cos(int16_t x)                              /* :2103-2108 */
{
    x = (int16_t)((x * x + 4096) >> 13);
    x = (int16_t)((32767 - x) + ROUND_MUL16(x, (-7651 + ROUND_MUL16(x, (8277 + ROUND_MUL16(-626, x))))));
    return (int16_t)(1 + x);
}

static int celt_log2tan(int isin, int icos)                     /* :2110-2120 */
{
    int lc = opus_ilog((uint32_t)icos), ls = opus_ilog((uint32_t)isin);
    icos <<= 15 - lc;
    isin <<= 15 - ls;
    return (ls << 11) - (lc << 11) + ROUND_MUL16(isin, ROUND_MUL16(isin, -2597) + 7932) - ROUND_MUL16(icos, ROUND_MUL16(icos, -2597) + 7932);
}

static uint32_t celt_rng(celt_ctx *s) { s->seed = 1664525u * s->seed + 1013904223u; return s->seed; }

static void decode_coarse_energy(celt_ctx *s, rc_t *rc)         /* :2128-2176 */
{
    float prev[2] = { 0, 0 };
    float alpha, beta;
    const uint8_t *model;
    if (rc_tell(rc) + 3 <= (uint32_t)s->framebits && rc_p2model(rc, 3)) {
        alpha = 0;
        beta = 1.0f - 4915.0f / 32768.0f;
        model = celt_coarse_energy_dist + (s->duration * 2 + 1) * 42;
    } else {
        alpha = tabf(celt_alpha_coef_bits, s->duration);
        beta = 1.0f - tabf(celt_beta_coef_bits, s->duration);
        model = celt_coarse_energy_dist + (s->duration * 2 + 0) * 42;
    }
    for (int i = 0; i < CELT_MAX_BANDS; i++) {
        for (int j = 0; j < s->coded_channels; j++) {
            celt_chan *frame = &s->frame[j];
            float value;
            if (i < s->startband || i >= s->endband) {
                frame->energy[i] = 0.0f;
                continue;
            }
            const int available = s->framebits - (int)rc_tell(rc);
            if (available >= 15) {
                const int k = FFMIN(i, 20) << 1;
                value = (float)rc_laplace(rc, (uint32_t)model[k] << 7, model[k + 1] << 6);
            } else if (available >= 2) {
                const int x = (int)rc_getsymbol(rc, celt_model_energy_small);
                value = (float)((x >> 1) ^ -(x & 1));
            } else if (available >= 1) {
                value = -(float)rc_p2model(rc, 1);
            } else {
                value = -1;
            }
            frame->energy[i] = FFMAX(-9.0f, frame->energy[i]) * alpha + prev[j] + value;
            prev[j] += beta * value;
        }
    }
}

static void decode_fine_energy(celt_ctx *s, rc_t *rc)           /* :2178-2195 */
{
    for (int i = s->startband; i < s->endband; i++) {
        if (!s->fine_bits[i]) continue;
        for (int j = 0; j < s->coded_channels; j++) {
            const int q2 = (int)getrawbits(rc, (uint32_t)s->fine_bits[i]);
            const float offset = (q2 + 0.5f) * (float)(1 << (14 - s->fine_bits[i])) / 16384.0f - 0.5f;
            s->frame[j].energy[i] += offset;
        }
    }
}

static void decode_final_energy(celt_ctx *s, rc_t *rc, int bits_left)   /* :2197-2216 */
{
    for (int priority = 0; priority < 2; priority++) {
        for (int i = s->startband; i < s->endband && bits_left >= s->coded_channels; i++) {
            if (s->fine_priority[i] != priority || s->fine_bits[i] >= CELT_MAX_FINE_BITS) continue;
            for (int j = 0; j < s->coded_channels; j++) {
                const int q2 = (int)getrawbits(rc, 1);
                const float offset = (q2 - 0.5f) * (float)(1 << (14 - s->fine_bits[i] - 1)) / 16384.0f;
                s->frame[j].energy[i] += offset;
                bits_left--;
            }
        }
    }
}

#define TF_SELECT(d, t, sel, c) celt_tf_select[(((d) * 2 + (t)) * 2 + (sel)) * 2 + (c)]

static void decode_tf_changes(celt_ctx *s, rc_t *rc, int transient)     /* :2218-2243 */
{
    int diff = 0, tf_select = 0, tf_changed = 0;
    int bits = transient ? 2 : 4;
    int consumed = (int)rc_tell(rc);
    const int tf_select_bit = (s->duration != 0 && consumed + bits + 1 <= s->framebits);
    for (int i = s->startband; i < s->endband; i++) {
        if (consumed + bits + tf_select_bit <= s->framebits) {
            diff ^= (int)rc_p2model(rc, (uint32_t)bits);
            consumed = (int)rc_tell(rc);
            tf_changed |= diff;
        }
        s->tf_change[i] = diff;
        bits = transient ? 4 : 5;
    }
    if (tf_select_bit && TF_SELECT(s->duration, transient, 0, tf_changed) != TF_SELECT(s->duration, transient, 1, tf_changed))
        tf_select = (int)rc_p2model(rc, 1);
    for (int i = s->startband; i < s->endband; i++) s->tf_change[i] = TF_SELECT(s->duration, transient, tf_select, s->tf_change[i]);
}

static void decode_allocation(celt_ctx *s, rc_t *rc)            /* :2245-2575 */
{
    int cap[CELT_MAX_BANDS], boost[CELT_MAX_BANDS], threshold[CELT_MAX_BANDS], bits1[CELT_MAX_BANDS], bits2[CELT_MAX_BANDS],
        trim_offset[CELT_MAX_BANDS];
    int skip_startband = s->startband, dynalloc = 6, alloctrim = 5, extrabits = 0;
    int skip_bit = 0, intensitystereo_bit = 0, dualstereo_bit = 0;
    int remaining, bandbits, low, high, total, done, totalbits, consumed, i, j;
    const int C = s->coded_channels, LM = s->duration;

    consumed = (int)rc_tell(rc);
    s->spread = SPREAD_NORMAL;
    if (consumed + 4 <= s->framebits) s->spread = (int)rc_getsymbol(rc, celt_model_spread);

    for (i = 0; i < CELT_MAX_BANDS; i++)
        cap[i] = (celt_static_caps[(LM * 2 + (C - 1)) * 21 + i] + 64) * celt_freq_range[i] << (C - 1) << LM >> 2;

    totalbits = s->framebits << 3;
    consumed = (int)rc_tell_frac(rc);
    for (i = s->startband; i < s->endband; i++) {
        int quanta, band_dynalloc;
        boost[i] = 0;
        quanta = celt_freq_range[i] << (C - 1) << LM;
        quanta = FFMIN(quanta << 3, FFMAX(6 << 3, quanta));
        band_dynalloc = dynalloc;
        while (consumed + (band_dynalloc << 3) < totalbits && boost[i] < cap[i]) {
            const int add = (int)rc_p2model(rc, (uint32_t)band_dynalloc);
            consumed = (int)rc_tell_frac(rc);
            if (!add) break;
            boost[i] += quanta;
            totalbits -= quanta;
            band_dynalloc = 1;
        }
        if (boost[i]) dynalloc = FFMAX(2, dynalloc - 1);
    }

    if (consumed + (6 << 3) <= totalbits) alloctrim = (int)rc_getsymbol(rc, celt_model_alloc_trim);

    totalbits = (s->framebits << 3) - (int)rc_tell_frac(rc) - 1;
    s->anticollapse_bit = 0;
    if (s->blocks > 1 && LM >= 2 && totalbits >= ((LM + 2) << 3)) s->anticollapse_bit = 1 << 3;
    totalbits -= s->anticollapse_bit;

    if (totalbits >= 1 << 3) skip_bit = 1 << 3;
    totalbits -= skip_bit;

    if (C == 2) {
        intensitystereo_bit = celt_log2_frac[s->endband - s->startband];
        if (intensitystereo_bit <= totalbits) {
            totalbits -= intensitystereo_bit;
            if (totalbits >= 1 << 3) {
                dualstereo_bit = 1 << 3;
                totalbits -= 1 << 3;
            }
        } else {
            intensitystereo_bit = 0;
        }
    }

    for (i = s->startband; i < s->endband; i++) {
        const int trim = alloctrim - 5 - LM;
        const int band = celt_freq_range[i] * (s->endband - i - 1);
        const int duration = LM + 3;
        const int scale = duration + C - 1;
        threshold[i] = FFMAX(3 * celt_freq_range[i] << duration >> 4, C << 3);
        trim_offset[i] = trim * (band << scale) >> 6;
        if (celt_freq_range[i] << LM == 1) trim_offset[i] -= C << 3;
    }

    low = 1;
    high = CELT_VECTORS - 1;
    while (low <= high) {
        const int center = (low + high) >> 1;
        done = total = 0;
        for (i = s->endband - 1; i >= s->startband; i--) {
            bandbits = celt_freq_range[i] * celt_static_alloc[center * 21 + i] << (C - 1) << LM >> 2;
            if (bandbits) bandbits = FFMAX(0, bandbits + trim_offset[i]);
            bandbits += boost[i];
            if (bandbits >= threshold[i] || done) {
                done = 1;
                total += FFMIN(bandbits, cap[i]);
            } else if (bandbits >= C << 3) {
                total += C << 3;
            }
        }
        if (total > totalbits) high = center - 1;
        else low = center + 1;
    }
    high = low--;

    for (i = s->startband; i < s->endband; i++) {
        bits1[i] = celt_freq_range[i] * celt_static_alloc[low * 21 + i] << (C - 1) << LM >> 2;
        bits2[i] = high >= CELT_VECTORS ? cap[i] : celt_freq_range[i] * celt_static_alloc[high * 21 + i] << (C - 1) << LM >> 2;
        if (bits1[i]) bits1[i] = FFMAX(0, bits1[i] + trim_offset[i]);
        if (bits2[i]) bits2[i] = FFMAX(0, bits2[i] + trim_offset[i]);
        if (low) bits1[i] += boost[i];
        bits2[i] += boost[i];
        if (boost[i]) skip_startband = i;
        bits2[i] = FFMAX(0, bits2[i] - bits1[i]);
    }

    low = 0;
    high = 1 << CELT_ALLOC_STEPS;
    for (i = 0; i < CELT_ALLOC_STEPS; i++) {
        const int center = (low + high) >> 1;
        done = total = 0;
        for (j = s->endband - 1; j >= s->startband; j--) {
            bandbits = bits1[j] + (center * bits2[j] >> CELT_ALLOC_STEPS);
            if (bandbits >= threshold[j] || done) {
                done = 1;
                total += FFMIN(bandbits, cap[j]);
            } else if (bandbits >= C << 3) {
                total += C << 3;
            }
        }
        if (total > totalbits) high = center;
        else low = center;
    }

    done = total = 0;
    for (i = s->endband - 1; i >= s->startband; i--) {
        bandbits = bits1[i] + (low * bits2[i] >> CELT_ALLOC_STEPS);
        if (bandbits >= threshold[i] || done) done = 1;
        else bandbits = (bandbits >= C << 3) ? C << 3 : 0;
        bandbits = FFMIN(bandbits, cap[i]);
        s->pulses[i] = bandbits;
        total += bandbits;
    }

    for (s->codedbands = s->endband;; s->codedbands--) {
        int allocation;
        j = s->codedbands - 1;
        if (j == skip_startband) {
            totalbits += skip_bit;
            break;
        }
        remaining = totalbits - total;
        bandbits = remaining / (celt_freq_bands[j + 1] - celt_freq_bands[s->startband]);
        remaining -= bandbits * (celt_freq_bands[j + 1] - celt_freq_bands[s->startband]);
        allocation = s->pulses[j] + bandbits * celt_freq_range[j] + FFMAX(0, remaining - (celt_freq_bands[j] - celt_freq_bands[s->startband]));
        if (allocation >= FFMAX(threshold[j], (C + 1) << 3)) {
            if (rc_p2model(rc, 1)) break;
            total += 1 << 3;
            allocation -= 1 << 3;
        }
        total -= s->pulses[j];
        if (intensitystereo_bit) {
            total -= intensitystereo_bit;
            intensitystereo_bit = celt_log2_frac[j - s->startband];
            total += intensitystereo_bit;
        }
        total += s->pulses[j] = (allocation >= C << 3) ? C << 3 : 0;
    }

    s->intensitystereo = 0;
    s->dualstereo = 0;
    if (intensitystereo_bit) s->intensitystereo = s->startband + (int)rc_unimodel(rc, (uint32_t)(s->codedbands + 1 - s->startband));
    if (s->intensitystereo <= s->startband) totalbits += dualstereo_bit;
    else if (dualstereo_bit) s->dualstereo = (int)rc_p2model(rc, 1);

    remaining = totalbits - total;
    bandbits = remaining / (celt_freq_bands[s->codedbands] - celt_freq_bands[s->startband]);
    remaining -= bandbits * (celt_freq_bands[s->codedbands] - celt_freq_bands[s->startband]);
    for (i = s->startband; i < s->codedbands; i++) {
        const int bits = FFMIN(remaining, celt_freq_range[i]);
        s->pulses[i] += bits + bandbits * celt_freq_range[i];
        remaining -= bits;
    }

    for (i = s->startband; i < s->codedbands; i++) {
        const int N = celt_freq_range[i] << LM;
        const int prev_extra = extrabits;
        s->pulses[i] += extrabits;
        if (N > 1) {
            int dof, temp, offset, fine_bits, max_bits;
            extrabits = FFMAX(0, s->pulses[i] - cap[i]);
            s->pulses[i] -= extrabits;
            dof = N * C + (C == 2 && N > 2 && !s->dualstereo && i < s->intensitystereo);
            temp = dof * (celt_log_freq_range[i] + (LM << 3));
            offset = (temp >> 1) - dof * CELT_FINE_OFFSET;
            if (N == 2) offset += dof << 1;
            if (s->pulses[i] + offset < 2 * (dof << 3)) offset += temp >> 2;
            else if (s->pulses[i] + offset < 3 * (dof << 3)) offset += temp >> 3;
            fine_bits = (s->pulses[i] + offset + (dof << 2)) / (dof << 3);
            max_bits = FFMIN((s->pulses[i] >> 3) >> (C - 1), CELT_MAX_FINE_BITS);
            max_bits = FFMAX(max_bits, 0);
            s->fine_bits[i] = av_clip(fine_bits, 0, max_bits);
            s->fine_priority[i] = (s->fine_bits[i] * (dof << 3) >= s->pulses[i] + offset);
            s->pulses[i] -= s->fine_bits[i] << (C - 1) << 3;
        } else {
            extrabits = FFMAX(0, s->pulses[i] - (C << 3));
            s->pulses[i] -= extrabits;
            s->fine_bits[i] = 0;
            s->fine_priority[i] = 1;
        }
        if (extrabits > 0) {
            int fineextra = FFMIN(extrabits >> (C + 2), CELT_MAX_FINE_BITS - s->fine_bits[i]);
            s->fine_bits[i] += fineextra;
            fineextra <<= C + 2;
            s->fine_priority[i] = (fineextra >= extrabits - prev_extra);
            extrabits -= fineextra;
        }
    }
    s->remaining = extrabits;

    for (; i < s->endband; i++) {
        s->fine_bits[i] = s->pulses[i] >> (C - 1) >> 3;
        s->pulses[i] = 0;
        s->fine_priority[i] = s->fine_bits[i] < 1;
    }
}

/* ---- PVQ (dopus.d:2577-2915) ---------------------------------------------------------------------------------- */
static int bits2pulses(const uint8_t *cache, int bits)
{
    int low = 0, high = cache[0];
    bits--;
    for (int i = 0; i < 6; i++) {
        const int center = (low + high + 1) >> 1;
        if (cache[center] >= bits) high = center;
        else low = center;
    }
    return (bits - (low == 0 ? -1 : cache[low]) <= cache[high] - bits) ? low : high;
}

static int pulses2bits(const uint8_t *cache, int pulses) { return (pulses == 0) ? 0 : cache[pulses] + 1; }

static void exp_rotation1(float *X, uint32_t len, uint32_t stride, float c, float s)
{
    float *Xptr = X;
    int i;
    for (i = 0; i < (int)(len - stride); i++) {
        const float x1 = Xptr[0], x2 = Xptr[stride];
        Xptr[stride] = c * x2 + s * x1;
        *Xptr++ = c * x1 - s * x2;
    }
    i = (int)(len - 2 * stride - 1);
    if (i < 0) return;
    Xptr = &X[i];
    for (; i >= 0; i--) {
        const float x1 = Xptr[0], x2 = Xptr[stride];
        Xptr[stride] = c * x2 + s * x1;
        *Xptr-- = c * x1 - s * x2;
    }
}

static void exp_rotation(float *X, uint32_t len, uint32_t stride, uint32_t K, int spread)
{
    uint32_t stride2 = 0;
    if (2 * K >= len || spread == SPREAD_NONE) return;
    const float gain = (float)len / (float)(len + (20 - 5 * (uint32_t)spread) * K);
    const float theta = (float)(3.14159265358979323846264338327950288L * gain * gain / 4);      /* PI is a real */
    const float c = (float)cos((double)theta), s = (float)sin((double)theta);
    if (len >= stride << 3) {
        stride2 = 1;
        while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
    }
    len /= stride;
    for (uint32_t i = 0; i < stride; i++) {
        if (stride2) exp_rotation1(X + i * len, len, stride2, s, c);
        exp_rotation1(X + i * len, len, 1, c, s);
    }
}

static uint32_t extract_collapse_mask(const int *iy, uint32_t N, uint32_t B)
{
    if (B <= 1) return 1;
    const int N0 = (int)(N / B);
    uint32_t mask = 0;
    for (uint32_t i = 0; i < B; i++)
        for (int j = 0; j < N0; j++) mask |= (uint32_t)(iy[i * (uint32_t)N0 + (uint32_t)j] != 0) << i;
    return mask;
}

static void renormalize_vector(float *X, int N, float gain)
{
    float g = 1e-15f;
    for (int i = 0; i < N; i++) g += X[i] * X[i];
    g = gain / sqrtf(g);
    for (int i = 0; i < N; i++) X[i] *= g;
}

static void stereo_merge(float *X, float *Y, float mid, int N)
{
    float xp = 0, side = 0, E[2], gain[2];
    for (int i = 0; i < N; i++) {
        xp += X[i] * Y[i];
        side += Y[i] * Y[i];
    }
    xp *= mid;
    const float mid2 = mid;
    E[0] = mid2 * mid2 + side - 2 * xp;
    E[1] = mid2 * mid2 + side + 2 * xp;
    if (E[0] < 6e-4f || E[1] < 6e-4f) {
        for (int i = 0; i < N; i++) Y[i] = X[i];
        return;
    }
    gain[0] = 1.0f / sqrtf(E[0]);
    gain[1] = 1.0f / sqrtf(E[1]);
    for (int i = 0; i < N; i++) {
        const float v0 = mid * X[i], v1 = Y[i];
        X[i] = gain[0] * (v0 - v1);
        Y[i] = gain[1] * (v0 + v1);
    }
}

static void interleave_hadamard(float *tmp, float *X, int N0, int stride, int hadamard)
{
    const int N = N0 * stride;
    if (hadamard) {
        const uint8_t *ordery = celt_hadamard_ordery + stride - 2;
        for (int i = 0; i < stride; i++)
            for (int j = 0; j < N0; j++) tmp[j * stride + i] = X[ordery[i] * N0 + j];
    } else {
        for (int i = 0; i < stride; i++)
            for (int j = 0; j < N0; j++) tmp[j * stride + i] = X[i * N0 + j];
    }
    for (int i = 0; i < N; i++) X[i] = tmp[i];
}

static void deinterleave_hadamard(float *tmp, float *X, int N0, int stride, int hadamard)
{
    const int N = N0 * stride;
    if (hadamard) {
        const uint8_t *ordery = celt_hadamard_ordery + stride - 2;
        for (int i = 0; i < stride; i++)
            for (int j = 0; j < N0; j++) tmp[ordery[i] * N0 + j] = X[j * stride + i];
    } else {
        for (int i = 0; i < stride; i++)
            for (int j = 0; j < N0; j++) tmp[i * N0 + j] = X[j * stride + i];
    }
    for (int i = 0; i < N; i++) X[i] = tmp[i];
}

static void haar1(float *X, int N0, int stride)
{
    N0 >>= 1;
    for (int i = 0; i < stride; i++)
        for (int j = 0; j < N0; j++) {
            const float x0 = X[stride * (2 * j + 0) + i], x1 = X[stride * (2 * j + 1) + i];
            X[stride * (2 * j + 0) + i] = (float)((double)(x0 + x1) * 0.70710678118654752440);     /* M_SQRT1_2 is a double enum */
            X[stride * (2 * j + 1) + i] = (float)((double)(x0 - x1) * 0.70710678118654752440);
        }
}

static int compute_qn(int N, int b, int offset, int pulse_cap, int dualstereo)
{
    int N2 = 2 * N - 1;
    if (dualstereo && N == 2) N2--;
    int qb = FFMIN(FFMIN(b - pulse_cap - (4 << 3), (b + N2 * offset) / N2), 8 << 3);
    return (qb < (1 << 3 >> 1)) ? 1 : ((celt_qn_exp2[qb & 0x7] >> (14 - (qb >> 3))) + 1) >> 1 << 1;
}

#define PVQ_ROW(r) (celt_pvq_u + celt_pvq_u_row[r])
#define PVQ_U(n, k) (PVQ_ROW(FFMIN(n, k))[FFMAX(n, k)])
#define PVQ_V(n, k) (PVQ_U(n, k) + PVQ_U(n, (k) + 1))

static uint64_t cwrsi(uint32_t N, uint32_t K, uint32_t i, int *y)   /* :2809-2893 */
{
    uint64_t norm = 0;
    uint32_t p;
    int s, val, k0;
    while (N > 2) {
        uint32_t q;
        if (K >= N) {
            const uint32_t *row = PVQ_ROW(N);
            p = row[K + 1];
            s = -(i >= p ? 1 : 0);
            i -= p & (uint32_t)s;
            k0 = (int)K;
            q = row[N];
            if (q > i) {
                K = N;
                do {
                    p = PVQ_ROW(--K)[N];
                } while (p > i);
            } else {
                for (p = row[K]; p > i; p = row[K]) K--;
            }
            i -= p;
            val = (k0 - (int)K + s) ^ s;
            norm += (uint64_t)(val * val);
            *y++ = val;
        } else {
            p = PVQ_ROW(K)[N];
            q = PVQ_ROW(K + 1)[N];
            if (p <= i && i < q) {
                i -= p;
                *y++ = 0;
            } else {
                s = -(i >= q ? 1 : 0);
                i -= q & (uint32_t)s;
                k0 = (int)K;
                do p = PVQ_ROW(--K)[N];
                while (p > i);
                i -= p;
                val = (k0 - (int)K + s) ^ s;
                norm += (uint64_t)(val * val);
                *y++ = val;
            }
        }
        N--;
    }
    p = 2 * K + 1;
    s = -(i >= p ? 1 : 0);
    i -= p & (uint32_t)s;
    k0 = (int)K;
    K = (i + 1) / 2;
    if (K) i -= 2 * K - 1;
    val = (k0 - (int)K + s) ^ s;
    norm += (uint64_t)(val * val);
    *y++ = val;
    s = -(int)i;
    val = ((int)K + s) ^ s;
    norm += (uint64_t)(val * val);
    *y = val;
    return norm;
}

static uint32_t alg_unquant(rc_t *rc, float *X, uint32_t N, uint32_t K, int spread, uint32_t blocks, float gain)
{
    int y[176];
    const uint32_t idx = rc_unimodel(rc, PVQ_V(N, K));
    gain /= sqrtf((float)cwrsi(N, K, idx, y));                  /* celt_decode_pulses returns float (:2895) */
    for (uint32_t i = 0; i < N; i++) X[i] = gain * (float)y[i];
    exp_rotation(X, N, blocks, K, spread);
    return extract_collapse_mask(y, N, blocks);
}

static uint32_t decode_band(celt_ctx *s, rc_t *rc, const int band, float *X, float *Y, int N, int b, uint32_t blocks, float *lowband,
                            int duration, float *lowband_out, int level, float gain, float *lowband_scratch, int fill)   /* :2917-3266 */
{
    const uint8_t *cache;
    int dualstereo, split;
    int imid = 0, iside = 0;
    const uint32_t N0 = (uint32_t)N;
    int N_B, N_B0;
    int B0 = (int)blocks;
    int time_divide = 0, recombine = 0, inv = 0;
    float mid = 0, side = 0;
    const int longblocks = (B0 == 1);
    uint32_t cm = 0;

    N_B0 = N_B = N / (int)blocks;
    split = dualstereo = (Y != NULL);

    if (N == 1) {
        float *x = X;
        for (int i = 0; i <= dualstereo; i++) {
            int sign = 0;
            if (s->remaining2 >= 1 << 3) {
                sign = (int)getrawbits(rc, 1);
                s->remaining2 -= 1 << 3;
                b -= 1 << 3;
            }
            x[0] = sign ? -1.0f : 1.0f;
            x = Y;
        }
        if (lowband_out) lowband_out[0] = X[0];
        return 1;
    }

    if (!dualstereo && level == 0) {
        int tf_change = s->tf_change[band];
        if (tf_change > 0) recombine = tf_change;
        if (lowband && (recombine || ((N_B & 1) == 0 && tf_change < 0) || B0 > 1)) {
            for (int j = 0; j < N; j++) lowband_scratch[j] = lowband[j];
            lowband = lowband_scratch;
        }
        for (int k = 0; k < recombine; k++) {
            if (lowband) haar1(lowband, N >> k, 1 << k);
            fill = celt_bit_interleave[fill & 0xF] | celt_bit_interleave[fill >> 4] << 2;
        }
        blocks >>= recombine;
        N_B <<= recombine;
        while ((N_B & 1) == 0 && tf_change < 0) {
            if (lowband) haar1(lowband, N_B, (int)blocks);
            fill |= fill << blocks;
            blocks <<= 1;
            N_B >>= 1;
            time_divide++;
            tf_change++;
        }
        B0 = (int)blocks;
        N_B0 = N_B;
        if (B0 > 1 && lowband) deinterleave_hadamard(s->scratch, lowband, N_B >> recombine, B0 << recombine, longblocks);
    }

    cache = celt_cache_bits + celt_cache_index[(duration + 1) * CELT_MAX_BANDS + band];
    if (!dualstereo && duration >= 0 && b > cache[cache[0]] + 12 && N > 2) {
        N >>= 1;
        Y = X + N;
        split = 1;
        duration -= 1;
        if (blocks == 1) fill = (fill & 1) | (fill << 1);
        blocks = (blocks + 1) >> 1;
    }

    if (split) {
        int qn, itheta = 0, mbits, sbits, delta, qalloc, pulse_cap, offset, orig_fill, tell;
        pulse_cap = celt_log_freq_range[band] + duration * 8;
        offset = (pulse_cap >> 1) - (dualstereo && N == 2 ? CELT_QTHETA_OFFSET_TWOPHASE : CELT_QTHETA_OFFSET);
        qn = (dualstereo && band >= s->intensitystereo) ? 1 : compute_qn(N, b, offset, pulse_cap, dualstereo);
        tell = (int)rc_tell_frac(rc);
        if (qn != 1) {
            if (dualstereo && N > 2) itheta = (int)rc_stepmodel(rc, qn / 2);
            else if (dualstereo || B0 > 1) itheta = (int)rc_unimodel(rc, (uint32_t)qn + 1);
            else itheta = (int)rc_trimodel(rc, qn);
            itheta = itheta * 16384 / qn;
        } else if (dualstereo) {
            inv = (b > 2 << 3 && s->remaining2 > 2 << 3) ? (int)rc_p2model(rc, 2) : 0;
            itheta = 0;
        }
        qalloc = (int)rc_tell_frac(rc) - tell;
        b -= qalloc;

        orig_fill = fill;
        if (itheta == 0) {
            imid = 32767;
            iside = 0;
            fill = (int)av_mod_uintp2((uint32_t)fill, blocks);
            delta = -16384;
        } else if (itheta == 16384) {
            imid = 0;
            iside = 32767;
            fill &= ((1 << blocks) - 1) << blocks;
            delta = 16384;
        } else {
            imid = celt_cos((int16_t)itheta);
            iside = celt_cos((int16_t)(16384 - itheta));
            delta = ROUND_MUL16((N - 1) << 7, celt_log2tan(iside, imid));
        }
        mid = (float)imid / 32768.0f;
        side = (float)iside / 32768.0f;

        if (N == 2 && dualstereo) {
            int sign = 0;
            float tmp;
            float *x2, *y2;
            mbits = b;
            sbits = (itheta != 0 && itheta != 16384) ? 1 << 3 : 0;
            mbits -= sbits;
            const int c = (itheta > 8192);
            s->remaining2 -= qalloc + sbits;
            x2 = c ? Y : X;
            y2 = c ? X : Y;
            if (sbits) sign = (int)getrawbits(rc, 1);
            sign = 1 - 2 * sign;
            cm = decode_band(s, rc, band, x2, NULL, N, mbits, blocks, lowband, duration, lowband_out, level, gain, lowband_scratch, orig_fill);
            y2[0] = (float)-sign * x2[1];
            y2[1] = (float)sign * x2[0];
            X[0] *= mid;
            X[1] *= mid;
            Y[0] *= side;
            Y[1] *= side;
            tmp = X[0];
            X[0] = tmp - Y[0];
            Y[0] = tmp + Y[0];
            tmp = X[1];
            X[1] = tmp - Y[1];
            Y[1] = tmp + Y[1];
        } else {
            float *next_lowband2 = NULL, *next_lowband_out1 = NULL;
            int next_level = 0, rebalance;
            if (B0 > 1 && !dualstereo && (itheta & 0x3fff)) {
                if (itheta > 8192) delta -= delta >> (4 - duration);
                else delta = FFMIN(0, delta + (N << 3 >> (5 - duration)));
            }
            mbits = av_clip((b - delta) / 2, 0, b);
            sbits = b - mbits;
            s->remaining2 -= qalloc;
            if (lowband && !dualstereo) next_lowband2 = lowband + N;
            if (dualstereo) next_lowband_out1 = lowband_out;
            else next_level = level + 1;
            rebalance = s->remaining2;
            if (mbits >= sbits) {
                cm = decode_band(s, rc, band, X, NULL, N, mbits, blocks, lowband, duration, next_lowband_out1, next_level,
                                 dualstereo ? 1.0f : (gain * mid), lowband_scratch, fill);
                rebalance = mbits - (rebalance - s->remaining2);
                if (rebalance > 3 << 3 && itheta != 0) sbits += rebalance - (3 << 3);
                cm |= decode_band(s, rc, band, Y, NULL, N, sbits, blocks, next_lowband2, duration, NULL, next_level, gain * side, NULL,
                                  fill >> blocks) << ((B0 >> 1) & (dualstereo - 1));
            } else {
                cm = decode_band(s, rc, band, Y, NULL, N, sbits, blocks, next_lowband2, duration, NULL, next_level, gain * side, NULL,
                                 fill >> blocks) << ((B0 >> 1) & (dualstereo - 1));
                rebalance = sbits - (rebalance - s->remaining2);
                if (rebalance > 3 << 3 && itheta != 16384) mbits += rebalance - (3 << 3);
                cm |= decode_band(s, rc, band, X, NULL, N, mbits, blocks, lowband, duration, next_lowband_out1, next_level,
                                  dualstereo ? 1.0f : (gain * mid), lowband_scratch, fill);
            }
        }
    } else {
        uint32_t q = (uint32_t)bits2pulses(cache, b);
        uint32_t curr_bits = (uint32_t)pulses2bits(cache, (int)q);
        s->remaining2 -= (int)curr_bits;
        while (s->remaining2 < 0 && q > 0) {
            s->remaining2 += (int)curr_bits;
            curr_bits = (uint32_t)pulses2bits(cache, (int)--q);
            s->remaining2 -= (int)curr_bits;
        }
        if (q != 0) {
            cm = alg_unquant(rc, X, (uint32_t)N, (q < 8) ? q : (8 + (q & 7)) << ((q >> 3) - 1), s->spread, blocks, gain);
        } else {
            const uint32_t cm_mask = (1u << blocks) - 1;
            fill &= (int)cm_mask;
            if (!fill) {
                for (int j = 0; j < N; j++) X[j] = 0.0f;
            } else {
                if (!lowband) {
                    for (int j = 0; j < N; j++) X[j] = (float)(((int32_t)celt_rng(s)) >> 20);
                    cm = cm_mask;
                } else {
                    for (int j = 0; j < N; j++) X[j] = lowband[j] + ((celt_rng(s) & 0x8000) ? 1.0f / 256 : -1.0f / 256);
                    cm = (uint32_t)fill;
                }
                renormalize_vector(X, N, gain);
            }
        }
    }

    if (dualstereo) {
        if (N != 2) stereo_merge(X, Y, mid, N);
        if (inv)
            for (int j = 0; j < N; j++) Y[j] *= -1;
    } else if (level == 0) {
        if (B0 > 1) interleave_hadamard(s->scratch, X, N_B >> recombine, B0 << recombine, longblocks);
        N_B = N_B0;
        blocks = (uint32_t)B0;
        for (int k = 0; k < time_divide; k++) {
            blocks >>= 1;
            N_B <<= 1;
            cm |= cm >> blocks;
            haar1(X, N_B, (int)blocks);
        }
        for (int k = 0; k < recombine; k++) {
            cm = celt_bit_deinterleave[cm];
            haar1(X, (int)(N0 >> k), 1 << k);
        }
        blocks <<= recombine;
        if (lowband_out) {
            const float n = sqrtf((float)N0);
            for (uint32_t j = 0; j < N0; j++) lowband_out[j] = n * X[j];
        }
        cm = av_mod_uintp2(cm, blocks);
    }
    return cm;
}

static void denormalize(celt_ctx *s, celt_chan *frame, float *data)    /* :3268-3279 */
{
    for (int i = s->startband; i < s->endband; i++) {
        float *dst = data + (celt_freq_bands[i] << s->duration);
        const float norm = (float)exp2((double)(frame->energy[i] + tabf(celt_mean_energy_bits, i)));
        for (int j = 0; j < celt_freq_range[i] << s->duration; j++) dst[j] *= norm;
    }
}

static int parse_postfilter(celt_ctx *s, rc_t *rc, int consumed)        /* :3380-3418 */
{
    static const float taps[3][3] = { { 0.3066406250f, 0.2170410156f, 0.1296386719f }, { 0.4638671875f, 0.2680664062f, 0.0f },
                                      { 0.7998046875f, 0.1000976562f, 0.0f } };
    memset(s->frame[0].pf_gains_new, 0, sizeof(s->frame[0].pf_gains_new));
    memset(s->frame[1].pf_gains_new, 0, sizeof(s->frame[1].pf_gains_new));
    if (s->startband == 0 && consumed + 16 <= s->framebits) {
        const int has_postfilter = (int)rc_p2model(rc, 1);
        if (has_postfilter) {
            const int octave = (int)rc_unimodel(rc, 6);
            const int period = (16 << octave) + (int)getrawbits(rc, (uint32_t)(4 + octave)) - 1;
            const float gain = 0.09375f * (float)(getrawbits(rc, 3) + 1);
            const int tapset = (rc_tell(rc) + 2 <= (uint32_t)s->framebits) ? (int)rc_getsymbol(rc, celt_model_tapset) : 0;
            for (int i = 0; i < 2; i++) {
                celt_chan *frame = &s->frame[i];
                frame->pf_period_new = FFMAX(period, CELT_POSTFILTER_MINPERIOD);
                frame->pf_gains_new[0] = gain * taps[tapset][0];
                frame->pf_gains_new[1] = gain * taps[tapset][1];
                frame->pf_gains_new[2] = gain * taps[tapset][2];
            }
        }
        consumed = (int)rc_tell(rc);
    }
    return consumed;
}

static void process_anticollapse(celt_ctx *s, celt_chan *frame, float *X)   /* :3420-3470 */
{
    for (int i = s->startband; i < s->endband; i++) {
        int renormalize = 0;
        float prev[2];
        const int depth = (1 + s->pulses[i]) / (celt_freq_range[i] << s->duration);
        const float thresh = (float)exp2((double)(float)(-1.0 - (double)(0.125f * (float)depth)));      /* exp2f(-1.0 - 0.125f*depth) */
        const float sqrt_1 = 1.0f / sqrtf((float)(celt_freq_range[i] << s->duration));
        float *xptr = X + (celt_freq_bands[i] << s->duration);
        prev[0] = frame->prev_energy[0][i];
        prev[1] = frame->prev_energy[1][i];
        if (s->coded_channels == 1) {
            celt_chan *frame1 = &s->frame[1];
            prev[0] = FFMAX(prev[0], frame1->prev_energy[0][i]);
            prev[1] = FFMAX(prev[1], frame1->prev_energy[1][i]);
        }
        float Ediff = frame->energy[i] - FFMIN(prev[0], prev[1]);
        Ediff = FFMAX(0, Ediff);
        float r = (float)exp2((double)(1 - Ediff));
        if (s->duration == 3) r = (float)((double)r * 1.41421356237309504880);                         /* M_SQRT2 is a double enum */
        r = FFMIN(thresh, r) * sqrt_1;
        for (int k = 0; k < 1 << s->duration; k++) {
            if (!(frame->collapse_masks[i] & 1 << k)) {
                for (int j = 0; j < celt_freq_range[i]; j++) xptr[(j << s->duration) + k] = (celt_rng(s) & 0x8000) ? r : -r;
                renormalize = 1;
            }
        }
        if (renormalize) renormalize_vector(xptr, celt_freq_range[i] << s->duration, 1.0f);
    }
}

static void decode_bands(celt_ctx *s, rc_t *rc)                 /* :3472-3566 */
{
    float lowband_scratch[8 * 22];
    float norm[2 * 8 * 100];
    const int totalbits = (s->framebits << 3) - s->anticollapse_bit;
    int update_lowband = 1, lowband_offset = 0;
    memset(s->coeffs, 0, sizeof(s->coeffs));
    memset(norm, 0, sizeof(norm));                             /* (`= void` in the reference: read only after being written) */
    for (int i = s->startband; i < s->endband; i++) {
        const int band_offset = celt_freq_bands[i] << s->duration;
        const int band_size = celt_freq_range[i] << s->duration;
        float *X = s->coeffs[0] + band_offset;
        float *Y = (s->coded_channels == 2) ? s->coeffs[1] + band_offset : NULL;
        const int consumed = (int)rc_tell_frac(rc);
        float *norm2 = norm + 8 * 100;
        int effective_lowband = -1;
        uint32_t cm[2];
        int b;
        if (i != s->startband) s->remaining -= consumed;
        s->remaining2 = totalbits - consumed - 1;
        if (i <= s->codedbands - 1) {
            const int curr_balance = s->remaining / FFMIN(3, s->codedbands - i);
            b = (int)av_clip_uintp2(FFMIN(s->remaining2 + 1, s->pulses[i] + curr_balance), 14);
        } else {
            b = 0;
        }
        if (celt_freq_bands[i] - celt_freq_range[i] >= celt_freq_bands[s->startband] && (update_lowband || lowband_offset == 0)) lowband_offset = i;
        if (lowband_offset != 0 && (s->spread != SPREAD_AGGRESSIVE || s->blocks > 1 || s->tf_change[i] < 0)) {
            int foldstart, foldend;
            effective_lowband = FFMAX(celt_freq_bands[s->startband], celt_freq_bands[lowband_offset] - celt_freq_range[i]);
            foldstart = lowband_offset;
            while (celt_freq_bands[--foldstart] > effective_lowband) {}
            foldend = lowband_offset - 1;
            while (celt_freq_bands[++foldend] < effective_lowband + celt_freq_range[i]) {}
            cm[0] = cm[1] = 0;
            for (int j = foldstart; j < foldend; j++) {
                cm[0] |= s->frame[0].collapse_masks[j];
                cm[1] |= s->frame[s->coded_channels - 1].collapse_masks[j];
            }
        } else {
            cm[0] = cm[1] = (1u << s->blocks) - 1;
        }
        if (s->dualstereo && i == s->intensitystereo) {
            s->dualstereo = 0;
            for (int j = celt_freq_bands[s->startband] << s->duration; j < band_offset; j++) norm[j] = (norm[j] + norm2[j]) / 2;
        }
        if (s->dualstereo) {
            cm[0] = decode_band(s, rc, i, X, NULL, band_size, b / 2, (uint32_t)s->blocks,
                                effective_lowband != -1 ? norm + (effective_lowband << s->duration) : NULL, s->duration, norm + band_offset, 0,
                                1.0f, lowband_scratch, (int)cm[0]);
            cm[1] = decode_band(s, rc, i, Y, NULL, band_size, b / 2, (uint32_t)s->blocks,
                                effective_lowband != -1 ? norm2 + (effective_lowband << s->duration) : NULL, s->duration, norm2 + band_offset, 0,
                                1.0f, lowband_scratch, (int)cm[1]);
        } else {
            cm[0] = decode_band(s, rc, i, X, Y, band_size, b, (uint32_t)s->blocks,
                                effective_lowband != -1 ? norm + (effective_lowband << s->duration) : NULL, s->duration, norm + band_offset, 0,
                                1.0f, lowband_scratch, (int)(cm[0] | cm[1]));
            cm[1] = cm[0];
        }
        s->frame[0].collapse_masks[i] = (uint8_t)cm[0];
        s->frame[s->coded_channels - 1].collapse_masks[i] = (uint8_t)cm[1];
        s->remaining += s->pulses[i] + consumed;
        update_lowband = (b > band_size << 3);
    }
}

/* One CELT frame up to the transform seam (head and tail bookkeeping of ff_celt_decode_frame, :3568-3731, without the
 * per-channel iMDCT / post-filter / de-emphasis loop at :3672-3702, which is the transform stage).  Returns 0, or -1. */
typedef struct {
    int frame_size, blocks, silence;
    int pf_period_new[2];
    float pf_gains_new[2][3];
    float imdct_scale;
} celt_frame_out;

static int celt_decode_frame_head(celt_ctx *s, rc_t *rc, int coded_channels, int frame_size, int startband, int endband, celt_frame_out *fo)
{
    int consumed, silence = 0, transient = 0, anticollapse = 0;
    float imdct_scale = 1.0f;
    if (coded_channels != 1 && coded_channels != 2) return -1;
    if (startband < 0 || startband > endband || endband > CELT_MAX_BANDS) return -1;
    s->coded_channels = coded_channels;
    s->startband = startband;
    s->endband = endband;
    s->framebits = (int)rc->rb.bytes * 8;
    s->duration = av_log2((uint32_t)(frame_size / CELT_SHORT_BLOCKSIZE));
    if (s->duration > CELT_MAX_LOG_BLOCKS || frame_size != CELT_SHORT_BLOCKSIZE * (1 << s->duration)) return -1;
    if (!s->output_channels) s->output_channels = coded_channels;
    memset(s->frame[0].collapse_masks, 0, sizeof(s->frame[0].collapse_masks));
    memset(s->frame[1].collapse_masks, 0, sizeof(s->frame[1].collapse_masks));

    consumed = (int)rc_tell(rc);
    if (consumed >= s->framebits) silence = 1;
    else if (consumed == 1) silence = (int)rc_p2model(rc, 15);
    if (silence) {
        consumed = s->framebits;
        rc->total_read_bits += (uint32_t)(s->framebits - (int)rc_tell(rc));
    }
    consumed = parse_postfilter(s, rc, consumed);
    if (s->duration != 0 && consumed + 3 <= s->framebits) transient = (int)rc_p2model(rc, 3);
    s->blocks = transient ? 1 << s->duration : 1;
    s->blocksize = frame_size / s->blocks;

    if (coded_channels == 1)
        for (int i = 0; i < CELT_MAX_BANDS; i++) s->frame[0].energy[i] = FFMAX(s->frame[0].energy[i], s->frame[1].energy[i]);

    decode_coarse_energy(s, rc);
    decode_tf_changes(s, rc, transient);
    decode_allocation(s, rc);
    decode_fine_energy(s, rc);
    decode_bands(s, rc);
    if (s->anticollapse_bit) anticollapse = (int)getrawbits(rc, 1);
    decode_final_energy(s, rc, s->framebits - (int)rc_tell(rc));

    for (int i = 0; i < s->coded_channels; i++) {
        celt_chan *frame = &s->frame[i];
        if (anticollapse) process_anticollapse(s, frame, s->coeffs[i]);
        denormalize(s, frame, s->coeffs[i]);
    }
    if (s->output_channels < s->coded_channels) {
        for (int i = 0; i < frame_size; i++) s->coeffs[0][i] += s->coeffs[1][i] * 1.0f;     /* vector_fmac_scalar(.., 1.0f, ..) */
        imdct_scale = 0.5f;
    } else if (s->output_channels > s->coded_channels) {
        memcpy(s->coeffs[1], s->coeffs[0], (size_t)frame_size * sizeof(float));
    }
    if (silence) {
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < CELT_MAX_BANDS; j++) s->frame[i].energy[j] = CELT_ENERGY_SILENCE;
        memset(s->coeffs, 0, sizeof(s->coeffs));
    }

    fo->frame_size = frame_size;
    fo->blocks = s->blocks;
    fo->silence = silence;
    fo->imdct_scale = imdct_scale;
    for (int i = 0; i < 2; i++) {
        fo->pf_period_new[i] = s->frame[i].pf_period_new;
        memcpy(fo->pf_gains_new[i], s->frame[i].pf_gains_new, sizeof(fo->pf_gains_new[i]));
    }

    /* (the transform stage runs here in the reference) */

    if (coded_channels == 1) memcpy(s->frame[1].energy, s->frame[0].energy, sizeof(s->frame[0].energy));
    for (int i = 0; i < 2; i++) {
        celt_chan *frame = &s->frame[i];
        if (!transient) {
            memcpy(frame->prev_energy[1], frame->prev_energy[0], sizeof(frame->prev_energy[0]));
            memcpy(frame->prev_energy[0], frame->energy, sizeof(frame->prev_energy[0]));
        } else {
            for (int j = 0; j < CELT_MAX_BANDS; j++) frame->prev_energy[0][j] = FFMIN(frame->prev_energy[0][j], frame->energy[j]);
        }
        for (int j = 0; j < s->startband; j++) {
            frame->prev_energy[0][j] = CELT_ENERGY_SILENCE;
            frame->energy[j] = 0.0f;
        }
        for (int j = s->endband; j < CELT_MAX_BANDS; j++) {
            frame->prev_energy[0][j] = CELT_ENERGY_SILENCE;
            frame->energy[j] = 0.0f;
        }
    }
    s->seed = rc->range;
    return 0;
}

static void celt_flush(celt_ctx *s)                             /* ff_celt_flush, :3733-3760: the front-end half */
{
    for (int i = 0; i < 2; i++) {
        celt_chan *frame = &s->frame[i];
        for (int j = 0; j < CELT_MAX_BANDS; j++) frame->prev_energy[0][j] = frame->prev_energy[1][j] = CELT_ENERGY_SILENCE;
        memset(frame->energy, 0, sizeof(frame->energy));
        frame->pf_period_new = 0;
        memset(frame->pf_gains_new, 0, sizeof(frame->pf_gains_new));
    }
    s->seed = 0;
}

/* ---- packet framing (ff_opus_parse_packet, dopus.d:1048-1258; self_delimiting = false) --------------------------- */
typedef struct {
    int packet_size, data_size, code, stereo, vbr, config, frame_count;
    int frame_offset[MAX_FRAMES], frame_size[MAX_FRAMES];
    int frame_duration, mode, bandwidth;
} opus_packet;

static int lacing_16bit(const uint8_t **ptr, const uint8_t *end)
{
    if (*ptr >= end) return -1;
    int val = *(*ptr)++;
    if (val >= 252) {
        if (*ptr >= end) return -1;
        val += 4 * *(*ptr)++;
    }
    return val;
}

static int lacing_full(const uint8_t **ptr, const uint8_t *end)
{
    int val = 0;
    for (;;) {
        if (*ptr >= end || val > 0x7fffffff - 254) return -1;
        const int next = *(*ptr)++;
        val += next;
        if (next < 255) break;
        --val;
    }
    return val;
}

static int parse_packet(opus_packet *pkt, const uint8_t *buf, int buf_size)
{
    const uint8_t *ptr = buf, *end = buf + buf_size;
    int padding = 0, frame_bytes, i;
    memset(pkt, 0, sizeof(*pkt));
    if (buf_size < 1) return -1;
    i = *ptr++;
    pkt->code = i & 0x3;
    pkt->stereo = (i >> 2) & 0x1;
    pkt->config = (i >> 3) & 0x1F;
    if (pkt->code >= 2 && buf_size < 2) goto fail;
    switch (pkt->code) {
    case 0:
        pkt->frame_count = 1;
        pkt->vbr = 0;
        frame_bytes = (int)(end - ptr);
        if (frame_bytes > MAX_FRAME_SIZE) goto fail;
        pkt->frame_offset[0] = (int)(ptr - buf);
        pkt->frame_size[0] = frame_bytes;
        break;
    case 1:
        pkt->frame_count = 2;
        pkt->vbr = 0;
        frame_bytes = (int)(end - ptr);
        if ((frame_bytes & 1) != 0 || (frame_bytes >> 1) > MAX_FRAME_SIZE) goto fail;
        pkt->frame_offset[0] = (int)(ptr - buf);
        pkt->frame_size[0] = frame_bytes >> 1;
        pkt->frame_offset[1] = pkt->frame_offset[0] + pkt->frame_size[0];
        pkt->frame_size[1] = frame_bytes >> 1;
        break;
    case 2:
        pkt->frame_count = 2;
        pkt->vbr = 1;
        frame_bytes = lacing_16bit(&ptr, end);
        if (frame_bytes < 0) goto fail;
        pkt->frame_offset[0] = (int)(ptr - buf);
        pkt->frame_size[0] = frame_bytes;
        frame_bytes = (int)(end - ptr - pkt->frame_size[0]);
        if (frame_bytes < 0 || frame_bytes > MAX_FRAME_SIZE) goto fail;
        pkt->frame_offset[1] = pkt->frame_offset[0] + pkt->frame_size[0];
        pkt->frame_size[1] = frame_bytes;
        break;
    default:
        i = *ptr++;
        pkt->frame_count = i & 0x3F;
        padding = (i >> 6) & 0x01;
        pkt->vbr = (i >> 7) & 0x01;
        if (pkt->frame_count == 0 || pkt->frame_count > MAX_FRAMES) goto fail;
        if (padding) {
            padding = lacing_full(&ptr, end);
            if (padding < 0) goto fail;
        }
        if (pkt->vbr) {
            int total_bytes = 0;
            for (i = 0; i < pkt->frame_count - 1; i++) {
                frame_bytes = lacing_16bit(&ptr, end);
                if (frame_bytes < 0) goto fail;
                pkt->frame_size[i] = frame_bytes;
                total_bytes += frame_bytes;
            }
            frame_bytes = (int)(end - ptr - padding);
            if (total_bytes > frame_bytes) goto fail;
            pkt->frame_offset[0] = (int)(ptr - buf);
            for (i = 1; i < pkt->frame_count; i++) pkt->frame_offset[i] = pkt->frame_offset[i - 1] + pkt->frame_size[i - 1];
            pkt->frame_size[pkt->frame_count - 1] = frame_bytes - total_bytes;
        } else {
            frame_bytes = (int)(end - ptr - padding);
            if (frame_bytes < 0) goto fail;                     /* (the reference goes on to a negative frame size and asserts, :582) */
            if (frame_bytes % pkt->frame_count || frame_bytes / pkt->frame_count > MAX_FRAME_SIZE) goto fail;
            frame_bytes /= pkt->frame_count;
            pkt->frame_offset[0] = (int)(ptr - buf);
            pkt->frame_size[0] = frame_bytes;
            for (i = 1; i < pkt->frame_count; i++) {
                pkt->frame_offset[i] = pkt->frame_offset[i - 1] + pkt->frame_size[i - 1];
                pkt->frame_size[i] = frame_bytes;
            }
        }
        break;
    }
    pkt->packet_size = buf_size;
    pkt->data_size = pkt->packet_size - padding;
    pkt->frame_duration = opus_frame_duration[pkt->config];
    if (pkt->frame_duration * pkt->frame_count > MAX_PACKET_DUR) goto fail;
    if (pkt->config < 12) {
        pkt->mode = 0;                                          /* SILK */
        pkt->bandwidth = pkt->config >> 2;
    } else if (pkt->config < 16) {
        pkt->mode = 1;                                          /* hybrid */
        pkt->bandwidth = 3 + (pkt->config >= 14 ? 1 : 0);
    } else {
        pkt->mode = 2;                                          /* CELT */
        pkt->bandwidth = (pkt->config - 16) >> 2;
        if (pkt->bandwidth) ++pkt->bandwidth;                   /* skip medium band */
    }
    return 0;
fail:
    memset(pkt, 0, sizeof(*pkt));
    return -1;
}

/* ---- Ogg (this project's minimal reader) + file drive (dopus.d:7793-7829, :8062-8193, :6563-6712) ---------------- */
typedef struct {
    const uint8_t *d;
    size_t n, pos;
    /* current page */
    int nseg, seg, flags;
    const uint8_t *lacing, *body;
    uint64_t granule;
} ogg_rd;

static uint32_t ogg_crc(const uint8_t *p, size_t n)             /* the Ogg page checksum: CRC-32, polynomial 0x04c11db7, no reflection */
{
    static uint32_t tab[256];
    static int have;
    if (!have) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t r = i << 24;
            for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
            tab[i] = r;
        }
        have = 1;
    }
    uint32_t c = 0;
    for (size_t i = 0; i < n; i++) {
        const uint8_t b = (i >= 22 && i < 26) ? 0 : p[i];      /* the checksum field counts as zero */
        c = (c << 8) ^ tab[((c >> 24) ^ b) & 0xff];
    }
    return c;
}

/* a valid page at p (n bytes left)?  returns its total size, or 0 */
static size_t ogg_page_at(const uint8_t *p, size_t n)
{
    if (n < 27 || memcmp(p, "OggS", 4) || p[4] != 0 || (p[5] & ~0x07)) return 0;
    const int nseg = p[26];
    if (27 + (size_t)nseg > n) return 0;
    size_t len = 27 + (size_t)nseg;
    for (int i = 0; i < nseg; i++) len += p[27 + i];
    if (len > n) return 0;
    const uint32_t crc = p[22] | (p[23] << 8) | (p[24] << 16) | ((uint32_t)p[25] << 24);
    if (ogg_crc(p, len) != crc) return 0;
    return len;
}

static int ogg_next_page(ogg_rd *o)
{
    const size_t len = ogg_page_at(o->d + o->pos, o->n - o->pos);
    if (!len) return 0;
    const uint8_t *h = o->d + o->pos;
    const int nseg = h[26];
    o->flags = h[5];
    o->granule = 0;
    for (int i = 7; i >= 0; i--) o->granule = (o->granule << 8) | h[6 + i];
    o->nseg = nseg;
    o->seg = 0;
    o->lacing = h + 27;
    o->body = h + 27 + nseg;
    o->pos += len;
    return 1;
}

/* next packet into buf (packets may span pages); returns length, or -1 at the end of the data */
static long ogg_next_packet(ogg_rd *o, uint8_t **buf, size_t *cap)
{
    size_t len = 0;
    for (;;) {
        if (o->lacing == NULL || o->seg >= o->nseg) {
            if (!ogg_next_page(o)) return -1;
            if (o->nseg == 0) continue;
        }
        while (o->seg < o->nseg) {
            const int l = o->lacing[o->seg++];
            if (len + (size_t)l > *cap) {
                *cap = (len + (size_t)l) * 2 + 256;
                *buf = (uint8_t *)realloc(*buf, *cap);
                if (!*buf) return -1;
            }
            memcpy(*buf + len, o->body, (size_t)l);
            o->body += l;
            len += (size_t)l;
            if (l < 255) return (long)len;
        }
    }
}

static int16_t comment_gain(const uint8_t *c, size_t n)          /* OpusFileCtx.getGain, :8011-8059 (c points behind "OpusTags") */
{
    if (n < 4) return 0;
    uint32_t len = c[0] | (c[1] << 8) | (c[2] << 16) | ((uint32_t)c[3] << 24);
    if (len > n || n - len < 4) return 0;
    size_t cpos = 4 + len;
    if (cpos >= n || n - cpos < 4) return 0;
    uint32_t count = c[cpos] | (c[cpos + 1] << 8) | (c[cpos + 2] << 16) | ((uint32_t)c[cpos + 3] << 24);
    cpos += 4;
    while (count > 0 && cpos + 4 <= n) {
        len = c[cpos] | (c[cpos + 1] << 8) | (c[cpos + 2] << 16) | ((uint32_t)c[cpos + 3] << 24);
        cpos += 4;
        if (cpos > n || n - cpos < len) break;
        {
            const uint8_t *cmt = c + cpos;
            size_t cl = len;
            static const char name[] = "R128_TRACK_GAIN=";
            const size_t nl = sizeof(name) - 1;
            while (cl && cmt[0] <= ' ') { cmt++; cl--; }
            while (cl && cmt[cl - 1] <= ' ') cl--;
            if (cl > nl) {
                int ok = 1;
                for (size_t x = 0; x < nl; x++) {
                    char ch = (char)cmt[x];
                    if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
                    if (ch != name[x]) { ok = 0; break; }
                }
                if (ok) {
                    int neg = 0, v = 0;
                    cmt += nl;
                    cl -= nl;
                    if (cl && cmt[0] == '-') { neg = 1; cmt++; cl--; }
                    else if (cl && cmt[0] == '+') { cmt++; cl--; }
                    if (cl == 0) v = -1;
                    while (cl) {
                        const int ch = cmt[0];
                        cmt++;
                        cl--;
                        if (ch < '0' || ch > '9') { v = -1; break; }
                        v = v * 10 + ch - '0';
                        if ((neg && v > 32768) || (!neg && v > 32767)) { v = -1; break; }
                    }
                    if (v >= 0) return (int16_t)(neg ? -v : v);
                }
            }
        }
        cpos += len;
        --count;
    }
    return 0;
}

static int push_frame(afgo_opus_file *f, size_t *cap_frames, const celt_ctx *s, const celt_frame_out *fo)
{
    const int C = f->channels;
    if (f->n_frames == *cap_frames) {
        *cap_frames = *cap_frames ? *cap_frames * 2 : 64;
        f->frames = (afgo_celt_frame *)realloc(f->frames, *cap_frames * sizeof(afgo_celt_frame));
        if (!f->frames) return -1;
    }
    const uint64_t need = f->n_coeffs + (uint64_t)fo->frame_size * (uint64_t)C;
    f->coeffs = (float *)realloc(f->coeffs, (size_t)need * sizeof(float) + 16);
    if (!f->coeffs) return -1;
    afgo_celt_frame *fr = &f->frames[f->n_frames++];
    memset(fr, 0, sizeof(*fr));
    /* the record of channel 0; channel c of the same frame: coef_off + c * frame_size, out_off + c */
    fr->coef_off = f->n_coeffs;
    fr->out_off = f->pcm_frames * (uint64_t)C;
    fr->out_stride = (uint32_t)C;
    fr->frame_size = (uint16_t)fo->frame_size;
    fr->blocks = (uint8_t)fo->blocks;
    fr->imdct_scale = fo->imdct_scale;
    fr->pf_period_new = fo->pf_period_new[0];                   /* parse_postfilter writes both channels alike */
    memcpy(fr->pf_gains_new, fo->pf_gains_new[0], sizeof(fr->pf_gains_new));
    for (int c = 0; c < C; c++) {
        memcpy(f->coeffs + f->n_coeffs, s->coeffs[c], (size_t)fo->frame_size * sizeof(float));
        f->n_coeffs += (uint64_t)fo->frame_size;
    }
    f->pcm_frames += (uint64_t)fo->frame_size;
    return 0;
}

/* Whole file.  0 = opened (out->error says whether a later packet failed to parse, where the reference's read sets its
 * error flag, stream.d:452-456); -1 = not an Ogg Opus file the reference would open; -2 = out of memory; -3 = a file
 * the reference opens but this restatement does not cover (a SILK or hybrid packet). */
int afgo_opus_decode_file(const uint8_t *data, size_t size, afgo_opus_file *out)
{
    memset(out, 0, sizeof(*out));
    if (!data || size < 47) return -1;
    ogg_rd og;
    memset(&og, 0, sizeof(og));
    og.d = data;
    og.n = size;
    uint8_t *pkt = NULL;
    size_t cap = 0;
    long len = ogg_next_packet(&og, &pkt, &cap);
    /* opus_header, :7791-7818: the first packet sits on a BOS page, OpusHead of at least 19 bytes, version nibble 0
     * (the reference does not compare the "OpusHead" magic; a non-Opus BOS packet then fails at the tags check) */
    if (len < 19 || !(og.flags & 0x02) || (pkt[8] & 0xF0) != 0) { free(pkt); return -1; }
    const int channels = pkt[9];
    const int preskip = pkt[10] | (pkt[11] << 8);
    int gain_i = pkt[16] | (pkt[17] << 8);                      /* AV_RL16 returns a ushort (:516, :1311): read unsigned */
    const int map_type = pkt[18];
    len = ogg_next_packet(&og, &pkt, &cap);
    if (len < 8 || memcmp(pkt, "OpusTags", 8)) { free(pkt); return -1; }
    const int16_t cmtgain = len >= 12 ? comment_gain(pkt + 8, (size_t)len - 8) : 0;
    /* ff_opus_parse_extradata (:1270-1405) and opusOpen (:8164-8169): one stream, one or two channels */
    if (!channels || (map_type == 0 && channels > 2)) { free(pkt); return -1; }
    if (map_type != 0) { free(pkt); return -1; }               /* (mappings 1/2/255 with one stream: not restated, refused) */
    gain_i += cmtgain;
    if (gain_i < -32768) gain_i = -32768;
    else if (gain_i > 32767) gain_i = 32767;
    out->channels = channels;
    out->preskip = preskip;
    out->gain_i = gain_i;
    out->gain = gain_i ? (float)exp2(3.32192809488736234787 * (gain_i / (20.0 * 256))) : 1.0f;     /* ff_exp10, :66-69, :1316 */
    {   /* findLastPage (:7195-7340): granule position of the last valid page */
        uint64_t last = 0;
        size_t p = 0, l;
        while ((l = ogg_page_at(data + p, size - p)) != 0) {
            last = 0;
            for (int i = 7; i >= 0; i--) last = (last << 8) | data[p + 6 + i];
            p += l;
        }
        if (last < (uint64_t)preskip) { free(pkt); return -1; } /* :8157 */
        out->declared_frames = (int64_t)(last - (uint64_t)preskip);
    }

    celt_ctx *s = (celt_ctx *)calloc(1, sizeof(celt_ctx));
    if (!s) { free(pkt); return -2; }
    s->output_channels = channels;
    celt_flush(s);                                              /* ff_celt_init ends in ff_celt_flush (:3805) */
    size_t cap_frames = 0;
    int rc_err = 0, first = 1;
    {   /* opusOpen's last checks, and: a file with a SILK or hybrid packet anywhere is outside this restatement as a whole */
        ogg_rd scan = og;
        uint8_t *sp = NULL;
        size_t scap = 0;
        long sl;
        while (!rc_err && (sl = ogg_next_packet(&scan, &sp, &scap)) >= 0) {
            if (first && scan.granule < (uint64_t)preskip) rc_err = -1;     /* :8155: the page the first audio packet came from */
            else if (sl > 0 && (sp[0] >> 3) < 16) rc_err = -3;
            first = 0;
        }
        free(sp);
        if (first) rc_err = -1;                                 /* no audio packet: opusOpen's loadPacket fails (:8147) */
        if (rc_err) { free(s); free(pkt); return rc_err; }
    }
    while ((len = ogg_next_packet(&og, &pkt, &cap)) >= 0) {
        opus_packet op;
        if (parse_packet(&op, pkt, (int)len) < 0) { out->error = 1; break; }       /* opus_decode_packet fails: readFrame reports it */
        /* readFrame decodes into two halves of 960*3*2 floats (:7961, :8079-8082): a stereo packet of more than 60 ms
         * overruns them in the reference (undefined there); reported as a decode error here */
        if (channels == 2 && op.frame_count * op.frame_duration > 960 * 3) { out->error = 1; break; }
        for (int i = 0; i < op.frame_count; i++) {
            rc_t rc;
            celt_frame_out fo;
            const uint8_t *fd = pkt + op.frame_offset[i];
            const int fsize = op.frame_size[i];
            rc_init(&rc, fd, fsize);                            /* opus_decode_frame, :6359-6366 */
            raw_init(&rc, fd + fsize, (uint32_t)fsize);         /* :6446 */
            /* cannot fail for a CELT-only packet: channels, bands and frame size all come from the TOC tables */
            if (celt_decode_frame_head(s, &rc, op.stereo + 1, op.frame_duration, 0, celt_band_end[op.bandwidth], &fo) < 0) { rc_err = -1; break; }
            if (push_frame(out, &cap_frames, s, &fo) < 0) { rc_err = -2; break; }
        }
        if (rc_err) break;
    }
    free(s);
    free(pkt);
    if (rc_err) { afgo_opus_file_free(out); return rc_err; }
    return 0;
}

void afgo_opus_file_free(afgo_opus_file *f)
{
    free(f->frames);
    free(f->coeffs);
    memset(f, 0, sizeof(*f));
}
