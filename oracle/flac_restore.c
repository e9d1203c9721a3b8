/*
 * oracle/flac_restore.c -- CPU restatement of the FLAC LPC restore + channel
 * decorrelation.  TEST INFRASTRUCTURE ONLY (see afg_oracle.h).  PARITY
 * UNPINNED by reference vectors (the reference has none); pinned by the
 * encode->decode round trips of tests/test_oracle_flac.py.
 *
 * Follows source/audioformats/drflac.d of the reference:
 *   prediction_32 / _64           :1060-1140
 *   sample = residual + pred      :1235, :1264-1269
 *   warm-up, fixed coefficients   :1396-1441
 *   decorrelate/shift/interleave  :2885-2941
 * and stream.d:505-511 for the int32 -> float conversion.
 *
 * D integer arithmetic wraps on overflow; C signed overflow is undefined, so
 * every wrapping add/multiply/shift below is done on unsigned operands.
 */
#include "afg_oracle.h"
#include <stdlib.h>
#include <string.h>

/* int32 accumulator, wrapping; arithmetic shift of the final sum. */
int32_t afgo_flac_prediction_32(unsigned order, int shift, const int16_t *coef, const int32_t *p)
{
    uint32_t acc = 0;
    for (unsigned j = order; j >= 1; j--)            /* the fall-through switch runs from tap `order` down to 1 */
        acc += (uint32_t)(int32_t)coef[j - 1] * (uint32_t)p[-(int)j];
    return (int32_t)acc >> shift;
}

/* int64 accumulator; truncation to int32 after the shift. */
int32_t afgo_flac_prediction_64(unsigned order, int shift, const int16_t *coef, const int32_t *p)
{
    uint64_t acc = 0;
    for (unsigned j = order; j >= 1; j--)
        acc += (uint64_t)((int64_t)coef[j - 1] * (int64_t)p[-(int)j]);
    return (int32_t)(uint32_t)(uint64_t)((int64_t)acc >> shift);
}

void afgo_flac_restore_subframe(const afgo_flac_subframe *sf, int32_t *s, uint32_t block_size)
{
    /* samples [0, order) are the verbatim warm-up, drflac.d:1406-1410, :1419-1423 */
    for (uint32_t i = sf->order; i < block_size; i++) {
        int32_t pred = sf->use64
            ? afgo_flac_prediction_64(sf->order, sf->shift, sf->coef, s + i)
            : afgo_flac_prediction_32(sf->order, sf->shift, sf->coef, s + i);
        s[i] = (int32_t)((uint32_t)s[i] + (uint32_t)pred);                 /* :1235 */
    }
}

static int32_t shl32(int32_t v, unsigned sh)
{
    return (int32_t)((uint32_t)v << (sh & 31u));      /* D/x86 shift count is taken mod 32 */
}

void afgo_flac_transform(uint64_t n_frames, const afgo_flac_frame *frames,
                         const afgo_flac_subframe *subframes, const int32_t *res,
                         int32_t *out_i32, float *out_f32)
{
    const double factor = 1.0 / 2147483647.0;                            /* stream.d:507 */
    for (uint64_t f = 0; f < n_frames; f++) {
        const afgo_flac_frame *fr = &frames[f];
        uint32_t bs = fr->block_size;
        unsigned C = fr->channels;
        uint64_t sf_base = fr->sf_index;
        int32_t *dec = (int32_t *)malloc(sizeof(int32_t) * (size_t)bs * C);
        if (fr->res16) {                                                     /* int16 rows (a storage format of this project, not the
                                                                                reference's: include/afg.h) widened back */
            const int16_t *r16 = (const int16_t *)res + fr->in_off;
            const size_t row = ((size_t)bs + 7u) & ~(size_t)7u;
            for (unsigned c = 0; c < C; c++)
                for (uint32_t i = 0; i < bs; i++) dec[(size_t)c * bs + i] = r16[(size_t)c * row + i];
        } else {
            memcpy(dec, res + fr->in_off, sizeof(int32_t) * (size_t)bs * C);
        }
        for (unsigned c = 0; c < C; c++)
            afgo_flac_restore_subframe(&subframes[sf_base + c], dec + (size_t)c * bs, bs);

        unsigned unused = 32u - fr->bps;                                   /* :2883 */
        int32_t *o = out_i32 + fr->out_off;
        const afgo_flac_subframe *sf = &subframes[sf_base];
        const int32_t *d0 = dec, *d1 = dec + bs;
        switch (fr->assignment) {
        case AFGO_FLAC_LEFT_SIDE:                                          /* :2886-2897 */
            for (uint32_t i = 0; i < bs; i++) {
                int32_t left = d0[i], side = d1[i];
                int32_t right = (int32_t)((uint32_t)left - (uint32_t)side);
                o[i * 2 + 0] = shl32(left, unused + sf[0].wasted);
                o[i * 2 + 1] = shl32(right, unused + sf[1].wasted);
            }
            break;
        case AFGO_FLAC_RIGHT_SIDE:                                         /* :2899-2909 */
            for (uint32_t i = 0; i < bs; i++) {
                int32_t side = d0[i], right = d1[i];
                int32_t left = (int32_t)((uint32_t)right + (uint32_t)side);
                o[i * 2 + 0] = shl32(left, unused + sf[0].wasted);
                o[i * 2 + 1] = shl32(right, unused + sf[1].wasted);
            }
            break;
        case AFGO_FLAC_MID_SIDE:                                           /* :2911-2920 */
            for (uint32_t i = 0; i < bs; i++) {
                int32_t side = d1[i];
                int32_t mid = (int32_t)(((uint32_t)d0[i] << 1) | (uint32_t)(side & 0x01));
                int32_t l = (int32_t)((uint32_t)mid + (uint32_t)side) >> 1;
                int32_t r = (int32_t)((uint32_t)mid - (uint32_t)side) >> 1;
                o[i * 2 + 0] = shl32(l, unused + sf[0].wasted);
                o[i * 2 + 1] = shl32(r, unused + sf[1].wasted);
            }
            break;
        default:                                                           /* :2922-2940 */
            for (uint32_t i = 0; i < bs; i++)
                for (unsigned c = 0; c < C; c++)
                    o[(size_t)i * C + c] = shl32(dec[(size_t)c * bs + i], unused + sf[c].wasted);
            break;
        }
        if (out_f32) {
            float *of = out_f32 + fr->out_off;
            for (size_t i = 0; i < (size_t)bs * C; i++)
                of[i] = (float)((double)o[i] * factor);                    /* stream.d:510 */
        }
        free(dec);
    }
}
