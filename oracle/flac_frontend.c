/*
 * oracle/flac_frontend.c -- CPU restatement of the reference's FLAC front-end: the bit reader, the frame and subframe
 * headers, the fused Rice + prediction loop and the drflac_read_s32 delivery loop of source/audioformats/drflac.d, from
 * the bytes of a native FLAC file to the interleaved int32 samples AudioStream converts to float (stream.d:492-515).
 * TEST INFRASTRUCTURE ONLY (see afg_oracle.h): the product's own parser (audio-formats_amd/host/afg_flac_front.cpp, a
 * 64-bit window reader of a different build) is checked against this one; nothing shipped links or calls it.
 * PARITY UNPINNED by reference vectors (the reference holds none for FLAC, and no D compiler exists here); pinned by
 * the encoder round trips of tests/test_oracle_flac_frontend.py, where every file has a known PCM content.
 *
 * Follows, function by function (scalar, the reference's 32-bit cache: `version = DRFLAC_64BIT` is commented out,
 * drflac.d:127):
 *   drflac__reload_l1_cache_from_l2 / reload_cache / seek_bits / read_uint32 / read_int32      :707-868
 *   drflac__seek_past_next_set_bit, read_utf8_coded_number                                       :939-1043
 *   drflac__calculate_prediction_32 / _64 (oracle/flac_restore.c)                                :1060-1140
 *   drflac__decode_samples_with_residual__rice_32/_64, __unencoded, decode_samples_with_residual :1143-1328
 *   decode_samples__constant / verbatim / fixed / lpc                                            :1374-1441
 *   read_next_frame_header, read_subframe_header, decode_subframe, decode_frame                  :1444-1678
 *   decode_block_header, read_streaminfo, read_and_decode_metadata, init_private__native         :1887-2160
 *   drflac_read_s32 (whole frames + the misaligned path)                                         :2775-2960
 *
 * The reference is not defined everywhere on damaged input; the file record says where this restatement had to choose:
 *   AFGO_FLAC_F_IGNORED_FAILURE  drflac__decode_subframe drops the result of the sample decoders (:1591-1594): a
 *                                subframe whose decode fails half way is delivered with what the decode buffer held --
 *                                the samples decoded so far, then the previous frames' at those positions.  Restated
 *                                as the reference does it, on a buffer that starts zeroed.  The same flag marks a
 *                                subframe whose partitions hold fewer samples than the block (a partition order the
 *                                block size is not a multiple of, :1295-1327): its tail is stale in the same way.
 *   AFGO_FLAC_F_UNINITIALISED    ... and the buffer positions delivered had never been written: the reference reads
 *                                malloc'ed memory there (:2600-2607 clear only the struct), this restatement zeros.
 *   AFGO_FLAC_F_UNDEFINED        the frame needs an operation D leaves undefined or that writes outside the decode
 *                                buffer (a block size code of 0 shifts by -8, :1507; a block larger than STREAMINFO's
 *                                maximum or more channels than STREAMINFO's overrun the buffer, :1662; a reserved
 *                                sample-size code gives 255-bit reads, :1452; a negative LPC shift, :1430; a sample
 *                                size of 0 or less after the wasted bits, :1588).  The stream ends before that frame.
 * D integer arithmetic wraps; every wrapping operation below is done on unsigned operands, and shift counts the x86
 * would mask are masked.
 */
#include "afg_oracle.h"
#include <stdlib.h>
#include <string.h>

#define L2_LINES 1024                      /* DR_FLAC_BUFFER_SIZE / sizeof(uint), drflac.d:124, :334 */

typedef struct {
    const uint8_t *data;                   /* the memory stream behind ReadStruct (stream.d:2090-2120) */
    size_t size, cursor;
    size_t unaligned_bytes;                /* drflac_bs, drflac.d:306-335 */
    uint32_t unaligned_cache;
    size_t next_l2;
    size_t consumed;
    uint32_t l2[L2_LINES];
    uint32_t cache;
} bs_t;

static size_t rs_read(bs_t *bs, void *dst, size_t n)            /* ReadStruct.read over memory_read */
{
    size_t left = bs->size - bs->cursor;
    if (n > left) n = left;
    memcpy(dst, bs->data + bs->cursor, n);
    bs->cursor += n;
    return n;
}

static int rs_seek_cur(bs_t *bs, long off)                      /* flac_seek returns true whatever memory_seek says */
{
    long long t = (long long)bs->cursor + off;
    if (t < 0) return 1;                                          /* memory_seek: refused, cursor stays */
    if ((unsigned long long)t >= bs->size) t = (long long)bs->size;
    bs->cursor = (size_t)t;
    return 1;
}

static uint32_t be32(uint32_t v)
{
    const uint8_t *p = (const uint8_t *)&v;
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

static uint32_t sel_mask(unsigned bits) { return ~(0xFFFFFFFFu >> (bits & 31u)); }          /* :691-696 (x86 masks the count) */
static uint32_t sel_shift(const bs_t *bs, unsigned bits) { return (bs->cache & sel_mask(bits)) >> ((32u - bits) & 31u); }
#define BITS_REMAINING(bs) (32u - (unsigned)(bs)->consumed)

static int reload_l1_from_l2(bs_t *bs)                          /* :707-752 */
{
    if (bs->next_l2 < L2_LINES) {
        bs->cache = bs->l2[bs->next_l2++];
        return 1;
    }
    if (bs->unaligned_bytes > 0) return 0;
    size_t got = rs_read(bs, bs->l2, sizeof(bs->l2));
    bs->next_l2 = 0;
    if (got == sizeof(bs->l2)) {
        bs->cache = bs->l2[bs->next_l2++];
        return 1;
    }
    size_t lines = got / 4;
    bs->unaligned_bytes = got - lines * 4;
    if (bs->unaligned_bytes > 0) bs->unaligned_cache = bs->l2[lines];
    if (lines > 0) {
        size_t offset = L2_LINES - lines;
        for (size_t i = lines; i > 0; --i) bs->l2[i - 1 + offset] = bs->l2[i - 1];
        bs->next_l2 = offset;
        bs->cache = bs->l2[bs->next_l2++];
        return 1;
    }
    bs->next_l2 = L2_LINES;
    return 0;
}

static int reload_cache(bs_t *bs)                               /* :754-780 */
{
    if (reload_l1_from_l2(bs)) {
        bs->cache = be32(bs->cache);
        bs->consumed = 0;
        return 1;
    }
    size_t got = bs->unaligned_bytes;
    if (got == 0) return 0;
    bs->consumed = (4 - got) * 8;
    bs->cache = be32(bs->unaligned_cache);
    bs->cache &= sel_mask(32u - (unsigned)bs->consumed);
    bs->unaligned_bytes = 0;
    return 1;
}

static int seek_bits(bs_t *bs, size_t n)                        /* :790-832 */
{
    if (n <= BITS_REMAINING(bs)) {
        bs->consumed += n;
        bs->cache = n >= 32 ? 0 : bs->cache << n;                 /* n == 32 only with a full cache: everything leaves */
        return 1;
    }
    n -= BITS_REMAINING(bs);
    bs->consumed += BITS_REMAINING(bs);
    bs->cache = 0;
    size_t whole_bytes = n / 8;
    if (whole_bytes > 0) {
        size_t whole_lines = whole_bytes / 4;
        size_t lines_left = L2_LINES - bs->next_l2;
        if (whole_lines < lines_left) {
            whole_bytes -= whole_lines * 4;
            n -= whole_lines * 32;
            bs->next_l2 += whole_lines;
        } else {
            whole_bytes -= lines_left * 4;
            n -= lines_left * 32;
            bs->next_l2 += lines_left;
            if (whole_bytes > 0 && bs->unaligned_bytes == 0) {
                if (!rs_seek_cur(bs, (long)(int)whole_bytes)) return 0;
                n -= whole_bytes * 8;
            }
        }
    }
    if (n > 0) {
        if (!reload_cache(bs)) return 0;
        return seek_bits(bs, n);
    }
    return 1;
}

static int read_u32(bs_t *bs, unsigned count, uint32_t *out)    /* :834-868; count 1..32 (0: see the header) */
{
    if (bs->consumed == 32) {
        if (!reload_cache(bs)) return 0;
    }
    if (count <= BITS_REMAINING(bs)) {
        if (count < 32) {
            *out = sel_shift(bs, count);
            bs->consumed += count;
            bs->cache <<= count;
        } else {
            *out = bs->cache;
            bs->consumed = 32;
            bs->cache = 0;
        }
        return 1;
    }
    unsigned hi = BITS_REMAINING(bs), lo = count - hi;
    uint32_t result_hi = sel_shift(bs, hi);
    if (!reload_cache(bs)) return 0;
    *out = (result_hi << lo) | sel_shift(bs, lo);
    bs->consumed += lo;
    bs->cache <<= lo;
    return 1;
}

static int read_i32(bs_t *bs, unsigned count, int32_t *out)     /* :870-882 */
{
    uint32_t r;
    if (!read_u32(bs, count, &r)) return 0;
    uint32_t signbit = (r >> ((count - 1u) & 31u)) & 1u;
    if (count < 32) r |= (~signbit + 1u) << count;
    *out = (int32_t)r;
    return 1;
}

static const uint32_t k_bit_offset[16] = { 0, 4, 3, 3, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1 };

static int seek_past_next_set_bit(bs_t *bs, uint32_t *offset)   /* :939-976 */
{
    uint32_t zeros = 0;
    while (bs->cache == 0) {
        zeros += BITS_REMAINING(bs);
        if (!reload_cache(bs)) return 0;
    }
    uint32_t plus1 = k_bit_offset[sel_shift(bs, 4)];
    if (plus1 == 0) {
        if (bs->cache == 1) {
            plus1 = 32;
        } else {
            plus1 = 5;
            for (;;) {
                if (bs->cache & sel_mask(plus1)) break;
                plus1 += 1;
            }
        }
    }
    bs->consumed += plus1;
    bs->cache = plus1 >= 32 ? 0 : bs->cache << plus1;
    *offset = zeros + plus1 - 1;
    return 1;
}

static int read_utf8(bs_t *bs, uint64_t *number)                /* :1005-1043: no continuation-byte check */
{
    uint32_t b;
    uint8_t utf8[7] = { 0 };
    if (!read_u32(bs, 8, &b)) { *number = 0; return 0; }
    utf8[0] = (uint8_t)b;
    if ((utf8[0] & 0x80) == 0) { *number = utf8[0]; return 1; }
    int count;
    if ((utf8[0] & 0xE0) == 0xC0) count = 2;
    else if ((utf8[0] & 0xF0) == 0xE0) count = 3;
    else if ((utf8[0] & 0xF8) == 0xF0) count = 4;
    else if ((utf8[0] & 0xFC) == 0xF8) count = 5;
    else if ((utf8[0] & 0xFE) == 0xFC) count = 6;
    else if ((utf8[0] & 0xFF) == 0xFE) count = 7;
    else { *number = 0; return 0; }
    uint64_t result = (uint64_t)(utf8[0] & (0xFF >> (count + 1)));
    for (int i = 1; i < count; ++i) {
        if (!read_u32(bs, 8, &b)) { *number = 0; return 0; }
        utf8[i] = (uint8_t)b;
        result = (result << 6) | (utf8[i] & 0x3F);
    }
    *number = result;
    return 1;
}

/* ------------------------------------------------------------------------------------------------ residual decode */

/* the fused loop of :1151-1252; `wide` picks prediction_64 (:1254) or prediction_32 (:1255) */
static int rice_samples(bs_t *bs, uint32_t count, unsigned rice, unsigned order, int shift, const int16_t *coef, int32_t *out,
                        int wide, uint32_t *done)
{
    const uint32_t rice_mask = sel_mask(rice);
    const uint32_t hi_shift = (32u - rice) & 31u;
    for (int i = 0; i < (int)count; ++i) {
        uint32_t zeros = 0;
        while (bs->cache == 0) {
            zeros += BITS_REMAINING(bs);
            if (!reload_cache(bs)) return 0;
        }
        uint32_t decoded;
        uint32_t plus1 = k_bit_offset[sel_shift(bs, 4)];
        if (plus1 > 0) {
            decoded = (zeros + (plus1 - 1)) << rice;
        } else if (bs->cache == 1) {
            plus1 = 32;
            decoded = (zeros + 31u) << rice;
        } else {
            plus1 = 5;
            for (;;) {
                if (bs->cache & sel_mask(plus1)) {
                    decoded = (zeros + (plus1 - 1)) << rice;
                    break;
                }
                plus1 += 1;
            }
        }
        uint32_t bits_lo = 0;
        uint32_t rice_length = plus1 + rice;
        if (rice_length < BITS_REMAINING(bs)) {
            bits_lo = (bs->cache & (rice_mask >> plus1)) >> ((32u - rice_length) & 31u);
            bs->consumed += rice_length;
            bs->cache <<= rice_length;
        } else {
            bs->consumed += rice_length;
            bs->cache = plus1 >= 32 ? 0 : bs->cache << plus1;
            size_t count_lo = bs->consumed - 32;
            uint32_t result_hi = bs->cache & rice_mask;
            if (bs->next_l2 < L2_LINES) {
                bs->cache = be32(bs->l2[bs->next_l2++]);
            } else {
                if (!reload_cache(bs)) return 0;
            }
            bits_lo = (rice ? (result_hi >> hi_shift) : 0u) | sel_shift(bs, (unsigned)count_lo);
            bs->consumed = count_lo;
            bs->cache = count_lo >= 32 ? 0 : bs->cache << count_lo;
        }
        decoded |= bits_lo;
        decoded = (decoded >> 1) ^ (~(decoded & 1u) + 1u);
        int32_t pred = wide ? afgo_flac_prediction_64(order, shift, coef, out + i) : afgo_flac_prediction_32(order, shift, coef, out + i);
        out[i] = (int32_t)(decoded + (uint32_t)pred);
        *done += 1;
    }
    return 1;
}

static int unencoded_samples(bs_t *bs, uint32_t bps, uint32_t count, unsigned raw_bits, unsigned order, int shift,
                             const int16_t *coef, int32_t *out, uint32_t *done)        /* :1268-1283 (dead code: below) */
{
    for (uint32_t i = 0; i < count; ++i) {
        if (!read_i32(bs, raw_bits, out + i)) return 0;
        int32_t pred = bps > 16 ? afgo_flac_prediction_64(order, shift, coef, out + i) : afgo_flac_prediction_32(order, shift, coef, out + i);
        out[i] = (int32_t)((uint32_t)out[i] + (uint32_t)pred);
        *done += 1;
    }
    return 1;
}

/* :1288-1336.  `done` counts the samples written past the warm-up (for the ignored-failure bookkeeping). */
static int samples_with_residual(bs_t *bs, uint32_t bps, uint32_t block, unsigned order, int shift, const int16_t *coef,
                                 int32_t *decoded, uint32_t *done)
{
    uint32_t method;
    if (!read_u32(bs, 2, &method)) return 0;
    if (method != 0 && method != 1) return 0;
    decoded += order;
    uint32_t part_order;
    if (!read_u32(bs, 4, &part_order)) return 0;
    uint32_t in_part = (block / (1u << part_order)) - order;          /* wraps when the first partition is shorter than the order */
    uint32_t parts_left = 1u << part_order;
    for (;;) {
        uint32_t rice = 0;
        if (method == 0) {
            if (!read_u32(bs, 4, &rice)) return 0;
            if (rice == 16) rice = 0xFF;                                  /* never true of four bits: the escape code 15 is a Rice parameter */
        } else {
            if (!read_u32(bs, 5, &rice)) return 0;
            if (rice == 32) rice = 0xFF;                                  /* likewise 31 */
        }
        /* a count that wrapped would run over the block (and the buffer): the caller has ruled it out */
        if (rice != 0xFF) {
            if (!rice_samples(bs, in_part, rice, order, shift, coef, decoded, bps > 16, done)) return 0;
        } else {
            uint32_t raw = 0;
            if (!read_u32(bs, 5, &raw)) return 0;
            if (!unencoded_samples(bs, bps, in_part, raw, order, shift, coef, decoded, done)) return 0;
        }
        decoded += in_part;
        if (parts_left == 1) break;
        parts_left -= 1;
        in_part = block / (1u << part_order);
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------- frames */

typedef struct {
    uint8_t type, wasted, lpc_order;
    uint32_t bps;
    int32_t *samples;
} subframe_t;

typedef struct {
    uint32_t sample_rate;
    uint16_t block;
    uint8_t assignment, bps, crc8;
    uint8_t bs_code, bps_code;
} header_t;

static int read_frame_header(bs_t *bs, uint8_t streaminfo_bps, header_t *h)      /* :1444-1528 */
{
    static const uint32_t rates[12] = { 0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000 };
    static const uint8_t sizes[8] = { 0, 8, 12, 255, 16, 20, 24, 255 };
    uint32_t sync, v, strategy, bsize, rate, asg, bps;
    if (!read_u32(bs, 14, &sync)) return 0;
    if (sync != 0x3FFE) return 0;
    if (!read_u32(bs, 1, &v)) return 0;
    if (!read_u32(bs, 1, &strategy)) return 0;
    if (!read_u32(bs, 4, &bsize)) return 0;
    if (!read_u32(bs, 4, &rate)) return 0;
    if (!read_u32(bs, 4, &asg)) return 0;
    if (!read_u32(bs, 3, &bps)) return 0;
    if (!read_u32(bs, 1, &v)) return 0;
    uint64_t number;
    if (!read_utf8(bs, &number)) return 0;                                       /* sample or frame number: not used here */
    h->bs_code = (uint8_t)bsize;
    if (bsize == 1) h->block = 192;
    else if (bsize >= 2 && bsize <= 5) h->block = (uint16_t)(576u * (1u << (bsize - 2)));
    else if (bsize == 6) { if (!read_u32(bs, 8, &v)) return 0; h->block = (uint16_t)(v + 1); }
    else if (bsize == 7) { if (!read_u32(bs, 16, &v)) return 0; h->block = (uint16_t)(v + 1); }
    else if (bsize >= 8) h->block = (uint16_t)(256u * (1u << (bsize - 8)));
    else h->block = 0;                                                           /* code 0: 1 << -8, undefined (flagged by the caller) */
    if (rate <= 11) h->sample_rate = rates[rate];
    else if (rate == 12) { if (!read_u32(bs, 8, &v)) return 0; h->sample_rate = v * 1000; }
    else if (rate == 13) { if (!read_u32(bs, 16, &v)) return 0; h->sample_rate = v; }
    else if (rate == 14) { if (!read_u32(bs, 16, &v)) return 0; h->sample_rate = v * 10; }
    else return 0;
    h->assignment = (uint8_t)asg;
    h->bps_code = (uint8_t)bps;
    h->bps = sizes[bps];
    if (h->bps == 0) h->bps = streaminfo_bps;
    if (!read_u32(bs, 8, &v)) return 0;
    h->crc8 = (uint8_t)v;
    return 1;
}

static int read_subframe_header(bs_t *bs, subframe_t *sf)                        /* :1530-1569 */
{
    uint32_t header;
    if (!read_u32(bs, 8, &header)) return 0;
    if ((header & 0x80) != 0) return 0;
    int type = (int)((header & 0x7E) >> 1);
    enum { CONSTANT = 0, VERBATIM = 1, FIXED = 8, LPC = 32, RESERVED = 255 };
    if (type == 0) sf->type = CONSTANT;
    else if (type == 1) sf->type = VERBATIM;
    else if ((type & 0x20) != 0) { sf->type = LPC; sf->lpc_order = (uint8_t)((type & 0x1F) + 1); }
    else if ((type & 0x08) != 0) {
        sf->type = FIXED;
        sf->lpc_order = (uint8_t)(type & 0x07);
        if (sf->lpc_order > 4) { sf->type = RESERVED; sf->lpc_order = 0; }
    } else sf->type = RESERVED;
    if (sf->type == RESERVED) return 0;
    sf->wasted = 0;
    if ((header & 0x01) == 1) {
        uint32_t w;
        if (!seek_past_next_set_bit(bs, &w)) return 0;
        sf->wasted = (uint8_t)((uint8_t)w + 1);
    }
    return 1;
}

typedef struct {
    bs_t bs;
    uint32_t channels, sample_rate, max_block;
    uint8_t bps;
    uint64_t total_samples;
    header_t header;
    subframe_t sub[8];
    uint32_t remaining;                    /* currentFrame.samplesRemaining */
    int32_t *decoded;                      /* pDecodedSamples: max_block * channels */
    uint8_t *written;                      /* which buffer positions any frame has written (for AFGO_FLAC_F_UNINITIALISED) */
    uint32_t flags;
    int frame_flagged;                     /* the frame in hand walked one of the paths the file record names */
} flac_t;

/* 1: decoded (possibly with ignored failures), 0: the frame fails (the stream ends), -1: undefined in the reference */
static int decode_subframe(flac_t *f, int idx, int32_t *out)                     /* :1571-1599 */
{
    bs_t *bs = &f->bs;
    subframe_t *sf = &f->sub[idx];
    const uint32_t block = f->header.block;
    if (!read_subframe_header(bs, sf)) return 0;
    sf->bps = f->header.bps;
    if ((f->header.assignment == AFGO_FLAC_LEFT_SIDE || f->header.assignment == AFGO_FLAC_MID_SIDE) && idx == 1) sf->bps += 1;
    else if (f->header.assignment == AFGO_FLAC_RIGHT_SIDE && idx == 0) sf->bps += 1;
    sf->bps -= sf->wasted;
    sf->samples = out;
    if (sf->bps == 0 || sf->bps > 32) return -1;                                 /* 0 or "negative": reads of no / of 2^32 - k bits */
    uint32_t done = 0;                                                           /* samples of this subframe written so far */
    int ok = 1;
    if (sf->type == 0) {                                                         /* constant, :1374-1385 */
        int32_t v;
        ok = read_i32(bs, sf->bps, &v);
        if (ok) { for (uint32_t i = 0; i < block; ++i) out[i] = v; done = block; }
    } else if (sf->type == 1) {                                                  /* verbatim, :1387-1394 */
        for (uint32_t i = 0; i < block && ok; ++i) {
            int32_t v;
            ok = read_i32(bs, sf->bps, &v);
            if (ok) { out[i] = v; done++; }
        }
    } else {
        static const int16_t fixed[5][4] = { { 0, 0, 0, 0 }, { 1, 0, 0, 0 }, { 2, -1, 0, 0 }, { 3, -3, 1, 0 }, { 4, -6, 4, -1 } };
        int16_t coef[32];
        memset(coef, 0, sizeof(coef));
        const unsigned order = sf->lpc_order;
        int shift = 0;
        if (order > block) return -1;                                            /* the warm-up alone overruns the block's part of the buffer */
        for (unsigned i = 0; i < order && ok; ++i) {                             /* warm-up, :1406-1410, :1419-1423 */
            int32_t v;
            ok = read_i32(bs, sf->bps, &v);
            if (ok) { out[i] = v; done++; }
        }
        if (ok && sf->type == 32) {                                              /* LPC, :1425-1436 */
            uint32_t prec;
            ok = read_u32(bs, 4, &prec);
            if (ok && prec == 15) ok = 0;
            if (ok) {
                prec += 1;
                int32_t s;
                ok = read_i32(bs, 5, &s);
                shift = (int)(int8_t)s;
                for (unsigned i = 0; i < order && ok; ++i) {
                    int32_t c;
                    ok = read_i32(bs, prec, &c);
                    coef[i] = (int16_t)c;
                }
                if (ok && shift < 0) return -1;                                  /* >> by a negative count */
            }
        } else if (ok) {
            memcpy(coef, fixed[order], sizeof(fixed[0]));
        }
        if (ok) {
            /* the first partition must hold the warm-up, or the count wraps and the loop runs over the buffer */
            bs_t peek = *bs;                                                      /* (look at the partition order without consuming) */
            uint32_t method, po;
            if (read_u32(&peek, 2, &method) && (method == 0 || method == 1) && read_u32(&peek, 4, &po) && (block >> po) < order) return -1;
            uint32_t rdone = 0;
            ok = samples_with_residual(bs, sf->bps, block, order, shift, coef, out, &rdone);
            done += rdone;
        }
    }
    for (uint32_t i = 0; i < done; ++i) f->written[(size_t)(out - f->decoded) + i] = 1;
    if (!ok || done < block) {                                                   /* :1591-1594: the result is dropped; or 2^order partitions
                                                                                    of block >> order samples that do not add up to the block */
        f->flags |= AFGO_FLAC_F_IGNORED_FAILURE;
        f->frame_flagged = 1;
        for (uint32_t i = done; i < block; ++i)
            if (!f->written[(size_t)(out - f->decoded) + i]) f->flags |= AFGO_FLAC_F_UNINITIALISED;
    }
    return 1;
}

static uint32_t channel_count(uint8_t assignment)                               /* :1652-1656 */
{
    static const uint8_t lookup[11] = { 1, 2, 3, 4, 5, 6, 7, 8, 2, 2, 2 };
    return assignment <= 10 ? lookup[assignment] : 0;
}

static int read_and_decode_next_frame(flac_t *f)                                 /* :1682-1688 + decode_frame :1658-1672 */
{
    const uint32_t flags_before = f->flags;                                      /* a frame that is dropped leaves no flag */
    f->frame_flagged = 0;
    if (!read_frame_header(&f->bs, f->bps, &f->header)) return 0;
    memset(f->sub, 0, sizeof(f->sub));
    const uint32_t channels = channel_count(f->header.assignment);
    if (channels == 0 || f->header.bs_code == 0 || f->header.block == 0 || f->header.bps == 255 ||
        (uint64_t)f->header.block * channels > (uint64_t)f->max_block * f->channels) {
        f->flags |= AFGO_FLAC_F_UNDEFINED;                                       /* see the header of this file */
        return -1;
    }
    for (uint32_t i = 0; i < channels; ++i) {
        int r = decode_subframe(f, (int)i, f->decoded + (size_t)f->header.block * i);
        if (r < 0) { f->flags = flags_before | AFGO_FLAC_F_UNDEFINED; return -1; }
        if (!r) { f->flags = flags_before; return 0; }
    }
    if (!seek_bits(&f->bs, (BITS_REMAINING(&f->bs) & 7) + 16)) { f->flags = flags_before; return 0; }   /* padding + CRC-16, not checked (:1667) */
    f->remaining = f->header.block * channels;
    return 1;
}

/* one interleaved sample of the frame in hand: the switch of drflac__read_s32__misaligned (:2775-2835), which
 * drflac_read_s32's whole-frame loops (:2885-2941) agree with sample for sample */
static int32_t frame_sample(const flac_t *f, uint32_t channels, uint32_t index)
{
    const uint32_t t = index / channels, c = index % channels;
    const subframe_t *s = f->sub;
    uint32_t v;
    switch (f->header.assignment) {
    case AFGO_FLAC_LEFT_SIDE:
        v = c == 0 ? (uint32_t)s[0].samples[t] : (uint32_t)s[0].samples[t] - (uint32_t)s[1].samples[t];
        break;
    case AFGO_FLAC_RIGHT_SIDE:
        v = c == 0 ? (uint32_t)s[0].samples[t] + (uint32_t)s[1].samples[t] : (uint32_t)s[1].samples[t];
        break;
    case AFGO_FLAC_MID_SIDE: {
        const uint32_t side = (uint32_t)s[1].samples[t];
        const uint32_t mid = ((uint32_t)s[0].samples[t] << 1) | (side & 1u);
        v = (uint32_t)((int32_t)(c == 0 ? mid + side : mid - side) >> 1);
        break;
    }
    default:
        v = (uint32_t)s[c].samples[t];
        break;
    }
    return (int32_t)(v << (((32u - f->bps) + s[c].wasted) & 31u));
}

static uint32_t rd_be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int afgo_flac_decode_file(const uint8_t *data, size_t size, afgo_flac_file *out)
{
    memset(out, 0, sizeof(*out));
    flac_t *f = (flac_t *)calloc(1, sizeof(flac_t));
    if (!f) return -2;
    bs_t *bs = &f->bs;
    bs->data = data;
    bs->size = size;
    /* drflac__check_init_private :2543-2552, init_private__native :2131-2160, read_streaminfo :1901-1931 */
    uint8_t head[4 + 4 + 34];
    if (rs_read(bs, head, 4) != 4 || memcmp(head, "fLaC", 4) != 0) { free(f); return -1; }
    if (rs_read(bs, head + 4, 4) != 4) { free(f); return -1; }
    uint32_t bh = rd_be32(head + 4);
    int last = (int)(bh >> 31);
    if (((bh >> 24) & 0x7F) != 0 || (bh & 0xFFFFFF) != 34) { free(f); return -1; }
    if (rs_read(bs, head + 8, 34) != 34) { free(f); return -1; }
    const uint8_t *si = head + 8;
    f->max_block = ((uint32_t)si[2] << 8) | si[3];
    uint64_t props = ((uint64_t)rd_be32(si + 10) << 32) | rd_be32(si + 14);
    f->sample_rate = (uint32_t)((props & 0xFFFFF00000000000ull) >> 44);
    f->channels = (uint32_t)((props & 0x00000E0000000000ull) >> 41) + 1;
    f->bps = (uint8_t)(((props & 0x000001F000000000ull) >> 36) + 1);
    f->total_samples = (props & 0x0000000FFFFFFFFFull) * f->channels;
    /* read_and_decode_metadata without a callback :1933-2118: every block is skipped (the seek never fails: stream.d:2227) */
    while (!last) {
        uint8_t b[4];
        if (rs_read(bs, b, 4) != 4) { free(f); return -1; }
        bh = rd_be32(b);
        last = (int)(bh >> 31);
        rs_seek_cur(bs, (long)(bh & 0xFFFFFF));
    }
    /* drflac__init_from_info :2570-2588: both caches start empty */
    bs->next_l2 = L2_LINES;
    bs->consumed = 32;
    out->channels = f->channels;
    out->sample_rate = f->sample_rate;
    out->bps = f->bps;
    out->max_block = f->max_block;
    out->total_samples = f->total_samples;
    const size_t buf = (size_t)f->max_block * f->channels;
    f->decoded = (int32_t *)calloc(buf ? buf : 1, sizeof(int32_t));
    f->written = (uint8_t *)calloc(buf ? buf : 1, 1);
    size_t cap = 1 << 16;
    int32_t *pcm = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!f->decoded || !f->written || !pcm) { free(f->decoded); free(f->written); free(pcm); free(f); return -2; }
    uint64_t n = 0;
    out->first_flag_sample = UINT64_MAX;
    /* the delivery loop of drflac_read_s32 (:2846-2960), to the end of the stream */
    for (;;) {
        int r = read_and_decode_next_frame(f);
        if (r <= 0) {
            if (r < 0 && out->first_flag_sample == UINT64_MAX) out->first_flag_sample = n;
            break;
        }
        if (f->frame_flagged && out->first_flag_sample == UINT64_MAX) out->first_flag_sample = n;
        const uint32_t channels = channel_count(f->header.assignment);
        const uint32_t total = f->remaining;
        if (n + total > cap) {
            while (n + total > cap) cap *= 2;
            int32_t *np = (int32_t *)realloc(pcm, cap * sizeof(int32_t));
            if (!np) { free(f->decoded); free(f->written); free(pcm); free(f); return -2; }
            pcm = np;
        }
        for (uint32_t i = 0; i < total; ++i) pcm[n + i] = frame_sample(f, channels, i);
        n += total;
        out->n_frames++;
    }
    out->pcm = pcm;
    out->n_samples = n;
    out->flags = f->flags;
    free(f->decoded);
    free(f->written);
    free(f);
    return 0;
}

void afgo_flac_file_free(afgo_flac_file *file)
{
    free(file->pcm);
    memset(file, 0, sizeof(*file));
}
