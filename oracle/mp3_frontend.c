/*
 * mp3_frontend.c -- ORACLE (test infrastructure only): CPU restatement of the reference's MP3 Layer III
 * front-end -- frame sync, side info, scalefactors, Huffman + requantisation, stereo processing, reorder,
 * bit reservoir -- and of the way minimp3_ex drives it (ID3 skipping, Xing/Info tag, encoder delay /
 * padding trimming), followed by the transform-stage oracle of mp3_transform.c.
 *
 * PARITY UNPINNED: the reference ships no MP3 vectors and cannot be built here (D); no other MP3 decoder
 * exists in this image.  What pins this file: the Huffman books are verified complete prefix codes of the
 * ISO dimensions by the generator, tests decode a real MPEG-1 Layer III file (tests/golden/) and check the
 * result for continuity across frame boundaries, tonal content and exact agreement with the product's
 * independently written parser.
 *
 * Each function cites the reference lines it follows (source/audioformats/minimp3.d = "mp3:",
 * minimp3_ex.d = "ex:").  Only Layer III is handled (Layer I/II frames end decoding: documented gap).
 */
#include "afg_oracle.h"
#include "mp3_front_tables.h"

#include <stdlib.h>
#include <string.h>

#define HDR_SIZE 4
#define MAX_FREE_FORMAT_FRAME_SIZE 2304
#define MAX_FRAME_SYNC_MATCHES 10
#define MAX_BITRESERVOIR_BYTES 511
#define MAX_L3_FRAME_PAYLOAD_BYTES MAX_FREE_FORMAT_FRAME_SIZE
#define MAX_SCF (255 + (-1) * 4 - 210)                  /* mp3:149-151, BITS_DEQUANTIZER_OUT = -1 */
#define MAX_SCFI ((MAX_SCF + 3) & ~3)

static float f32(unsigned bits) { float f; memcpy(&f, &bits, 4); return f; }

/* ---- header helpers, mp3:67-146, :215-265 ---- */
static int h_mono(const uint8_t *h) { return (h[3] & 0xC0) == 0xC0; }
static int h_ms(const uint8_t *h) { return (h[3] & 0xE0) == 0x60; }
static int h_free(const uint8_t *h) { return (h[2] & 0xF0) == 0; }
static int h_crc(const uint8_t *h) { return !(h[1] & 1); }
static int h_mpeg1(const uint8_t *h) { return h[1] & 0x8; }
static int h_not25(const uint8_t *h) { return h[1] & 0x10; }
static int h_istereo(const uint8_t *h) { return h[3] & 0x10; }
static int h_msbit(const uint8_t *h) { return h[3] & 0x20; }
static int h_layer(const uint8_t *h) { return (h[1] >> 1) & 3; }
static int h_bitrate(const uint8_t *h) { return h[2] >> 4; }
static int h_sr(const uint8_t *h) { return (h[2] >> 2) & 3; }
static int h_my_sr(const uint8_t *h) { return h_sr(h) + (((h[1] >> 3) & 1) + ((h[1] >> 4) & 1)) * 3; }
static int h_576(const uint8_t *h) { return (h[1] & 14) == 2; }
static int h_l1(const uint8_t *h) { return (h[1] & 6) == 6; }

static int hdr_valid(const uint8_t *h)
{
    return h[0] == 0xff && ((h[1] & 0xF0) == 0xf0 || (h[1] & 0xFE) == 0xe2) && h_layer(h) != 0 &&
           h_bitrate(h) != 15 && h_sr(h) != 3;
}
static int hdr_compare(const uint8_t *h1, const uint8_t *h2)
{
    return hdr_valid(h2) && ((h1[1] ^ h2[1]) & 0xFE) == 0 && ((h1[2] ^ h2[2]) & 0x0C) == 0 && !(h_free(h1) ^ h_free(h2));
}
static unsigned hdr_kbps(const uint8_t *h) { return 2u * k_halfrate[!!h_mpeg1(h)][h_layer(h) - 1][h_bitrate(h)]; }
static unsigned hdr_hz(const uint8_t *h)
{
    static const unsigned g_hz[3] = { 44100, 48000, 32000 };
    return g_hz[h_sr(h)] >> (int)!h_mpeg1(h) >> (int)!h_not25(h);
}
static unsigned hdr_frame_samples(const uint8_t *h) { return h_l1(h) ? 384 : (1152 >> (int)h_576(h)); }
static int hdr_frame_bytes(const uint8_t *h, int free_format_size)
{
    int fb = (int)(hdr_frame_samples(h) * hdr_kbps(h) * 125 / hdr_hz(h));
    if (h_l1(h)) fb &= ~3;
    return fb ? fb : free_format_size;
}
static int hdr_padding(const uint8_t *h) { return (h[2] & 0x2) ? (h_l1(h) ? 4 : 1) : 0; }

/* ---- bit reader, mp3:185-207 ---- */
typedef struct { const uint8_t *buf; int pos, limit; } bs_t;
static void bs_init(bs_t *bs, const uint8_t *data, int bytes) { bs->buf = data; bs->pos = 0; bs->limit = bytes * 8; }
static uint32_t get_bits(bs_t *bs, int n)
{
    uint32_t next, cache = 0, s = (uint32_t)bs->pos & 7;
    int shl = n + (int)s;
    const uint8_t *p = bs->buf + (bs->pos >> 3);
    if ((bs->pos += n) > bs->limit) return 0;
    next = *p++ & (255u >> s);
    while ((shl -= 8) > 0) {
        cache |= next << shl;
        next = *p++;
    }
    return cache | (next >> -shl);
}

typedef struct {
    const uint8_t *sfbtab;
    uint16_t part_23_length, big_values, scalefac_compress;
    uint8_t global_gain, block_type, mixed_block_flag, n_long_sfb, n_short_sfb;
    uint8_t table_select[3], region_count[3], subblock_gain[3];
    uint8_t preflag, scalefac_scale, count1_table, scfsi;
} gr_info_t;

/* mp3:487-614 */
static int read_side_info(bs_t *bs, gr_info_t *gr, const uint8_t *hdr)
{
    unsigned tables, scfsi = 0;
    int main_data_begin, part_23_sum = 0;
    int sr_idx = h_my_sr(hdr);
    sr_idx -= (sr_idx != 0);
    int gr_count = h_mono(hdr) ? 1 : 2;
    if (h_mpeg1(hdr)) {
        gr_count *= 2;
        main_data_begin = (int)get_bits(bs, 9);
        scfsi = get_bits(bs, 7 + gr_count);
    } else {
        main_data_begin = (int)(get_bits(bs, 8 + gr_count) >> gr_count);
    }
    do {
        if (h_mono(hdr)) scfsi <<= 4;
        gr->part_23_length = (uint16_t)get_bits(bs, 12);
        part_23_sum += gr->part_23_length;
        gr->big_values = (uint16_t)get_bits(bs, 9);
        if (gr->big_values > 288) return -1;
        gr->global_gain = (uint8_t)get_bits(bs, 8);
        gr->scalefac_compress = (uint16_t)get_bits(bs, h_mpeg1(hdr) ? 4 : 9);
        gr->sfbtab = k_sfb_long[sr_idx];
        gr->n_long_sfb = 22;
        gr->n_short_sfb = 0;
        if (get_bits(bs, 1)) {
            gr->block_type = (uint8_t)get_bits(bs, 2);
            if (!gr->block_type) return -1;
            gr->mixed_block_flag = (uint8_t)get_bits(bs, 1);
            gr->region_count[0] = 7;
            gr->region_count[1] = 255;
            if (gr->block_type == 2) {
                scfsi &= 0x0F0F;
                if (!gr->mixed_block_flag) {
                    gr->region_count[0] = 8;
                    gr->sfbtab = k_sfb_short[sr_idx];
                    gr->n_long_sfb = 0;
                    gr->n_short_sfb = 39;
                } else {
                    gr->sfbtab = k_sfb_mixed[sr_idx];
                    gr->n_long_sfb = h_mpeg1(hdr) ? 8 : 6;
                    gr->n_short_sfb = 30;
                }
            }
            tables = get_bits(bs, 10);
            tables <<= 5;
            gr->subblock_gain[0] = (uint8_t)get_bits(bs, 3);
            gr->subblock_gain[1] = (uint8_t)get_bits(bs, 3);
            gr->subblock_gain[2] = (uint8_t)get_bits(bs, 3);
        } else {
            gr->block_type = 0;
            gr->mixed_block_flag = 0;
            tables = get_bits(bs, 15);
            gr->region_count[0] = (uint8_t)get_bits(bs, 4);
            gr->region_count[1] = (uint8_t)get_bits(bs, 3);
            gr->region_count[2] = 255;
        }
        gr->table_select[0] = (uint8_t)(tables >> 10);
        gr->table_select[1] = (uint8_t)((tables >> 5) & 31);
        gr->table_select[2] = (uint8_t)(tables & 31);
        gr->preflag = h_mpeg1(hdr) ? (uint8_t)get_bits(bs, 1) : (uint8_t)(gr->scalefac_compress >= 500);
        gr->scalefac_scale = (uint8_t)get_bits(bs, 1);
        gr->count1_table = (uint8_t)get_bits(bs, 1);
        gr->scfsi = (uint8_t)((scfsi >> 12) & 15);
        scfsi <<= 4;
        gr++;
    } while (--gr_count);
    if (part_23_sum + bs->pos > bs->limit + main_data_begin * 8) return -1;
    return main_data_begin;
}

/* mp3:616-649 */
static void read_scalefactors(uint8_t *scf, uint8_t *ist_pos, const uint8_t *scf_size, const uint8_t *scf_count, bs_t *bs, int scfsi)
{
    for (int i = 0; i < 4 && scf_count[i]; i++, scfsi *= 2) {
        int cnt = scf_count[i];
        if (scfsi & 8) {
            memcpy(scf, ist_pos, (size_t)cnt);
        } else {
            int bits = scf_size[i];
            if (!bits) {
                memset(scf, 0, (size_t)cnt);
                memset(ist_pos, 0, (size_t)cnt);
            } else {
                int max_scf = (scfsi < 0) ? (1 << bits) - 1 : -1;
                for (int k = 0; k < cnt; k++) {
                    int s = (int)get_bits(bs, bits);
                    ist_pos[k] = (uint8_t)(s == max_scf ? -1 : s);
                    scf[k] = (uint8_t)s;
                }
            }
        }
        ist_pos += cnt;
        scf += cnt;
    }
    scf[0] = scf[1] = scf[2] = 0;
}

/* mp3:651-662 */
static float ldexp_q2(float y, int exp_q2)
{
    int e;
    do {
        e = exp_q2 < 30 * 4 ? exp_q2 : 30 * 4;
        y *= f32(k_expfrac_bits[e & 3]) * (float)(1 << 30 >> (e >> 2));
    } while ((exp_q2 -= e) > 0);
    return y;
}

/* mp3:664-719 */
static void decode_scalefactors(const uint8_t *hdr, uint8_t *ist_pos, bs_t *bs, const gr_info_t *gr, float *scf, int ch)
{
    const uint8_t *scf_partition = k_scf_partitions[!!gr->n_short_sfb + !gr->n_long_sfb];
    uint8_t scf_size[4], iscf[40];
    int i, scf_shift = gr->scalefac_scale + 1, gain_exp, scfsi = gr->scfsi;
    float gain;
    if (h_mpeg1(hdr)) {
        int part = k_scfc_decode[gr->scalefac_compress];
        scf_size[1] = scf_size[0] = (uint8_t)(part >> 2);
        scf_size[3] = scf_size[2] = (uint8_t)(part & 3);
    } else {
        int k, modprod, sfc, ist = h_istereo(hdr) && ch;
        sfc = gr->scalefac_compress >> ist;
        for (k = ist * 3 * 4; sfc >= 0; sfc -= modprod, k += 4) {
            for (modprod = 1, i = 3; i >= 0; i--) {
                scf_size[i] = (uint8_t)(sfc / modprod % k_scf_mod[k + i]);
                modprod *= k_scf_mod[k + i];
            }
        }
        scf_partition += k;
        scfsi = -16;
    }
    read_scalefactors(iscf, ist_pos, scf_size, scf_partition, bs, scfsi);
    if (gr->n_short_sfb) {
        int sh = 3 - scf_shift;
        for (i = 0; i < gr->n_short_sfb; i += 3) {
            iscf[gr->n_long_sfb + i + 0] = (uint8_t)(iscf[gr->n_long_sfb + i + 0] + (gr->subblock_gain[0] << sh));
            iscf[gr->n_long_sfb + i + 1] = (uint8_t)(iscf[gr->n_long_sfb + i + 1] + (gr->subblock_gain[1] << sh));
            iscf[gr->n_long_sfb + i + 2] = (uint8_t)(iscf[gr->n_long_sfb + i + 2] + (gr->subblock_gain[2] << sh));
        }
    } else if (gr->preflag) {
        for (i = 0; i < 10; i++) iscf[11 + i] = (uint8_t)(iscf[11 + i] + k_preamp[i]);
    }
    gain_exp = gr->global_gain + (-1) * 4 - 210 - (h_ms(hdr) ? 2 : 0);
    gain = ldexp_q2((float)(1 << (MAX_SCFI / 4)), MAX_SCFI - gain_exp);
    for (i = 0; i < (int)(gr->n_long_sfb + gr->n_short_sfb); i++) scf[i] = ldexp_q2(gain, iscf[i] << scf_shift);
}

/* mp3:727-746 */
static float pow_43(int x)
{
    float frac;
    int sign, mult = 256;
    if (x < 129) return f32(k_pow43_bits[16 + x]);
    if (x < 1024) {
        mult = 16;
        x <<= 3;
    }
    sign = 2 * x & 64;
    frac = (float)((x & 63) - sign) / (float)((x & ~63) + sign);
    return f32(k_pow43_bits[16 + ((x + sign) >> 6)]) * (1.0f + frac * ((4.0f / 3) + frac * (2.0f / 9))) * (float)mult;
}

/* mp3:748-883.  The cache / refill discipline is the reference's (a 32-bit window refilled byte-wise after every
 * pair and after every linbits field); the code books are searched length by length in canonical lists. */
typedef struct { uint32_t cache; int sh; const uint8_t *next; } hcache;
#define FLUSH(n) do { hc.cache <<= (n); hc.sh += (n); } while (0)
#define REFILL() do { while (hc.sh >= 0) { hc.cache |= (uint32_t)*hc.next++ << hc.sh; hc.sh -= 8; } } while (0)

static int book_lookup(int book, uint32_t cache, int *len)
{
    const afgo_huff_code *e = k_huff_codes + k_huff_first[book];
    int n = k_huff_count[book];
    for (int i = 0; i < n; i++) {
        if ((cache >> (32 - e[i].len)) == e[i].code) {
            *len = e[i].len;
            return e[i].x | (e[i].y << 4);
        }
    }
    *len = 0;
    return 0;
}

static void huffman(float *dst, bs_t *bs, const gr_info_t *gr, const float *scf, int layer3gr_limit)
{
    float one = 0.0f;
    int ireg = 0, big_val_cnt = gr->big_values;
    const uint8_t *sfb = gr->sfbtab;
    hcache hc;
    hc.next = bs->buf + bs->pos / 8;
    hc.cache = (((hc.next[0] * 256u + hc.next[1]) * 256u + hc.next[2]) * 256u + hc.next[3]) << (bs->pos & 7);
    hc.sh = (bs->pos & 7) - 8;
    hc.next += 4;
    int pairs_to_decode, np;
    while (big_val_cnt > 0) {
        int tab_num = gr->table_select[ireg];
        int sfb_cnt = gr->region_count[ireg++];
        int book = k_book_of_table[tab_num];
        int linbits = k_linbits[tab_num];
        do {
            np = *sfb++ / 2;
            pairs_to_decode = big_val_cnt < np ? big_val_cnt : np;
            one = *scf++;
            do {
                int len = 0, leaf = 0;
                if (book) leaf = book_lookup(book, hc.cache, &len);
                FLUSH(len);
                for (int j = 0; j < 2; j++, dst++, leaf >>= 4) {
                    int lsb = leaf & 0x0F;
                    if (linbits && lsb == 15) {
                        lsb += (int)(hc.cache >> (32 - linbits));
                        FLUSH(linbits);
                        REFILL();
                        *dst = one * pow_43(lsb) * ((int32_t)hc.cache < 0 ? -1 : 1);
                    } else {
                        *dst = f32(k_pow43_bits[16 + lsb - 16 * (int)(hc.cache >> 31)]) * one;
                    }
                    FLUSH(lsb ? 1 : 0);
                }
                REFILL();
            } while (--pairs_to_decode);
        } while ((big_val_cnt -= np) > 0 && --sfb_cnt >= 0);
    }
    for (np = 1 - big_val_cnt;; dst += 4) {
        const unsigned char (*c1)[3] = k_count1_codes[gr->count1_table ? 1 : 0];
        int len = 0, flags = 0;
        for (int i = 0; i < 16; i++) {
            if ((hc.cache >> (32 - c1[i][0])) == c1[i][1]) {
                len = c1[i][0];
                flags = c1[i][2];
                break;
            }
        }
        FLUSH(len);
        if (((hc.next - bs->buf) * 8 - 24 + hc.sh) > layer3gr_limit) break;
        if (!--np) { np = *sfb++ / 2; if (!np) break; one = *scf++; }
        if (flags & 8) { dst[0] = ((int32_t)hc.cache < 0) ? -one : one; FLUSH(1); }
        if (flags & 4) { dst[1] = ((int32_t)hc.cache < 0) ? -one : one; FLUSH(1); }
        if (!--np) { np = *sfb++ / 2; if (!np) break; one = *scf++; }
        if (flags & 2) { dst[2] = ((int32_t)hc.cache < 0) ? -one : one; FLUSH(1); }
        if (flags & 1) { dst[3] = ((int32_t)hc.cache < 0) ? -one : one; FLUSH(1); }
        REFILL();
    }
    bs->pos = layer3gr_limit;
}

/* mp3:885-896 */
static void midside_stereo(float *left, int n)
{
    float *right = left + 576;
    for (int i = 0; i < n; i++) {
        float a = left[i], b = right[i];
        left[i] = a + b;
        right[i] = a - b;
    }
}
/* mp3:898-906 */
static void intensity_stereo_band(float *left, int n, float kl, float kr)
{
    for (int i = 0; i < n; i++) {
        left[i + 576] = left[i] * kr;
        left[i] = left[i] * kl;
    }
}
/* mp3:908-926 */
static void stereo_top_band(const float *right, const uint8_t *sfb, int nbands, int *max_band)
{
    max_band[0] = max_band[1] = max_band[2] = -1;
    for (int i = 0; i < nbands; i++) {
        for (int k = 0; k < sfb[i]; k += 2) {
            if (right[k] != 0 || right[k + 1] != 0) {
                max_band[i % 3] = i;
                break;
            }
        }
        right += sfb[i];
    }
}
/* mp3:928-962 */
static void stereo_process(float *left, const uint8_t *ist_pos, const uint8_t *sfb, const uint8_t *hdr, int *max_band, int mpeg2_sh)
{
    unsigned max_pos = h_mpeg1(hdr) ? 7 : 64;
    for (unsigned i = 0; sfb[i]; i++) {
        unsigned ipos = ist_pos[i];
        if ((int)i > max_band[i % 3] && ipos < max_pos) {
            float kl, kr, s = h_msbit(hdr) ? 1.41421356f : 1;
            if (h_mpeg1(hdr)) {
                kl = f32(k_pan_bits[2 * ipos]);
                kr = f32(k_pan_bits[2 * ipos + 1]);
            } else {
                kl = 1;
                kr = ldexp_q2(1, (int)((ipos + 1) >> 1 << mpeg2_sh));
                if (ipos & 1) {
                    kl = kr;
                    kr = 1;
                }
            }
            intensity_stereo_band(left, sfb[i], kl * s, kr * s);
        } else if (h_msbit(hdr)) {
            midside_stereo(left, sfb[i]);
        }
        left += sfb[i];
    }
}
/* mp3:964-983 */
static void intensity_stereo(float *left, uint8_t *ist_pos, const gr_info_t *gr, const uint8_t *hdr)
{
    int max_band[3], n_sfb = gr->n_long_sfb + gr->n_short_sfb;
    int max_blocks = gr->n_short_sfb ? 3 : 1;
    stereo_top_band(left + 576, gr->sfbtab, n_sfb, max_band);
    if (gr->n_long_sfb) {
        int m = max_band[0] > max_band[1] ? max_band[0] : max_band[1];
        m = m > max_band[2] ? m : max_band[2];
        max_band[0] = max_band[1] = max_band[2] = m;
    }
    for (int i = 0; i < max_blocks; i++) {
        int default_pos = h_mpeg1(hdr) ? 3 : 0;
        int itop = n_sfb - max_blocks + i;
        int prev = itop - max_blocks;
        ist_pos[itop] = (uint8_t)(max_band[i] >= prev ? default_pos : ist_pos[prev]);
    }
    stereo_process(left, ist_pos, gr->sfbtab, hdr, max_band, gr[1].scalefac_compress & 1);
}
/* mp3:985-1000 */
static void reorder(float *grbuf, float *scratch, const uint8_t *sfb)
{
    int len;
    float *src = grbuf, *dst = scratch;
    for (; 0 != (len = *sfb); sfb += 3, src += 2 * len) {
        for (int i = 0; i < len; i++, src++) {
            *dst++ = src[0 * len];
            *dst++ = src[1 * len];
            *dst++ = src[2 * len];
        }
    }
    memcpy(grbuf, scratch, (size_t)(dst - scratch) * sizeof(float));
}

/* ---- decoder state, mp3:37-46 (the transform part lives in afgo_mp3_state) ---- */
typedef struct {
    afgo_mp3_state tr;
    int reserv, free_format_bytes;
    uint8_t header[4], reserv_buf[511];
} dec_t;

typedef struct {
    bs_t bs;
    uint8_t maindata[MAX_BITRESERVOIR_BYTES + MAX_L3_FRAME_PAYLOAD_BYTES + 2048];   /* zeroed slack: a damaged granule may read far past its data */
    gr_info_t gr_info[4];
    float grbuf[2][576];
    float scf[40];
    float syn[(18 + 15) * 64];
    uint8_t ist_pos[2][39];
} scratch_t;

/* mp3:1170-1197 */
static void save_reservoir(dec_t *h, scratch_t *s)
{
    int pos = (int)((unsigned)(s->bs.pos + 7) / 8u);
    int remains = (int)((unsigned)s->bs.limit / 8u) - pos;
    if (remains > MAX_BITRESERVOIR_BYTES) {
        pos += remains - MAX_BITRESERVOIR_BYTES;
        remains = MAX_BITRESERVOIR_BYTES;
    }
    if (remains > 0) memmove(h->reserv_buf, s->maindata + pos, (size_t)remains);
    h->reserv = remains;
}
static int restore_reservoir(dec_t *h, bs_t *bs, scratch_t *s, int main_data_begin)
{
    int frame_bytes = (bs->limit - bs->pos) / 8;
    int bytes_have = h->reserv < main_data_begin ? h->reserv : main_data_begin;
    memcpy(s->maindata, h->reserv_buf + (h->reserv - main_data_begin > 0 ? h->reserv - main_data_begin : 0), (size_t)bytes_have);
    memcpy(s->maindata + bytes_have, bs->buf + bs->pos / 8, (size_t)frame_bytes);
    bs_init(&s->bs, s->maindata, bytes_have + frame_bytes);
    return h->reserv >= main_data_begin;
}

/* growing record planes */
typedef struct {
    afgo_mp3_file *f;
    size_t cap_blocks, cap_pcm, cap_streams;
    int new_stream;                 /* the decoder state was reset: the next granule opens a new run */
} sink_t;

static int grow(void **p, size_t *cap, size_t need, size_t elem)
{
    if (need <= *cap) return 1;
    size_t nc = *cap ? *cap * 2 : 64;
    while (nc < need) nc *= 2;
    void *q = realloc(*p, nc * elem);
    if (!q) return 0;
    *p = q;
    *cap = nc;
    return 1;
}

/* mp3:1199-1230 up to the seam, then the transform oracle (mp3:1226-1228 + :1553) */
static int l3_decode_granule(dec_t *h, scratch_t *s, gr_info_t *gr_info, int nch, float *pcm, sink_t *sink)
{
    uint32_t flags[2] = { 0, 0 };
    for (int ch = 0; ch < nch; ch++) {
        int layer3gr_limit = s->bs.pos + gr_info[ch].part_23_length;
        decode_scalefactors(h->header, s->ist_pos[ch], &s->bs, gr_info + ch, s->scf, ch);
        huffman(s->grbuf[ch], &s->bs, gr_info + ch, s->scf, layer3gr_limit);
    }
    if (h_istereo(h->header)) intensity_stereo(s->grbuf[0], s->ist_pos[1], gr_info, h->header);
    else if (h_ms(h->header)) midside_stereo(s->grbuf[0], 576);
    for (int ch = 0; ch < nch; ch++) {
        const gr_info_t *g = gr_info + ch;
        int aa_bands = 31;
        int n_long_bands = (g->mixed_block_flag ? 2 : 0) << (int)(h_my_sr(h->header) == 2);
        if (g->n_short_sfb) {
            aa_bands = n_long_bands - 1;
            reorder(s->grbuf[ch] + n_long_bands * 18, s->syn, g->sfbtab + g->n_long_sfb);
        }
        flags[ch] = (uint32_t)g->block_type | ((uint32_t)n_long_bands << 8) | ((uint32_t)(aa_bands + 1) << 16);
    }
    if (!sink) {                                   /* index scan of the open step: nothing is kept */
        afgo_mp3_granule(&h->tr, s->grbuf[0], flags, nch, pcm);
        return 1;
    }
    /* record the seam */
    afgo_mp3_file *f = sink->f;
    if (sink->new_stream || f->n_streams == 0) {
        if (!grow((void **)&f->stream_granules, &sink->cap_streams, f->n_streams + 1, sizeof(uint32_t))) return 0;
        f->stream_granules[f->n_streams++] = 0;
        sink->new_stream = 0;
    }
    size_t need = (size_t)f->n_blocks + (size_t)nch;
    size_t cap2 = sink->cap_blocks;
    if (!grow((void **)&f->coef, &sink->cap_blocks, need, 576 * sizeof(float))) return 0;
    if (!grow((void **)&f->flags, &cap2, need, sizeof(uint32_t))) return 0;
    for (int ch = 0; ch < nch; ch++) {
        memcpy(f->coef + (f->n_blocks + (size_t)ch) * 576, s->grbuf[ch], 576 * sizeof(float));
        f->flags[f->n_blocks + (size_t)ch] = flags[ch];
    }
    f->n_blocks += (uint64_t)nch;
    f->stream_granules[f->n_streams - 1]++;
    afgo_mp3_granule(&h->tr, s->grbuf[0], flags, nch, pcm);
    return 1;
}

/* mp3:1436-1484 */
static int match_frame(const uint8_t *hdr, int mp3_bytes, int frame_bytes)
{
    int i, nmatch;
    for (i = 0, nmatch = 0; nmatch < MAX_FRAME_SYNC_MATCHES; nmatch++) {
        i += hdr_frame_bytes(hdr + i, frame_bytes) + hdr_padding(hdr + i);
        if (i + HDR_SIZE > mp3_bytes) return nmatch > 0;
        if (!hdr_compare(hdr, hdr + i)) return 0;
    }
    return 1;
}
static int find_frame(const uint8_t *mp3, int mp3_bytes, int *free_format_bytes, int *ptr_frame_bytes)
{
    int i, k;
    for (i = 0; i < mp3_bytes - HDR_SIZE; i++, mp3++) {
        if (hdr_valid(mp3)) {
            int frame_bytes = hdr_frame_bytes(mp3, *free_format_bytes);
            int frame_and_padding = frame_bytes + hdr_padding(mp3);
            for (k = HDR_SIZE; !frame_bytes && k < MAX_FREE_FORMAT_FRAME_SIZE && i + 2 * k < mp3_bytes - HDR_SIZE; k++) {
                if (hdr_compare(mp3, mp3 + k)) {
                    int fb = k - hdr_padding(mp3);
                    int nextfb = fb + hdr_padding(mp3 + k);
                    if (i + k + nextfb + HDR_SIZE > mp3_bytes || !hdr_compare(mp3, mp3 + k + nextfb)) continue;
                    frame_and_padding = k;
                    frame_bytes = fb;
                    *free_format_bytes = fb;
                }
            }
            if ((frame_bytes && i + frame_and_padding <= mp3_bytes && match_frame(mp3, mp3_bytes - i, frame_bytes)) ||
                (!i && frame_and_padding == mp3_bytes)) {
                *ptr_frame_bytes = frame_and_padding;
                return i;
            }
            *free_format_bytes = 0;
        }
    }
    *ptr_frame_bytes = 0;
    return mp3_bytes;
}


/* ---- Layer I / II, mp3:286-484 -------------------------------------------------------------------------------- */
typedef struct { uint8_t tab_offset, code_tab_width, band_count; } l12_alloc_t;
typedef struct {
    float scf[3 * 64];
    uint8_t total_bands, stereo_bands, bitalloc[64], scfcod[64];
} l12_scale_info;

static const l12_alloc_t *l12_subband_alloc_table(const uint8_t *hdr, l12_scale_info *sci)     /* :286-346 */
{
    static const l12_alloc_t g_alloc_L1[] = { { 76, 4, 32 } };
    static const l12_alloc_t g_alloc_L2M2[] = { { 60, 4, 4 }, { 44, 3, 7 }, { 44, 2, 19 } };
    static const l12_alloc_t g_alloc_L2M1[] = { { 0, 4, 3 }, { 16, 4, 8 }, { 32, 3, 12 }, { 40, 2, 7 } };
    static const l12_alloc_t g_alloc_L2M1_lowrate[] = { { 44, 4, 2 }, { 44, 3, 10 } };
    const l12_alloc_t *alloc;
    int mode = (hdr[3] >> 6) & 3;
    int nbands, stereo_bands = (mode == 3) ? 0 : (mode == 1) ? ((((hdr[3]) >> 4) & 3) << 2) + 4 : 32;
    if (h_l1(hdr)) {
        alloc = g_alloc_L1;
        nbands = 32;
    } else if (!h_mpeg1(hdr)) {
        alloc = g_alloc_L2M2;
        nbands = 30;
    } else {
        int sample_rate_idx = (hdr[2] >> 2) & 3;
        unsigned kbps = hdr_kbps(hdr) >> (int)(mode != 3);
        if (!kbps) kbps = 192;                            /* free-format */
        alloc = g_alloc_L2M1;
        nbands = 27;
        if (kbps < 56) {
            alloc = g_alloc_L2M1_lowrate;
            nbands = sample_rate_idx == 2 ? 12 : 8;
        } else if (kbps >= 96 && sample_rate_idx != 1) {
            nbands = 30;
        }
    }
    sci->total_bands = (uint8_t)nbands;
    sci->stereo_bands = (uint8_t)(stereo_bands < nbands ? stereo_bands : nbands);
    return alloc;
}

static void l12_read_scalefactors(bs_t *bs, uint8_t *pba, uint8_t *scfcod, int bands, float *scf)      /* :348-384 */
{
    /* double literals converted to float, as the D initialiser does */
    static const float g_deq_L12[18 * 3] = {
        (float)3.17891e-07, (float)2.52311e-07, (float)2.00259e-07, (float)1.36239e-07, (float)1.08133e-07, (float)8.58253e-08,
        (float)6.35783e-08, (float)5.04621e-08, (float)4.00518e-08, (float)3.07637e-08, (float)2.44172e-08, (float)1.93799e-08,
        (float)1.51377e-08, (float)1.20148e-08, (float)9.53615e-09, (float)7.50925e-09, (float)5.96009e-09, (float)4.73053e-09,
        (float)3.7399e-09, (float)2.96836e-09, (float)2.35599e-09, (float)1.86629e-09, (float)1.48128e-09, (float)1.17569e-09,
        (float)9.32233e-10, (float)7.39914e-10, (float)5.8727e-10, (float)4.65889e-10, (float)3.69776e-10, (float)2.93492e-10,
        (float)2.32888e-10, (float)1.84843e-10, (float)1.4671e-10, (float)1.1643e-10, (float)9.24102e-11, (float)7.3346e-11,
        (float)5.82112e-11, (float)4.62023e-11, (float)3.66708e-11, (float)2.91047e-11, (float)2.31004e-11, (float)1.83348e-11,
        (float)1.45521e-11, (float)1.155e-11, (float)9.16727e-12, (float)3.17891e-07, (float)2.52311e-07, (float)2.00259e-07,
        (float)1.90735e-07, (float)1.51386e-07, (float)1.20155e-07, (float)1.05964e-07, (float)8.41035e-08, (float)6.6753e-08
    };
    for (int i = 0; i < bands; i++) {
        float s = 0;
        int ba = *pba++;
        int mask = ba ? 4 + ((19 >> scfcod[i]) & 3) : 0;
        for (int m = 4; m; m >>= 1) {
            if (mask & m) {
                int b = (int)get_bits(bs, 6);
                s = g_deq_L12[ba * 3 - 6 + b % 3] * (float)(1 << 21 >> b / 3);
            }
            *scf++ = s;
        }
    }
}

static void l12_read_scale_info(const uint8_t *hdr, bs_t *bs, l12_scale_info *sci)                    /* :386-435 */
{
    static const uint8_t g_bitalloc_code_tab[] = {
        0,17, 3, 4, 5,6,7, 8,9,10,11,12,13,14,15,16,
        0,17,18, 3,19,4,5, 6,7, 8, 9,10,11,12,13,16,
        0,17,18, 3,19,4,5,16,
        0,17,18,16,
        0,17,18,19, 4,5,6, 7,8, 9,10,11,12,13,14,15,
        0,17,18, 3,19,4,5, 6,7, 8, 9,10,11,12,13,14,
        0, 2, 3, 4, 5,6,7, 8,9,10,11,12,13,14,15,16
    };
    const l12_alloc_t *subband_alloc = l12_subband_alloc_table(hdr, sci);
    int i, k = 0, ba_bits = 0;
    const uint8_t *ba_code_tab = g_bitalloc_code_tab;
    for (i = 0; i < sci->total_bands; i++) {
        uint8_t ba;
        if (i == k) {
            k += subband_alloc->band_count;
            ba_bits = subband_alloc->code_tab_width;
            ba_code_tab = g_bitalloc_code_tab + subband_alloc->tab_offset;
            subband_alloc++;
        }
        ba = ba_code_tab[get_bits(bs, ba_bits)];
        sci->bitalloc[2 * i] = ba;
        if (i < sci->stereo_bands) ba = ba_code_tab[get_bits(bs, ba_bits)];
        sci->bitalloc[2 * i + 1] = sci->stereo_bands ? ba : 0;
    }
    for (i = 0; i < 2 * sci->total_bands; i++) {
        uint8_t temp = h_l1(hdr) ? 2 : (uint8_t)get_bits(bs, 2);
        sci->scfcod[i] = sci->bitalloc[i] ? temp : 6;
    }
    l12_read_scalefactors(bs, sci->bitalloc, sci->scfcod, sci->total_bands * 2, sci->scf);
    for (i = sci->stereo_bands; i < sci->total_bands; i++) sci->bitalloc[2 * i + 1] = 0;
}

static int l12_dequantize_granule(float *grbuf, bs_t *bs, l12_scale_info *sci, int group_size)         /* :437-471 */
{
    int i, j, k, choff = 576;
    for (j = 0; j < 4; j++) {
        float *dst = grbuf + group_size * j;
        for (i = 0; i < 2 * sci->total_bands; i++) {
            int ba = sci->bitalloc[i];
            if (ba != 0) {
                if (ba < 17) {
                    int half = (1 << (ba - 1)) - 1;
                    for (k = 0; k < group_size; k++) dst[k] = (float)((int)get_bits(bs, ba) - half);
                } else {
                    unsigned mod = (2u << (ba - 17)) + 1;                 /* 3, 5, 9 */
                    unsigned code = get_bits(bs, (int)(mod + 2 - (mod >> 3)));  /* 5, 7, 10 */
                    for (k = 0; k < group_size; k++, code /= mod) dst[k] = (float)((int)(code % mod - mod / 2));
                }
            }
            dst += choff;
            choff = 18 - choff;
        }
    }
    return group_size * 4;
}

static void l12_apply_scf_384(l12_scale_info *sci, const float *scf, float *dst)                        /* :473-485 */
{
    memcpy(dst + 576 + sci->stereo_bands * 18, dst + sci->stereo_bands * 18, (size_t)(sci->total_bands - sci->stereo_bands) * 18 * sizeof(float));
    for (int i = 0; i < sci->total_bands; i++, dst += 18, scf += 6) {
        for (int k = 0; k < 12; k++) {
            dst[k + 0] *= scf[0];
            dst[k + 576] *= scf[3];
        }
    }
}

/* records one Layer I/II synthesis granule (12 time slots, layout [ch][band * 18 + slot], mp3:1563-1566) like a Layer III
 * one: 576 floats per channel with flag AFGO_MP3_L12 (its slots 12..17 are zero and are NOT part of the signal) */
#define AFGO_MP3_L12 0x40000000u
static int l12_record(sink_t *sink, const float *grbuf, int nch)
{
    afgo_mp3_file *f = sink->f;
    if (sink->new_stream || f->n_streams == 0) {
        if (!grow((void **)&f->stream_granules, &sink->cap_streams, f->n_streams + 1, sizeof(uint32_t))) return 0;
        f->stream_granules[f->n_streams++] = 0;
        sink->new_stream = 0;
    }
    size_t need = (size_t)f->n_blocks + (size_t)nch, cap2 = sink->cap_blocks;
    if (!grow((void **)&f->coef, &sink->cap_blocks, need, 576 * sizeof(float))) return 0;
    if (!grow((void **)&f->flags, &cap2, need, sizeof(uint32_t))) return 0;
    for (int ch = 0; ch < nch; ch++) {
        memcpy(f->coef + (f->n_blocks + (size_t)ch) * 576, grbuf + 576 * ch, 576 * sizeof(float));
        f->flags[f->n_blocks + (size_t)ch] = AFGO_MP3_L12;
    }
    f->n_blocks += (uint64_t)nch;
    f->stream_granules[f->n_streams - 1]++;
    return 1;
}

typedef struct { int frame_bytes, frame_offset, channels, hz, layer, bitrate_kbps; } frame_info_t;

/* mp3:1491-1581.  pcm == NULL: header only (returns the frame's sample count). */
static int decode_frame(dec_t *dec, const uint8_t *mp3, int mp3_bytes, float *pcm, frame_info_t *info, sink_t *sink)
{
    int i = 0, igr, frame_size = 0, success = 1;
    const uint8_t *hdr;
    bs_t bs_frame;
    static _Thread_local scratch_t scratch;      /* (a stack variable in the reference; one per thread: bench.py times files on a thread pool) */
    if (mp3_bytes > 4 && dec->header[0] == 0xff && hdr_compare(dec->header, mp3)) {
        frame_size = hdr_frame_bytes(mp3, dec->free_format_bytes) + hdr_padding(mp3);
        if (frame_size != mp3_bytes && (frame_size + HDR_SIZE > mp3_bytes || !hdr_compare(mp3, mp3 + frame_size))) frame_size = 0;
    }
    if (!frame_size) {
        memset(dec, 0, sizeof(*dec));
        if (sink) sink->new_stream = 1;
        i = find_frame(mp3, mp3_bytes, &dec->free_format_bytes, &frame_size);
        if (!frame_size || i + frame_size > mp3_bytes) {
            info->frame_bytes = i;
            return 0;
        }
    }
    hdr = mp3 + i;
    memcpy(dec->header, hdr, HDR_SIZE);
    info->frame_bytes = i + frame_size;
    info->frame_offset = i;
    info->channels = h_mono(hdr) ? 1 : 2;
    info->hz = (int)hdr_hz(hdr);
    info->layer = 4 - h_layer(hdr);
    info->bitrate_kbps = (int)hdr_kbps(hdr);
    if (!pcm) return (int)hdr_frame_samples(hdr);
    bs_init(&bs_frame, hdr + HDR_SIZE, frame_size - HDR_SIZE);
    if (h_crc(hdr)) get_bits(&bs_frame, 16);
    if (info->layer != 3) {                                            /* Layer I / II, mp3:1557-1578 */
        l12_scale_info sci;
        float lins[(12 + 15) * 64];
        /* (records of a frame that fails below are dropped again: the reference discards the whole frame, :1573-1577) */
        const uint64_t blocks_at_entry = sink ? sink->f->n_blocks : 0;
        const uint32_t streams_at_entry = sink ? sink->f->n_streams : 0;
        const uint32_t last_at_entry = (sink && sink->f->n_streams) ? sink->f->stream_granules[sink->f->n_streams - 1] : 0;
        memset(&sci, 0, sizeof(sci));
        l12_read_scale_info(hdr, &bs_frame, &sci);
        memset(scratch.grbuf, 0, sizeof(scratch.grbuf));
        for (i = 0, igr = 0; igr < 3; igr++) {
            if (12 == (i += l12_dequantize_granule(scratch.grbuf[0] + i, &bs_frame, &sci, info->layer | 1))) {
                i = 0;
                l12_apply_scf_384(&sci, sci.scf + igr, scratch.grbuf[0]);
                if (sink && !l12_record(sink, scratch.grbuf[0], info->channels)) return -2;
                memset(lins, 0, sizeof(lins));                          /* (uninitialised scratch in the reference, see afgo_mp3_granule) */
                afgo_mp3_synth_granule(dec->tr.qmf_state, scratch.grbuf[0], 12, info->channels, pcm, lins);
                memset(scratch.grbuf, 0, sizeof(scratch.grbuf));
                pcm += 384 * info->channels;
            }
            if (bs_frame.pos > bs_frame.limit) {
                dec->header[0] = 0;                                    /* mp3dec_init */
                if (sink) {
                    sink->f->n_blocks = blocks_at_entry;
                    sink->f->n_streams = streams_at_entry;
                    if (streams_at_entry) sink->f->stream_granules[streams_at_entry - 1] = last_at_entry;
                }
                return 0;
            }
        }
        return success * (int)hdr_frame_samples(dec->header);
    }
    memset(&scratch.maindata, 0, sizeof(scratch.maindata));           /* the reference leaves the tail undefined */
    memset(scratch.ist_pos, 0, sizeof(scratch.ist_pos));              /* ... and ist_pos: uninitialised stack in the reference
                                                                         (minimp3.d:179-181); zero at every frame here and in the product */
    int main_data_begin = read_side_info(&bs_frame, scratch.gr_info, hdr);
    if (main_data_begin < 0 || bs_frame.pos > bs_frame.limit) {
        dec->header[0] = 0;                                            /* mp3dec_init, mp3:1486-1489 */
        return 0;
    }
    success = restore_reservoir(dec, &bs_frame, &scratch, main_data_begin);
    if (success) {
        for (igr = 0; igr < (h_mpeg1(hdr) ? 2 : 1); igr++, pcm += 576 * info->channels) {
            memset(scratch.grbuf, 0, sizeof(scratch.grbuf));
            if (!l3_decode_granule(dec, &scratch, scratch.gr_info + igr * info->channels, info->channels, pcm, sink)) return -2;
        }
    }
    save_reservoir(dec, &scratch);
    return success * (int)hdr_frame_samples(dec->header);
}

/* ---- ID3 / APE, ex:93-142 ---- */
static void skip_id3v1(const uint8_t *buf, size_t *pbuf_size)
{
    size_t n = *pbuf_size;
    if (n >= 128 && !memcmp(buf + n - 128, "TAG", 3)) {
        n -= 128;
        if (n >= 227 && !memcmp(buf + n - 227, "TAG+", 4)) n -= 227;
    }
    if (n > 32 && !memcmp(buf + n - 32, "APETAGEX", 8)) {
        n -= 32;
        const uint8_t *tag = buf + n + 8 + 4;
        uint32_t tag_size = ((uint32_t)tag[3] << 24) | ((uint32_t)tag[2] << 16) | ((uint32_t)tag[1] << 8) | tag[0];
        if (n >= tag_size) n -= tag_size;
    }
    *pbuf_size = n;
}
static size_t skip_id3v2(const uint8_t *buf, size_t n)
{
    if (n >= 10 && !memcmp(buf, "ID3", 3) && !((buf[5] & 15) || (buf[6] & 0x80) || (buf[7] & 0x80) || (buf[8] & 0x80) || (buf[9] & 0x80))) {
        size_t sz = (size_t)(((buf[6] & 0x7f) << 21) | ((buf[7] & 0x7f) << 14) | ((buf[8] & 0x7f) << 7) | (buf[9] & 0x7f)) + 10;
        if (buf[5] & 16) sz += 10;
        return sz;
    }
    return 0;
}

/* ex:144-190 */
static int check_vbrtag(const uint8_t *frame, int frame_size, uint32_t *frames, int *delay, int *padding)
{
    bs_t bs;
    gr_info_t gr_info[4];
    bs_init(&bs, frame + HDR_SIZE, frame_size - HDR_SIZE);
    if (h_crc(frame)) get_bits(&bs, 16);
    if (read_side_info(&bs, gr_info, frame) < 0) return 0;
    const uint8_t *tag = frame + HDR_SIZE + bs.pos / 8;
    if (memcmp("Xing", tag, 4) && memcmp("Info", tag, 4)) return 0;
    int flags = tag[7];
    if (!(flags & 1)) return -1;
    tag += 8;
    *frames = ((uint32_t)tag[0] << 24) | ((uint32_t)tag[1] << 16) | ((uint32_t)tag[2] << 8) | tag[3];
    tag += 4;
    if (flags & 2) tag += 4;
    if (flags & 4) tag += 100;
    if (flags & 8) tag += 4;
    *delay = *padding = 0;
    if (*tag) {
        tag += 21;
        if (tag - frame + 14 >= frame_size) return 0;
        *delay = ((tag[0] << 4) | (tag[1] >> 4)) + (528 + 1);
        *padding = (((tag[1] & 0xF) << 8) | tag[2]) - (528 + 1);
    }
    return 1;
}

/* Whole-file drive: mp3dec_ex_open (ex:566-639, index scan or VBR tag) then mp3dec_ex_read to the end
 * (ex:787-888) over the in-memory file (the callback-I/O windowing of ex:490-564 sees the same frames
 * whenever ten consecutive frames fit its 16 KiB look-ahead). */
int afgo_mp3_decode_file(const uint8_t *data, size_t size, afgo_mp3_file *f)
{
    memset(f, 0, sizeof(*f));
    if (!data || size < 10) return -1;
    const uint8_t *buf = data;
    size_t buf_size = size;
    {   /* mp3dec_skip_id3, ex:127-142 */
        size_t id3 = skip_id3v2(buf, buf_size);
        if (id3) {
            if (id3 >= buf_size) id3 = buf_size;
            buf += id3;
            buf_size -= id3;
        }
        skip_id3v1(buf, &buf_size);
    }
    if (!buf_size) return -1;
    dec_t *dec = (dec_t *)calloc(1, sizeof(dec_t));
    float *frame_pcm = (float *)malloc(sizeof(float) * 2304);
    sink_t sink;
    memset(&sink, 0, sizeof(sink));
    sink.f = f;
    if (!dec || !frame_pcm) { free(dec); free(frame_pcm); return -2; }

    /* ---- open: mp3dec_iterate_buf + mp3dec_load_index ---- */
    uint64_t start_offset = 0, samples = 0, detected_samples = 0;
    int to_skip = 0, have_info = 0, num_frames = 0, buffer_samples = 0, vbr = 0, free_format = 0;
    frame_info_t first;
    memset(&first, 0, sizeof(first));
    {
        const uint8_t *p = buf;
        size_t left = buf_size;
        for (;;) {
            int ffb = 0, frame_size = 0;
            int i = find_frame(p, (int)(left > 0x7fffffff ? 0x7fffffff : left), &ffb, &frame_size);
            p += i;
            left -= (size_t)i;
            if (i && !frame_size) continue;
            if (!frame_size) break;
            frame_info_t fi;
            fi.channels = h_mono(p) ? 1 : 2;
            fi.hz = (int)hdr_hz(p);
            fi.layer = 4 - h_layer(p);
            fi.bitrate_kbps = (int)hdr_kbps(p);
            fi.frame_bytes = frame_size;
            fi.frame_offset = 0;
            uint64_t offset = (uint64_t)(p - buf);
            if (!have_info) {                                          /* ex:570-603 */
                have_info = 1;
                first = fi;
                start_offset = offset;
                free_format = ffb;
                if (fi.layer == 3) {
                    uint32_t frames = 0;
                    int delay = 0, padding = 0;
                    int ret = check_vbrtag(p, frame_size, &frames, &delay, &padding);
                    if (ret) start_offset = offset + (uint64_t)frame_size;
                    if (ret > 0) {
                        padding *= fi.channels;
                        to_skip = delay * fi.channels;
                        samples = (uint64_t)hdr_frame_samples(p) * (uint64_t)fi.channels * (uint64_t)frames;
                        if (samples >= (uint64_t)to_skip) samples -= (uint64_t)to_skip;
                        if (padding > 0 && samples >= (uint64_t)padding) samples -= (uint64_t)padding;
                        detected_samples = samples;
                        vbr = 1;
                        break;
                    } else if (ret < 0) {
                        p += frame_size;
                        left -= (size_t)frame_size;
                        continue;
                    }
                }
            }
            num_frames++;
            if (!buffer_samples && num_frames < 256) {                  /* ex:616-621 */
                frame_info_t tmp;
                buffer_samples = decode_frame(dec, p, (int)(left > 0x7fffffff ? 0x7fffffff : left), frame_pcm, &tmp, NULL);
                if (buffer_samples < 0) buffer_samples = 0;
                samples += (uint64_t)buffer_samples * (uint64_t)fi.channels;
            } else {
                samples += (uint64_t)hdr_frame_samples(p) * (uint64_t)fi.channels;
            }
            p += frame_size;
            left -= (size_t)frame_size;
        }
    }
    (void)free_format;
    if (!have_info) { free(dec); free(frame_pcm); return -1; }
    /* the index scan decoded with a sink-less decoder: forget everything it recorded (nothing) and reset */
    memset(dec, 0, sizeof(*dec));                                       /* mp3dec_init + the memset of mp3:1508 */
    f->channels = first.channels;
    f->hz = first.hz;
    f->layer = first.layer;
    f->vbr_tag_found = vbr;
    f->start_delay = to_skip;
    f->detected_samples = detected_samples;
    f->samples = samples;

    /* ---- read everything: mp3dec_ex_read ---- */
    uint64_t offset = start_offset, cur_sample = 0;
    sink.new_stream = 1;
    for (;;) {
        if (detected_samples && cur_sample >= detected_samples) break;
        uint64_t left = buf_size - offset;
        if (!left) break;
        frame_info_t fi;
        memset(&fi, 0, sizeof(fi));
        const uint64_t blocks_before = f->n_blocks;
        const uint32_t streams_before = f->n_streams;
        const uint32_t last_before = f->n_streams ? f->stream_granules[f->n_streams - 1] : 0;
        int n = decode_frame(dec, buf + offset, (int)(left > 0x7fffffff ? 0x7fffffff : left), frame_pcm, &fi, &sink);
        if (n == -2) { free(dec); free(frame_pcm); return -2; }
        if (n < 0 || first.hz != fi.hz || first.layer != fi.layer || first.channels != fi.channels) {   /* MP3D_E_DECODE, ex:851-857 */
            f->n_blocks = blocks_before;                   /* nothing of this frame is delivered: drop its records */
            f->n_streams = streams_before;
            if (streams_before) f->stream_granules[streams_before - 1] = last_before;
            break;
        }
        if (n) {
            int buffer = n * fi.channels, consumed = 0;
            if (to_skip) {
                int skip = buffer < to_skip ? buffer : to_skip;
                consumed += skip;
                to_skip -= skip;
            }
            size_t to_copy = (size_t)(buffer - consumed);
            if (detected_samples && cur_sample + to_copy >= detected_samples) to_copy = (size_t)(detected_samples - cur_sample);
            if (!grow((void **)&f->pcm, &sink.cap_pcm, (size_t)f->pcm_samples + to_copy, sizeof(float))) { free(dec); free(frame_pcm); return -2; }
            memcpy(f->pcm + f->pcm_samples, frame_pcm + consumed, to_copy * sizeof(float));
            f->pcm_samples += to_copy;
            cur_sample += to_copy;
        } else if (to_skip) {
            int frame_samples = (int)hdr_frame_samples(buf + offset) * fi.channels;
            to_skip -= frame_samples < to_skip ? frame_samples : to_skip;
        }
        offset += (uint64_t)fi.frame_bytes;
    }
    free(dec);
    free(frame_pcm);
    return 0;
}

void afgo_mp3_file_free(afgo_mp3_file *f)
{
    free(f->stream_granules);
    free(f->coef);
    free(f->flags);
    free(f->pcm);
    memset(f, 0, sizeof(*f));
}
