/*
 * vorbis_frontend.c -- ORACLE (test infrastructure only): CPU restatement of the reference's Ogg Vorbis
 * front-end (stb_vorbis2.d): Ogg page / segment / packet walking, the three header packets (codebooks,
 * floors, residues, mappings, modes), and per audio packet the mode / window decision, floor 1 decode,
 * residue decode (types 0, 1, 2), inverse coupling and floor curve multiplication -- everything up to the
 * transform seam (stb_vorbis2.d:2526) -- plus the sample bookkeeping of the pull API (first frame discarded,
 * truncation by the last page's granule position, stream length from the last page).
 *
 * PARITY UNPINNED: the reference ships no Vorbis vectors and cannot be built here (D).  What pins this file:
 * a real Ogg Vorbis file (tests/golden/, MathJax earcon) must decode to the same sound as the MP3 encoding of
 * the same earcon, and the product's independently written parser must agree bit for bit.
 *
 * The control flow is the reference's (its streaming state machine over get8()), only the I/O is a memory
 * buffer.  "vb:" = source/audioformats/stb_vorbis2.d.
 */
#include "afg_oracle.h"
#include "vorbis_front_tables.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX_CH 16
#define FAST_LEN 10                       /* STB_VORBIS_FAST_HUFFMAN_LENGTH, vb:221 */
#define FAST_SIZE (1 << FAST_LEN)
#define NO_CODE 255
#define EOP (-1)
#define INVALID_BITS (-1)
#define PAGEFLAG_continued_packet 1
#define PAGEFLAG_first_page 2
#define PAGEFLAG_last_page 4

typedef struct {
    int dimensions, entries;
    uint8_t *codeword_lengths;
    float minimum_value, delta_value;
    uint8_t value_bits, lookup_type, sequence_p, sparse;
    uint32_t lookup_values;
    float *multiplicands;
    uint32_t *codewords;
    int16_t fast_huffman[FAST_SIZE];
    uint32_t *sorted_codewords;
    int *sorted_values_base, *sorted_values;
    int sorted_entries;
} Codebook;

typedef struct {
    uint8_t partitions, partition_class_list[32], class_dimensions[16], class_subclasses[16], class_masterbooks[16];
    int16_t subclass_books[16][8];
    uint16_t Xlist[31 * 8 + 2];
    uint8_t sorted_order[31 * 8 + 2], neighbors[31 * 8 + 2][2];
    uint8_t floor1_multiplier, rangebits;
    int values;
} Floor1;

typedef struct {
    uint32_t begin, end, part_size;
    uint8_t classifications, classbook;
    uint8_t **classdata;
    int16_t (*residue_books)[8];
} Residue;

typedef struct { uint8_t magnitude, angle, mux; } MappingChannel;
typedef struct {
    uint16_t coupling_steps;
    MappingChannel *chan;
    uint8_t submaps, submap_floor[15], submap_residue[15];
} Mapping;
typedef struct { uint8_t blockflag, mapping; uint16_t windowtype, transformtype; } Mode;

typedef struct {
    /* memory "file" */
    const uint8_t *data;
    size_t size, pos;
    int eof, error;
    unsigned sample_rate;
    int channels;
    uint32_t stream_len, first_audio_page_offset;
    int blocksize[2], blocksize_0, blocksize_1;
    int codebook_count;
    Codebook *codebooks;
    int floor_count;
    uint16_t floor_types[64];
    Floor1 *floor_config;
    int residue_count;
    uint16_t residue_types[64];
    Residue *residue_config;
    int mapping_count;
    Mapping *mapping;
    int mode_count;
    Mode mode_config[64];
    uint32_t total_samples;
    float *channel_buffers[MAX_CH];
    int16_t *finalY[MAX_CH];
    int previous_length;
    uint32_t current_loc;
    int current_loc_valid;
    int last_page, segment_count;
    uint8_t segments[255], page_flag, bytes_in_seg, first_decode;
    int next_seg, last_seg, last_seg_which;
    uint32_t acc;
    int valid_bits, packet_bytes, end_seg_with_known_loc;
    uint32_t known_loc_for_packet;
    int discard_samples_deferred;
    uint32_t crc_table[256];
    int seek_clears_eof;             /* 0: the reference (its set_file_offset leaves eof alone); 1: upstream stb_vorbis */
} vorb;

static float f32(unsigned bits) { float f; memcpy(&f, &bits, 4); return f; }
static int verror(vorb *f, int e) { f->error = e; return 0; }

/* ---- leaf helpers, vb:599-898 ---- */
static void crc32_init(vorb *f)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t s = i << 24;
        for (int j = 0; j < 8; ++j) s = (s << 1) ^ (s >= (1U << 31) ? 0x04c11db7u : 0);
        f->crc_table[i] = s;
    }
}
static uint32_t crc32_update(vorb *f, uint32_t crc, uint8_t byte) { return (crc << 8) ^ f->crc_table[byte ^ (crc >> 24)]; }

static uint32_t bit_reverse(uint32_t n)
{
    n = ((n & 0xAAAAAAAAu) >> 1) | ((n & 0x55555555u) << 1);
    n = ((n & 0xCCCCCCCCu) >> 2) | ((n & 0x33333333u) << 2);
    n = ((n & 0xF0F0F0F0u) >> 4) | ((n & 0x0F0F0F0Fu) << 4);
    n = ((n & 0xFF00FF00u) >> 8) | ((n & 0x00FF00FFu) << 8);
    return (n >> 16) | (n << 16);
}
static int ilog(int32_t n)
{
    static const signed char log2_4[16] = { 0, 1, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4 };
    if (n < 0) return 0;
    if (n < (1 << 14))
        if (n < (1 << 4)) return 0 + log2_4[n];
        else if (n < (1 << 9)) return 5 + log2_4[n >> 5];
        else return 10 + log2_4[n >> 10];
    else if (n < (1 << 24))
        if (n < (1 << 19)) return 15 + log2_4[n >> 15];
        else return 20 + log2_4[n >> 20];
    else if (n < (1 << 29)) return 25 + log2_4[n >> 25];
    else return 30 + log2_4[n >> 30];
}
static float float32_unpack(uint32_t x)                 /* vb:662-671 */
{
    uint32_t mantissa = x & 0x1fffff, sign = x & 0x80000000u, e = (x & 0x7fe00000u) >> 21;
    double res = sign ? -(double)mantissa : (double)mantissa;
    return (float)ldexp((float)res, (int)e - 788);
}
static void add_entry(Codebook *c, uint32_t huff_code, int symbol, int count, int len, uint32_t *values)
{
    if (!c->sparse) {
        c->codewords[symbol] = huff_code;
    } else {
        c->codewords[count] = huff_code;
        c->codeword_lengths[count] = (uint8_t)len;
        values[count] = (uint32_t)symbol;
    }
}
static int compute_codewords(Codebook *c, uint8_t *len, int n, uint32_t *values)    /* vb:691-737 */
{
    int i, k, m = 0;
    uint32_t available[32];
    memset(available, 0, sizeof(available));
    for (k = 0; k < n; ++k) if (len[k] < NO_CODE) break;
    if (k == n) return 1;
    add_entry(c, 0, k, m++, len[k], values);
    for (i = 1; i <= len[k]; ++i) available[i] = 1U << (32 - i);
    for (i = k + 1; i < n; ++i) {
        uint32_t res;
        int z = len[i], y;
        if (z == NO_CODE) continue;
        while (z > 0 && !available[z]) --z;
        if (z == 0) return 0;
        res = available[z];
        available[z] = 0;
        add_entry(c, bit_reverse(res), i, m++, len[i], values);
        if (z != len[i])
            for (y = len[i]; y > z; --y) available[y] = res + (1U << (32 - y));
    }
    return 1;
}
static void compute_accelerated_huffman(Codebook *c)     /* vb:739-759 */
{
    int i, len;
    for (i = 0; i < FAST_SIZE; ++i) c->fast_huffman[i] = -1;
    len = c->sparse ? c->sorted_entries : c->entries;
    if (len > 32767) len = 32767;
    for (i = 0; i < len; ++i) {
        if (c->codeword_lengths[i] <= FAST_LEN) {
            uint32_t z = c->sparse ? bit_reverse(c->sorted_codewords[i]) : c->codewords[i];
            while (z < FAST_SIZE) {
                c->fast_huffman[z] = (int16_t)i;
                z += 1U << c->codeword_lengths[i];
            }
        }
    }
}
static int u32cmp(const void *p, const void *q)
{
    uint32_t x = *(const uint32_t *)p, y = *(const uint32_t *)q;
    return x < y ? -1 : x > y;
}
static int include_in_sort(Codebook *c, uint8_t len)
{
    if (c->sparse) return 1;
    if (len == NO_CODE) return 0;
    return len > FAST_LEN;
}
static void compute_sorted_huffman(Codebook *c, uint8_t *lengths, uint32_t *values)    /* vb:776-828 */
{
    int i, len;
    if (!c->sparse) {
        int k = 0;
        for (i = 0; i < c->entries; ++i)
            if (include_in_sort(c, lengths[i])) c->sorted_codewords[k++] = bit_reverse(c->codewords[i]);
    } else {
        for (i = 0; i < c->sorted_entries; ++i) c->sorted_codewords[i] = bit_reverse(c->codewords[i]);
    }
    qsort(c->sorted_codewords, (size_t)c->sorted_entries, sizeof(uint32_t), u32cmp);
    c->sorted_codewords[c->sorted_entries] = 0xffffffffu;
    len = c->sparse ? c->sorted_entries : c->entries;
    for (i = 0; i < len; ++i) {
        uint8_t huff_len = c->sparse ? lengths[values[i]] : lengths[i];
        if (include_in_sort(c, huff_len)) {
            uint32_t code = bit_reverse(c->codewords[i]);
            int x = 0, n = c->sorted_entries;
            while (n > 1) {
                int m = x + (n >> 1);
                if (c->sorted_codewords[m] <= code) { x = m; n -= (n >> 1); }
                else n >>= 1;
            }
            if (c->sparse) {
                c->sorted_values[x] = (int)values[i];
                c->codeword_lengths[x] = huff_len;
            } else {
                c->sorted_values[x] = i;
            }
        }
    }
}
static int lookup1_values(int entries, int dim)          /* vb:838-848 */
{
    int r = (int)floor(exp((float)log((float)entries) / dim));
    if ((int)floor(pow((float)r + 1, dim)) <= entries) ++r;
    if (pow((float)r + 1, dim) <= entries) return -1;
    if ((int)floor(pow((float)r, dim)) > entries) return -1;
    return r;
}
static void neighbors(uint16_t *x, int n, int *plow, int *phigh)
{
    int low = -1, high = 65536;
    for (int i = 0; i < n; ++i) {
        if (x[i] > low && x[i] < x[n]) { *plow = i; low = x[i]; }
        if (x[i] < high && x[i] > x[n]) { *phigh = i; high = x[i]; }
    }
}
typedef struct { uint16_t x, id; } floor_ordering;
static int point_compare(const void *p, const void *q)
{
    const floor_ordering *a = (const floor_ordering *)p, *b = (const floor_ordering *)q;
    return a->x < b->x ? -1 : a->x > b->x;
}

/* ---- byte I/O over the memory buffer, vb:925-980 ---- */
static uint8_t get8(vorb *z)
{
    if (z->pos < z->size) return z->data[z->pos++];
    z->eof = 1;
    return 0;
}
static uint32_t get32(vorb *f)
{
    uint32_t x = get8(f);
    x += (uint32_t)get8(f) << 8;
    x += (uint32_t)get8(f) << 16;
    x += (uint32_t)get8(f) << 24;
    return x;
}
static int getn(vorb *z, uint8_t *data, int n)
{
    if (z->pos + (size_t)n <= z->size) {
        memcpy(data, z->data + z->pos, (size_t)n);
        z->pos += (size_t)n;
        return 1;
    }
    z->pos = z->size;                                   /* a short read consumes what is there */
    z->eof = 1;
    return 0;
}
static int skip_bytes(vorb *z, int n)
{
    if (z->pos + (size_t)n > z->size) return 0;
    z->pos += (size_t)n;
    return 1;
}
static int set_file_offset(vorb *f, uint32_t loc)
{
    if (f->seek_clears_eof) f->eof = 0;
    if (loc <= f->size) { f->pos = loc; return 1; }      /* the reference: a successful seek leaves the eof flag alone */
    f->eof = 1;
    f->pos = f->size;
    return 0;
}

/* ---- pages / packets, vb:984-1152 ---- */
static int capture_pattern(vorb *f)
{
    if (0x4f != get8(f)) return 0;
    if (0x67 != get8(f)) return 0;
    if (0x67 != get8(f)) return 0;
    if (0x53 != get8(f)) return 0;
    return 1;
}
static int start_page_no_capturepattern(vorb *f)
{
    uint32_t loc0, loc1, n;
    if (0 != get8(f)) return verror(f, 1);
    f->page_flag = get8(f);
    loc0 = get32(f);
    loc1 = get32(f);
    get32(f);
    n = get32(f);
    f->last_page = (int)n;
    get32(f);
    f->segment_count = get8(f);
    if (!getn(f, f->segments, f->segment_count)) return verror(f, 2);
    f->end_seg_with_known_loc = -2;
    if (loc0 != ~0U || loc1 != ~0U) {
        int i;
        for (i = f->segment_count - 1; i >= 0; --i)
            if (f->segments[i] < 255) break;
        if (i >= 0) {
            f->end_seg_with_known_loc = i;
            f->known_loc_for_packet = loc0;
        }
    }
    f->next_seg = 0;
    return 1;
}
static int start_page(vorb *f)
{
    if (!capture_pattern(f)) return verror(f, 3);
    return start_page_no_capturepattern(f);
}
static int start_packet(vorb *f)
{
    while (f->next_seg == -1) {
        if (!start_page(f)) return 0;
        if (f->page_flag & PAGEFLAG_continued_packet) return verror(f, 4);
    }
    f->last_seg = 0;
    f->valid_bits = 0;
    f->packet_bytes = 0;
    f->bytes_in_seg = 0;
    return 1;
}
static int maybe_start_packet(vorb *f)
{
    if (f->next_seg == -1) {
        int x = get8(f);
        if (f->eof) return 0;
        if (0x4f != x) return verror(f, 3);
        if (0x67 != get8(f)) return verror(f, 3);
        if (0x67 != get8(f)) return verror(f, 3);
        if (0x53 != get8(f)) return verror(f, 3);
        if (!start_page_no_capturepattern(f)) return 0;
        if (f->page_flag & PAGEFLAG_continued_packet) {
            f->last_seg = 0;
            f->bytes_in_seg = 0;
            return verror(f, 4);
        }
    }
    return start_packet(f);
}
static int next_segment(vorb *f)
{
    int len;
    if (f->last_seg) return 0;
    if (f->next_seg == -1) {
        f->last_seg_which = f->segment_count - 1;
        if (!start_page(f)) { f->last_seg = 1; return 0; }
        if (!(f->page_flag & PAGEFLAG_continued_packet)) return verror(f, 4);
    }
    len = f->segments[f->next_seg++];
    if (len < 255) {
        f->last_seg = 1;
        f->last_seg_which = f->next_seg - 1;
    }
    if (f->next_seg >= f->segment_count) f->next_seg = -1;
    f->bytes_in_seg = (uint8_t)len;
    return len;
}
static int get8_packet_raw(vorb *f)
{
    if (!f->bytes_in_seg) {
        if (f->last_seg) return EOP;
        else if (!next_segment(f)) return EOP;
    }
    --f->bytes_in_seg;
    ++f->packet_bytes;
    return get8(f);
}
static int get8_packet(vorb *f)
{
    int x = get8_packet_raw(f);
    f->valid_bits = 0;
    return x;
}
static int get32_packet(vorb *f)
{
    uint32_t x = (uint32_t)get8_packet(f);
    x += (uint32_t)get8_packet(f) << 8;
    x += (uint32_t)get8_packet(f) << 16;
    x += (uint32_t)get8_packet(f) << 24;
    return (int)x;
}
static void flush_packet(vorb *f) { while (get8_packet_raw(f) != EOP) {} }

static uint32_t get_bits(vorb *f, int n)                 /* vb:1154-1184 */
{
    uint32_t z;
    if (f->valid_bits < 0) return 0;
    if (f->valid_bits < n) {
        if (n > 24) {
            z = get_bits(f, 24);
            z += get_bits(f, n - 24) << 24;
            return z;
        }
        if (f->valid_bits == 0) f->acc = 0;
        while (f->valid_bits < n) {
            int z2 = get8_packet_raw(f);
            if (z2 == EOP) {
                f->valid_bits = INVALID_BITS;
                return 0;
            }
            f->acc += (uint32_t)z2 << f->valid_bits;
            f->valid_bits += 8;
        }
    }
    z = f->acc & ((1U << n) - 1);
    f->acc >>= n;
    f->valid_bits -= n;
    return z;
}
static void prep_huffman(vorb *f)                        /* vb:1189-1203 */
{
    if (f->valid_bits <= 24) {
        if (f->valid_bits == 0) f->acc = 0;
        do {
            int z;
            if (f->last_seg && !f->bytes_in_seg) return;
            z = get8_packet_raw(f);
            if (z == EOP) return;
            f->acc += (uint32_t)z << f->valid_bits;
            f->valid_bits += 8;
        } while (f->valid_bits <= 24);
    }
}

/* ---- codebook decode, vb:1211-1444 ---- */
static int codebook_decode_scalar_raw(vorb *f, Codebook *c)
{
    prep_huffman(f);
    if (c->codewords == NULL && c->sorted_codewords == NULL) return -1;
    if (c->entries > 8 ? c->sorted_codewords != NULL : !c->codewords) {
        uint32_t code = bit_reverse(f->acc);
        int x = 0, n = c->sorted_entries, len;
        while (n > 1) {
            int m = x + (n >> 1);
            if (c->sorted_codewords[m] <= code) { x = m; n -= (n >> 1); }
            else n >>= 1;
        }
        if (!c->sparse) x = c->sorted_values[x];
        len = c->codeword_lengths[x];
        if (f->valid_bits >= len) {
            f->acc >>= len;
            f->valid_bits -= len;
            return x;
        }
        f->valid_bits = 0;
        return -1;
    }
    for (int i = 0; i < c->entries; ++i) {
        if (c->codeword_lengths[i] == NO_CODE) continue;
        if (c->codewords[i] == (f->acc & ((1U << c->codeword_lengths[i]) - 1))) {
            if (f->valid_bits >= c->codeword_lengths[i]) {
                f->acc >>= c->codeword_lengths[i];
                f->valid_bits -= c->codeword_lengths[i];
                return i;
            }
            f->valid_bits = 0;
            return -1;
        }
    }
    verror(f, 5);
    f->valid_bits = 0;
    return -1;
}
static int codebook_decode_scalar(vorb *f, Codebook *c)
{
    int i;
    if (f->valid_bits < FAST_LEN) prep_huffman(f);
    i = (int)(f->acc & (FAST_SIZE - 1));
    i = c->fast_huffman[i];
    if (i >= 0) {
        f->acc >>= c->codeword_lengths[i];
        f->valid_bits -= c->codeword_lengths[i];
        if (f->valid_bits < 0) { f->valid_bits = 0; return -1; }
        return i;
    }
    return codebook_decode_scalar_raw(f, c);
}
static int DECODE(vorb *f, Codebook *c)                 /* vb:1293-1298 */
{
    int var = codebook_decode_scalar(f, c);
    if (c->sparse && c->sorted_values) var = c->sorted_values[var];
    return var;
}
static int codebook_decode_start(vorb *f, Codebook *c)
{
    int z = -1;
    if (c->lookup_type == 0) verror(f, 5);
    else {
        z = codebook_decode_scalar(f, c);
        if (z < 0) {
            if (!f->bytes_in_seg)
                if (f->last_seg) return z;
            verror(f, 5);
        }
    }
    return z;
}
static int codebook_decode(vorb *f, Codebook *c, float *output, int len)
{
    int i, z = codebook_decode_start(f, c);
    if (z < 0) return 0;
    if (len > c->dimensions) len = c->dimensions;
    z *= c->dimensions;
    if (c->sequence_p) {
        float last = 0;
        for (i = 0; i < len; ++i) {
            float val = c->multiplicands[z + i] + last;
            output[i] += val;
            last = val + c->minimum_value;
        }
    } else {
        float last = 0;
        for (i = 0; i < len; ++i) output[i] += c->multiplicands[z + i] + last;
    }
    return 1;
}
static int codebook_decode_step(vorb *f, Codebook *c, float *output, int len, int step)
{
    int i, z = codebook_decode_start(f, c);
    float last = 0;
    if (z < 0) return 0;
    if (len > c->dimensions) len = c->dimensions;
    z *= c->dimensions;
    for (i = 0; i < len; ++i) {
        float val = c->multiplicands[z + i] + last;
        output[i * step] += val;
        if (c->sequence_p) last = val;
    }
    return 1;
}
static int codebook_decode_deinterleave_repeat(vorb *f, Codebook *c, float **outputs, int ch, int *c_inter_p, int *p_inter_p,
                                               int len, int total_decode)
{
    int c_inter = *c_inter_p, p_inter = *p_inter_p, i, z, effective = c->dimensions;
    if (c->lookup_type == 0) return verror(f, 5);
    while (total_decode > 0) {
        float last = 0;
        z = codebook_decode_scalar(f, c);
        if (z < 0) {
            if (!f->bytes_in_seg)
                if (f->last_seg) return 0;
            return verror(f, 5);
        }
        if (c_inter + p_inter * ch + effective > len * ch) effective = len * ch - (p_inter * ch - c_inter);
        z *= c->dimensions;
        if (c->sequence_p) {
            for (i = 0; i < effective; ++i) {
                float val = c->multiplicands[z + i] + last;
                if (outputs[c_inter]) outputs[c_inter][p_inter] += val;
                if (++c_inter == ch) { c_inter = 0; ++p_inter; }
                last = val;
            }
        } else {
            for (i = 0; i < effective; ++i) {
                float val = c->multiplicands[z + i] + last;
                if (outputs[c_inter]) outputs[c_inter][p_inter] += val;
                if (++c_inter == ch) { c_inter = 0; ++p_inter; }
            }
        }
        total_decode -= effective;
    }
    *c_inter_p = c_inter;
    *p_inter_p = p_inter;
    return 1;
}

/* ---- floor 1 + residue, vb:1446-1713, :2255-2284 ---- */
static int predict_point(int x, int x0, int x1, int y0, int y1)
{
    int dy = y1 - y0, adx = x1 - x0;
    int err = abs(dy) * (x - x0);
    int off = err / adx;
    return dy < 0 ? y0 - off : y0 + off;
}
static void draw_line(float *output, int x0, int y0, int x1, int y1, int n)
{
    int dy = y1 - y0, adx = x1 - x0, ady = abs(dy), base, x = x0, y = y0, err = 0, sy;
    base = dy / adx;
    if (dy < 0) sy = base - 1;
    else sy = base + 1;
    ady -= abs(base) * adx;
    if (x1 > n) x1 = n;
    if (x < x1) {
        output[x] *= f32(k_inverse_db_bits[y & 255]);
        for (++x; x < x1; ++x) {
            err += ady;
            if (err >= adx) { err -= adx; y += sy; }
            else y += base;
            output[x] *= f32(k_inverse_db_bits[y & 255]);
        }
    }
}
static int residue_decode(vorb *f, Codebook *book, float *target, int offset, int n, int rtype)
{
    int k;
    if (rtype == 0) {
        int step = n / book->dimensions;
        for (k = 0; k < step; ++k)
            if (!codebook_decode_step(f, book, target + offset + k, n - offset - k, step)) return 0;
    } else {
        for (k = 0; k < n;) {
            if (!codebook_decode(f, book, target + offset, n - k)) return 0;
            k += book->dimensions;
            offset += book->dimensions;
        }
    }
    return 1;
}
static void decode_residue(vorb *f, float **residue_buffers, int ch, int n, int rn, uint8_t *do_not_decode)
{
    int i, j, pass;
    Residue *r = f->residue_config + rn;
    int rtype = f->residue_types[rn];
    int classwords = f->codebooks[r->classbook].dimensions;
    unsigned actual_size = rtype == 2 ? (unsigned)n * 2 : (unsigned)n;
    unsigned limit_r_begin = r->begin < actual_size ? r->begin : actual_size;
    unsigned limit_r_end = r->end < actual_size ? r->end : actual_size;
    int n_read = (int)(limit_r_end - limit_r_begin);
    int part_read = n_read / (int)r->part_size;
    uint8_t ***part_classdata = (uint8_t ***)calloc((size_t)f->channels, sizeof(uint8_t **));
    for (i = 0; i < f->channels; ++i) part_classdata[i] = (uint8_t **)calloc((size_t)part_read + 1, sizeof(uint8_t *));

    for (i = 0; i < ch; ++i)
        if (!do_not_decode[i]) memset(residue_buffers[i], 0, sizeof(float) * (size_t)n);

    if (rtype == 2 && ch != 1) {
        for (j = 0; j < ch; ++j)
            if (!do_not_decode[j]) break;
        if (j == ch) goto done;
        for (pass = 0; pass < 8; ++pass) {
            int pcount = 0, class_set = 0;
            while (pcount < part_read) {
                int z = (int)r->begin + pcount * (int)r->part_size;
                int c_inter = ch == 2 ? (z & 1) : z % ch, p_inter = ch == 2 ? z >> 1 : z / ch;
                if (pass == 0) {
                    int q = DECODE(f, f->codebooks + r->classbook);
                    if (q == EOP) goto done;
                    part_classdata[0][class_set] = r->classdata[q];
                }
                for (i = 0; i < classwords && pcount < part_read; ++i, ++pcount) {
                    int z2 = (int)r->begin + pcount * (int)r->part_size;
                    int c2 = part_classdata[0][class_set][i];
                    int b = r->residue_books[c2][pass];
                    if (b >= 0) {
                        if (!codebook_decode_deinterleave_repeat(f, f->codebooks + b, residue_buffers, ch, &c_inter, &p_inter, n, (int)r->part_size))
                            goto done;
                    } else {
                        z2 += (int)r->part_size;
                        c_inter = ch == 2 ? (z2 & 1) : z2 % ch;
                        p_inter = ch == 2 ? z2 >> 1 : z2 / ch;
                    }
                }
                ++class_set;
            }
        }
        goto done;
    }
    for (pass = 0; pass < 8; ++pass) {
        int pcount = 0, class_set = 0;
        while (pcount < part_read) {
            if (pass == 0) {
                for (j = 0; j < ch; ++j) {
                    if (!do_not_decode[j]) {
                        int temp = DECODE(f, f->codebooks + r->classbook);
                        if (temp == EOP) goto done;
                        part_classdata[j][class_set] = r->classdata[temp];
                    }
                }
            }
            for (i = 0; i < classwords && pcount < part_read; ++i, ++pcount) {
                for (j = 0; j < ch; ++j) {
                    if (!do_not_decode[j]) {
                        int c5 = part_classdata[j][class_set][i];
                        int b = r->residue_books[c5][pass];
                        if (b >= 0) {
                            if (!residue_decode(f, f->codebooks + b, residue_buffers[j], (int)r->begin + pcount * (int)r->part_size,
                                                (int)r->part_size, rtype))
                                goto done;
                        }
                    }
                }
            }
            ++class_set;
        }
    }
done:
    for (i = 0; i < f->channels; ++i) free(part_classdata[i]);
    free(part_classdata);
}
static int do_floor(vorb *f, Mapping *map, int i, int n, float *target, int16_t *finalY)
{
    int n2 = n >> 1;
    int s = map->chan[i].mux, fl = map->submap_floor[s];
    Floor1 *g = &f->floor_config[fl];
    int j, q, lx = 0, ly = finalY[0] * g->floor1_multiplier;
    for (q = 1; q < g->values; ++q) {
        j = g->sorted_order[q];
        if (finalY[j] >= 0) {
            int hy = finalY[j] * g->floor1_multiplier, hx = g->Xlist[j];
            if (lx != hx) draw_line(target, lx, ly, hx, hy, n2);
            lx = hx;
            ly = hy;
        }
    }
    if (lx < n2)
        for (j = lx; j < n2; ++j) target[j] *= f32(k_inverse_db_bits[ly]);
    return 1;
}

/* ---- setup, vb:2669-3266 ---- */
static int start_decoder(vorb *f)
{
    uint8_t header[6], x, y;
    int len, i, j, k, longest_floorlist = 0;
    f->first_decode = 1;
    if (!start_page(f)) return 0;
    if (!(f->page_flag & PAGEFLAG_first_page)) return verror(f, 6);
    if (f->page_flag & PAGEFLAG_last_page) return verror(f, 6);
    if (f->page_flag & PAGEFLAG_continued_packet) return verror(f, 6);
    if (f->segment_count != 1) return verror(f, 6);
    if (f->segments[0] != 30) return verror(f, 6);
    if (get8(f) != 1) return verror(f, 6);
    if (!getn(f, header, 6)) return verror(f, 2);
    if (memcmp(header, "vorbis", 6)) return verror(f, 6);
    if (get32(f) != 0) return verror(f, 6);
    f->channels = get8(f);
    if (!f->channels) return verror(f, 6);
    if (f->channels > MAX_CH) return verror(f, 7);
    f->sample_rate = get32(f);
    if (!f->sample_rate) return verror(f, 6);
    get32(f); get32(f); get32(f);
    x = get8(f);
    {
        int log0 = x & 15, log1 = x >> 4;
        f->blocksize_0 = 1 << log0;
        f->blocksize_1 = 1 << log1;
        if (log0 < 6 || log0 > 13) return verror(f, 8);
        if (log1 < 6 || log1 > 13) return verror(f, 8);
        if (log0 > log1) return verror(f, 8);
    }
    x = get8(f);
    if (!(x & 1)) return verror(f, 6);

    if (!start_page(f)) return 0;
    if (!start_packet(f)) return 0;
    if (!next_segment(f)) return 0;
    if (get8_packet(f) != 3) return verror(f, 8);
    for (i = 0; i < 6; ++i) header[i] = (uint8_t)get8_packet(f);
    if (memcmp(header, "vorbis", 6)) return verror(f, 8);
    /* (the vendor string and the comments are copied into setup_malloc'ed memory, vb:2741-2765; setup_malloc (vb:556-567)
       rounds its int size up to a multiple of four and answers NULL for one that is not positive -- the open fails with
       outofmem.  Positive sizes are taken to succeed.) */
#define AFGO_ALLOCATES(sz) ((int32_t)(((uint32_t)(sz) + 3u) & ~3u) > 0)
    len = get32_packet(f);
    if (!AFGO_ALLOCATES((uint32_t)len + 1u)) return verror(f, 3);
    for (i = 0; i < len; ++i) get8_packet(f);
    {
        int n_comments = get32_packet(f);
        if (n_comments > 0 && !AFGO_ALLOCATES(8u * (uint32_t)n_comments)) return verror(f, 3);
        for (i = 0; i < n_comments; ++i) {
            len = get32_packet(f);
            if (!AFGO_ALLOCATES((uint32_t)len + 1u)) return verror(f, 3);
            for (j = 0; j < len; ++j) get8_packet(f);
        }
    }
#undef AFGO_ALLOCATES
    x = (uint8_t)get8_packet(f);
    if (!(x & 1)) return verror(f, 8);
    if (!skip_bytes(f, f->bytes_in_seg)) return verror(f, 8);
    f->bytes_in_seg = 0;
    do {
        len = next_segment(f);
        if (!skip_bytes(f, len)) return verror(f, 8);
        f->bytes_in_seg = 0;
    } while (len);

    if (!start_packet(f)) return 0;
    crc32_init(f);
    if (get8_packet(f) != 5) return verror(f, 8);
    for (i = 0; i < 6; ++i) header[i] = (uint8_t)get8_packet(f);
    if (memcmp(header, "vorbis", 6)) return verror(f, 8);

    f->codebook_count = (int)get_bits(f, 8) + 1;
    f->codebooks = (Codebook *)calloc((size_t)f->codebook_count, sizeof(Codebook));
    for (i = 0; i < f->codebook_count; ++i) {
        uint32_t *values = NULL;
        int ordered, sorted_count, total = 0;
        uint8_t *lengths;
        Codebook *c = f->codebooks + i;
        x = (uint8_t)get_bits(f, 8); if (x != 0x42) return verror(f, 8);
        x = (uint8_t)get_bits(f, 8); if (x != 0x43) return verror(f, 8);
        x = (uint8_t)get_bits(f, 8); if (x != 0x56) return verror(f, 8);
        x = (uint8_t)get_bits(f, 8);
        c->dimensions = (int)(get_bits(f, 8) << 8) + x;
        x = (uint8_t)get_bits(f, 8);
        y = (uint8_t)get_bits(f, 8);
        c->entries = (int)(get_bits(f, 8) << 16) + (y << 8) + x;
        ordered = (int)get_bits(f, 1);
        c->sparse = ordered ? 0 : (uint8_t)get_bits(f, 1);
        if (c->dimensions == 0 && c->entries != 0) return verror(f, 8);
        lengths = (uint8_t *)calloc((size_t)c->entries + 1, 1);
        if (!c->sparse) c->codeword_lengths = lengths;
        if (ordered) {
            int current_entry = 0, current_length = (int)get_bits(f, 5) + 1;
            while (current_entry < c->entries) {
                int limit = c->entries - current_entry;
                int n = (int)get_bits(f, ilog(limit));
                if (current_length >= 32) return verror(f, 8);
                if (current_entry + n > c->entries) return verror(f, 8);
                memset(lengths + current_entry, current_length, (size_t)n);
                current_entry += n;
                ++current_length;
            }
        } else {
            for (j = 0; j < c->entries; ++j) {
                int present = c->sparse ? (int)get_bits(f, 1) : 1;
                if (present) {
                    lengths[j] = (uint8_t)(get_bits(f, 5) + 1);
                    ++total;
                    if (lengths[j] == 32) return verror(f, 8);
                } else {
                    lengths[j] = NO_CODE;
                }
            }
        }
        if (c->sparse && total >= c->entries >> 2) {
            c->codeword_lengths = lengths;
            c->sparse = 0;
        }
        if (c->sparse) {
            sorted_count = total;
        } else {
            sorted_count = 0;
            for (j = 0; j < c->entries; ++j)
                if (lengths[j] > FAST_LEN && lengths[j] != NO_CODE) ++sorted_count;
        }
        c->sorted_entries = sorted_count;
        if (!c->sparse) {
            c->codewords = (uint32_t *)calloc((size_t)c->entries + 1, sizeof(uint32_t));
        } else if (c->sorted_entries) {
            c->codeword_lengths = (uint8_t *)calloc((size_t)c->sorted_entries, 1);
            c->codewords = (uint32_t *)calloc((size_t)c->sorted_entries, sizeof(uint32_t));
            values = (uint32_t *)calloc((size_t)c->sorted_entries, sizeof(uint32_t));
        }
        if (!compute_codewords(c, lengths, c->entries, values)) return verror(f, 8);
        if (c->sorted_entries) {
            c->sorted_codewords = (uint32_t *)calloc((size_t)c->sorted_entries + 1, sizeof(uint32_t));
            c->sorted_values_base = (int *)calloc((size_t)c->sorted_entries + 1, sizeof(int));
            c->sorted_values = c->sorted_values_base + 1;
            c->sorted_values[-1] = -1;
            compute_sorted_huffman(c, lengths, values);
        }
        if (c->sparse) {
            free(values);
            free(c->codewords);
            free(lengths);
            c->codewords = NULL;
        }
        compute_accelerated_huffman(c);
        c->lookup_type = (uint8_t)get_bits(f, 4);
        if (c->lookup_type > 2) return verror(f, 8);
        if (c->lookup_type > 0) {
            uint16_t *mults;
            c->minimum_value = float32_unpack(get_bits(f, 32));
            c->delta_value = float32_unpack(get_bits(f, 32));
            c->value_bits = (uint8_t)(get_bits(f, 4) + 1);
            c->sequence_p = (uint8_t)get_bits(f, 1);
            if (c->lookup_type == 1) {
                int values2 = lookup1_values(c->entries, c->dimensions);
                if (values2 < 0) return verror(f, 8);
                c->lookup_values = (uint32_t)values2;
            } else {
                c->lookup_values = (uint32_t)c->entries * (uint32_t)c->dimensions;
            }
            if (c->lookup_values == 0) return verror(f, 8);
            mults = (uint16_t *)calloc(c->lookup_values, sizeof(uint16_t));
            for (j = 0; j < (int)c->lookup_values; ++j) {
                int q = (int)get_bits(f, c->value_bits);
                if (q == EOP) { free(mults); return verror(f, 8); }
                mults[j] = (uint16_t)q;
            }
            if (c->lookup_type == 1) {
                int len2, sparse = c->sparse;
                float last = 0;
                if (sparse && c->sorted_entries == 0) { free(mults); continue; }
                len2 = sparse ? c->sorted_entries : c->entries;
                c->multiplicands = (float *)calloc((size_t)len2 * (size_t)c->dimensions + 1, sizeof(float));
                for (j = 0; j < len2; ++j) {
                    unsigned z = sparse ? (unsigned)c->sorted_values[j] : (unsigned)j;
                    unsigned div = 1;
                    for (k = 0; k < c->dimensions; ++k) {
                        int off = (int)((z / div) % c->lookup_values);
                        float val = mults[off] * c->delta_value + c->minimum_value + last;
                        c->multiplicands[j * c->dimensions + k] = val;
                        if (c->sequence_p) last = val;
                        if (k + 1 < c->dimensions) {
                            if (div > 0xffffffffu / c->lookup_values) { free(mults); return verror(f, 8); }
                            div *= c->lookup_values;
                        }
                    }
                }
                c->lookup_type = 2;
            } else {
                float last = 0;
                c->multiplicands = (float *)calloc(c->lookup_values + 1, sizeof(float));
                for (j = 0; j < (int)c->lookup_values; ++j) {
                    float val = mults[j] * c->delta_value + c->minimum_value + last;
                    c->multiplicands[j] = val;
                    if (c->sequence_p) last = val;
                }
            }
            free(mults);
        }
    }
    x = (uint8_t)(get_bits(f, 6) + 1);
    for (i = 0; i < x; ++i)
        if (get_bits(f, 16) != 0) return verror(f, 8);

    f->floor_count = (int)get_bits(f, 6) + 1;
    f->floor_config = (Floor1 *)calloc((size_t)f->floor_count, sizeof(Floor1));
    for (i = 0; i < f->floor_count; ++i) {
        f->floor_types[i] = (uint16_t)get_bits(f, 16);
        if (f->floor_types[i] > 1) return verror(f, 8);
        if (f->floor_types[i] == 0) return verror(f, 9);             /* floor 0: not supported by the reference */
        floor_ordering p[31 * 8 + 2];
        Floor1 *g = &f->floor_config[i];
        int max_class = -1;
        g->partitions = (uint8_t)get_bits(f, 5);
        for (j = 0; j < g->partitions; ++j) {
            g->partition_class_list[j] = (uint8_t)get_bits(f, 4);
            if (g->partition_class_list[j] > max_class) max_class = g->partition_class_list[j];
        }
        for (j = 0; j <= max_class; ++j) {
            g->class_dimensions[j] = (uint8_t)(get_bits(f, 3) + 1);
            g->class_subclasses[j] = (uint8_t)get_bits(f, 2);
            if (g->class_subclasses[j]) {
                g->class_masterbooks[j] = (uint8_t)get_bits(f, 8);
                if (g->class_masterbooks[j] >= f->codebook_count) return verror(f, 8);
            }
            for (k = 0; k < 1 << g->class_subclasses[j]; ++k) {
                g->subclass_books[j][k] = (int16_t)((int)get_bits(f, 8) - 1);
                if (g->subclass_books[j][k] >= f->codebook_count) return verror(f, 8);
            }
        }
        g->floor1_multiplier = (uint8_t)(get_bits(f, 2) + 1);
        g->rangebits = (uint8_t)get_bits(f, 4);
        g->Xlist[0] = 0;
        g->Xlist[1] = (uint16_t)(1 << g->rangebits);
        g->values = 2;
        for (j = 0; j < g->partitions; ++j) {
            int c = g->partition_class_list[j];
            for (k = 0; k < g->class_dimensions[c]; ++k) {
                g->Xlist[g->values] = (uint16_t)get_bits(f, g->rangebits);
                ++g->values;
            }
        }
        for (j = 0; j < g->values; ++j) { p[j].x = g->Xlist[j]; p[j].id = (uint16_t)j; }
        qsort(p, (size_t)g->values, sizeof(p[0]), point_compare);
        for (j = 0; j < g->values - 1; ++j)
            if (p[j].x == p[j + 1].x) return verror(f, 8);
        for (j = 0; j < g->values; ++j) g->sorted_order[j] = (uint8_t)p[j].id;
        for (j = 2; j < g->values; ++j) {
            int low = 0, hi = 0;
            neighbors(g->Xlist, j, &low, &hi);
            g->neighbors[j][0] = (uint8_t)low;
            g->neighbors[j][1] = (uint8_t)hi;
        }
        if (g->values > longest_floorlist) longest_floorlist = g->values;
    }

    f->residue_count = (int)get_bits(f, 6) + 1;
    f->residue_config = (Residue *)calloc((size_t)f->residue_count, sizeof(Residue));
    for (i = 0; i < f->residue_count; ++i) {
        uint8_t residue_cascade[64];
        Residue *r = f->residue_config + i;
        f->residue_types[i] = (uint16_t)get_bits(f, 16);
        if (f->residue_types[i] > 2) return verror(f, 8);
        r->begin = get_bits(f, 24);
        r->end = get_bits(f, 24);
        if (r->end < r->begin) return verror(f, 8);
        r->part_size = get_bits(f, 24) + 1;
        r->classifications = (uint8_t)(get_bits(f, 6) + 1);
        r->classbook = (uint8_t)get_bits(f, 8);
        if (r->classbook >= f->codebook_count) return verror(f, 8);
        for (j = 0; j < r->classifications; ++j) {
            uint8_t high_bits = 0, low_bits = (uint8_t)get_bits(f, 3);
            if (get_bits(f, 1)) high_bits = (uint8_t)get_bits(f, 5);
            residue_cascade[j] = (uint8_t)(high_bits * 8 + low_bits);
        }
        r->residue_books = (int16_t (*)[8])calloc(r->classifications, sizeof(int16_t[8]));
        for (j = 0; j < r->classifications; ++j) {
            for (k = 0; k < 8; ++k) {
                if (residue_cascade[j] & (1 << k)) {
                    r->residue_books[j][k] = (int16_t)get_bits(f, 8);
                    if (r->residue_books[j][k] >= f->codebook_count) return verror(f, 8);
                } else {
                    r->residue_books[j][k] = -1;
                }
            }
        }
        {
            int entries = f->codebooks[r->classbook].entries, classwords = f->codebooks[r->classbook].dimensions;
            r->classdata = (uint8_t **)calloc((size_t)entries + 1, sizeof(uint8_t *));
            for (j = 0; j < entries; ++j) {
                int temp = j;
                r->classdata[j] = (uint8_t *)calloc((size_t)classwords + 1, 1);
                for (k = classwords - 1; k >= 0; --k) {
                    r->classdata[j][k] = (uint8_t)(temp % r->classifications);
                    temp /= r->classifications;
                }
            }
        }
    }

    f->mapping_count = (int)get_bits(f, 6) + 1;
    f->mapping = (Mapping *)calloc((size_t)f->mapping_count, sizeof(Mapping));
    for (i = 0; i < f->mapping_count; ++i) {
        Mapping *m = f->mapping + i;
        int mapping_type = (int)get_bits(f, 16);
        if (mapping_type != 0) return verror(f, 8);
        m->chan = (MappingChannel *)calloc((size_t)f->channels, sizeof(MappingChannel));
        if (get_bits(f, 1)) m->submaps = (uint8_t)(get_bits(f, 4) + 1);
        else m->submaps = 1;
        if (get_bits(f, 1)) {
            m->coupling_steps = (uint16_t)(get_bits(f, 8) + 1);
            if (m->coupling_steps > f->channels) return verror(f, 8);
            for (k = 0; k < m->coupling_steps; ++k) {
                m->chan[k].magnitude = (uint8_t)get_bits(f, ilog(f->channels - 1));
                m->chan[k].angle = (uint8_t)get_bits(f, ilog(f->channels - 1));
                if (m->chan[k].magnitude >= f->channels) return verror(f, 8);
                if (m->chan[k].angle >= f->channels) return verror(f, 8);
                if (m->chan[k].magnitude == m->chan[k].angle) return verror(f, 8);
            }
        } else {
            m->coupling_steps = 0;
        }
        if (get_bits(f, 2)) return verror(f, 8);
        if (m->submaps > 1) {
            for (j = 0; j < f->channels; ++j) {
                m->chan[j].mux = (uint8_t)get_bits(f, 4);
                if (m->chan[j].mux >= m->submaps) return verror(f, 8);
            }
        } else {
            for (j = 0; j < f->channels; ++j) m->chan[j].mux = 0;
        }
        for (j = 0; j < m->submaps; ++j) {
            get_bits(f, 8);
            m->submap_floor[j] = (uint8_t)get_bits(f, 8);
            m->submap_residue[j] = (uint8_t)get_bits(f, 8);
            if (m->submap_floor[j] >= f->floor_count) return verror(f, 8);
            if (m->submap_residue[j] >= f->residue_count) return verror(f, 8);
        }
    }

    f->mode_count = (int)get_bits(f, 6) + 1;
    for (i = 0; i < f->mode_count; ++i) {
        Mode *m = f->mode_config + i;
        m->blockflag = (uint8_t)get_bits(f, 1);
        m->windowtype = (uint16_t)get_bits(f, 16);
        m->transformtype = (uint16_t)get_bits(f, 16);
        m->mapping = (uint8_t)get_bits(f, 8);
        if (m->windowtype != 0) return verror(f, 8);
        if (m->transformtype != 0) return verror(f, 8);
        if (m->mapping >= f->mapping_count) return verror(f, 8);
    }
    flush_packet(f);
    f->previous_length = 0;
    for (i = 0; i < f->channels; ++i) {
        f->channel_buffers[i] = (float *)calloc((size_t)f->blocksize_1, sizeof(float));
        f->finalY[i] = (int16_t *)calloc((size_t)longest_floorlist + 1, sizeof(int16_t));
    }
    f->blocksize[0] = f->blocksize_0;
    f->blocksize[1] = f->blocksize_1;
    if (f->next_seg == -1) f->first_audio_page_offset = (uint32_t)f->pos;
    else f->first_audio_page_offset = 0;
    return 1;
}

/* ---- audio packets, vb:2300-2597 ---- */
static int decode_initial(vorb *f, int *p_left_start, int *p_left_end, int *p_right_start, int *p_right_end, int *mode, unsigned *flags)
{
    Mode *m;
    int i, n, prev, next, window_center;
retry:
    if (f->eof) return 0;
    if (!maybe_start_packet(f)) return 0;
    if (get_bits(f, 1) != 0) {
        while (EOP != get8_packet(f)) {}
        goto retry;
    }
    i = (int)get_bits(f, ilog(f->mode_count - 1));
    if (i == EOP) return 0;
    if (i >= f->mode_count) return 0;
    *mode = i;
    m = f->mode_config + i;
    if (m->blockflag) {
        n = f->blocksize_1;
        prev = (int)get_bits(f, 1);
        next = (int)get_bits(f, 1);
    } else {
        prev = next = 0;
        n = f->blocksize_0;
    }
    *flags = (m->blockflag ? 1u : 0u) | (prev ? 2u : 0u) | (next ? 4u : 0u);
    window_center = n >> 1;
    if (m->blockflag && !prev) {
        *p_left_start = (n - f->blocksize_0) >> 2;
        *p_left_end = (n + f->blocksize_0) >> 2;
    } else {
        *p_left_start = 0;
        *p_left_end = window_center;
    }
    if (m->blockflag && !next) {
        *p_right_start = (n * 3 - f->blocksize_0) >> 2;
        *p_right_end = (n * 3 + f->blocksize_0) >> 2;
    } else {
        *p_right_start = window_center;
        *p_right_end = n;
    }
    return 1;
}

/* decode up to the seam: channel_buffers hold the spectra; *len / *p_left as the reference leaves them */
static int decode_packet_rest(vorb *f, int *len, Mode *m, int left_start, int left_end, int right_start, int right_end, int *p_left)
{
    Mapping *map;
    int i, j, k, n, n2;
    int zero_channel[256], really_zero_channel[256];
    (void)left_end;
    n = f->blocksize[m->blockflag];
    map = &f->mapping[m->mapping];
    n2 = n >> 1;
    for (i = 0; i < f->channels; ++i) {
        int s = map->chan[i].mux, fl = map->submap_floor[s];
        zero_channel[i] = 0;
        Floor1 *g = &f->floor_config[fl];
        if (get_bits(f, 1)) {
            int16_t *finalY;
            uint8_t step2_flag[256];
            static const int range_list[4] = { 256, 128, 86, 64 };
            int range = range_list[g->floor1_multiplier - 1];
            int offset = 2;
            finalY = f->finalY[i];
            finalY[0] = (int16_t)get_bits(f, ilog(range) - 1);
            finalY[1] = (int16_t)get_bits(f, ilog(range) - 1);
            for (j = 0; j < g->partitions; ++j) {
                int pclass = g->partition_class_list[j];
                int cdim = g->class_dimensions[pclass];
                int cbits = g->class_subclasses[pclass];
                int csub = (1 << cbits) - 1;
                int cval = 0;
                if (cbits) cval = DECODE(f, f->codebooks + g->class_masterbooks[pclass]);
                for (k = 0; k < cdim; ++k) {
                    int book = g->subclass_books[pclass][cval & csub];
                    cval = cval >> cbits;
                    if (book >= 0) finalY[offset++] = (int16_t)DECODE(f, f->codebooks + book);
                    else finalY[offset++] = 0;
                }
            }
            if (f->valid_bits == INVALID_BITS) { zero_channel[i] = 1; continue; }
            step2_flag[0] = step2_flag[1] = 1;
            for (j = 2; j < g->values; ++j) {
                int low = g->neighbors[j][0], high = g->neighbors[j][1];
                int pred = predict_point(g->Xlist[j], g->Xlist[low], g->Xlist[high], finalY[low], finalY[high]);
                int val = finalY[j], highroom = range - pred, lowroom = pred, room;
                if (highroom < lowroom) room = highroom * 2;
                else room = lowroom * 2;
                if (val) {
                    step2_flag[low] = step2_flag[high] = 1;
                    step2_flag[j] = 1;
                    if (val >= room)
                        if (highroom > lowroom) finalY[j] = (int16_t)(val - lowroom + pred);
                        else finalY[j] = (int16_t)(pred - val + highroom - 1);
                    else if (val & 1) finalY[j] = (int16_t)(pred - ((val + 1) >> 1));
                    else finalY[j] = (int16_t)(pred + (val >> 1));
                } else {
                    step2_flag[j] = 0;
                    finalY[j] = (int16_t)pred;
                }
            }
            for (j = 0; j < g->values; ++j)
                if (!step2_flag[j]) finalY[j] = -1;
        } else {
            zero_channel[i] = 1;
        }
    }
    memcpy(really_zero_channel, zero_channel, sizeof(int) * (size_t)f->channels);
    for (i = 0; i < map->coupling_steps; ++i)
        if (!zero_channel[map->chan[i].magnitude] || !zero_channel[map->chan[i].angle])
            zero_channel[map->chan[i].magnitude] = zero_channel[map->chan[i].angle] = 0;

    for (i = 0; i < map->submaps; ++i) {
        float *residue_buffers[MAX_CH];
        uint8_t do_not_decode[256];
        int ch = 0;
        for (j = 0; j < f->channels; ++j) {
            if (map->chan[j].mux == i) {
                if (zero_channel[j]) { do_not_decode[ch] = 1; residue_buffers[ch] = NULL; }
                else { do_not_decode[ch] = 0; residue_buffers[ch] = f->channel_buffers[j]; }
                ++ch;
            }
        }
        decode_residue(f, residue_buffers, ch, n2, map->submap_residue[i], do_not_decode);
    }
    for (i = map->coupling_steps - 1; i >= 0; --i) {
        float *m_ = f->channel_buffers[map->chan[i].magnitude], *a = f->channel_buffers[map->chan[i].angle];
        for (j = 0; j < n2; ++j) {
            float a2, m2;
            if (m_[j] > 0)
                if (a[j] > 0) { m2 = m_[j]; a2 = m_[j] - a[j]; }
                else { a2 = m_[j]; m2 = m_[j] + a[j]; }
            else if (a[j] > 0) { m2 = m_[j]; a2 = m_[j] + a[j]; }
            else { a2 = m_[j]; m2 = m_[j] - a[j]; }
            m_[j] = m2;
            a[j] = a2;
        }
    }
    for (i = 0; i < f->channels; ++i) {
        if (really_zero_channel[i]) memset(f->channel_buffers[i], 0, sizeof(float) * (size_t)n2);
        else do_floor(f, map, i, n, f->channel_buffers[i], f->finalY[i]);
    }
    /* ---- the seam (inverse_mdct, vb:2526-2527) is here; the bookkeeping below is vb:2531-2596 ---- */
    flush_packet(f);
    if (f->first_decode) {
        f->current_loc = 0u - (uint32_t)n2;
        f->discard_samples_deferred = n - right_end;
        f->current_loc_valid = 1;
        f->first_decode = 0;
    } else if (f->discard_samples_deferred) {
        if (f->discard_samples_deferred >= right_start - left_start) {
            f->discard_samples_deferred -= (right_start - left_start);
            left_start = right_start;
            *p_left = left_start;
        } else {
            left_start += f->discard_samples_deferred;
            *p_left = left_start;
            f->discard_samples_deferred = 0;
        }
    }
    if (f->last_seg_which == f->end_seg_with_known_loc) {
        if (f->current_loc_valid && (f->page_flag & PAGEFLAG_last_page)) {
            uint32_t current_end = f->known_loc_for_packet;
            if (current_end < f->current_loc + (uint32_t)(right_end - left_start)) {
                if (current_end < f->current_loc) *len = 0;
                else *len = (int)(current_end - f->current_loc);
                *len += left_start;
                if (*len > right_end) *len = right_end;
                f->current_loc += (uint32_t)*len;
                return 1;
            }
        }
        f->current_loc = f->known_loc_for_packet - (uint32_t)(n2 - left_start);
        f->current_loc_valid = 1;
    }
    if (f->current_loc_valid) f->current_loc += (uint32_t)(right_start - left_start);
    *len = right_end;
    return 1;
}

/* vb:3397-3466 / :3797-3868 */
static uint32_t find_page(vorb *f, uint32_t *end, uint32_t *last)
{
    static const uint8_t ogg_page_header[4] = { 0x4f, 0x67, 0x67, 0x53 };
    for (;;) {
        int n;
        if (f->eof) return 0;
        n = get8(f);
        if (n == 0x4f) {
            uint32_t retry_loc = (uint32_t)f->pos;
            int i;
            if (retry_loc - 25 > f->stream_len) return 0;
            for (i = 1; i < 4; ++i)
                if (get8(f) != ogg_page_header[i]) break;
            if (f->eof) return 0;
            if (i == 4) {
                uint8_t header[27];
                uint32_t i2, crc, goal, len;
                for (i2 = 0; i2 < 4; ++i2) header[i2] = ogg_page_header[i2];
                for (; i2 < 27; ++i2) header[i2] = get8(f);
                if (f->eof) return 0;
                if (header[4] != 0) goto invalid;
                goal = header[22] + ((uint32_t)header[23] << 8) + ((uint32_t)header[24] << 16) + ((uint32_t)header[25] << 24);
                for (i2 = 22; i2 < 26; ++i2) header[i2] = 0;
                crc = 0;
                for (i2 = 0; i2 < 27; ++i2) crc = crc32_update(f, crc, header[i2]);
                len = 0;
                for (i2 = 0; i2 < header[26]; ++i2) {
                    int s = get8(f);
                    crc = crc32_update(f, crc, (uint8_t)s);
                    len += (uint32_t)s;
                }
                if (len && f->eof) return 0;
                for (i2 = 0; i2 < len; ++i2) crc = crc32_update(f, crc, get8(f));
                if (crc == goal) {
                    if (end) *end = (uint32_t)f->pos;
                    if (last) *last = (header[5] & 0x04) ? 1 : 0;
                    set_file_offset(f, retry_loc - 1);
                    return 1;
                }
            }
        invalid:
            set_file_offset(f, retry_loc);
        }
    }
}
static uint32_t stream_length_in_samples(vorb *f)
{
    uint32_t restore_offset, previous_safe, end, last_page_loc;
    if (!f->total_samples) {
        uint32_t last, lo, hi;
        uint8_t header[6];
        restore_offset = (uint32_t)f->pos;
        if (f->stream_len >= 65536 && f->stream_len - 65536 >= f->first_audio_page_offset) previous_safe = f->stream_len - 65536;
        else previous_safe = f->first_audio_page_offset;
        set_file_offset(f, previous_safe);
        if (!find_page(f, &end, &last)) {
            f->total_samples = 0xffffffffu;
            goto done;
        }
        last_page_loc = (uint32_t)f->pos;
        while (!last) {
            set_file_offset(f, end);
            if (!find_page(f, &end, &last)) break;
            last_page_loc = (uint32_t)f->pos;
        }
        set_file_offset(f, last_page_loc);
        getn(f, header, 6);
        lo = get32(f);
        hi = get32(f);
        if (lo == 0xffffffffu && hi == 0xffffffffu) {
            f->total_samples = 0xffffffffu;
            goto done;
        }
        if (hi) lo = 0xfffffffeu;
        f->total_samples = lo;
    done:
        set_file_offset(f, restore_offset);
    }
    return f->total_samples == 0xffffffffu ? 0 : f->total_samples;
}

static void free_all(vorb *f)
{
    for (int i = 0; i < f->residue_count && f->residue_config; i++) {      /* (reads the class book's entry count: first) */
        Residue *r = f->residue_config + i;
        if (r->classdata && f->codebooks)
            for (int j = 0; j < f->codebooks[r->classbook].entries; j++) free(r->classdata[j]);
        free(r->classdata);
        free(r->residue_books);
    }
    free(f->residue_config);
    for (int i = 0; i < f->codebook_count && f->codebooks; i++) {
        Codebook *c = f->codebooks + i;
        free(c->codeword_lengths); free(c->multiplicands); free(c->codewords); free(c->sorted_codewords); free(c->sorted_values_base);
    }
    free(f->codebooks);
    free(f->floor_config);
    for (int i = 0; i < f->mapping_count && f->mapping; i++) free(f->mapping[i].chan);
    free(f->mapping);
    for (int i = 0; i < MAX_CH; i++) { free(f->channel_buffers[i]); free(f->finalY[i]); }
}

static int grow(void **p, size_t *cap, size_t need, size_t elem)
{
    if (need <= *cap) return 1;
    size_t nc = *cap ? *cap * 2 : 256;
    while (nc < need) nc *= 2;
    void *q = realloc(*p, nc * elem);
    if (!q) return 0;
    *p = q;
    *cap = nc;
    return 1;
}

/* stb_vorbis_open (start_decoder + vorbis_pump_first_frame, vb:3897-3918), the stream length (vb:3797), then
 * stb_vorbis_get_frame_float until it fails (vb:3876-3895). */
int afgo_vorbis_decode_file(const uint8_t *data, size_t size, afgo_vorbis_file *out)
{
    return afgo_vorbis_decode_file_ex(data, size, out, 0);
}

/* seek_clears_eof = 1 gives upstream stb_vorbis' set_file_offset (it resets the eof flag); the D port dropped that
 * line (vb:967-980), so after the stream-length scan of a file without a last-page flag has run into the end of
 * the data, the reference's decoder believes it is at eof and delivers nothing more. */
int afgo_vorbis_decode_file_ex(const uint8_t *data, size_t size, afgo_vorbis_file *out, int seek_clears_eof)
{
    memset(out, 0, sizeof(*out));
    vorb *f = (vorb *)calloc(1, sizeof(vorb));
    if (!f) return -2;
    f->seek_clears_eof = seek_clears_eof;
    f->data = data;
    f->size = size;
    f->stream_len = (uint32_t)size;
    f->next_seg = -1;                                    /* vorbis_init, vb:3330-3341 */
    if (!start_decoder(f)) { free_all(f); free(f); return -1; }
    out->channels = f->channels;
    out->sample_rate = f->sample_rate;
    out->blocksize0 = f->blocksize_0;
    out->blocksize1 = f->blocksize_1;
    size_t cap_p = 0, cap_s = 0, cap_l = 0, cap_n = 0;
    int first = 1;
    for (;;) {
        int mode, left, left_end, right, right_end, len;
        unsigned flags;
        if (!decode_initial(f, &left, &left_end, &right, &right_end, &mode, &flags)) break;
        Mode *m = f->mode_config + mode;
        const int n = f->blocksize[m->blockflag];
        const int left_start = left;
        if (!decode_packet_rest(f, &len, m, left, left_end, right, right_end, &left)) break;
        /* record the packet */
        size_t need = (size_t)out->spec_floats + (size_t)f->channels * (size_t)(n / 2);
        if (!grow((void **)&out->pflags, &cap_p, out->n_packets + 1, 1) || !grow((void **)&out->spec, &cap_s, need, sizeof(float)) ||
            !grow((void **)&out->take_from, &cap_l, out->n_packets + 1, sizeof(int32_t)) ||
            !grow((void **)&out->take_count, &cap_n, out->n_packets + 1, sizeof(int32_t))) {
            free_all(f); free(f); return -2;
        }
        out->pflags[out->n_packets] = (uint8_t)flags;
        for (int c = 0; c < f->channels; c++)
            memcpy(out->spec + out->spec_floats + (size_t)c * (size_t)(n / 2), f->channel_buffers[c], sizeof(float) * (size_t)(n / 2));
        out->spec_floats = need;
        /* vorbis_finish_frame's return (vb:2606-2657): samples [left, min(len, right)) once a previous frame exists */
        int r = right;
        if (len < r) r = len;
        int count = first ? 0 : r - left;
        if (count < 0) count = 0;
        out->take_from[out->n_packets] = first ? 0 : left - left_start;
        out->take_count[out->n_packets] = count;
        out->n_packets++;
        out->pcm_frames += (uint64_t)count;
        first = 0;
        if (out->n_packets == 1) out->total_samples = stream_length_in_samples(f);   /* the caller asks right after opening */
    }
    if (out->n_packets == 0) out->total_samples = stream_length_in_samples(f);
    free_all(f);
    free(f);
    return 0;
}

void afgo_vorbis_file_free(afgo_vorbis_file *f)
{
    free(f->pflags);
    free(f->spec);
    free(f->take_from);
    free(f->take_count);
    memset(f, 0, sizeof(*f));
}

/* ---- the same tail driven by records (checker of the device floor stage) ---- */
void afgo_vorbis_floor(uint64_t n_packets, const afgo_vorbis_floor_packet *packets, const afgo_vorbis_floor_curve *curves,
                       const int32_t *points, const uint8_t *steps, float *spec)
{
    uint64_t p;
    for (p = 0; p < n_packets; ++p) {
        const afgo_vorbis_floor_packet *k = packets + p;
        const int n2 = (int)k->n2;
        float *base = spec + k->spec_off;
        uint32_t s, c;
        int j;
        /* INVERSE COUPLING, vb:2493-2514 (steps are stored in the order the loop visits them) */
        for (s = 0; s < k->n_steps; ++s) {
            float *m_ = base + (size_t)steps[2 * (k->step_off + s)] * (size_t)n2;
            float *a = base + (size_t)steps[2 * (k->step_off + s) + 1] * (size_t)n2;
            for (j = 0; j < n2; ++j) {
                float a2, m2;
                if (m_[j] > 0)
                    if (a[j] > 0) { m2 = m_[j]; a2 = m_[j] - a[j]; }
                    else { a2 = m_[j]; m2 = m_[j] + a[j]; }
                else
                    if (a[j] > 0) { m2 = m_[j]; a2 = m_[j] + a[j]; }
                    else { a2 = m_[j]; m2 = m_[j] - a[j]; }
                m_[j] = m2;
                a[j] = a2;
            }
        }
        /* finish decoding the floors, vb:2516-2523 with do_floor, vb:2255-2284 */
        for (c = 0; c < k->channels; ++c) {
            float *target = base + (size_t)c * (size_t)n2;
            const afgo_vorbis_floor_curve *cv = curves + k->curve_index + c;
            const int32_t *pt = points + 2 * (size_t)cv->point_off;
            uint32_t q;
            int lx, ly;
            if (cv->n_points == 0) {
                memset(target, 0, sizeof(float) * (size_t)n2);
                continue;
            }
            lx = 0;
            ly = pt[1];
            for (q = 1; q < cv->n_points; ++q) {
                const int hx = pt[2 * q], hy = pt[2 * q + 1];
                if (lx != hx) draw_line(target, lx, ly, hx, hy, n2);
                lx = hx;
                ly = hy;
            }
            if (lx < n2)
                for (j = lx; j < n2; ++j) target[j] *= f32(k_inverse_db_bits[ly & 255]);
        }
    }
}
