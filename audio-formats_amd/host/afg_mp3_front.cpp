// afg_mp3_front.cpp -- host front-end for MPEG-1/2/2.5 Layer III files.
//
// Everything the reference does ahead of the transform seam (minimp3.d:1226): frame sync
// (minimp3.d:1436-1484), header / side info (:487-614), scalefactors (:616-719), Huffman decoding and
// requantisation (:721-883), mid/side and intensity stereo (:885-983), short-block reorder (:985-1000),
// the bit reservoir (:1170-1197), and the file-level drive of minimp3_ex (ID3v2 / ID3v1 / APE skipping
// :93-142, Xing/Info tag and LAME delay/padding :144-190, :566-621, delivery order and trimming
// :787-888).  Output: the transform-stage records of include/afg.h (576 floats + one flag word per
// granule-channel) and a copy plan that turns the device's PCM plane into what mp3dec_ex_read returns.
//
// Organisation (not the reference's): the main-data stream of a file is treated as one byte queue that
// frames append to and granules consume from; a granule's spectrum is decoded by spectral-line index
// against precomputed band boundaries with two-level lookup tables for the code books
// (mp3_front_tables.h, generated); every float is produced by the reference's expression trees, so the
// records are bit-identical to the oracle's.  Layer I / II frames are not handled (the stream ends there).
#include "afg_mp3_front.h"

#include "mp3_front_tables.h"
#include "mp3_l12_tables.h"

#include <algorithm>
#include <cstring>

namespace afg_mp3 {
namespace {

inline float bits_f32(unsigned b)
{
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

// ---- frame header ------------------------------------------------------------------------------
struct Header {
    const uint8_t *h;
    bool mpeg1() const { return (h[1] & 0x08) != 0; }
    bool mpeg25() const { return (h[1] & 0x10) == 0; }
    int layer_code() const { return (h[1] >> 1) & 3; }              // 1 = Layer III
    bool crc() const { return (h[1] & 1) == 0; }
    int bitrate_index() const { return h[2] >> 4; }
    int rate_index() const { return (h[2] >> 2) & 3; }
    bool padding() const { return (h[2] & 2) != 0; }
    bool mono() const { return (h[3] & 0xC0) == 0xC0; }
    bool ms_stereo() const { return (h[3] & 0xE0) == 0x60; }       // joint stereo with the M/S bit
    bool ms_bit() const { return (h[3] & 0x20) != 0; }
    bool intensity() const { return (h[3] & 0x10) != 0; }
    bool free_format() const { return (h[2] & 0xF0) == 0; }
    bool layer1() const { return (h[1] & 6) == 6; }
    bool valid() const
    {
        return h[0] == 0xff && ((h[1] & 0xF0) == 0xf0 || (h[1] & 0xFE) == 0xe2) && layer_code() != 0 &&
               bitrate_index() != 15 && rate_index() != 3;
    }
    bool same_stream(const uint8_t *o) const                          // hdr_compare, minimp3.d:222-228
    {
        const Header b{ o };
        return b.valid() && ((h[1] ^ o[1]) & 0xFE) == 0 && ((h[2] ^ o[2]) & 0x0C) == 0 && free_format() == b.free_format();
    }
    unsigned kbps() const { return 2u * k_halfrate[mpeg1() ? 1 : 0][layer_code() - 1][bitrate_index()]; }
    unsigned hz() const
    {
        static const unsigned base[3] = { 44100, 48000, 32000 };
        return base[rate_index()] >> (mpeg1() ? 0 : 1) >> (mpeg25() ? 1 : 0);
    }
    unsigned samples() const { return layer1() ? 384u : (1152u >> (((h[1] & 14) == 2) ? 1 : 0)); }
    int bytes(int free_format_size) const
    {
        int n = (int)(samples() * kbps() * 125 / hz());
        if (layer1()) n &= ~3;
        return n ? n : free_format_size;
    }
    int pad_bytes() const { return padding() ? (layer1() ? 4 : 1) : 0; }
    int band_table() const                                            // index into the sfb tables, minimp3.d:534
    {
        int i = rate_index() + (((h[1] >> 3) & 1) + ((h[1] >> 4) & 1)) * 3;
        return i - (i != 0);
    }
    bool low_rate_25() const { return rate_index() + (((h[1] >> 3) & 1) + ((h[1] >> 4) & 1)) * 3 == 2; }   // :1218
};

constexpr int kMaxFreeFormat = 2304, kSyncMatches = 10, kReservoir = 511;

bool confirm_sync(const uint8_t *p, int avail, int frame_bytes)       // mp3d_match_frame
{
    const Header first{ p };
    int at = 0;
    for (int n = 0; n < kSyncMatches; n++) {
        const Header cur{ p + at };
        at += cur.bytes(frame_bytes) + cur.pad_bytes();
        if (at + 4 > avail) return n > 0;
        if (!first.same_stream(p + at)) return false;
    }
    return true;
}

// mp3d_find_frame: offset of the next frame in [p, p + avail) (avail if none), its size in *size
int next_frame(const uint8_t *p, int avail, int *free_bytes, int *size)
{
    for (int i = 0; i < avail - 4; i++, p++) {
        const Header hd{ p };
        if (!hd.valid()) continue;
        int fb = hd.bytes(*free_bytes);
        int total = fb + hd.pad_bytes();
        for (int k = 4; !fb && k < kMaxFreeFormat && i + 2 * k < avail - 4; k++) {
            if (!hd.same_stream(p + k)) continue;
            const int cand = k - hd.pad_bytes();
            const int nxt = cand + Header{ p + k }.pad_bytes();
            if (i + k + nxt + 4 > avail || !hd.same_stream(p + k + nxt)) continue;
            total = k;
            fb = cand;
            *free_bytes = cand;
        }
        if ((fb && i + total <= avail && confirm_sync(p, avail - i, fb)) || (!i && total == avail)) {
            *size = total;
            return i;
        }
        *free_bytes = 0;
    }
    *size = 0;
    return avail;
}

// ---- bits ----------------------------------------------------------------------------------------
// Side info / scalefactor reader with the reference's overrun rule (a read past the end yields 0 and
// still advances, minimp3.d:192-207).
struct Bits {
    const uint8_t *p;
    int pos, limit;
    Bits(const uint8_t *d, int bytes) : p(d), pos(0), limit(bytes * 8) {}
    uint32_t get(int n)
    {
        const int at = pos;
        pos += n;
        if (pos > limit || n == 0) return 0;
        uint64_t w = 0;
        const uint8_t *q = p + (at >> 3);
        for (int i = 0; i < 5; i++) w = (w << 8) | q[i];            // callers keep 8 readable bytes behind the data
        return (uint32_t)((w >> (40 - (at & 7) - n)) & ((1ull << n) - 1));
    }
};

// Spectrum reader: a 64-bit window kept at >= 32 valid bits.
struct Window {
    const uint8_t *base, *next;
    uint64_t w;
    int have;                 // valid bits in w (left-aligned)
    int consumed;             // bits consumed since `base`
    Window(const uint8_t *d, int bitpos) : base(d), next(d + (bitpos >> 3)), w(0), have(0), consumed(bitpos & ~7)
    {
        fill();
        drop(bitpos & 7);
    }
    void fill()
    {
        while (have <= 56) {
            w |= (uint64_t)*next++ << (56 - have);
            have += 8;
        }
    }
    uint32_t peek(int n) const { return n ? (uint32_t)(w >> (64 - n)) : 0; }
    void drop(int n)
    {
        w <<= n;
        have -= n;
        consumed += n;
    }
    bool top() const { return (int64_t)w < 0; }
};

struct Granule {                       // side info of one granule-channel
    const uint8_t *bands;              // widths, 0-terminated
    int part23, big_values, sf_compress, global_gain, block_type, mixed;
    int n_long, n_short;
    int table[3], region[3], sub_gain[3];
    int preflag, sf_scale, count1_table, scfsi;
};

// L3_read_side_info; returns main_data_begin or -1
int read_side_info(Bits &b, Granule *g, const Header &hd)
{
    const int bt = hd.band_table();
    int n = hd.mono() ? 1 : 2, begin;
    unsigned scfsi = 0;
    int sum = 0;
    if (hd.mpeg1()) {
        n *= 2;
        begin = (int)b.get(9);
        scfsi = b.get(7 + n);
    } else {
        begin = (int)(b.get(8 + n) >> n);
    }
    for (int k = 0; k < n; k++, g++) {
        if (hd.mono()) scfsi <<= 4;
        g->part23 = (int)b.get(12);
        sum += g->part23;
        g->big_values = (int)b.get(9);
        if (g->big_values > 288) return -1;
        g->global_gain = (int)b.get(8);
        g->sf_compress = (int)b.get(hd.mpeg1() ? 4 : 9);
        g->bands = k_sfb_long[bt];
        g->n_long = 22;
        g->n_short = 0;
        unsigned tables;
        if (b.get(1)) {
            g->block_type = (int)b.get(2);
            if (!g->block_type) return -1;
            g->mixed = (int)b.get(1);
            g->region[0] = 7;
            g->region[1] = 255;
            if (g->block_type == 2) {
                scfsi &= 0x0F0F;
                if (!g->mixed) {
                    g->region[0] = 8;
                    g->bands = k_sfb_short[bt];
                    g->n_long = 0;
                    g->n_short = 39;
                } else {
                    g->bands = k_sfb_mixed[bt];
                    g->n_long = hd.mpeg1() ? 8 : 6;
                    g->n_short = 30;
                }
            }
            tables = b.get(10) << 5;
            for (int i = 0; i < 3; i++) g->sub_gain[i] = (int)b.get(3);
        } else {
            g->block_type = 0;
            g->mixed = 0;
            tables = b.get(15);
            g->region[0] = (int)b.get(4);
            g->region[1] = (int)b.get(3);
            g->region[2] = 255;
        }
        g->table[0] = (int)(tables >> 10);
        g->table[1] = (int)((tables >> 5) & 31);
        g->table[2] = (int)(tables & 31);
        g->preflag = hd.mpeg1() ? (int)b.get(1) : (g->sf_compress >= 500);
        g->sf_scale = (int)b.get(1);
        g->count1_table = (int)b.get(1);
        g->scfsi = (int)((scfsi >> 12) & 15);
        scfsi <<= 4;
    }
    if (sum + b.pos > b.limit + begin * 8) return -1;
    return begin;
}

// L3_ldexp_q2: y * 2^(-exp/4) built from quarter-octave steps (float products in this order)
float scale_q2(float y, int exp_q2)
{
    int e;
    do {
        e = std::min(30 * 4, exp_q2);
        y *= bits_f32(k_expfrac_bits[e & 3]) * (float)(1 << 30 >> (e >> 2));
    } while ((exp_q2 -= e) > 0);
    return y;
}

// L3_pow_43
float pow43(int x)
{
    if (x < 129) return bits_f32(k_pow43_bits[16 + x]);
    int mult = 256;
    if (x < 1024) {
        mult = 16;
        x <<= 3;
    }
    const int sign = 2 * x & 64;
    const float frac = (float)((x & 63) - sign) / (float)((x & ~63) + sign);
    return bits_f32(k_pow43_bits[16 + ((x + sign) >> 6)]) * (1.0f + frac * ((4.0f / 3) + frac * (2.0f / 9))) * (float)mult;
}

constexpr int kMaxScf = 255 - 4 - 210, kMaxScfi = (kMaxScf + 3) & ~3;

// L3_decode_scalefactors: band scales `scale[0 .. n_long + n_short)`, intensity positions in ist[]
void band_scales(const Header &hd, uint8_t *ist, Bits &b, const Granule &g, float *scale, int ch)
{
    const uint8_t *parts = k_scf_partitions[(g.n_short ? 1 : 0) + (g.n_long ? 0 : 1)];
    uint8_t width[4], iscf[40];
    int scfsi = g.scfsi;
    const int shift = g.sf_scale + 1;
    if (hd.mpeg1()) {
        const int part = k_scfc_decode[g.sf_compress];
        width[0] = width[1] = (uint8_t)(part >> 2);
        width[2] = width[3] = (uint8_t)(part & 3);
    } else {
        const int ist_ch = (hd.intensity() && ch) ? 1 : 0;
        int sfc = g.sf_compress >> ist_ch, k = ist_ch * 12, prod;
        for (; sfc >= 0; sfc -= prod, k += 4) {
            prod = 1;
            for (int i = 3; i >= 0; i--) {
                width[i] = (uint8_t)(sfc / prod % k_scf_mod[k + i]);
                prod *= k_scf_mod[k + i];
            }
        }
        parts += k;
        scfsi = -16;
    }
    {   // L3_read_scalefactors
        uint8_t *s = iscf, *ip = ist;
        for (int i = 0; i < 4 && parts[i]; i++, scfsi *= 2) {
            const int cnt = parts[i];
            if (scfsi & 8) {
                std::memcpy(s, ip, (size_t)cnt);
            } else if (!width[i]) {
                std::memset(s, 0, (size_t)cnt);
                std::memset(ip, 0, (size_t)cnt);
            } else {
                const int none = (scfsi < 0) ? (1 << width[i]) - 1 : -1;
                for (int k = 0; k < cnt; k++) {
                    const int v = (int)b.get(width[i]);
                    ip[k] = (uint8_t)(v == none ? -1 : v);
                    s[k] = (uint8_t)v;
                }
            }
            ip += cnt;
            s += cnt;
        }
        s[0] = s[1] = s[2] = 0;
    }
    if (g.n_short) {
        const int sh = 3 - shift;
        for (int i = 0; i < g.n_short; i += 3)
            for (int w = 0; w < 3; w++) iscf[g.n_long + i + w] = (uint8_t)(iscf[g.n_long + i + w] + (g.sub_gain[w] << sh));
    } else if (g.preflag) {
        for (int i = 0; i < 10; i++) iscf[11 + i] = (uint8_t)(iscf[11 + i] + k_preamp[i]);
    }
    const int gain_exp = g.global_gain - 4 - 210 - (hd.ms_stereo() ? 2 : 0);
    const float gain = scale_q2((float)(1 << (kMaxScfi / 4)), kMaxScfi - gain_exp);
    for (int i = 0; i < g.n_long + g.n_short; i++) scale[i] = scale_q2(gain, iscf[i] << shift);
}

// L3_huffman by spectral line: big-value pairs of the three regions, then count1 quads up to the end of
// part2_3 (a quad that ends past it is discarded) or the end of the band table.
// With `qline` the Huffman values themselves are recorded (sign included) instead of the requantised lines: the device
// requantises them (afg_mp3_requant_hip).
void spectrum(float *line, const uint8_t *main_data, int *bitpos, const Granule &g, const float *scale, int limit,
              int16_t *qline = nullptr)
{
    int ends[41], nb = 0, acc = 0;
    for (; g.bands[nb]; nb++) ends[nb] = (acc += g.bands[nb]);
    const int total = acc;                                  // 576 for every table
    Window win(main_data, *bitpos);
    int i = 0, band = 0;
    const int big_end = std::min(2 * g.big_values, total);
    int first_band = 0;
    for (int r = 0; r < 3 && i < big_end; r++) {
        const int last_band = std::min(first_band + g.region[r], nb - 1);      // a region spans region[r] + 1 bands
        const int stop = std::min(ends[last_band], big_end);
        first_band = last_band + 1;
        const int t = g.table[r], book = k_book_of_table[t], extra = k_linbits[t];
        const unsigned *lut = k_huff_lut + k_huff_lut_base[book];
        while (i < stop) {
            while (i >= ends[band]) band++;
            const float one = scale[band];
            int xy = 0;
            if (book) {
                unsigned e = lut[win.peek(AFG_MP3_HUFF_L1)];
                if (e & 0x80000000u) {
                    const int sub = (int)((e >> 24) & 0x7f);
                    e = lut[(e & 0xffffff) + ((win.peek(AFG_MP3_HUFF_L1 + sub)) & ((1u << sub) - 1))];
                }
                win.drop((int)(e >> 8));
                xy = (int)(e & 0xff);
            }
            for (int j = 0; j < 2; j++, xy >>= 4) {
                int v = xy & 15;
                if (extra && v == 15) {
                    v += (int)win.peek(extra);
                    win.drop(extra);
                    win.fill();
                    if (qline) qline[i + j] = (int16_t)(win.top() ? -v : v);
                    else line[i + j] = one * pow43(v) * (win.top() ? -1 : 1);
                } else if (qline) {
                    qline[i + j] = (int16_t)((v && win.top()) ? -v : v);
                } else {
                    line[i + j] = bits_f32(k_pow43_bits[16 + v - (win.top() ? 16 : 0)]) * one;
                }
                if (v) win.drop(1);
            }
            win.fill();
            i += 2;
        }
        if (i >= big_end) break;
    }
    // count1 region.  The reference tracks "pairs left in the current band" starting from wherever the
    // big values stopped; by line index that is simply the band containing the line.
    const unsigned char *c1 = k_count1_lut + (g.count1_table ? 64 : 0);
    i = big_end;
    // when the big values overshoot a band the reference keeps counting from its own pair counter; for a
    // well-formed granule 2*big_values lands on a pair boundary inside a band, handled here by index
    for (;; i += 4) {
        const unsigned e = c1[win.peek(6)];
        win.drop((int)(e >> 4));
        if (win.consumed > limit) break;
        if (i >= total) break;
        while (band < nb && i >= ends[band]) band++;
        if (band >= nb) break;
        float one = scale[band];
        auto unit = [&](int at) {                          // a count1 line: +-1 (requantised: +-one)
            if (qline) qline[at] = (int16_t)(win.top() ? -1 : 1);
            else line[at] = win.top() ? -one : one;
            win.drop(1);
        };
        if (e & 8) unit(i);
        if (e & 4) unit(i + 1);
        if (i + 2 >= total) break;
        int b2 = band;
        while (b2 < nb && i + 2 >= ends[b2]) b2++;
        if (b2 >= nb) break;
        one = scale[b2];
        if (e & 2) unit(i + 2);
        if (e & 1) unit(i + 3);
        win.fill();
    }
    *bitpos = limit;
}

void mid_side(float *l, int n)
{
    float *r = l + 576;
    for (int i = 0; i < n; i++) {
        const float a = l[i], b = r[i];
        l[i] = a + b;
        r[i] = a - b;
    }
}

// L3_intensity_stereo + L3_stereo_process, in two halves: the plan (which band gets which treatment: needs to know where
// the right channel still has energy) and its application to the lines.  The quantised path ships the plan to the device.
struct StereoPlan {
    uint8_t type[40];                  // per band of the left channel's table: 0 leave, 1 mid/side, 2 intensity
    float fl[40], fr[40];
};

template <typename RightNonzero>
void intensity_plan(StereoPlan &plan, uint8_t *ist, const Granule *g, const Header &hd, RightNonzero right_nonzero)
{
    const int nb = g->n_long + g->n_short, blocks = g->n_short ? 3 : 1;
    int top[3] = { -1, -1, -1 };
    {   // last band of the right channel with energy, per window
        int at = 0;
        for (int b = 0; b < nb; b++) {
            for (int k = 0; k < g->bands[b]; k += 2)
                if (right_nonzero(at + k) || right_nonzero(at + k + 1)) { top[b % 3] = b; break; }
            at += g->bands[b];
        }
    }
    if (g->n_long) top[0] = top[1] = top[2] = std::max(std::max(top[0], top[1]), top[2]);
    for (int i = 0; i < blocks; i++) {
        const int itop = nb - blocks + i, prev = itop - blocks;
        ist[itop] = (uint8_t)(top[i] >= prev ? (hd.mpeg1() ? 3 : 0) : ist[prev]);
    }
    const int lsf_shift = g[1].sf_compress & 1;
    const unsigned max_pos = hd.mpeg1() ? 7 : 64;
    std::memset(&plan, 0, sizeof(plan));
    for (int b = 0; g->bands[b]; b++) {
        const unsigned ipos = ist[b];
        if (b > top[b % 3] && ipos < max_pos) {
            const float s = hd.ms_bit() ? 1.41421356f : 1;
            float kl, kr;
            if (hd.mpeg1()) {
                kl = bits_f32(k_pan_bits[2 * ipos]);
                kr = bits_f32(k_pan_bits[2 * ipos + 1]);
            } else {
                kl = 1;
                kr = scale_q2(1, (int)((ipos + 1) >> 1 << lsf_shift));
                if (ipos & 1) {
                    kl = kr;
                    kr = 1;
                }
            }
            plan.type[b] = 2;
            plan.fl[b] = kl * s;
            plan.fr[b] = kr * s;
        } else if (hd.ms_bit()) {
            plan.type[b] = 1;
        }
    }
}

void intensity(float *left, uint8_t *ist, const Granule *g, const Header &hd)
{
    StereoPlan plan;
    const float *r = left + 576;
    intensity_plan(plan, ist, g, hd, [&](int k) { return r[k] != 0; });
    float *l = left;
    for (int b = 0; g->bands[b]; b++) {
        const int n = g->bands[b];
        if (plan.type[b] == 2) {
            const float fl = plan.fl[b], fr = plan.fr[b];
            for (int i = 0; i < n; i++) {
                l[i + 576] = l[i] * fr;
                l[i] = l[i] * fl;
            }
        } else if (plan.type[b] == 1) {
            mid_side(l, n);
        }
        l += n;
    }
}

// L3_reorder: [band][window][line] -> [band][line][window].  For MPEG-2.5 at 8 kHz a mixed block starts its
// short part at line 72 (4 subbands, minimp3.d:1218) while the band table's short part starts at line 48, so the
// reference walks 24 lines past the channel's 576 (minimp3.d:1223): into the next channel's spectrum for channel 0,
// into the scalefactor array behind grbuf for channel 1.  The work area below has the same layout so that even
// this corner gives the reference's numbers.
struct Work {
    float x[2][576];
    float scale[40];                   // directly behind the spectra, as mp3dec_scratch_t.scf (minimp3.d:179-181)
    float slack[24];
};

// which of the 24 scalefactor-band tables a granule's `bands` points into: kind * 8 + sample-rate row
int band_table_id(const uint8_t *bands)
{
    const uint8_t *bases[3] = { &k_sfb_long[0][0], &k_sfb_short[0][0], &k_sfb_mixed[0][0] };
    const size_t row_bytes[3] = { sizeof(k_sfb_long[0]), sizeof(k_sfb_short[0]), sizeof(k_sfb_mixed[0]) };
    for (int kind = 0; kind < 3; kind++)
        for (int row = 0; row < 8; row++)
            if (bands == bases[kind] + (size_t)row * row_bytes[kind]) return kind * 8 + row;
    return -1;
}

void interleave_windows(float *x, const uint8_t *bands)
{
    float tmp[640];
    float *d = tmp;
    const float *s = x;
    for (int len; (len = *bands) != 0; bands += 3, s += 3 * len)
        for (int i = 0; i < len; i++) {
            *d++ = s[i];
            *d++ = s[len + i];
            *d++ = s[2 * len + i];
        }
    std::memcpy(x, tmp, (size_t)(d - tmp) * sizeof(float));
}

// ---- ID3 / APE --------------------------------------------------------------------------------------
size_t id3v2_size(const uint8_t *p, size_t n)
{
    if (n >= 10 && !std::memcmp(p, "ID3", 3) && !((p[5] & 15) || (p[6] & 0x80) || (p[7] & 0x80) || (p[8] & 0x80) || (p[9] & 0x80))) {
        size_t sz = (size_t)(((p[6] & 0x7f) << 21) | ((p[7] & 0x7f) << 14) | ((p[8] & 0x7f) << 7) | (p[9] & 0x7f)) + 10;
        if (p[5] & 16) sz += 10;
        return sz;
    }
    return 0;
}
void trim_trailing_tags(const uint8_t *p, size_t *n)
{
    size_t m = *n;
    if (m >= 128 && !std::memcmp(p + m - 128, "TAG", 3)) {
        m -= 128;
        if (m >= 227 && !std::memcmp(p + m - 227, "TAG+", 4)) m -= 227;
    }
    if (m > 32 && !std::memcmp(p + m - 32, "APETAGEX", 8)) {
        m -= 32;
        const uint8_t *t = p + m + 12;
        const uint32_t sz = ((uint32_t)t[3] << 24) | ((uint32_t)t[2] << 16) | ((uint32_t)t[1] << 8) | t[0];
        if (m >= sz) m -= sz;
    }
    *n = m;
}

// mp3dec_check_vbrtag: 1 tag with frame count, -1 tag without, 0 none
int info_tag(const uint8_t *frame, int size, uint32_t *frames, int *delay, int *padding)
{
    const Header hd{ frame };
    Bits b(frame + 4, size - 4);
    if (hd.crc()) b.get(16);
    Granule g[4];
    if (read_side_info(b, g, hd) < 0) return 0;
    const uint8_t *t = frame + 4 + b.pos / 8;
    if (std::memcmp(t, "Xing", 4) && std::memcmp(t, "Info", 4)) return 0;
    const int fl = t[7];
    if (!(fl & 1)) return -1;
    t += 8;
    *frames = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    t += 4;
    if (fl & 2) t += 4;
    if (fl & 4) t += 100;
    if (fl & 8) t += 4;
    *delay = *padding = 0;
    if (*t) {
        t += 21;
        if (t - frame + 14 >= size) return 0;
        *delay = ((t[0] << 4) | (t[1] >> 4)) + 529;
        *padding = (((t[1] & 0xF) << 8) | t[2]) - 529;
    }
    return 1;
}

// ---- one file --------------------------------------------------------------------------------------
struct Decoder {                           // what persists between frames on the host side
    uint8_t header[4] = { 0, 0, 0, 0 };
    int free_bytes = 0;
    std::vector<uint8_t> reservoir;        // unread main data of earlier frames (<= 511 bytes)
    void reset()
    {
        header[0] = 0;
        free_bytes = 0;
        reservoir.clear();
    }
};


// ---- Layer I / II (minimp3.d:286-484) -------------------------------------------------------------------------------
// Quantiser codes: 0 = no samples, 2..16 = bits per sample ((1 << bits) - 1 levels), kG3 / kG5 / kG9 = three consecutive
// samples in one 5 / 7 / 10-bit codeword of 3 / 5 / 9 levels each.  The allocation classes below are the columns of
// ISO 11172-3 tables B.2a-d and 13818-3 table B.1 (what an allocation index means in a given subband).
enum : uint8_t { kG3 = 17, kG5 = 18, kG9 = 19 };
const uint8_t kAllocA[16] = { 0, kG3, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };          // B.2a/b subbands 0-2
const uint8_t kAllocB[16] = { 0, kG3, kG5, 3, kG9, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16 };       // B.2a/b subbands 3-10
const uint8_t kAllocC[8] = { 0, kG3, kG5, 3, kG9, 4, 5, 16 };                                     // B.2a/b subbands 11-22
const uint8_t kAllocD[4] = { 0, kG3, kG5, 16 };                                                   // B.2a/b subbands 23-
const uint8_t kAllocE[16] = { 0, kG3, kG5, kG9, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 };      // B.2c/d; B.1 subbands 4- (first 8 / 4)
const uint8_t kAllocF[16] = { 0, kG3, kG5, 3, kG9, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14 };       // B.1 subbands 0-3
const uint8_t kAllocL1[16] = { 0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16 };          // Layer I: index + 1 bits
struct AllocRun { const uint8_t *quant; uint8_t width, bands; };

struct L12Frame {
    int bands = 0, two_channel_bands = 0;       // coded subbands; those below two_channel_bands carry both channels separately
    uint8_t quant[32][2];
    float factor[32][2][3];                     // per (subband, channel): the scale of each third of the frame

    void read(const Header &hd, Bits &in)       // L12_read_scale_info
    {
        static const AllocRun l1[] = { { kAllocL1, 4, 32 } };
        static const AllocRun lsf[] = { { kAllocF, 4, 4 }, { kAllocE, 3, 7 }, { kAllocE, 2, 19 } };
        static const AllocRun full[] = { { kAllocA, 4, 3 }, { kAllocB, 4, 8 }, { kAllocC, 3, 12 }, { kAllocD, 2, 7 } };
        static const AllocRun low[] = { { kAllocE, 4, 2 }, { kAllocE, 3, 10 } };
        const int mode = (hd.h[3] >> 6) & 3;
        const int bound = mode == 3 ? 0 : mode == 1 ? (((hd.h[3] >> 4) & 3) << 2) + 4 : 32;
        const AllocRun *run;
        if (hd.layer1()) {
            run = l1;
            bands = 32;
        } else if (!hd.mpeg1()) {
            run = lsf;
            bands = 30;
        } else {
            unsigned per_channel = hd.kbps() >> (mode != 3 ? 1 : 0);
            if (!per_channel) per_channel = 192;                 // free format
            run = full;
            bands = 27;
            if (per_channel < 56) {
                run = low;
                bands = hd.rate_index() == 2 ? 12 : 8;
            } else if (per_channel >= 96 && hd.rate_index() != 1) {
                bands = 30;
            }
        }
        two_channel_bands = std::min(bound, bands);
        std::memset(quant, 0, sizeof(quant));
        std::memset(factor, 0, sizeof(factor));
        uint8_t scf_quant[32][2];                                // what the scalefactor pass sees (a shared allocation counts for both)
        for (int b = 0, left = 0; b < bands; b++) {
            if (!left) left = run->bands; 
            uint8_t q = run->quant[in.get(run->width)];
            scf_quant[b][0] = q;
            if (b < two_channel_bands) q = run->quant[in.get(run->width)];
            scf_quant[b][1] = two_channel_bands ? q : 0;
            if (--left == 0) run++;
        }
        uint8_t select[32][2];                                   // scalefactor selection information
        for (int b = 0; b < bands; b++)
            for (int c = 0; c < 2; c++) {
                const uint8_t sel = hd.layer1() ? 2 : (uint8_t)in.get(2);
                select[b][c] = scf_quant[b][c] ? sel : 6;
            }
        for (int b = 0; b < bands; b++)
            for (int c = 0; c < 2; c++) {
                const int q = scf_quant[b][c];
                const int present = q ? 4 + ((19 >> select[b][c]) & 3) : 0;       // which of the three are transmitted
                float f = 0.0f;
                for (int third = 0, bit = 4; third < 3; third++, bit >>= 1) {
                    if (present & bit) {
                        const int idx = (int)in.get(6);
                        f = bits_f32(k_l12_deq_bits[q * 3 - 6 + idx % 3]) * (float)(1 << 21 >> idx / 3);
                    }
                    factor[b][c][third] = f;
                }
            }
        for (int b = 0; b < bands; b++) {
            quant[b][0] = scf_quant[b][0];
            quant[b][1] = b < two_channel_bands ? scf_quant[b][1] : 0;            // above the bound: one set of samples for both
        }
    }

    // four sample groups of `per_group` slots each, starting at slot `at` (L12_dequantize_granule)
    void samples(Bits &in, float (*g)[576], int at, int per_group) const
    {
        for (int j = 0; j < 4; j++, at += per_group)
            for (int b = 0; b < bands; b++)
                for (int c = 0; c < 2; c++) {
                    const int q = quant[b][c];
                    if (!q) continue;
                    float *dst = g[c] + b * 18 + at;
                    if (q < kG3) {
                        const int half = (1 << (q - 1)) - 1;
                        for (int k = 0; k < per_group; k++) dst[k] = (float)((int)in.get(q) - half);
                    } else {
                        const unsigned levels = (2u << (q - kG3)) + 1;
                        unsigned code = in.get((int)(levels + 2 - (levels >> 3)));
                        for (int k = 0; k < per_group; k++, code /= levels) dst[k] = (float)((int)(code % levels) - (int)(levels / 2));
                    }
                }
    }

    // scale the 12 slots of a finished granule with the factors of frame third `third` (L12_apply_scf_384)
    void scale(float (*g)[576], int third) const
    {
        for (int b = two_channel_bands; b < bands; b++) std::memcpy(g[1] + b * 18, g[0] + b * 18, 18 * sizeof(float));
        for (int b = 0; b < bands; b++)
            for (int k = 0; k < 12; k++) {
                g[0][b * 18 + k] *= factor[b][0][third];
                g[1][b * 18 + k] *= factor[b][1][third];
            }
    }
};

struct FrameResult {
    int consumed = 0;          // bytes to advance
    int samples = 0;           // per channel, 0 = nothing decoded
    int channels = 0, hz = 0, layer = 0;
    bool stop = false;         // allocation trouble, or a layer the chosen record format does not carry
    uint64_t pcm_at = 0;       // Layer I / II: float offset of this frame's first sample in the PCM plane of `out`
};

// mp3dec_decode_frame up to the seam; records go to `out` when given (the open scan passes nullptr)
FrameResult frame(Decoder &d, const uint8_t *p, int avail, File *out, bool *fresh_state)
{
    FrameResult r;
    int at = 0, size = 0;
    const Header kept{ d.header };
    if (avail > 4 && d.header[0] == 0xff && kept.same_stream(p)) {
        const Header hd{ p };
        size = hd.bytes(d.free_bytes) + hd.pad_bytes();
        if (size != avail && (size + 4 > avail || !hd.same_stream(p + size))) size = 0;
    }
    if (!size) {
        d.reset();                                         // memset(dec, 0): transform state and reservoir start over
        *fresh_state = true;
        at = next_frame(p, avail, &d.free_bytes, &size);
        if (!size || at + size > avail) {
            r.consumed = at;
            return r;
        }
    }
    const uint8_t *fp = p + at;
    const Header hd{ fp };
    std::memcpy(d.header, fp, 4);
    r.consumed = at + size;
    r.channels = hd.mono() ? 1 : 2;
    r.hz = (int)hd.hz();
    r.layer = 4 - hd.layer_code();
    if (r.layer != 3 && out && out->quantised) {           // the device requantiser is Layer III's: the caller takes the float path
        out->q_unsupported = true;
        r.stop = true;
        return r;
    }
    // side info; the frame body is copied so that short frames can be read with slack behind them
    uint8_t body[kMaxFreeFormat + 16];
    const int body_bytes = std::min(size - 4, kMaxFreeFormat);
    std::memcpy(body, fp + 4, (size_t)body_bytes);
    std::memset(body + body_bytes, 0, 16);
    Bits sb(body, body_bytes);
    if (hd.crc()) sb.get(16);
    if (r.layer != 3) {
        // Layer I / II (minimp3.d:1557-1578): allocation + scalefactors, then three thirds of the frame; a Layer II third is a
        // synthesis granule of its own (3 x 4 groups of 3 slots), Layer I gathers its three thirds (4 slots each) into one
        L12Frame info;
        info.read(hd, sb);
        const int per_group = hd.layer1() ? 1 : 3, nch = r.channels;
        float done[3][2][576];
        int n_done = 0, filled = 0;
        std::memset(done, 0, sizeof(done));
        for (int third = 0; third < 3; third++) {
            info.samples(sb, done[n_done], filled, per_group);
            filled += 4 * per_group;
            if (filled == 12) {
                info.scale(done[n_done], third);
                n_done++;
                filled = 0;
            }
            if (sb.pos > sb.limit) {
                d.header[0] = 0;                           // mp3dec_init: nothing of this frame is kept, the next call resynchronises
                return r;
            }
        }
        if (out) {
            if (*fresh_state || out->run_granules.empty()) {
                out->flush_l12(nch);                       // the run before ends in a padded block
                if (out->run_granules.empty()) out->continues_previous = !*fresh_state;
                out->run_granules.push_back(0);
                *fresh_state = false;
            }
            out->layer = r.layer;
            r.pcm_at = out->l12_pcm_at();
            for (int k = 0; k < n_done; k++) out->add_l12(done[k], nch);
        } else {
            *fresh_state = false;
        }
        r.samples = (int)hd.samples();
        return r;
    }
    Granule gi[4];
    const int begin = read_side_info(sb, gi, hd);
    if (begin < 0 || sb.pos > sb.limit) {
        d.header[0] = 0;                                   // mp3dec_init: the next call resynchronises
        return r;
    }
    // main data = the last `begin` bytes of the reservoir + this frame's payload
    const int payload = (sb.limit - sb.pos) / 8;
    const int have = std::min((int)d.reservoir.size(), begin);
    // (a damaged granule can claim far more big values than it has bits for: the spectrum reader may run up to
    // 288 pairs x 47 bits past the data; it finds zeros there)
    std::vector<uint8_t> md((size_t)have + (size_t)payload + 2048, 0);
    if (have) std::memcpy(md.data(), d.reservoir.data() + (d.reservoir.size() - (size_t)have), (size_t)have);
    std::memcpy(md.data() + have, body + sb.pos / 8, (size_t)payload);
    const bool ok = (int)d.reservoir.size() >= begin;
    const int md_bytes = have + payload;
    int bitpos = 0;
    if (ok) {
        const int ngr = hd.mpeg1() ? 2 : 1, nch = r.channels;
        Work wk;
        std::memset(&wk, 0, sizeof(wk));
        float (&x)[2][576] = wk.x;
        float *scale = wk.scale;
        // Intensity positions persist across the granules of a frame (scfsi copies them, minimp3.d:632) and start from zero
        // in every frame.  The reference keeps them in an uninitialised stack scratch (mp3dec_scratch_t.ist_pos): a stream
        // that reads one it never wrote (scfsi in granule 0; a long right channel under a short left one, whose plan walks
        // 39 bands) gets whatever the stack held there.  Zero is the convention here and in the oracle.
        uint8_t ist[2][40];
        std::memset(ist, 0, sizeof(ist));
        for (int gr = 0; gr < ngr; gr++) {
            std::memset(x, 0, sizeof(wk.x));
            const Granule *g = gi + gr * nch;
            if (out && out->quantised) {
                // ---- quantised records: Huffman values + what the device needs to requantise them ----
                int16_t qv[2][576];
                std::memset(qv, 0, sizeof(qv));
                afg_mp3_qgranule rec;
                std::memset(&rec, 0, sizeof(rec));
                rec.nch = (uint8_t)nch;
                rec.sdesc = AFG_MP3_NO_SDESC;
                uint32_t fl[2] = { 0, 0 };
                int tab[2] = { 0, 0 };
                const QTables &qt = qtables();
                for (int ch = 0; ch < nch; ch++) {
                    const int limit = bitpos + g[ch].part23;
                    Bits sfb(md.data(), md_bytes);
                    sfb.pos = bitpos;
                    band_scales(hd, ist[ch], sfb, g[ch], scale, ch);
                    bitpos = sfb.pos;
                    spectrum(nullptr, md.data(), &bitpos, g[ch], scale, limit, qv[ch]);
                    std::memcpy(rec.scale[ch], scale, sizeof(rec.scale[ch]));
                    tab[ch] = band_table_id(g[ch].bands);
                    const int n_long_bands = (g[ch].mixed ? 2 : 0) << (hd.low_rate_25() ? 1 : 0);
                    int long_lines = 0;
                    for (int b = 0; b < g[ch].n_long; b++) long_lines += g[ch].bands[b];
                    if (tab[ch] < 0 || (g[ch].n_short && long_lines != n_long_bands * 18)) out->q_unsupported = true;
                    rec.table[ch] = (uint8_t)((tab[ch] < 0 ? 0 : tab[ch]) | (g[ch].n_short ? 0x80 : 0));
                    fl[ch] = AFG_MP3_FLAGS(g[ch].block_type, n_long_bands, g[ch].n_short ? n_long_bands - 1 : 31);
                }
                // The reference tests the intensity bit of the header whatever the channel mode (HDR_TEST_I_STEREO,
                // minimp3.d:100): a MONO frame that carries it -- damaged files do -- has L3_intensity_stereo run over its one
                // channel and the scratch row behind it (:1207-1210).  The float front-end does the same; the device
                // requantiser has no such case: the file takes the float path.
                if (nch == 1 && hd.intensity()) out->q_unsupported = true;
                // a line of the right channel is nonzero exactly when its value is and its band's scale is (the
                // requantised magnitude of a nonzero value is at least 1)
                if (hd.intensity()) {
                    StereoPlan plan;
                    const uint8_t *rb = tab[1] >= 0 ? qt.band_of_line[tab[1]] : qt.band_of_line[0];
                    intensity_plan(plan, ist[1], g, hd, [&](int k) { return qv[1][k] != 0 && rec.scale[1][rb[k]] != 0; });
                    afg_mp3_sdesc sd;
                    std::memcpy(sd.type, plan.type, sizeof(sd.type));
                    std::memcpy(sd.fl, plan.fl, sizeof(sd.fl));
                    std::memcpy(sd.fr, plan.fr, sizeof(sd.fr));
                    rec.stereo = 2;
                    rec.sdesc = (uint32_t)out->sdesc.size();
                    out->sdesc.push_back(sd);
                } else if (hd.ms_stereo()) {
                    rec.stereo = 1;
                }
                {   // AFG_MP3_NZ_BANDS: the last line that can be nonzero.  With stereo processing a value of either channel
                    // can make both outputs nonzero; a reordered line stays inside its window group.
                    int top[2] = { -1, -1 };
                    for (int ch = 0; ch < nch; ch++)
                        for (int i = 575; i >= 0; i--)
                            if (qv[ch][i]) { top[ch] = i; break; }
                    if (rec.stereo) top[0] = top[1] = std::max(top[0], top[1]);
                    for (int ch = 0; ch < nch; ch++) {
                        int last = top[ch];
                        if (last >= 0 && g[ch].n_short) last = qt.group_end[tab[ch] < 0 ? 0 : tab[ch]][last] - 1;
                        fl[ch] |= AFG_MP3_NZ_BANDS((last + 18) / 18);
                    }
                }
                if (*fresh_state || out->run_granules.empty()) {
                    if (out->run_granules.empty()) out->continues_previous = !*fresh_state;
                    out->run_granules.push_back(0);
                    *fresh_state = false;
                }
                out->run_granules.back()++;
                out->push_q(qv, nch, rec, fl);
                continue;
            }
            for (int ch = 0; ch < nch; ch++) {
                const int limit = bitpos + g[ch].part23;
                Bits sfb(md.data(), md_bytes);
                sfb.pos = bitpos;
                band_scales(hd, ist[ch], sfb, g[ch], scale, ch);
                bitpos = sfb.pos;
                spectrum(x[ch], md.data(), &bitpos, g[ch], scale, limit);
            }
            if (hd.intensity()) intensity(x[0], ist[1], g, hd);
            else if (hd.ms_stereo()) mid_side(x[0], 576);
            uint32_t fl[2] = { 0, 0 };
            for (int ch = 0; ch < nch; ch++) {
                int aa = 31;
                const int n_long_bands = (g[ch].mixed ? 2 : 0) << (hd.low_rate_25() ? 1 : 0);
                if (g[ch].n_short) {
                    aa = n_long_bands - 1;
                    interleave_windows(x[ch] + n_long_bands * 18, g[ch].bands + g[ch].n_long);
                }
                fl[ch] = AFG_MP3_FLAGS(g[ch].block_type, n_long_bands, aa);
            }
            if (out) {
                if (*fresh_state || out->run_granules.empty()) {
                    if (out->run_granules.empty()) out->continues_previous = !*fresh_state;
                    out->run_granules.push_back(0);
                    *fresh_state = false;
                }
                out->run_granules.back()++;
                for (int ch = 0; ch < nch; ch++) out->push(x[ch], fl[ch]);
            } else {
                *fresh_state = false;
            }
        }
        r.samples = (int)hd.samples();
    }
    {   // L3_save_reservoir: unread main data, at most 511 bytes
        int pos = (bitpos + 7) / 8, remains = md_bytes - pos;
        if (!ok) {                                         // nothing was consumed: everything stays
            pos = 0;
            remains = md_bytes;
        }
        if (remains > kReservoir) {
            pos += remains - kReservoir;
            remains = kReservoir;
        }
        std::vector<uint8_t> keep;
        if (remains > 0) keep.assign(md.begin() + pos, md.begin() + pos + remains);
        d.reservoir.swap(keep);
    }
    return r;
}

}  // namespace

const QTables &qtables()
{
    static const QTables tabs = [] {
        QTables t;
        std::memset(&t, 0, sizeof(t));
        const uint8_t *bases[3] = { &k_sfb_long[0][0], &k_sfb_short[0][0], &k_sfb_mixed[0][0] };
        const size_t row_bytes[3] = { sizeof(k_sfb_long[0]), sizeof(k_sfb_short[0]), sizeof(k_sfb_mixed[0]) };
        for (int kind = 0; kind < 3; kind++)
            for (int row = 0; row < 8; row++) {
                const uint8_t *bands = bases[kind] + (size_t)row * row_bytes[kind];
                uint8_t *bol = t.band_of_line[kind * 8 + row];
                uint16_t *dst = t.dst_of_src[kind * 8 + row];
                uint16_t *gend = t.group_end[kind * 8 + row];
                for (int i = 0; i < 576; i++) { dst[i] = (uint16_t)i; gend[i] = (uint16_t)(i + 1); }
                int at = 0, nb = 0;
                for (; bands[nb] && at < 576; nb++)
                    for (int k = 0; k < bands[nb] && at < 576; k++) bol[at++] = (uint8_t)nb;
                // the short part: groups of three equal widths (one per window), as L3_reorder walks them.  Long tables have
                // none; short tables are all groups; mixed tables start their groups after the long bands (8 at MPEG-1
                // rates, 6 below: L3_read_side_info; rows 5..7 of the tables are the MPEG-1 rates)
                const int first = kind == 0 ? nb : kind == 1 ? 0 : (row >= 5 ? 8 : 6);
                int s0 = 0;
                for (int b = 0; b < first && b < nb; b++) s0 += bands[b];
                for (int b = first; b + 2 < nb + 0 && s0 < 576; b += 3) {
                    const int len = bands[b];
                    for (int w = 0; w < 3; w++)
                        for (int k = 0; k < len; k++)
                            if (s0 + w * len + k < 576 && s0 + 3 * k + w < 576) {
                                dst[s0 + w * len + k] = (uint16_t)(s0 + 3 * k + w);
                                gend[s0 + w * len + k] = (uint16_t)std::min(576, s0 + 3 * len);
                            }
                    s0 += 3 * len;
                }
            }
        for (int i = 0; i < 145; i++) t.pow43[i] = bits_f32(k_pow43_bits[i]);
        return t;
    }();
    return tabs;
}

bool looks_like_mp3(const uint8_t *data, size_t size)
{
    if (!data || size < 10) return false;
    if (!std::memcmp(data, "RIFF", 4) || !std::memcmp(data, "RF64", 4) || !std::memcmp(data, "OggS", 4) ||
        !std::memcmp(data, "fLaC", 4) || !std::memcmp(data, "qoaf", 4))
        return false;                                      // formats the reference probes before MP3 (stream.d:1586-1706)
    size_t n = size;
    const size_t id3 = id3v2_size(data, n);
    if (id3 >= n) return false;
    const uint8_t *p = data + id3;
    n -= id3;
    trim_trailing_tags(p, &n);
    int fb = 0, sz = 0;
    next_frame(p, (int)std::min<size_t>(n, 0x7fffffff), &fb, &sz);
    return sz != 0;
}

size_t max_blocks(const uint8_t *data, size_t size)
{
    if (!data || size < 10) return 0;
    const uint8_t *buf = data;
    size_t n = size;
    size_t id3 = id3v2_size(buf, n);
    if (id3) {
        id3 = std::min(id3, n);
        buf += id3;
        n -= id3;
    }
    trim_trailing_tags(buf, &n);
    // every frame the decoder can accept starts at a position that holds a valid header, and frames do not
    // overlap: the blocks of ALL such positions bound the result (false positives are ~1 per 3 KB of noise)
    size_t blocks = 0;
    for (size_t at = 0; at + 4 <= n; at++) {
        if (buf[at] != 0xff) continue;
        const Header hd{ buf + at };
        // (Layer III: 2 or 1 granules; Layer II: 36 slots = 2 blocks; Layer I: 12 slots, at most one block with a run's padding)
        if (hd.valid()) blocks += (size_t)(hd.layer_code() == 1 ? (hd.mpeg1() ? 2 : 1) : hd.layer1() ? 1 : 2) * (hd.mono() ? 1 : 2);
    }
    return blocks + 8;
}

bool parse_file_into(const uint8_t *data, size_t size, File &f, float *coef, uint32_t *flags, size_t cap);

// mp3dec_ex_open (index scan or Xing/Info tag) followed by mp3dec_ex_read to the end of the stream
bool parse_file(const uint8_t *data, size_t size, File &f) { return parse_file_into(data, size, f, nullptr, nullptr, 0); }

namespace {

// mp3dec_ex_open (index scan or Xing/Info tag) followed by mp3dec_ex_read, resumable: `open` does the scan, `run` decodes
// frames until the stream ends or `max_frames` have been consumed.  A whole file (parse_file_into) and a chunked
// stream (Reader) run the same code.
struct Walk {
    std::vector<uint8_t> padded;
    const uint8_t *buf = nullptr;
    size_t n = 0;
    Decoder dec;
    uint64_t start = 0, declared = 0, detected = 0;
    int to_skip = 0;
    bool tagged = false;
    int ch0 = 0, hz0 = 0, layer0 = 0;
    bool fresh = true, done = false;
    uint64_t off = 0, cur = 0;

    bool open(const uint8_t *data, size_t size)
    {
        if (!data || size < 10) return false;
        const uint8_t *src = data;
        n = size;
        {
            size_t id3 = id3v2_size(src, n);
            if (id3) {
                id3 = std::min(id3, n);
                src += id3;
                n -= id3;
            }
            trim_trailing_tags(src, &n);
        }
        if (!n) return false;
        // keep 16 readable bytes behind the data for the windowed readers
        padded.assign(n + 16, 0);
        std::memcpy(padded.data(), src, n);
        buf = padded.data();

        int counted = 0, probe = 0;
        bool have = false;
        {
            const uint8_t *p = buf;
            size_t left = n;
            for (;;) {
                int fb = 0, sz = 0;
                const int i = next_frame(p, (int)std::min<size_t>(left, 0x7fffffff), &fb, &sz);
                p += i;
                left -= (size_t)i;
                if (i && !sz) continue;
                if (!sz) break;
                const Header hd{ p };
                const int nch = hd.mono() ? 1 : 2;
                const uint64_t at = (uint64_t)(p - buf);
                if (!have) {
                    have = true;
                    ch0 = nch;
                    hz0 = (int)hd.hz();
                    layer0 = 4 - hd.layer_code();
                    start = at;
                    if (layer0 == 3) {
                        uint32_t frames = 0;
                        int delay = 0, padding = 0;
                        const int t = info_tag(p, sz, &frames, &delay, &padding);
                        if (t) start = at + (uint64_t)sz;
                        if (t > 0) {
                            padding *= nch;
                            to_skip = delay * nch;
                            declared = (uint64_t)hd.samples() * (uint64_t)nch * frames;
                            if (declared >= (uint64_t)to_skip) declared -= (uint64_t)to_skip;
                            if (padding > 0 && declared >= (uint64_t)padding) declared -= (uint64_t)padding;
                            detected = declared;
                            tagged = true;
                            break;
                        }
                        if (t < 0) {
                            p += sz;
                            left -= (size_t)sz;
                            continue;
                        }
                    }
                }
                counted++;
                if (!probe && counted < 256) {                 // early frames may lack their reservoir: count what decodes
                    bool fr_fresh = false;
                    const FrameResult fr = frame(dec, p, (int)std::min<size_t>(left, 0x7fffffff), nullptr, &fr_fresh);
                    probe = fr.stop ? 0 : fr.samples;
                    declared += (uint64_t)probe * (uint64_t)nch;
                } else {
                    declared += (uint64_t)hd.samples() * (uint64_t)nch;
                }
                p += sz;
                left -= (size_t)sz;
            }
        }
        if (!have) return false;
        dec.reset();
        fresh = true;
        off = start;
        cur = 0;
        done = false;
        return true;
    }

    void describe(File &f) const
    {
        f.channels = ch0;
        f.hz = hz0;
        f.layer = layer0;
        f.tagged = tagged;
        f.start_delay = to_skip;
        f.detected_samples = detected;
        f.declared_samples = declared;
    }

    // frames into `f` until the stream ends or max_frames were consumed; copies are relative to f's own blocks
    void run(File &f, int max_frames)
    {
        uint64_t chunk_samples = 0;
        // (a Layer I chunk goes on until its slots fill whole blocks: three frames make two)
        for (int k = 0; (k < max_frames || f.l12_slots) && !done; k++) {
            if (detected && cur >= detected) { done = true; break; }
            const uint64_t left = n - off;
            if (!left) { done = true; break; }
            const uint64_t blocks_before = f.blocks();
            const std::vector<uint32_t> runs_before = f.run_granules;
            const int slots_before = f.l12_slots;
            float acc_before[2][576];
            if (slots_before) std::memcpy(acc_before, f.l12_acc, sizeof(acc_before));
            const FrameResult fr = frame(dec, buf + off, (int)std::min<uint64_t>(left, 0x7fffffff), &f, &fresh);
            if (fr.stop || fr.hz != hz0 || fr.layer != layer0 || fr.channels != ch0) {
                // MP3D_E_DECODE (minimp3_ex.d:851-857; also what "no further frame" turns into, since the frame info
                // stays zero then): the stream ends here; records of this frame are dropped
                f.truncate(blocks_before);
                f.run_granules = runs_before;
                f.l12_slots = slots_before;
                if (slots_before) std::memcpy(f.l12_acc, acc_before, sizeof(acc_before));
                done = true;
                break;
            }
            if (fr.samples) {
                const int total = fr.samples * fr.channels;
                int used = 0;
                if (to_skip) {
                    used = std::min(total, to_skip);
                    to_skip -= used;
                }
                uint64_t take = (uint64_t)(total - used);
                if (detected && cur + take >= detected) take = detected - cur;
                const uint64_t first = fr.layer != 3 ? fr.pcm_at : blocks_before * 576;
                if (take) f.copies.push_back(Copy{ first + (uint64_t)used, take });
                cur += take;
                chunk_samples += take;
            } else if (to_skip) {
                const int fs = (int)Header{ buf + off }.samples() * fr.channels;
                to_skip -= std::min(fs, to_skip);
            }
            off += (uint64_t)fr.consumed;
        }
        if (done) f.flush_l12(ch0);
        f.pcm_samples += chunk_samples;
        // trailing granules that no copy refers to are still part of their run (the device needs whole runs)
    }
};

}  // namespace

bool parse_file_into(const uint8_t *data, size_t size, File &f, float *coef_dst, uint32_t *flags_dst, size_t cap)
{
    const bool quantised = f.quantised;                    // the caller's choices survive the reset
    int16_t *const ext_q = f.ext_q;
    afg_mp3_qgranule *const ext_qgr = f.ext_qgr;
    f = File();
    f.quantised = quantised;
    f.ext_q = ext_q;
    f.ext_qgr = ext_qgr;
    f.ext_coef = coef_dst;
    f.ext_flags = flags_dst;
    f.ext_cap = cap;
    Walk w;
    if (!w.open(data, size)) return false;
    w.describe(f);
    while (!w.done) w.run(f, 1 << 20);
    f.pcm_samples = w.cur;
    return true;
}

// ---- chunked reading (the AudioStream surface decodes as the caller pulls, stream.d:429-637, minimp3_ex.d:787-888) ----
struct Reader::Impl {
    Walk w;
};

Reader::Reader() : p(new Impl) {}
Reader::~Reader() { delete p; }

bool Reader::open(const uint8_t *data, size_t size, File &meta)
{
    *p = Impl();
    if (!p->w.open(data, size)) return false;
    meta = File();
    p->w.describe(meta);
    return true;
}

bool Reader::more(File &out, int max_frames, bool *continues)
{
    out = File();
    p->w.describe(out);
    if (p->w.done) return false;
    p->w.run(out, max_frames);
    // does the first run of this chunk go on from the previous chunk's state?  Known only once its first frame has been
    // looked at: a frame that failed at the end of the previous chunk resets the decoder at the next call (minimp3.d:1507)
    *continues = out.continues_previous;
    return out.blocks() != 0 || !p->w.done;
}

}  // namespace afg_mp3
