// afg_flac_front.cpp -- host front-ends for native FLAC files and QOA files.
//
//   FLAC  container + frame/subframe headers + Rice residuals   (reference drflac.d:680-1695,
//         :1887-2153; the prediction half of drflac.d:1235 is NOT done here: residuals and
//         subframe parameters become afg_flac_subframe / afg_flac_frame records)
//   QOA   file/frame headers only (reference qoa.d:413-486): the device reads the raw bytes
#include "afg_flac_front.h"

#include <algorithm>
#include <cstring>

namespace afg_front {
namespace {

// ---------------------------------------------------------------------------------------------
// MSB-first bit reader over memory (the role of drflac_bs, drflac.d:680-1002)
// ---------------------------------------------------------------------------------------------
struct BitReader {
    const uint8_t *p;
    size_t nbits, pos = 0;
    bool fail = false;
    size_t real_bytes;             // bytes that exist behind p (nbits can come to lie past them: rice_low)
    unsigned phase;                // bits of a 32-bit line in front of p: the reference's reader takes the stream in lines
                                   // counted from the first frame (drflac.d:707-780)
    bool phantom = false;
    BitReader(const uint8_t *data, size_t bytes, size_t lead_bytes = 0)
        : p(data), nbits(bytes * 8), real_bytes(bytes), phase((unsigned)(lead_bytes & 3) * 8) {}
    uint32_t at(size_t i) const { return i < real_bytes ? p[i] : 0u; }
    uint32_t bit()
    {
        if (pos >= nbits) { fail = true; return 0; }
        uint32_t b = (at(pos >> 3) >> (7 - (pos & 7))) & 1u;
        pos++;
        return b;
    }
    uint64_t bits(unsigned n)      // n <= 57
    {
        if (n == 0) return 0;
        if (pos + n > nbits) {
            // drflac__read_uint32 (drflac.d:834-856): a read that runs from a whole line into the partial last line (1-3 bytes)
            // adds what it takes there to the line's count of consumed bits, which started at 32 minus the line's own -- when
            // it takes more than the line holds the count passes 32, "bits remaining" (an unsigned difference) becomes
            // enormous, and every later read of this kind succeeds with zeros.  The frame in hand is delivered, the sync
            // code of the next one is not found.  (A stream that ends on a line boundary, or a read that starts inside the
            // last line, fails as one expects.)
            const size_t end_all = real_bytes * 8 + phase, last = end_all & ~(size_t)31;
            if (!phantom && nbits == real_bytes * 8 && (end_all & 31) != 0 && pos + phase < last && n <= 32) nbits = (size_t)1 << 62;
            else { fail = true; pos = nbits; return 0; }
        }
        uint64_t v = 0;
        size_t byte = pos >> 3;
        unsigned have = 0;
        uint64_t acc = 0;
        unsigned skip = (unsigned)(pos & 7);
        while (have < n + skip) { acc = (acc << 8) | at(byte++); have += 8; }
        v = (acc >> (have - n - skip)) & ((n == 64) ? ~0ull : ((1ull << n) - 1));
        pos += n;
        return v;
    }
    // The low k bits of a Rice symbol whose stop bit has just been read, as the reference's fused loop gets them
    // (drflac.d:1166-1236): when the symbol reaches the end of the 32-bit line its stop bit sits in, the loop fetches the NEXT
    // line -- even for zero bits of it ("riceLength < bits remaining", else the straddling branch).  Two things follow at the
    // end of a stream.  A symbol that ends on the last bit of the data fails (there is no next line).  And when the next line
    // is the partial last one (1-3 bytes), that branch overwrites the count of bits the line holds with the count it has
    // just taken from it: the line is a whole one from then on, its missing bytes zeros.  Files cut short get there.
    bool rice_low(unsigned k, uint32_t &low)
    {
        const size_t s = pos - 1 + phase, end_all = nbits + phase;     // stop bit and end of the stream, in line coordinates
        size_t le = ((s >> 5) + 1) << 5;                                // end of the stop bit's line
        if (le > end_all) le = end_all;
        if (s + 1 + k >= le) {
            if (le >= end_all) { fail = true; pos = nbits; low = 0; return false; }
            if (!phantom && le + 32 > real_bytes * 8 + phase) { phantom = true; nbits = le + 32 - phase; }
        }
        low = (uint32_t)bits(k);
        return !fail;
    }
    int64_t sbits(unsigned n)      // two's complement, drflac__read_int32 (drflac.d:858-870)
    {
        uint64_t v = bits(n);
        if (n == 0) return 0;
        uint64_t sign = 1ull << (n - 1);
        return (int64_t)((v ^ sign)) - (int64_t)sign;
    }
    bool unary(uint32_t &zeros)    // counts zeros up to and including the terminating one
    {
        zeros = 0;
        for (;;) {
            if (pos >= nbits || (pos >> 3) >= real_bytes + 4) { fail = true; return false; }      // (zeros without end: no set bit will come)
            unsigned skip = (unsigned)(pos & 7);
            uint32_t byte = (uint32_t)(at(pos >> 3) << skip) & 0xffu;
            if (byte) {
                unsigned lz = (unsigned)__builtin_clz(byte) - 24;
                zeros += lz;
                pos += lz + 1;
                return true;
            }
            zeros += 8 - skip;
            pos += 8 - skip;
        }
    }
    void align() { pos = (pos + 7) & ~(size_t)7; }
    size_t byte_pos() const { return pos >> 3; }
};

// ---------------------------------------------------------------------------------------------
// FLAC: native container -> records
// ---------------------------------------------------------------------------------------------

// STREAMINFO + metadata walk, drflac.d:1901-1931, :1933-2118 (ID3-prefixed FLAC is not supported by
// the reference either, drflac.d:12-13)
bool flac_open(const uint8_t *d, size_t n, FlacInfo &fi)
{
    if (n < 4 + 4 + 34 || d[0] != 'f' || d[1] != 'L' || d[2] != 'a' || d[3] != 'C') return false;
    size_t pos = 4;
    bool got = false;
    for (;;) {
        if (pos + 4 > n) return false;
        const bool last = (d[pos] & 0x80) != 0;
        const unsigned type = d[pos] & 0x7f;
        const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
        pos += 4;
        if (pos == 8 && (type != 0 || len != 34)) return false;    // the first block is STREAMINFO, 34 bytes (drflac.d:2139-2141)
        if (type == 0 && pos == 8) {
            BitReader br(d + pos, len);
            br.bits(16);                               // min block size
            fi.max_block = (uint32_t)br.bits(16);
            br.bits(24); br.bits(24);                  // min/max frame size
            fi.sample_rate = (uint32_t)br.bits(20);
            fi.channels = (uint32_t)br.bits(3) + 1;
            fi.bps = (uint32_t)br.bits(5) + 1;
            fi.total_samples = br.bits(36);
            got = true;
        }
        // The blocks behind STREAMINFO are skipped by a seek that never fails and stops at the end of the data
        // (drflac.d:2103-2106, stream.d:2227-2239): a block longer than the file is no error -- if it is the last one the
        // stream opens and holds no frames (a STREAMINFO whose last-block flag a damaged byte cleared makes the first
        // frame's bytes such a block).
        pos = len > n - pos ? n : pos + len;
        if (last) break;
    }
    fi.first_frame = pos;
    // (STREAMINFO's rate and sample size are not validated by the reference: a damaged 0 Hz or 3-bit claim opens, frames
    //  decode by their own headers and drflac_read_s32 shifts by 32 minus the claim, :2883)
    return got;
}


// drflac__read_utf8_coded_number, drflac.d:1005-1043
bool read_utf8(BitReader &br, uint64_t &out)
{
    uint32_t b0 = (uint32_t)br.bits(8);
    if (br.fail) return false;
    int extra;
    if ((b0 & 0x80) == 0) { out = b0; return true; }
    else if ((b0 & 0xE0) == 0xC0) { extra = 1; out = b0 & 0x1F; }
    else if ((b0 & 0xF0) == 0xE0) { extra = 2; out = b0 & 0x0F; }
    else if ((b0 & 0xF8) == 0xF0) { extra = 3; out = b0 & 0x07; }
    else if ((b0 & 0xFC) == 0xF8) { extra = 4; out = b0 & 0x03; }
    else if ((b0 & 0xFE) == 0xFC) { extra = 5; out = b0 & 0x01; }
    else if (b0 == 0xFE) { extra = 6; out = 0; }
    else return false;
    for (int i = 0; i < extra; i++) {
        uint32_t b = (uint32_t)br.bits(8);
        if (br.fail) return false;                                  // (continuation bytes are not validated: drflac.d:1033-1039)
        out = (out << 6) | (b & 0x3F);
    }
    return true;
}

// residual of one subframe into dst[order .. block_size), drflac.d:1279-1328 (+ Rice :1166-1224)
// (two builds of this one function, chosen once at load: lzcnt / shlx / movbe shorten the loop by a quarter)
#if defined(__x86_64__) && defined(__clang__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("default", "arch=x86-64-v3")))
#endif
bool flac_residual(BitReader &br, uint32_t block_size, uint32_t order, int32_t *dst)
{
    const unsigned method = (unsigned)br.bits(2);
    if (method > 1) return false;                                  // :1287
    const unsigned part_order = (unsigned)br.bits(4);
    const uint32_t nparts = 1u << part_order;
    if ((block_size >> part_order) < order && part_order) return false;
    uint32_t i = order;
    for (uint32_t part = 0; part < nparts; part++) {
        uint32_t count = block_size >> part_order;
        if (part == 0) {
            if (count < order) return false;
            count -= order;                                        // :1295
        }
        unsigned k = (unsigned)br.bits(method == 0 ? 4 : 5);
        // The reference tests the parameter against 16 / 32 (drflac.d:1301, :1304), values a 4- / 5-bit
        // field never takes, so its unencoded-partition branch (:1313-1321) is dead and the FLAC escape
        // codes 15 / 31 are decoded as plain Rice parameters.  Results must be identical to the
        // reference's, so the same happens here (HISTORY.md 4, "FLAC front-end"); flip kSpecEscape to get
        // the format's own behaviour.
        constexpr bool kSpecEscape = false;
        const bool escape = kSpecEscape && ((method == 0 && k == 15) || (method == 1 && k == 31));
        if (br.fail) return false;
        if (!escape) {
            // The hot loop of the whole front-end.  Fast path: a 64-bit big-endian window `w` whose top `avail` bits are the
            // stream at `pos` -- the unary part is a count of leading zeros, the k low bits a shift, and the window moves on
            // by a shift (the dependent chain per symbol is clz + shift, not load + swap + clz); it is topped up from
            // memory whenever eight whole bytes are left at the position.  Anything else -- a symbol that does not end
            // inside the window (a run of 50 zeros), the last bytes of the buffer -- takes the bit reader.  Same values.
            // (the last bytes of the stream go symbol by symbol through BitReader::rice_low: what the reference does there)
            const size_t bytes = br.real_bytes > 12 ? br.real_bytes - 12 : 0;
            size_t pos = br.pos;
            uint32_t j = 0;
            // Groups first: one load of eight bytes (at least 57 stream bits) feeds `per` symbols -- as many as fit nearly always
            // at this parameter (a symbol is k + 1 bits and a quotient that averages one to two) -- so the load, the byte
            // swap and the shift by the bit offset are paid once per group and the chain from symbol to symbol is a count of
            // leading zeros and a shift.  The trip count is fixed per partition (a predictable loop); a symbol that does
            // not end inside the window, which is rare, goes through the bit reader and the groups resume behind it.
            {
                const unsigned per = k <= 5 ? 6u : 57u / (k + 4);
                while (j + per <= count) {
                    const size_t bp = pos >> 3;
                    if (bp + 8 > bytes) break;
                    const unsigned s = (unsigned)(pos & 7);
                    uint64_t x;
                    std::memcpy(&x, br.p + bp, 8);
                    uint64_t w = __builtin_bswap64(x) << s;
                    unsigned left = 64 - s;
                    int32_t *const o = dst + i;
                    unsigned n = 0;
                    for (; n < per; n++) {
                        if (!w) break;
                        const unsigned lz = (unsigned)__builtin_clzll(w);
                        const unsigned need = lz + 1 + k;
                        if (need > left) break;
                        const uint32_t low = k ? (uint32_t)((w << (lz + 1)) >> (64 - k)) : 0u;      // (lz + 1 <= 63 when k > 0)
                        w = need < 64 ? w << need : 0;
                        left -= need;
                        const uint32_t v = (lz << k) | low;
                        o[n] = (int32_t)((v >> 1) ^ (~(v & 1) + 1));                              // zig-zag, :1224
                    }
                    pos += (64 - s) - left;
                    i += n;
                    j += n;
                    if (n < per) {
                        uint32_t q, low;
                        br.pos = pos;
                        if (!br.unary(q) || !br.rice_low(k, low)) return false;
                        pos = br.pos;
                        const uint32_t v = (q << k) | low;
                        dst[i++] = (int32_t)((v >> 1) ^ (~(v & 1) + 1));
                        j++;
                    }
                }
            }
            uint64_t w = 0;
            unsigned avail = 0;
            for (; j < count; j++) {
                if (avail < 48) {
                    const size_t bp = pos >> 3;
                    const unsigned s = (unsigned)(pos & 7);
                    w = 0;
                    avail = 0;
                    if (bp + 8 <= bytes) {
                        uint64_t x;
                        std::memcpy(&x, br.p + bp, 8);
                        w = __builtin_bswap64(x) << s;
                        avail = 64 - s;
                    }
                }
                uint32_t q, low;
                const unsigned lz = w ? (unsigned)__builtin_clzll(w) : 64u;
                const unsigned need = lz + 1 + k;
                if (need <= avail) {
                    q = lz;
                    low = k ? (uint32_t)((w << (lz + 1)) >> (64 - k)) : 0u;     // (lz + 1 <= 63 when k > 0)
                    w = need < 64 ? w << need : 0;
                    avail -= need;
                    pos += need;
                } else {
                    br.pos = pos;
                    if (!br.unary(q) || !br.rice_low(k, low)) return false;
                    pos = br.pos;
                    avail = 0;
                }
                const uint32_t v = (q << k) | low;
                dst[i++] = (int32_t)((v >> 1) ^ (~(v & 1) + 1));   // zig-zag, :1224
            }
            br.pos = pos;
        } else {
            const unsigned raw = (unsigned)br.bits(5);
            for (uint32_t j = 0; j < count; j++) dst[i++] = (int32_t)br.sbits(raw);
        }
        if (br.fail) return false;
    }
    return i == block_size;
}

const int16_t kFixedCoef[5][4] = { { 0, 0, 0, 0 }, { 1, 0, 0, 0 }, { 2, -1, 0, 0 }, { 3, -3, 1, 0 }, { 4, -6, 4, -1 } };   // :1397-1403

// one frame: header (drflac.d:1444-1528), subframes (:1530-1599), padding + CRC-16 (:1673)
bool flac_frame(BitReader &br, const FlacInfo &fi, FlacRecords &rec)
{
    static const uint32_t bpsTable[8] = { 0, 8, 12, 255, 16, 20, 24, 255 };
    if (br.bits(14) != 0x3FFE || br.fail) return false;
    br.bits(1);
    br.bits(1);                                                    // blocking strategy: number is skipped either way
    const unsigned bsCode = (unsigned)br.bits(4), srCode = (unsigned)br.bits(4);
    const unsigned asg = (unsigned)br.bits(4), bpsCode = (unsigned)br.bits(3);
    br.bits(1);
    uint64_t number;
    if (!read_utf8(br, number)) return false;
    uint32_t bs;
    if (bsCode == 0) return false;                                 // reserved (undefined shift in the reference)
    else if (bsCode == 1) bs = 192;
    else if (bsCode <= 5) bs = 576u << (bsCode - 2);
    else if (bsCode == 6) bs = (uint32_t)br.bits(8) + 1;
    else if (bsCode == 7) bs = (uint32_t)br.bits(16) + 1;
    else bs = 256u << (bsCode - 8);
    if (srCode == 12) br.bits(8);
    else if (srCode == 13 || srCode == 14) br.bits(16);
    else if (srCode == 15) return false;
    uint32_t bps = bpsTable[bpsCode];
    if (bps == 255) return false;                                  // reserved
    if (bps == 0) bps = fi.bps;
    br.bits(8);                                                    // CRC-8 (stored, unused: :1450)
    if (br.fail) return false;
    uint32_t C;
    if (asg <= 7) C = asg + 1;
    else if (asg <= 10) C = 2;
    else return false;
    // The reference decodes a frame with the channel count of ITS header into a buffer sized for STREAMINFO's
    // (drflac.d:1658-1662, :2594): fewer channels than the stream's are delivered as they come, more -- or a block
    // longer than the declared maximum -- overrun that buffer, which ends the stream here.
    if (bs == 0 || bs > 65535 || (uint64_t)bs * C > (uint64_t)fi.max_block * fi.channels) return false;

    afg_flac_frame fr;
    std::memset(&fr, 0, sizeof(fr));
    fr.in_off = rec.res_size();
    fr.out_off = rec.out_samples;
    fr.block_size = bs;
    fr.sf_index = (uint32_t)rec.subframes.size();
    fr.channels = (uint8_t)C;
    fr.assignment = (uint8_t)(asg <= 7 ? AFG_FLAC_INDEPENDENT : asg);
    fr.bps = (uint8_t)fi.bps;                                      // drflac_read_s32 shifts by 32 - STREAMINFO bps (:2883)
    int32_t *const plane = rec.res_grow((size_t)bs * C);
    if (!plane) return false;
    for (uint32_t c = 0; c < C; c++) {
        int32_t *dst = plane + (size_t)c * bs;
        afg_flac_subframe sf;
        std::memset(&sf, 0, sizeof(sf));
        const unsigned hdr = (unsigned)br.bits(8);                 // :1530-1569
        if (br.fail || (hdr & 0x80)) return false;
        const unsigned type = (hdr & 0x7E) >> 1;
        unsigned wasted = 0;
        if (hdr & 1) {
            uint32_t z;
            if (!br.unary(z)) return false;
            wasted = (uint8_t)((uint8_t)z + 1);                    // (the reference keeps the count in a byte: 276 zeros are 21 wasted bits, :1563)
        }
        unsigned sbps = bps;                                       // side channels carry one extra bit, :1578-1585
        if ((asg == AFG_FLAC_LEFT_SIDE || asg == AFG_FLAC_MID_SIDE) && c == 1) sbps++;
        else if (asg == AFG_FLAC_RIGHT_SIDE && c == 0) sbps++;
        if (wasted >= sbps) return false;
        sbps -= wasted;
        sf.wasted = (uint8_t)wasted;
        sf.use64 = sbps > 16;                                      // :1308
        if (type == 0) {                                           // constant, :1375-1385
            const int32_t v = (int32_t)br.sbits(sbps);
            std::fill(dst, dst + bs, v);
        } else if (type == 1) {                                    // verbatim, :1387-1394
            for (uint32_t i = 0; i < bs; i++) dst[i] = (int32_t)br.sbits(sbps);
        } else if (type & 0x20) {                                  // LPC, :1417-1441
            const unsigned order = (type & 0x1F) + 1;
            if (order > bs) return false;
            for (unsigned i = 0; i < order; i++) dst[i] = (int32_t)br.sbits(sbps);
            const unsigned prec = (unsigned)br.bits(4);
            if (prec == 15) return false;
            const int shift = (int)br.sbits(5);
            if (shift < 0) return false;                           // undefined in the reference: rejected
            for (unsigned i = 0; i < order; i++) sf.coef[i] = (int16_t)br.sbits(prec + 1);
            sf.order = (uint8_t)order;
            sf.shift = (uint8_t)shift;
            if (!flac_residual(br, bs, order, dst)) return false;
        } else if (type & 0x08) {                                  // fixed, :1396-1415
            const unsigned order = type & 0x07;
            if (order > 4 || order > bs) return false;
            for (unsigned i = 0; i < order; i++) dst[i] = (int32_t)br.sbits(sbps);
            for (unsigned i = 0; i < order; i++) sf.coef[i] = kFixedCoef[order][i];
            sf.order = (uint8_t)order;
            sf.shift = 0;
            if (!flac_residual(br, bs, order, dst)) return false;
        } else {
            return false;                                          // reserved
        }
        if (br.fail) return false;
        rec.subframes.push_back(sf);
    }
    br.align();
    // CRC-16, not verified (drflac.d:108, :1673).  At the very end of the data the reference does not even need it to
    // be there: its drflac__seek_bits hands the whole bytes it cannot find in its caches to the client's seek
    // (drflac.d:812-819), which AudioStream's callback answers with success at any offset (stream.d:2227-2239), so a
    // last frame whose CRC-16 is cut short -- or missing -- is delivered like any other (oracle/flac_frontend.c:
    // seek_bits; tests/test_oracle_flac_frontend.py::test_last_frame_without_its_crc).
    if (br.nbits - br.pos < 16) br.pos = br.nbits;
    else br.bits(16);
    if (br.fail) return false;
    if (rec.pack16 && bs >= 8 && (fr.in_off & 3) == 0) {
        uint32_t seen = 0;
        const size_t n = (size_t)bs * C;
        for (size_t i = 0; i < n; i++) seen |= (uint32_t)(plane[i] ^ (plane[i] >> 31));
        if (seen < 32768u) {
            // forward in place: the int16 rows (padded to 8) never overtake the int32 words still to be read (bs >= 8)
            // (byte-wise copies: the int16 stores overlap int32 words of the same buffer, which plain typed accesses
            // would leave to the optimiser's no-alias assumptions)
            unsigned char *rows = (unsigned char *)plane;
            const size_t row = (size_t)AFG_FLAC_ROW16(bs);
            for (uint32_t c = 0; c < C; c++) {
                const unsigned char *src = (const unsigned char *)(plane + (size_t)c * bs);
                unsigned char *dst = rows + (size_t)c * row * sizeof(int16_t);
                for (uint32_t i = 0; i < bs; i++) {
                    int32_t w;
                    std::memcpy(&w, src + (size_t)i * sizeof(int32_t), sizeof(w));
                    const int16_t h = (int16_t)w;
                    std::memcpy(dst + (size_t)i * sizeof(int16_t), &h, sizeof(h));
                }
                std::memset(dst + (size_t)bs * sizeof(int16_t), 0, (row - bs) * sizeof(int16_t));
            }
            fr.res16 = 1;
            fr.in_off *= 2;
        }
    }
    rec.frames.push_back(fr);
    rec.out_samples += (uint64_t)bs * C;
    return true;
}

// ---------------------------------------------------------------------------------------------
// QOA: locate frames (qoa.d:413-486); everything else happens on the device
// ---------------------------------------------------------------------------------------------

uint64_t be64(const uint8_t *p)
{
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}

}  // namespace

// whole file -> records; stops at the first frame that does not parse (the reference's read loop
// does the same: drflac.d:2860)
bool flac_parse(const uint8_t *d, size_t n, FlacInfo &fi, FlacRecords &rec) { return flac_parse_into(d, n, fi, rec, nullptr, 0); }

size_t flac_res_bound(const uint8_t *d, size_t n)
{
    FlacInfo fi;
    if (!flac_open(d, n, fi) || !fi.total_samples || !fi.channels) return 0;
    // STREAMINFO's length is only a claim: the staged plane is also bounded by what the bytes can hold.  A frame takes
    // at least 10 bytes (sync + header + CRC-8, one byte of subframe header and one of payload per channel, CRC-16) and
    // carries at most max_block samples per channel, so a 42-byte file declaring 2^36 samples reserves a few blocks,
    // not terabytes of page-locked memory.  Past the hard cap the file takes the grow-as-parsed path instead.
    const uint64_t block = fi.max_block ? fi.max_block : 65535;
    const uint64_t by_bytes = ((uint64_t)n / 10 + 1) * block;
    const uint64_t frames = std::min<uint64_t>(fi.total_samples + block, by_bytes);
    const uint64_t words = frames * fi.channels;
    return words > ((uint64_t)1 << 31) ? 0 : (size_t)words;
}

bool flac_open_info(const uint8_t *d, size_t n, FlacInfo &fi) { return flac_open(d, n, fi); }

int flac_parse_frames(const uint8_t *d, size_t n, const FlacInfo &fi, FlacRecords &rec, size_t *pos, int max_frames, bool *ended)
{
    *ended = false;
    if (fi.first_frame + *pos >= n) { *ended = true; return 0; }
    const size_t avail = n - fi.first_frame - *pos;
    BitReader br(d + fi.first_frame + *pos, avail, *pos);
    int got = 0;
    while (got < max_frames) {
        if (br.byte_pos() + 2 >= avail) { *ended = true; break; }
        const size_t keep_f = rec.frames.size(), keep_s = rec.subframes.size(), keep_r = rec.res_size();
        const size_t before = br.byte_pos();
        if (!flac_frame(br, fi, rec)) {                        // the stream ends at the first frame that does not parse
            rec.frames.resize(keep_f);
            rec.subframes.resize(keep_s);
            rec.res_truncate(keep_r);
            *ended = true;
            *pos += before;
            return got;
        }
        got++;
    }
    *pos += br.byte_pos();
    return got;
}

bool flac_parse_into(const uint8_t *d, size_t n, FlacInfo &fi, FlacRecords &rec, int32_t *res_dst, size_t cap)
{
    if (!flac_open(d, n, fi)) return false;
    rec.ext_res = res_dst;
    rec.ext_cap = cap;
    BitReader br(d + fi.first_frame, n - fi.first_frame);
    while (br.byte_pos() + 2 < n - fi.first_frame) {
        const size_t keep_f = rec.frames.size(), keep_s = rec.subframes.size(), keep_r = rec.res_size();
        if (!flac_frame(br, fi, rec)) {
            rec.frames.resize(keep_f);
            rec.subframes.resize(keep_s);
            rec.res_truncate(keep_r);
            break;
        }
    }
    return true;
}

bool qoa_parse(const uint8_t *d, size_t n, QoaInfo &qi, std::vector<afg_qoa_frame> &frames)
{
    if (n < 16) return false;                                      // QOA_MIN_FILESIZE
    const uint64_t fh = be64(d);
    if ((fh >> 32) != 0x716f6166u) return false;                   // 'qoaf'
    qi.samples = (uint32_t)(fh & 0xffffffffu);
    if (!qi.samples) return false;
    const uint64_t first = be64(d + 8);
    qi.channels = (uint32_t)((first >> 56) & 0xff);
    qi.samplerate = (uint32_t)((first >> 32) & 0xffffff);
    if (qi.channels == 0 || qi.channels > 8 || qi.samplerate == 0) return false;
    size_t pos = 8;
    uint64_t out = 0;
    // qoa_decode_frame as the READER it is (qoa.d:455-534): header, LMS state and ceil(samples / 20) slices per channel are
    // consumed from the cursor, and the next frame starts where that ends -- the frame-size field is only compared with
    // the bytes left (:477) and with the sample count (:481-486), it positions nothing.
    while (n - pos >= 8 + 16 * (size_t)qi.channels) {              // :460
        const uint64_t h = be64(d + pos);
        const int ch = (int)((h >> 56) & 0xff), sr = (int)((h >> 32) & 0xffffff);
        const int smp = (int)((h >> 16) & 0xffff), fsz = (int)(h & 0xffff);
        const int num_slices = (fsz - 8 - 16 * ch) / 8;            // (may be negative: then no sample count passes :484)
        if ((long long)(n - pos - 8) < (long long)fsz - 8) break;  // :477
        if (ch != (int)qi.channels || sr != (int)qi.samplerate || smp * ch > num_slices * 20) break;   // :481-486
        if (smp == 0 || smp > 5120) break;                         // an empty frame ends reading (:813); a longer one overruns the reference's buffer (:786)
        const size_t used = 8 + 16 * (size_t)ch + 8 * (size_t)ch * (size_t)((smp + 19) / 20);
        if (used > n - pos) break;                                 // a slice read fails (:512): the frame is not delivered
        afg_qoa_frame fr;
        std::memset(&fr, 0, sizeof(fr));
        fr.byte_off = pos;
        fr.out_off = out;
        fr.samples = (uint16_t)smp;
        fr.channels = (uint8_t)ch;
        frames.push_back(fr);
        out += (uint64_t)smp * ch;
        pos += used;
    }
    return !frames.empty();
}

}  // namespace afg_front
