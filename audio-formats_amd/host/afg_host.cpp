// afg_host.cpp -- host front-ends and the AudioStream-shaped surface of the C ABI.
//
// What stays on the host (bitstream / entropy parsing) for the formats whose
// front-ends exist so far:
//   FLAC  container + frame/subframe headers + Rice residuals   (reference drflac.d:680-1695,
//         :1887-2153; the prediction half of drflac.d:1235 is NOT done here: residuals and
//         subframe parameters become afg_flac_subframe / afg_flac_frame records)
//   QOA   file/frame headers only (reference qoa.d:413-486): the device reads the raw bytes
// and the outer surface mirroring AudioStream (stream.d:150-170 openFromMemory, :295-412
// getters, :429-637 readSamplesFloat): parse the file into transform-stage records, restore the
// samples on the device, serve interleaved floats.
#include "../csrc/afg_common.h"
#include "afg_mp3_front.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <memory>
#include <new>
#include <thread>
#include <vector>

namespace {

// error strings of the reference (internals.d:16-23; stream.d:1379)
const char *const kErrorUnknownFormat = "Cannot decode stream: unrecognized encoding.";
const char *const kErrorDecodingError = "Decoder encountered an error";
const char *const kErrorDecoderInitializationFailed = "Decoder initialization failed";
const char *const kErrorNotInitialized = "Stream not initialized";

// ---------------------------------------------------------------------------------------------
// MSB-first bit reader over memory (the role of drflac_bs, drflac.d:680-1002)
// ---------------------------------------------------------------------------------------------
struct BitReader {
    const uint8_t *p;
    size_t nbits, pos = 0;
    bool fail = false;
    BitReader(const uint8_t *data, size_t bytes) : p(data), nbits(bytes * 8) {}
    uint32_t bit()
    {
        if (pos >= nbits) { fail = true; return 0; }
        uint32_t b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u;
        pos++;
        return b;
    }
    uint64_t bits(unsigned n)      // n <= 57
    {
        if (n == 0) return 0;
        if (pos + n > nbits) { fail = true; pos = nbits; return 0; }
        uint64_t v = 0;
        size_t byte = pos >> 3;
        unsigned have = 0;
        uint64_t acc = 0;
        unsigned skip = (unsigned)(pos & 7);
        while (have < n + skip) { acc = (acc << 8) | p[byte++]; have += 8; }
        v = (acc >> (have - n - skip)) & ((n == 64) ? ~0ull : ((1ull << n) - 1));
        pos += n;
        return v;
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
            if (pos >= nbits) { fail = true; return false; }
            unsigned skip = (unsigned)(pos & 7);
            uint32_t byte = (uint32_t)(p[pos >> 3] << skip) & 0xffu;
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
struct FlacInfo {
    uint32_t sample_rate = 0, channels = 0, bps = 0, max_block = 0;
    uint64_t total_samples = 0;      // per channel (STREAMINFO), 0 = unknown
    size_t first_frame = 0;
};

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
        if (pos + len > n) return false;
        if (type == 0) {
            if (len < 34) return false;
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
        pos += len;
        if (last) break;
    }
    fi.first_frame = pos;
    return got && fi.sample_rate != 0 && fi.bps >= 4;
}

struct FlacRecords {
    std::vector<afg_flac_frame> frames;
    std::vector<afg_flac_subframe> subframes;
    std::vector<int32_t> res;
    uint64_t out_samples = 0;       // interleaved samples
};

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
        if (br.fail || (b & 0xC0) != 0x80) return false;
        out = (out << 6) | (b & 0x3F);
    }
    return true;
}

// residual of one subframe into dst[order .. block_size), drflac.d:1279-1328 (+ Rice :1166-1224)
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
        // reference's, so the same happens here (DESIGN.md "FLAC front-end"); flip kSpecEscape to get
        // the format's own behaviour.
        constexpr bool kSpecEscape = false;
        const bool escape = kSpecEscape && ((method == 0 && k == 15) || (method == 1 && k == 31));
        if (br.fail) return false;
        if (!escape) {
            for (uint32_t j = 0; j < count; j++) {
                uint32_t q;
                if (!br.unary(q)) return false;
                uint32_t v = (q << k) | (uint32_t)br.bits(k);
                dst[i++] = (int32_t)((v >> 1) ^ (~(v & 1) + 1));   // zig-zag, :1224
            }
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
    if (C != fi.channels || bs == 0 || bs > 65535) return false;

    afg_flac_frame fr;
    std::memset(&fr, 0, sizeof(fr));
    fr.in_off = rec.res.size();
    fr.out_off = rec.out_samples;
    fr.block_size = bs;
    fr.sf_index = (uint32_t)rec.subframes.size();
    fr.channels = (uint8_t)C;
    fr.assignment = (uint8_t)(asg <= 7 ? AFG_FLAC_INDEPENDENT : asg);
    fr.bps = (uint8_t)fi.bps;                                      // drflac_read_s32 shifts by 32 - STREAMINFO bps (:2883)
    rec.res.resize(rec.res.size() + (size_t)bs * C);
    for (uint32_t c = 0; c < C; c++) {
        int32_t *dst = rec.res.data() + fr.in_off + (size_t)c * bs;
        afg_flac_subframe sf;
        std::memset(&sf, 0, sizeof(sf));
        const unsigned hdr = (unsigned)br.bits(8);                 // :1530-1569
        if (br.fail || (hdr & 0x80)) return false;
        const unsigned type = (hdr & 0x7E) >> 1;
        unsigned wasted = 0;
        if (hdr & 1) {
            uint32_t z;
            if (!br.unary(z)) return false;
            wasted = z + 1;
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
    br.bits(16);                                                   // CRC-16, not verified (drflac.d:108, :1673)
    if (br.fail) return false;
    rec.frames.push_back(fr);
    rec.out_samples += (uint64_t)bs * C;
    return true;
}

// whole file -> records; stops at the first frame that does not parse (the reference's read loop
// does the same: drflac.d:2860)
bool flac_parse(const uint8_t *d, size_t n, FlacInfo &fi, FlacRecords &rec)
{
    if (!flac_open(d, n, fi)) return false;
    BitReader br(d + fi.first_frame, n - fi.first_frame);
    while (br.byte_pos() + 2 < n - fi.first_frame) {
        const size_t keep_f = rec.frames.size(), keep_s = rec.subframes.size(), keep_r = rec.res.size();
        if (!flac_frame(br, fi, rec)) {
            rec.frames.resize(keep_f);
            rec.subframes.resize(keep_s);
            rec.res.resize(keep_r);
            break;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// QOA: locate frames (qoa.d:413-486); everything else happens on the device
// ---------------------------------------------------------------------------------------------
struct QoaInfo {
    uint32_t channels = 0, samplerate = 0, samples = 0;
};

uint64_t be64(const uint8_t *p)
{
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
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
    while (pos + 8 + 16 * (size_t)qi.channels <= n) {
        const uint64_t h = be64(d + pos);
        const uint32_t ch = (uint32_t)((h >> 56) & 0xff), sr = (uint32_t)((h >> 32) & 0xffffff);
        const uint32_t smp = (uint32_t)((h >> 16) & 0xffff), fsz = (uint32_t)(h & 0xffff);
        if (fsz < 8 + 16 * ch || pos + fsz > n) break;
        const uint32_t slices = (fsz - 8 - 16 * ch) / 8;
        if (ch != qi.channels || sr != qi.samplerate || smp * ch > slices * 20 || smp == 0) break;   // qoa.d:478-486
        afg_qoa_frame fr;
        std::memset(&fr, 0, sizeof(fr));
        fr.byte_off = pos;
        fr.out_off = out;
        fr.samples = (uint16_t)smp;
        fr.channels = (uint8_t)ch;
        frames.push_back(fr);
        out += (uint64_t)smp * ch;
        pos += fsz;
    }
    return !frames.empty();
}

// ---------------------------------------------------------------------------------------------
// decoded files: one result plane for a whole batch
// ---------------------------------------------------------------------------------------------
struct Decoded {
    int status = AFG_OK;
    const char *message = nullptr;
    int format = AFG_FORMAT_UNKNOWN;
    int channels = 0;
    float samplerate = 0;
    int64_t frames = 0;                 // frames actually decoded
    int64_t declared_frames = AFG_UNKNOWN_LENGTH;
    size_t pcm_off = 0;                 // float offset of this file's interleaved PCM in the result plane
};

struct Parsed {
    int format = AFG_FORMAT_UNKNOWN;
    FlacInfo fi;
    FlacRecords flac;
    QoaInfo qi;
    std::vector<afg_qoa_frame> qoa;
    afg_mp3::File mp3;
};

// startDecoding's probe order for the formats handled here (stream.d:1586-1838): FLAC, QOA, then MP3 (whose
// detection is the weakest: a frame-sync search, which is why the reference tries it after the containers)
void parse_file(const uint8_t *d, size_t n, Parsed &p)
{
    if (flac_parse(d, n, p.fi, p.flac)) { p.format = AFG_FORMAT_FLAC; return; }
    p.flac = FlacRecords();
    if (qoa_parse(d, n, p.qi, p.qoa)) { p.format = AFG_FORMAT_QOA; return; }
    if (afg_mp3::looks_like_mp3(d, n) && afg_mp3::parse_file(d, n, p.mp3)) { p.format = AFG_FORMAT_MP3; return; }
    p.mp3 = afg_mp3::File();
}

struct DeviceBuf {
    void *p = nullptr;
    ~DeviceBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) { p = nullptr; afg::set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return AFG_ERR_OOM; }
        return AFG_OK;
    }
};

// page-locked host memory: H2D / D2H run at PCIe rate without a staging copy
struct PinnedBuf {
    void *p = nullptr;
    ~PinnedBuf() { release(); }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; }
    int alloc(size_t bytes)
    {
        hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; afg::set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return AFG_ERR_OOM; }
        return AFG_OK;
    }
};

template <typename F>
void parallel_for(size_t n, unsigned threads, F fn)
{
    if (n == 0) return;
    threads = (unsigned)std::min<size_t>(std::max(1u, threads), n);
    std::atomic<size_t> next{ 0 };
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) return;
            fn(i);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
}

struct BatchOut {
    std::vector<Decoded> files;
    PinnedBuf plane;                    // all PCM of the batch, FLAC files first then QOA files
    size_t plane_floats = 0;
};

// Device stage for a set of parsed files: every FLAC record of the batch in one launch, every QOA frame
// in another; inputs are gathered (by `threads` host threads) into one page-locked buffer per kind and
// the results come back as one plane.
int decode_parsed(std::vector<Parsed> &parsed, const uint8_t *const *data, const size_t *len, unsigned threads, BatchOut &out)
{
    const size_t nf = parsed.size();
    out.files.assign(nf, Decoded());
    // ---- layout ----
    std::vector<size_t> res_base(nf, 0), fr_base(nf, 0), sf_base(nf, 0), qbyte_base(nf, 0), qfr_base(nf, 0);
    size_t res_total = 0, fr_total = 0, sf_total = 0, flac_out = 0, qbytes = 0, qframes = 0, qoa_out = 0;
    for (size_t i = 0; i < nf; i++) {
        Parsed &p = parsed[i];
        if (p.format != AFG_FORMAT_FLAC) continue;
        res_base[i] = res_total; fr_base[i] = fr_total; sf_base[i] = sf_total;
        out.files[i].pcm_off = flac_out;
        res_total += p.flac.res.size(); fr_total += p.flac.frames.size(); sf_total += p.flac.subframes.size();
        flac_out += p.flac.out_samples;
    }
    for (size_t i = 0; i < nf; i++) {
        Parsed &p = parsed[i];
        if (p.format != AFG_FORMAT_QOA) continue;
        qbyte_base[i] = qbytes; qfr_base[i] = qframes;
        out.files[i].pcm_off = flac_out + qoa_out;
        qbytes += (len[i] + 15) & ~(size_t)15;
        qframes += p.qoa.size();
        qoa_out += p.qoa.back().out_off + (size_t)p.qoa.back().samples * p.qoa.back().channels;
    }
    std::vector<size_t> mp3_blk_base(nf, 0);
    size_t mp3_blocks = 0, mp3_out = 0, mp3_runs = 0;
    for (size_t i = 0; i < nf; i++) {
        Parsed &p = parsed[i];
        if (p.format != AFG_FORMAT_MP3) continue;
        mp3_blk_base[i] = mp3_blocks;
        out.files[i].pcm_off = flac_out + qoa_out + mp3_out;
        mp3_blocks += p.mp3.flags.size();
        mp3_out += (size_t)p.mp3.pcm_samples;
        mp3_runs += p.mp3.run_granules.size();
    }
    out.plane_floats = flac_out + qoa_out + mp3_out;
    if (out.plane_floats == 0) goto metadata;
    {
        if (int rc = out.plane.alloc(out.plane_floats * sizeof(float))) return rc;
        DeviceBuf d_out;
        if (int rc = d_out.alloc(out.plane_floats * sizeof(float))) return rc;
        hipStream_t stream = nullptr;
        // ---- FLAC ----
        if (flac_out) {
            const size_t rec_bytes = fr_total * sizeof(afg_flac_frame) + sf_total * sizeof(afg_flac_subframe);
            const size_t rec_pad = (rec_bytes + 15) & ~(size_t)15;
            PinnedBuf h_in;
            DeviceBuf d_in;
            if (int rc = h_in.alloc(rec_pad + res_total * 4)) return rc;
            if (int rc = d_in.alloc(rec_pad + res_total * 4)) return rc;
            afg_flac_frame *hf = (afg_flac_frame *)h_in.p;
            afg_flac_subframe *hs = (afg_flac_subframe *)(hf + fr_total);
            int32_t *hr = (int32_t *)((uint8_t *)h_in.p + rec_pad);
            parallel_for(nf, threads, [&](size_t i) {
                Parsed &p = parsed[i];
                if (p.format != AFG_FORMAT_FLAC) return;
                for (size_t k = 0; k < p.flac.frames.size(); k++) {
                    afg_flac_frame f = p.flac.frames[k];
                    f.in_off += res_base[i]; f.out_off += out.files[i].pcm_off; f.sf_index += (uint32_t)sf_base[i];
                    hf[fr_base[i] + k] = f;
                }
                std::memcpy(hs + sf_base[i], p.flac.subframes.data(), p.flac.subframes.size() * sizeof(afg_flac_subframe));
                std::memcpy(hr + res_base[i], p.flac.res.data(), p.flac.res.size() * 4);
                std::vector<int32_t>().swap(p.flac.res);               // the residual plane is the big one: drop it early
            });
            AFG_HIP_CHECK(hipMemcpyAsync(d_in.p, h_in.p, rec_pad + res_total * 4, hipMemcpyHostToDevice, stream));
            const afg_flac_frame *df = (const afg_flac_frame *)d_in.p;
            const afg_flac_subframe *ds = (const afg_flac_subframe *)(df + fr_total);
            const int32_t *dr = (const int32_t *)((const uint8_t *)d_in.p + rec_pad);
            if (int rc = afg_flac_transform_hip(fr_total, df, ds, dr, nullptr, (float *)d_out.p, stream)) return rc;
            AFG_HIP_CHECK(hipStreamSynchronize(stream));
        }
        // ---- QOA ----
        if (qoa_out) {
            const size_t rec_pad = (qframes * sizeof(afg_qoa_frame) + 15) & ~(size_t)15;
            PinnedBuf h_in;
            DeviceBuf d_in;
            if (int rc = h_in.alloc(rec_pad + qbytes)) return rc;
            if (int rc = d_in.alloc(rec_pad + qbytes)) return rc;
            afg_qoa_frame *hq = (afg_qoa_frame *)h_in.p;
            uint8_t *hb = (uint8_t *)h_in.p + rec_pad;
            parallel_for(nf, threads, [&](size_t i) {
                Parsed &p = parsed[i];
                if (p.format != AFG_FORMAT_QOA) return;
                for (size_t k = 0; k < p.qoa.size(); k++) {
                    afg_qoa_frame f = p.qoa[k];
                    f.byte_off += qbyte_base[i]; f.out_off += out.files[i].pcm_off;
                    hq[qfr_base[i] + k] = f;
                }
                std::memcpy(hb + qbyte_base[i], data[i], len[i]);
            });
            AFG_HIP_CHECK(hipMemcpyAsync(d_in.p, h_in.p, rec_pad + qbytes, hipMemcpyHostToDevice, stream));
            if (int rc = afg_qoa_transform_hip(qframes, (const afg_qoa_frame *)d_in.p, (const uint8_t *)d_in.p + rec_pad, nullptr,
                                               (float *)d_out.p, stream))
                return rc;
            AFG_HIP_CHECK(hipStreamSynchronize(stream));
        }
        if (flac_out + qoa_out) {
            AFG_HIP_CHECK(hipMemcpyAsync(out.plane.p, d_out.p, (flac_out + qoa_out) * sizeof(float), hipMemcpyDeviceToHost, stream));
            AFG_HIP_CHECK(hipStreamSynchronize(stream));
        }
        // ---- MP3: spectra of every decoded granule -> PCM plane -> the samples mp3dec_ex_read would deliver ----
        if (mp3_blocks) {
            const size_t coef_bytes = mp3_blocks * 576 * sizeof(float), flag_bytes = (mp3_blocks * 4 + 15) & ~(size_t)15;
            PinnedBuf h_in, h_pcm;
            DeviceBuf d_in, d_pcm;
            if (int rc = h_in.alloc(coef_bytes + flag_bytes)) return rc;
            if (int rc = d_in.alloc(coef_bytes + flag_bytes)) return rc;
            if (int rc = h_pcm.alloc(coef_bytes)) return rc;
            if (int rc = d_pcm.alloc(coef_bytes)) return rc;
            float *hc = (float *)h_in.p;
            uint32_t *hfl = (uint32_t *)((uint8_t *)h_in.p + coef_bytes);
            std::vector<uint32_t> granules;
            std::vector<uint8_t> channels;
            granules.reserve(mp3_runs);
            channels.reserve(mp3_runs);
            for (size_t i = 0; i < nf; i++) {
                if (parsed[i].format != AFG_FORMAT_MP3) continue;
                for (uint32_t g : parsed[i].mp3.run_granules) {
                    granules.push_back(g);
                    channels.push_back((uint8_t)parsed[i].mp3.channels);
                }
            }
            parallel_for(nf, threads, [&](size_t i) {
                Parsed &p = parsed[i];
                if (p.format != AFG_FORMAT_MP3) return;
                std::memcpy(hc + mp3_blk_base[i] * 576, p.mp3.coef.data(), p.mp3.coef.size() * sizeof(float));
                std::memcpy(hfl + mp3_blk_base[i], p.mp3.flags.data(), p.mp3.flags.size() * sizeof(uint32_t));
                std::vector<float>().swap(p.mp3.coef);
            });
            afg_mp3_plan *plan = nullptr;
            if (int rc = afg_mp3_plan_create(&plan, (uint32_t)granules.size(), granules.data(), channels.data(), 0)) return rc;
            hipError_t e = hipMemcpyAsync(d_in.p, h_in.p, coef_bytes + flag_bytes, hipMemcpyHostToDevice, stream);
            int rc = e == hipSuccess ? afg_mp3_transform_hip(plan, (const float *)d_in.p, (const uint32_t *)((const uint8_t *)d_in.p + coef_bytes),
                                                             (float *)d_pcm.p, nullptr, stream)
                                     : AFG_ERR_HIP;
            if (!rc) e = hipMemcpyAsync(h_pcm.p, d_pcm.p, coef_bytes, hipMemcpyDeviceToHost, stream);
            if (!rc && e == hipSuccess) e = hipStreamSynchronize(stream);
            afg_mp3_plan_destroy(plan);
            if (rc) return rc;
            if (e != hipSuccess) { afg::set_error("MP3 stage failed: %s", hipGetErrorString(e)); return AFG_ERR_HIP; }
            parallel_for(nf, threads, [&](size_t i) {
                const Parsed &p = parsed[i];
                if (p.format != AFG_FORMAT_MP3) return;
                const float *src = (const float *)h_pcm.p + mp3_blk_base[i] * 576;
                float *dst = (float *)out.plane.p + out.files[i].pcm_off;
                for (const afg_mp3::Copy &c : p.mp3.copies) {
                    std::memcpy(dst, src + c.src, (size_t)c.count * sizeof(float));
                    dst += c.count;
                }
            });
        }
    }
metadata:
    for (size_t i = 0; i < nf; i++) {
        Parsed &p = parsed[i];
        Decoded &dcd = out.files[i];
        dcd.format = p.format;
        if (p.format == AFG_FORMAT_FLAC) {
            dcd.channels = (int)p.fi.channels;
            dcd.samplerate = (float)p.fi.sample_rate;
            dcd.frames = (int64_t)(p.flac.out_samples / p.fi.channels);
            dcd.declared_frames = (int64_t)p.fi.total_samples;      // totalSampleCount / channels, stream.d:1631
        } else if (p.format == AFG_FORMAT_MP3) {
            dcd.channels = p.mp3.channels;
            dcd.samplerate = (float)p.mp3.hz;
            dcd.frames = (int64_t)(p.mp3.pcm_samples / (uint64_t)p.mp3.channels);
            dcd.declared_frames = (int64_t)(p.mp3.declared_samples / (uint64_t)p.mp3.channels);   // stream.d:1737
        } else if (p.format == AFG_FORMAT_QOA) {
            dcd.channels = (int)p.qi.channels;
            dcd.samplerate = (float)p.qi.samplerate;
            dcd.frames = (int64_t)(p.qoa.back().out_off / p.qi.channels) + p.qoa.back().samples;
            dcd.declared_frames = (int64_t)p.qi.samples;
        } else {
            dcd.status = AFG_ERR_UNSUPPORTED;
            dcd.message = kErrorUnknownFormat;
        }
    }
    return AFG_OK;
}

}  // namespace

struct afg_stream {
    const char *error = kErrorNotInitialized;      // stream.d:1379
    Decoded d;
    std::vector<float> pcm;
    int64_t position = 0;
};

extern "C" {

afg_stream *afg_open_from_memory(const uint8_t *data, size_t length)
{
    afg_stream *s = new (std::nothrow) afg_stream;
    if (!s) return nullptr;
    if (!data || length == 0) { s->error = kErrorUnknownFormat; return s; }
    std::vector<Parsed> parsed(1);
    parse_file(data, length, parsed[0]);
    if (parsed[0].format == AFG_FORMAT_UNKNOWN) { s->error = kErrorUnknownFormat; return s; }
    if (afg::require_device() != AFG_OK) { s->error = kErrorDecoderInitializationFailed; return s; }
    BatchOut out;
    const uint8_t *dp[1] = { data };
    const size_t lp[1] = { length };
    if (decode_parsed(parsed, dp, lp, 1, out) != AFG_OK) { s->error = kErrorDecodingError; return s; }
    s->d = out.files[0];
    if (out.plane_floats) s->pcm.assign((const float *)out.plane.p, (const float *)out.plane.p + out.plane_floats);
    s->error = s->d.status == AFG_OK ? nullptr : kErrorDecodingError;
    return s;
}

int afg_is_error(const afg_stream *s) { return !s || s->error != nullptr; }
const char *afg_error_message(const afg_stream *s) { return s ? s->error : kErrorNotInitialized; }
int afg_get_format(const afg_stream *s) { return (s && !s->error) ? s->d.format : AFG_FORMAT_UNKNOWN; }
int afg_get_num_channels(const afg_stream *s) { return (s && !s->error) ? s->d.channels : 0; }
float afg_get_samplerate(const afg_stream *s) { return (s && !s->error) ? s->d.samplerate : 0.0f; }

int64_t afg_get_length_in_frames(const afg_stream *s)
{
    if (!s || s->error) return AFG_UNKNOWN_LENGTH;
    return s->d.declared_frames;       // stream.d:404-407: whatever the container declares (FLAC: may be 0)
}

int afg_read_samples_float(afg_stream *s, float *out, int frames)
{
    if (!s || s->error || frames <= 0) return 0;
    // stream.d:498: a FLAC stream stops once the position equals the declared length (a STREAMINFO that
    // declares 0 samples therefore reads nothing); the check is made on entry only, like the reference's.
    if (s->d.format == AFG_FORMAT_FLAC && s->position == s->d.declared_frames) return 0;
    const int64_t n = std::max<int64_t>(0, std::min<int64_t>(s->d.frames - s->position, frames));
    if (out && n) std::memcpy(out, s->pcm.data() + s->position * s->d.channels, (size_t)n * s->d.channels * sizeof(float));
    s->position += n;
    return (int)n;
}

void afg_close(afg_stream *s) { delete s; }

namespace {
struct FlacParsedOwner {
    FlacRecords rec;
};
}  // namespace

int afg_flac_parse(const uint8_t *data, size_t length, afg_flac_parsed *out)
{
    if (!out) return AFG_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    if (!data) return AFG_ERR_INVALID;
    auto *own = new (std::nothrow) FlacParsedOwner;
    if (!own) return AFG_ERR_OOM;
    FlacInfo fi;
    if (!flac_parse(data, length, fi, own->rec)) {
        delete own;
        afg::set_error("afg_flac_parse: not a native FLAC stream");
        return AFG_ERR_UNSUPPORTED;
    }
    out->sample_rate = fi.sample_rate;
    out->channels = fi.channels;
    out->bps = fi.bps;
    out->max_block = fi.max_block;
    out->total_samples = fi.total_samples;
    out->n_frames = own->rec.frames.size();
    out->n_subframes = own->rec.subframes.size();
    out->n_res = own->rec.res.size();
    out->out_samples = own->rec.out_samples;
    out->frames = own->rec.frames.data();
    out->subframes = own->rec.subframes.data();
    out->res = own->rec.res.data();
    out->owner = own;
    return AFG_OK;
}

void afg_flac_parsed_free(afg_flac_parsed *p)
{
    if (!p) return;
    delete (FlacParsedOwner *)p->owner;
    std::memset(p, 0, sizeof(*p));
}

int afg_mp3_parse(const uint8_t *data, size_t length, afg_mp3_parsed *out)
{
    if (!out) return AFG_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    if (!data) return AFG_ERR_INVALID;
    auto *own = new (std::nothrow) afg_mp3::File;
    if (!own) return AFG_ERR_OOM;
    if (!afg_mp3::parse_file(data, length, *own)) {
        delete own;
        afg::set_error("afg_mp3_parse: no MPEG Layer III stream found");
        return AFG_ERR_UNSUPPORTED;
    }
    static_assert(sizeof(afg_mp3::Copy) == sizeof(afg_mp3_copy), "copy plan layout");
    out->channels = own->channels;
    out->hz = own->hz;
    out->tagged = own->tagged ? 1 : 0;
    out->start_delay = own->start_delay;
    out->detected_samples = own->detected_samples;
    out->declared_samples = own->declared_samples;
    out->pcm_samples = own->pcm_samples;
    out->n_runs = own->run_granules.size();
    out->n_blocks = own->flags.size();
    out->n_copies = own->copies.size();
    out->run_granules = own->run_granules.data();
    out->coef = own->coef.data();
    out->flags = own->flags.data();
    out->copies = (afg_mp3_copy *)own->copies.data();
    out->owner = own;
    return AFG_OK;
}

void afg_mp3_parsed_free(afg_mp3_parsed *p)
{
    if (!p) return;
    delete (afg_mp3::File *)p->owner;
    std::memset(p, 0, sizeof(*p));
}

int afg_qoa_parse(const uint8_t *data, size_t length, uint32_t *channels, uint32_t *samplerate, uint32_t *samples,
                  afg_qoa_frame *frames, size_t frame_cap, size_t *n_frames)
{
    if (!data) return AFG_ERR_INVALID;
    QoaInfo qi;
    std::vector<afg_qoa_frame> fr;
    if (!qoa_parse(data, length, qi, fr)) {
        afg::set_error("afg_qoa_parse: not a QOA file");
        return AFG_ERR_UNSUPPORTED;
    }
    if (channels) *channels = qi.channels;
    if (samplerate) *samplerate = qi.samplerate;
    if (samples) *samples = qi.samples;
    if (n_frames) *n_frames = fr.size();
    if (frames) std::memcpy(frames, fr.data(), std::min(frame_cap, fr.size()) * sizeof(afg_qoa_frame));
    return AFG_OK;
}

int afg_batch_decode(const uint8_t *const *data, const size_t *length, int n_files, int n_threads, afg_batch_result *out)
{
    if (!out || n_files < 0 || (n_files && (!data || !length))) return AFG_ERR_INVALID;
    out->n_files = 0; out->items = nullptr; out->owner = nullptr;
    if (n_files == 0) return AFG_OK;
    if (int rc = afg::require_device()) return rc;
    std::vector<Parsed> parsed((size_t)n_files);
    const unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::max(1u, std::thread::hardware_concurrency());
    parallel_for((size_t)n_files, nt, [&](size_t i) {
        if (data[i] && length[i]) parse_file(data[i], length[i], parsed[i]);
    });

    BatchOut *owner = new (std::nothrow) BatchOut;
    if (!owner) return AFG_ERR_OOM;
    int rc = decode_parsed(parsed, data, length, nt, *owner);
    if (rc) { delete owner; return rc; }
    afg_batch_item *items = (afg_batch_item *)std::calloc((size_t)n_files, sizeof(afg_batch_item));
    if (!items) { delete owner; return AFG_ERR_OOM; }
    for (int i = 0; i < n_files; i++) {
        const Decoded &d = owner->files[(size_t)i];
        items[i].status = d.status;
        items[i].message = d.message;
        items[i].format = d.format;
        items[i].channels = d.channels;
        items[i].samplerate = d.samplerate;
        items[i].frames = d.frames;
        items[i].pcm = (d.status == AFG_OK && d.frames > 0) ? (float *)owner->plane.p + d.pcm_off : nullptr;
    }
    out->n_files = n_files;
    out->items = items;
    out->owner = owner;
    return AFG_OK;
}

void afg_batch_free(afg_batch_result *r)
{
    if (!r) return;
    std::free(r->items);
    delete (BatchOut *)r->owner;
    r->items = nullptr; r->owner = nullptr; r->n_files = 0;
}

}  // extern "C"
