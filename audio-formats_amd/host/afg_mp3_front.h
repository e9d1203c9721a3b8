// afg_mp3_front.h -- host front-end for MPEG Layer III files (see afg_mp3_front.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/afg.h"

namespace afg_mp3 {

struct Copy {                    // `count` floats of the PCM plane starting at float `src` are delivered, in order
    uint64_t src, count;
};

struct File {
    int channels = 0, hz = 0;
    int layer = 3;                       // 1 / 2: the blocks hold subband samples (AFG_MP3_SUBBAND), see add_l12
    bool tagged = false;                 // Xing / Info tag found (minimp3_ex.d:586-603)
    int start_delay = 0;                 // samples (channels included) dropped at the start
    uint64_t detected_samples = 0;       // 0: deliver to the end of the data
    uint64_t declared_samples = 0;       // mp3dec_ex_t.samples (AudioStream length = this / channels)
    uint64_t pcm_samples = 0;            // floats the copy plan delivers
    std::vector<uint32_t> run_granules;  // granules per run of continuous decoder state (a resync starts a new one)
    bool continues_previous = false;     // chunked reading: the first run goes on from the decoder state the previous chunk left
    std::vector<float> coef;             // 576 floats per granule-channel, order [granule][channel]
    std::vector<uint32_t> flags;         // AFG_MP3_FLAGS per granule-channel
    std::vector<Copy> copies;

    // Optional caller-provided record storage (the batch path parses straight into its page-locked staging
    // buffer).  When set, coef / flags above stay empty; `overflow` reports that `ext_cap` blocks were not enough.
    float *ext_coef = nullptr;
    uint32_t *ext_flags = nullptr;
    size_t ext_cap = 0, n_blocks = 0;
    bool overflow = false;

    // Quantised mode (set `quantised` before parsing): instead of dequantised lines the parser records
    // the Huffman values (int16, 576 per block) and per granule what afg_mp3_requant_hip needs to turn them into the
    // same floats on the device: band scales, band tables, the stereo plan.  coef stays empty.
    bool quantised = false;
    bool q_unsupported = false;          // a granule the device requantiser does not cover was met (MPEG-2.5 8 kHz mixed
                                         // blocks, whose reorder walks outside the channel: minimp3.d:1218-1223)
    std::vector<int16_t> q;
    std::vector<afg_mp3_qgranule> qgr;
    std::vector<afg_mp3_sdesc> sdesc;
    int16_t *ext_q = nullptr;            // optional caller storage, like ext_coef: ext_cap blocks of 576 values ...
    afg_mp3_qgranule *ext_qgr = nullptr; // ... and ext_cap granule records (record k describes the granule whose first
                                         // block is k: slots of second channels stay unused)
    size_t n_granules = 0;

    size_t blocks() const { return n_blocks; }
    // one granule in quantised mode: values of its nch channels, its record (q_off / coef_off filled here), its flags
    void push_q(const int16_t (*qv)[576], int nch, afg_mp3_qgranule rec, const uint32_t *fl)
    {
        if (ext_q) {
            if (n_blocks + (size_t)nch > ext_cap) { overflow = true; return; }
            std::memcpy(ext_q + n_blocks * 576, qv, (size_t)nch * 576 * sizeof(int16_t));
            rec.q_off = rec.coef_off = (uint64_t)n_blocks * 576;
            ext_qgr[n_blocks] = rec;
            for (int c = 0; c < nch; c++) ext_flags[n_blocks + (size_t)c] = fl[c];
        } else {
            rec.q_off = rec.coef_off = (uint64_t)n_blocks * 576;
            q.insert(q.end(), &qv[0][0], &qv[0][0] + (size_t)nch * 576);
            qgr.push_back(rec);
            for (int c = 0; c < nch; c++) flags.push_back(fl[c]);
        }
        n_blocks += (size_t)nch;
        n_granules++;
    }
    void push(const float *x, uint32_t fl)
    {
        // AFG_MP3_NZ_BANDS: subbands above the last line whose bit pattern is not +0.0 need not be fetched by the device
        int last = 575;
        while (last >= 0) {
            uint32_t bits;
            std::memcpy(&bits, x + last, 4);
            if (bits) break;
            last--;
        }
        fl |= AFG_MP3_NZ_BANDS((last + 18) / 18);
        if (ext_coef) {
            if (n_blocks >= ext_cap) { overflow = true; return; }
            for (int i = 0; i < 576; i++) ext_coef[n_blocks * 576 + i] = x[i];
            ext_flags[n_blocks] = fl;
        } else {
            coef.insert(coef.end(), x, x + 576);
            flags.push_back(fl);
        }
        n_blocks++;
    }
    // Layer I / II (minimp3.d:1557-1578): a frame yields synthesis granules of 12 time slots (32 subband samples each).  The
    // synthesis filterbank runs slot pair by slot pair over a 15-slot history and knows nothing of granule boundaries, so
    // the slots of a run are packed into the transform stage's 18-slot blocks (index band * 18 + slot): three granules
    // make two blocks.  A run that ends inside a block is padded with silent slots whose output is not delivered.
    float l12_acc[2][576];
    int l12_slots = 0;                   // slots gathered in l12_acc
    uint64_t l12_pcm_at() const { return (uint64_t)n_blocks * 576 + (uint64_t)l12_slots * 32 * (uint64_t)channels; }
    void l12_emit(int nch)
    {
        for (int c = 0; c < nch; c++) push(l12_acc[c], AFG_MP3_SUBBAND);
        if (!run_granules.empty()) run_granules.back()++;
        l12_slots = 0;
    }
    void add_l12(const float (*g)[576], int nch)           // one granule: slots 0..11 of g[ch][band * 18 + slot]
    {
        for (int s = 0; s < 12; s++) {
            if (l12_slots == 0) std::memset(l12_acc, 0, sizeof(l12_acc));
            for (int c = 0; c < nch; c++)
                for (int b = 0; b < 32; b++) l12_acc[c][b * 18 + l12_slots] = g[c][b * 18 + s];
            if (++l12_slots == 18) l12_emit(nch);
        }
    }
    void flush_l12(int nch)
    {
        if (l12_slots) l12_emit(nch);
    }
    void truncate(size_t nb)
    {
        const size_t n_blocks_seen = n_blocks;
        n_blocks = nb;
        if (quantised) {
            if (ext_q) {
                for (size_t k = nb; k < n_blocks_seen; k++) ext_qgr[k].nch = 0;       // slots of dropped granules: unused again
            }
            if (!ext_q) {
                q.resize(nb * 576);
                flags.resize(nb);
                while (!qgr.empty() && qgr.back().q_off >= (uint64_t)nb * 576) qgr.pop_back();
            }
            n_granules = channels ? nb / (size_t)channels : 0;
        } else if (!ext_coef) {
            coef.resize(nb * 576);
            flags.resize(nb);
        }
    }
};

bool looks_like_mp3(const uint8_t *data, size_t size);
bool parse_file(const uint8_t *data, size_t size, File &out);      // false: no Layer III stream found
// as above into caller-provided storage of `cap` blocks (out.overflow set if that was too small)
bool parse_file_into(const uint8_t *data, size_t size, File &out, float *coef, uint32_t *flags, size_t cap);
// an upper bound for the number of granule-channel blocks parse_file will produce (frame-header walk only)
size_t max_blocks(const uint8_t *data, size_t size);

// Chunked reading for the AudioStream surface (the reference decodes as the caller pulls: stream.d:429-637,
// minimp3_ex.d:787-888): `open` does mp3dec_ex_open's scan (tags, Xing/Info, length), `more` decodes the next frames
// into fresh records whose copy plan is relative to that chunk.  *continues tells whether the chunk's first run goes on
// from the decoder state the previous chunk ended with (the device keeps it in an AFG_MP3_STATE_FLOATS blob).
class Reader {
public:
    Reader();
    ~Reader();
    Reader(const Reader &) = delete;
    Reader &operator=(const Reader &) = delete;
    bool open(const uint8_t *data, size_t size, File &meta);
    bool more(File &out, int max_frames, bool *continues);      // false: the stream has ended and out holds nothing
private:
    struct Impl;
    Impl *p;
};

// Tables of the device requantiser, derived from the scalefactor-band tables (24 = 3 kinds x 8 sample-rate rows):
// band of every line, and where a line of the short part goes when L3_reorder interleaves the windows.
struct QTables {
    uint8_t band_of_line[24][576];
    uint16_t dst_of_src[24][576];
    uint16_t group_end[24][576];         // one past the window group a line belongs to (line + 1 in the long part): the
                                         // reorder permutes lines inside their group only
    float pow43[145];                    // g_pow43 (minimp3.d:722-735): 16 negative entries, then 0 .. 128
};
const QTables &qtables();

}  // namespace afg_mp3
