// afg_flac_front.h -- host front-ends for native FLAC files and QOA files (see afg_flac_front.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/afg.h"

namespace afg_front {

struct FlacInfo {
    uint32_t sample_rate = 0, channels = 0, bps = 0, max_block = 0;
    uint64_t total_samples = 0;      // per channel (STREAMINFO), 0 = unknown
    size_t first_frame = 0;
};

struct FlacRecords {
    std::vector<afg_flac_frame> frames;
    std::vector<afg_flac_subframe> subframes;
    std::vector<int32_t> res;
    uint64_t out_samples = 0;       // interleaved samples
    // Optional external destination of the residual plane (the batch path parses straight into a page-locked staging
    // buffer): when set, `res` stays empty; `overflow` reports that `ext_cap` words were not enough.
    int32_t *ext_res = nullptr;
    size_t ext_cap = 0, n_res = 0;
    bool overflow = false;
    // pack16 (set before parsing): a frame whose residuals and warm-up samples all fit 16 bits is rewritten in place as
    // int16 rows (afg_flac_frame.res16, AFG_FLAC_ROW16): the device then reads half the bytes.  The frame keeps the words
    // it was parsed into (the second half is unused); the plane itself must start on a 16-byte boundary.
    bool pack16 = false;
    size_t res_size() const { return ext_res ? n_res : res.size(); }
    const int32_t *res_data() const { return ext_res ? ext_res : res.data(); }
    int32_t *res_grow(size_t words)          // room for `words` more residuals, or nullptr (external buffer full)
    {
        if (ext_res) {
            if (n_res + words > ext_cap) { overflow = true; return nullptr; }
            int32_t *p = ext_res + n_res;
            n_res += words;
            return p;
        }
        res.resize(res.size() + words);
        return res.data() + res.size() - words;
    }
    void res_truncate(size_t words)
    {
        if (ext_res) n_res = words;
        else res.resize(words);
    }
};

// whole file -> records; stops at the first frame that does not parse (drflac.d:2860).  false: not FLAC.
bool flac_parse(const uint8_t *d, size_t n, FlacInfo &fi, FlacRecords &rec);
bool flac_parse_into(const uint8_t *d, size_t n, FlacInfo &fi, FlacRecords &rec, int32_t *res_dst, size_t cap);
// Chunked reading for the AudioStream surface: the container walk alone (STREAMINFO, first frame), then up to max_frames
// frames from byte *pos behind the first frame (advanced past what was parsed).  *ended: no further frame parses.
bool flac_open_info(const uint8_t *d, size_t n, FlacInfo &fi);
int flac_parse_frames(const uint8_t *d, size_t n, const FlacInfo &fi, FlacRecords &rec, size_t *pos, int max_frames, bool *ended);
// Residual words a well-formed file needs ((STREAMINFO total + one block) x channels); 0: not FLAC or length unknown.
size_t flac_res_bound(const uint8_t *d, size_t n);

struct QoaInfo {
    uint32_t channels = 0, samplerate = 0, samples = 0;
};

bool qoa_parse(const uint8_t *d, size_t n, QoaInfo &qi, std::vector<afg_qoa_frame> &frames);

}  // namespace afg_front
