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
};

// whole file -> records; stops at the first frame that does not parse (drflac.d:2860).  false: not FLAC.
bool flac_parse(const uint8_t *d, size_t n, FlacInfo &fi, FlacRecords &rec);

struct QoaInfo {
    uint32_t channels = 0, samplerate = 0, samples = 0;
};

bool qoa_parse(const uint8_t *d, size_t n, QoaInfo &qi, std::vector<afg_qoa_frame> &frames);

}  // namespace afg_front
