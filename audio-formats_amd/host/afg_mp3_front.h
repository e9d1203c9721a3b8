// afg_mp3_front.h -- host front-end for MPEG Layer III files (see afg_mp3_front.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/afg.h"

namespace afg_mp3 {

struct Copy {                    // `count` floats of the PCM plane starting at float `src` are delivered, in order
    uint64_t src, count;
};

struct File {
    int channels = 0, hz = 0;
    bool tagged = false;                 // Xing / Info tag found (minimp3_ex.d:586-603)
    int start_delay = 0;                 // samples (channels included) dropped at the start
    uint64_t detected_samples = 0;       // 0: deliver to the end of the data
    uint64_t declared_samples = 0;       // mp3dec_ex_t.samples (AudioStream length = this / channels)
    uint64_t pcm_samples = 0;            // floats the copy plan delivers
    std::vector<uint32_t> run_granules;  // granules per run of continuous decoder state (a resync starts a new one)
    std::vector<float> coef;             // 576 floats per granule-channel, order [granule][channel]
    std::vector<uint32_t> flags;         // AFG_MP3_FLAGS per granule-channel
    std::vector<Copy> copies;
};

bool looks_like_mp3(const uint8_t *data, size_t size);
bool parse_file(const uint8_t *data, size_t size, File &out);      // false: no Layer III stream found

}  // namespace afg_mp3
