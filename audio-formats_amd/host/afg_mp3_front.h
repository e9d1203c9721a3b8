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
    bool tagged = false;                 // Xing / Info tag found (minimp3_ex.d:586-603)
    int start_delay = 0;                 // samples (channels included) dropped at the start
    uint64_t detected_samples = 0;       // 0: deliver to the end of the data
    uint64_t declared_samples = 0;       // mp3dec_ex_t.samples (AudioStream length = this / channels)
    uint64_t pcm_samples = 0;            // floats the copy plan delivers
    std::vector<uint32_t> run_granules;  // granules per run of continuous decoder state (a resync starts a new one)
    std::vector<float> coef;             // 576 floats per granule-channel, order [granule][channel]
    std::vector<uint32_t> flags;         // AFG_MP3_FLAGS per granule-channel
    std::vector<Copy> copies;

    // Optional caller-provided record storage (the batch path parses straight into its page-locked staging
    // buffer).  When set, coef / flags above stay empty; `overflow` reports that `ext_cap` blocks were not enough.
    float *ext_coef = nullptr;
    uint32_t *ext_flags = nullptr;
    size_t ext_cap = 0, n_blocks = 0;
    bool overflow = false;

    size_t blocks() const { return n_blocks; }
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
    void truncate(size_t nb)
    {
        n_blocks = nb;
        if (!ext_coef) {
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

}  // namespace afg_mp3
