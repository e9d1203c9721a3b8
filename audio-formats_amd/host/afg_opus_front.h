// afg_opus_front.h -- host front-end for Ogg Opus files, CELT-only packets (see afg_opus_front.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/afg.h"

namespace afg_opus {

struct File {
    int channels = 0;                      // OpusHead channel count (1 or 2): the decoder's output channels
    int preskip = 0;                       // OpusHead pre-skip (reported, not dropped: the reference keeps those samples too)
    int gain_i = 0;                        // header gain + R128_TRACK_GAIN, clamped to int16 (Q7.8 dB); 0 = no scaling
    float gain = 1.0f;                     // 10^(gain_i / (20 * 256)), what the decoder multiplies its floats by
    int64_t declared_frames = 0;           // last page's granule position - preskip: the stream length
    uint64_t pcm_frames = 0;               // frames the recorded packets decode to
    bool error = false;                    // a packet failed to parse: the records end there and a read reports an error
    // One record per CELT frame, addressed for channel 0 of interleaved output (out_off = frame * channels, out_stride =
    // channels); channel c of the same frame reads coef_off + c * frame_size and writes out_off + c.
    std::vector<afg_celt_frame> frames;
    std::vector<float> coeffs;             // per frame [channel][frame_size]: denormalised MDCT coefficients
    // What `open` learns from the packets' TOC bytes: upper bounds of what decoding the stream will record (exact unless a
    // packet turns out to be unframeable).  The batch path sizes its page-locked staging with them.
    size_t bound_frames = 0, bound_coeffs = 0;
    // Optional external destination (the batch path decodes straight into page-locked staging): when set, `frames` and
    // `coeffs` stay empty, records go to ext_frames[0 .. n_frames) and coefficients to ext_coeffs[0 .. n_coeffs); `overflow`
    // reports that the capacities (the bounds above) were not enough.
    afg_celt_frame *ext_frames = nullptr;
    float *ext_coeffs = nullptr;
    size_t ext_frames_cap = 0, ext_coeffs_cap = 0, n_frames = 0, n_coeffs = 0;
    bool overflow = false;
    const afg_celt_frame *frame_records() const { return ext_frames ? ext_frames : frames.data(); }
};

enum Status { kOpened = 0, kNotOpus = 1, kUnsupported = 2 };

// Whole file.  kNotOpus: not an Ogg Opus stream the reference would open.  kUnsupported: it is one, but holds what this
// front-end does not decode (SILK or hybrid packets).
Status parse_file(const uint8_t *data, size_t size, File &out);
// The same into caller-provided storage (out.ext_* set by this call); sizes from a previous `Reader::open`.
Status parse_file_into(const uint8_t *data, size_t size, File &out, afg_celt_frame *frames, size_t frames_cap, float *coeffs, size_t coeffs_cap);

// Chunked reading for the AudioStream surface: `open` reads the two header packets and the stream length (and looks at
// every packet's TOC byte, so that an unsupported file is refused at open rather than in the middle of a read), `more`
// decodes the next audio packets into fresh records (out_off counted from the chunk's first frame).  The CELT layer's
// own memory (band energies, the noise seed) lives in the reader; the transform stage's (overlap, post-filter,
// de-emphasis) in the afg_celt_state blobs the caller keeps on the device between chunks.
class Reader {
public:
    Reader();
    ~Reader();
    Reader(const Reader &) = delete;
    Reader &operator=(const Reader &) = delete;
    Status open(const uint8_t *data, size_t size, File &meta);
    bool more(File &out, int max_packets);     // false: the stream has ended (out holds nothing to deliver)
private:
    struct Impl;
    Impl *p;
};

}  // namespace afg_opus
