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
};

enum Status { kOpened = 0, kNotOpus = 1, kUnsupported = 2 };

// Whole file.  kNotOpus: not an Ogg Opus stream the reference would open.  kUnsupported: it is one, but holds what this
// front-end does not decode (SILK or hybrid packets).
Status parse_file(const uint8_t *data, size_t size, File &out);

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
