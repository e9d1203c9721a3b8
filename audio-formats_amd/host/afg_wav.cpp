// afg_wav.cpp -- WAV writer of the output side (reference wav.d:365-701, WAVEncoder), host only.
//
// Header exactly as the reference lays it out: "RIFF" <len> "WAVE" "fmt " 16 <tag> <channels> <rate> <bytes/s>
// <block align> <bits> "data" <len>, no pad byte after an odd-sized data chunk (the reference writes none), lengths as
// finalizeEncoding computes them (:571-606).  Sample conversions are writeSamples' (:482-527) with dither off.
#include "../../include/afg.h"

#include <cstdint>
#include <cstring>

namespace {

int sample_size(int format)
{
    switch (format) {
    case AFG_WAV_S8: return 1;
    case AFG_WAV_S16LE: return 2;
    case AFG_WAV_S24LE: return 3;
    case AFG_WAV_FP32LE: return 4;
    case AFG_WAV_FP64LE: return 8;
    default: return 0;
    }
}

void put16(uint8_t *&p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p += 2; }
void put32(uint8_t *&p, uint32_t v) { put16(p, v & 0xffff); put16(p, v >> 16); }

}  // namespace

extern "C" {

uint64_t afg_wav_encoded_size(uint64_t frames, uint32_t channels, int format)
{
    const int ss = sample_size(format);
    if (!ss || channels > 1024) return 0;                                  // wav.d:400-405
    return 44 + frames * (uint64_t)channels * (uint64_t)ss;
}

uint64_t afg_wav_encode(const float *samples, uint64_t frames, uint32_t channels, uint32_t samplerate, int format,
                        uint8_t *out, uint64_t cap)
{
    const uint64_t size = afg_wav_encoded_size(frames, channels, format);
    if (!size || !out || cap < size || (!samples && frames * channels)) return 0;
    const int ss = sample_size(format);
    const uint32_t frame_size = (uint32_t)ss * channels;
    const uint64_t data_bytes = (uint64_t)frame_size * frames;
    uint8_t *p = out;
    std::memcpy(p, "RIFF", 4); p += 4;
    put32(p, (uint32_t)(4 + (4 + 4 + 16) + (4 + 4 + data_bytes)));        // :573
    std::memcpy(p, "WAVE", 4); p += 4;
    std::memcpy(p, "fmt ", 4); p += 4;
    put32(p, 16);
    put16(p, format <= AFG_WAV_S24LE ? 1 : 3);                            // LinearPCM / FloatingPointIEEE
    put16(p, channels & 0xffff);
    put32(p, samplerate);
    put32(p, (uint32_t)((uint64_t)samplerate * frame_size));
    put16(p, frame_size & 0xffff);
    put16(p, (uint32_t)ss * 8);
    std::memcpy(p, "data", 4); p += 4;
    put32(p, (uint32_t)data_bytes);
    const uint64_t n = frames * channels;
    switch (format) {
    case AFG_WAV_S8:
        for (uint64_t i = 0; i < n; i++) { const double x = samples[i]; *p++ = (uint8_t)(int8_t)(int)(128.5 + x * 127.0); }      // :486-487
        break;
    case AFG_WAV_S16LE:
        for (uint64_t i = 0; i < n; i++) {
            const double x = samples[i];
            const int s = (int)(32768.5 + x * 32767.0) - 32768;            // :501-502
            put16(p, (uint32_t)s & 0xffff);
        }
        break;
    case AFG_WAV_S24LE:
        for (uint64_t i = 0; i < n; i++) {
            const double x = samples[i];
            const int s = (int)(8388608.5 + x * 8388607.0) - 8388608;      // :517-518
            p[0] = (uint8_t)s; p[1] = (uint8_t)(s >> 8); p[2] = (uint8_t)(s >> 16); p += 3;
        }
        break;
    case AFG_WAV_FP32LE:
        std::memcpy(p, samples, n * 4); p += n * 4;
        break;
    default:
        for (uint64_t i = 0; i < n; i++) { const double x = samples[i]; std::memcpy(p, &x, 8); p += 8; }
        break;
    }
    return (uint64_t)(p - out);
}

}  // extern "C"
