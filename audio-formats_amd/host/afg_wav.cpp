// afg_wav.cpp -- WAV writer of the output side (reference wav.d:365-701, WAVEncoder), host only.
//
// Header exactly as the reference lays it out: "RIFF" <len> "WAVE" "fmt " 16 <tag> <channels> <rate> <bytes/s>
// <block align> <bits> "data" <len>, no pad byte after an odd-sized data chunk (the reference writes none), lengths as
// finalizeEncoding computes them (:571-606).  Sample conversions are writeSamples' (:482-527); the TPDF dither of
// :674-701 (on by default in the reference, EncodingOptions.enableDither stream.d:66) draws from libc rand() there:
// afg_wav_encode_dithered takes the generator as a callback so that the result is reproducible and testable.
#include "../../include/afg.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
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

namespace {

int libc_rand(void *) { return std::rand(); }

// TPDFDither.process for one sample (wav.d:680-699): scale to LSBs, add the offset and two uniform draws, floor,
// scale back, clamp.  Two draws per sample, in this order.
struct Dither {
    afg_rand_fn fn;
    void *user;
    double rand_max;
    double one(double x, double scale) const
    {
        const double TUNE0 = 0.25, TUNE1 = TUNE0 * 0.5;
        x *= scale;
        x += (0.5 - 0.5 * (TUNE0 + TUNE1));
        x += TUNE0 * (fn(user) / rand_max);
        x += TUNE1 * (fn(user) / rand_max);
        x = std::floor(x);
        x /= scale;
        if (x < -1.0) x = -1.0;
        if (x > 1.0) x = 1.0;
        return x;
    }
};

uint64_t wav_write(const float *samples, uint64_t frames, uint32_t channels, uint32_t samplerate, int format,
                   const Dither *dither, uint8_t *out, uint64_t cap);

}  // namespace

uint64_t afg_wav_encode(const float *samples, uint64_t frames, uint32_t channels, uint32_t samplerate, int format,
                        uint8_t *out, uint64_t cap)
{
    return wav_write(samples, frames, channels, samplerate, format, nullptr, out, cap);
}

uint64_t afg_wav_encode_dithered(const float *samples, uint64_t frames, uint32_t channels, uint32_t samplerate, int format,
                                 afg_rand_fn rng, void *rng_user, uint32_t rng_max, uint8_t *out, uint64_t cap)
{
    Dither d;
    d.fn = rng ? rng : libc_rand;
    d.user = rng_user;
    d.rand_max = rng ? (double)rng_max : (double)RAND_MAX;
    if (rng && rng_max == 0) return 0;
    return wav_write(samples, frames, channels, samplerate, format, &d, out, cap);
}

}  // extern "C"

namespace {

uint64_t wav_write(const float *samples, uint64_t frames, uint32_t channels, uint32_t samplerate, int format,
                   const Dither *dither, uint8_t *out, uint64_t cap)
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
        for (uint64_t i = 0; i < n; i++) {
            double x = samples[i];
            if (dither) x = dither->one(x, 127.0);                         // ditherInput(.., 127.0f), :483
            *p++ = (uint8_t)(int8_t)(int)(128.5 + x * 127.0);              // :486-487
        }
        break;
    case AFG_WAV_S16LE:
        for (uint64_t i = 0; i < n; i++) {
            double x = samples[i];
            if (dither) x = dither->one(x, 32767.0);                       // :497
            const int s = (int)(32768.5 + x * 32767.0) - 32768;            // :501-502
            put16(p, (uint32_t)s & 0xffff);
        }
        break;
    case AFG_WAV_S24LE:
        for (uint64_t i = 0; i < n; i++) {
            double x = samples[i];
            if (dither) x = dither->one(x, 8388607.0);                     // :513
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

}  // namespace
