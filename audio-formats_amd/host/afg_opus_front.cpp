// afg_opus_front.cpp -- host front-end for Ogg Opus files whose packets are CELT-only.
//
// The part of the reference's Opus decoder that stays on the host (entropy decoding is serial per
// stream): Ogg pages and lacing, OpusHead / OpusTags (dopus.d:7791-7829, :8120-8193; output gain :1311-1316 with the
// R128_TRACK_GAIN comment :8011-8059), packet framing (ff_opus_parse_packet, :1081-1258), the range decoder (:809-1034)
// and the CELT frame decoder up to the denormalised coefficients (coarse / fine / final band energies :2128-2216, tf
// changes :2218-2243, bit allocation :2245-2575, PVQ shapes with spreading, splitting and folding :2577-3266, post-filter
// parameters :3380-3418, anti-collapse :3420-3470, the band loop :3472-3566, frame head and memory update :3568-3678,
// :3704-3731).  Everything after that seam -- inverse MDCT, window, post-filter, de-emphasis (:3680-3702), gain, the
// int16 round trip -- runs on the device (afg_celt_transform_hip, afg_opus_output_gain_hip).
//
// Not decoded: SILK and hybrid packets (a file that holds one is refused with kUnsupported) and multistream mappings
// (the reference refuses those itself, :8164-8169).
//
// Float readings where the D source leaves room: cos / sin / exp2 of a float are evaluated in double and rounded;
// products with the double constants M_SQRT1_2 / M_SQRT2 in double; PI * gain * gain / 4 in long double (PI is a real).
// This file is built with -ffp-contract=off like the rest of the host code: the reference has no fused multiply-adds.
#include "afg_opus_front.h"

#include <cmath>
#include <cstring>
#include <memory>

namespace afg_opus {
namespace {

#include "opus_front_tables.h"

constexpr int kBands = 21;
constexpr int kShortBlock = 120;
constexpr int kMaxFrame = 960;
constexpr int kMaxFineBits = 8;
constexpr float kSilenceEnergy = -28.0f;

inline float table_f32(const uint32_t *bits, int i)
{
    float f;
    std::memcpy(&f, bits + i, 4);
    return f;
}

inline int ilog2_floor(uint32_t v)          // av_log2: floor(log2(v)), 0 for v == 0
{
    return v ? 31 - __builtin_clz(v) : 0;
}
inline int ilog(uint32_t v) { return v ? 32 - __builtin_clz(v) : 0; }          // opus_ilog
inline int imin(int a, int b) { return a < b ? a : b; }
inline int imax(int a, int b) { return a > b ? a : b; }
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
inline int round_mul16(int a, int b) { return (a * b + 16384) >> 15; }

uint32_t isqrt(uint32_t a)                   // ff_sqrt is floor(sqrt(a)) on every argument the decoder passes
{
    uint32_t r = (uint32_t)std::sqrt((double)a);
    while ((uint64_t)r * r > a) r--;
    while ((uint64_t)(r + 1) * (r + 1) <= a) r++;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Range decoder (RFC 6716 section 4.1 as the reference implements it): the coded symbols are read
// from the front of the frame, raw bits from its back.
// ---------------------------------------------------------------------------------------------
class RangeDecoder {
public:
    void start(const uint8_t *data, int size)
    {
        buf_ = data;
        size_ = size > 0 ? (uint32_t)size : 0;
        at_ = 0;
        last_ = next_byte();
        rng_ = 128;
        val_ = 127 - (uint32_t)(last_ >> 1);
        bits_ = 9;
        refill();
        tail_ = data + size_;
        tail_left_ = size_;
        window_ = 0;
        window_bits_ = 0;
    }
    uint32_t frame_bits() const { return size_ * 8; }
    uint32_t range() const { return rng_; }
    uint32_t tell() const { return bits_ - (uint32_t)ilog2_floor(rng_) - 1; }
    uint32_t tell_frac() const                       // in 1/8 bit
    {
        uint32_t lg = (uint32_t)ilog2_floor(rng_) + 1;
        uint32_t r = rng_ >> (lg - 16);
        for (int i = 0; i < 3; i++) {
            r = r * r >> 15;
            const uint32_t bit = r >> 16;
            lg = lg << 1 | bit;
            r >>= bit;
        }
        return (bits_ << 3) - lg;
    }
    void skip_to_end() { bits_ += frame_bits() - tell(); }       // a silent frame consumes the whole budget

    unsigned cdf(const uint16_t *model)              // model[0] = total, then the cumulative counts
    {
        const uint32_t total = model[0];
        const uint16_t *c = model + 1;
        const uint32_t scale = rng_ / total;
        uint32_t sym = val_ / scale + 1;
        sym = total - (sym < total ? sym : total);
        unsigned k = 0;
        while (c[k] <= sym) k++;
        take(scale, k ? c[k - 1] : 0, c[k], total);
        return k;
    }
    unsigned bit_logp(unsigned bits)                 // 1 with probability 2^-bits
    {
        const uint32_t scale = rng_ >> bits;
        unsigned k;
        if (val_ >= scale) {
            val_ -= scale;
            rng_ -= scale;
            k = 0;
        } else {
            rng_ = scale;
            k = 1;
        }
        refill();
        return k;
    }
    uint32_t raw(unsigned count)
    {
        while (tail_left_ && window_bits_ < count) {
            window_ |= (uint32_t)*--tail_ << window_bits_;
            window_bits_ += 8;
            tail_left_--;
        }
        const uint32_t v = window_ & ((1u << count) - 1);
        window_ >>= count;
        window_bits_ -= count;
        bits_ += count;
        return v;
    }
    uint32_t uniform(uint32_t size)
    {
        const unsigned bits = (unsigned)ilog(size - 1);
        const uint32_t total = bits > 8 ? ((size - 1) >> (bits - 8)) + 1 : size;
        const uint32_t scale = rng_ / total;
        uint32_t k = val_ / scale + 1;
        k = total - (k < total ? k : total);
        take(scale, k, k + 1, total);
        if (bits > 8) {
            k = k << (bits - 8) | raw(bits - 8);
            return k < size - 1 ? k : size - 1;
        }
        return k;
    }
    int laplace(uint32_t first, int decay)
    {
        int value = 0;
        const uint32_t scale = rng_ >> 15;
        uint32_t center = val_ / scale + 1, low = 0, width = first;
        center = 32768u - (center < 32768u ? center : 32768u);
        if (center >= width) {
            value = 1;
            low = width;
            width = 1 + ((32768u - 32u - width) * (uint32_t)(16384 - decay) >> 15);
            while (width > 1 && center >= low + 2 * width) {
                value++;
                width *= 2;
                low += width;
                width = (((width - 2) * (uint32_t)decay) >> 15) + 1;
            }
            if (width <= 1) {
                const int distance = (int)((center - low) >> 1);
                value += distance;
                low += 2u * (uint32_t)distance;
            }
            if (center < low + width) value = -value;
            else low += width;
        }
        const uint32_t high = low + width;
        take(scale, low, high < 32768u ? high : 32768u, 32768u);
        return value;
    }
    uint32_t step(int k0)                            // probability 3 up to k0, 1 beyond
    {
        const uint32_t n = (uint32_t)k0 + 1, total = n * 3 + (uint32_t)k0;
        const uint32_t scale = rng_ / total;
        uint32_t sym = val_ / scale + 1;
        sym = total - (sym < total ? sym : total);
        const uint32_t k = sym < n * 3 ? sym / 3 : sym - n * 2;
        if (k <= (uint32_t)k0) take(scale, 3 * k, 3 * (k + 1), total);
        else take(scale, (k - 1 - (uint32_t)k0) + 3 * n, (k - (uint32_t)k0) + 3 * n, total);
        return k;
    }
    uint32_t triangular(int qn)
    {
        const uint32_t half = (uint32_t)(qn >> 1) + 1, total = half * half;
        const uint32_t scale = rng_ / total;
        uint32_t center = val_ / scale + 1;
        center = total - (center < total ? center : total);
        uint32_t k, low, width;
        if (center < total >> 1) {
            k = (isqrt(8 * center + 1) - 1) >> 1;
            low = k * (k + 1) >> 1;
            width = k + 1;
        } else {
            k = (2 * (uint32_t)(qn + 1) - isqrt(8 * (total - center - 1) + 1)) >> 1;
            low = total - (((uint32_t)qn + 1 - k) * ((uint32_t)qn + 2 - k) >> 1);
            width = (uint32_t)qn + 1 - k;
        }
        take(scale, low, low + width, total);
        return k;
    }

private:
    int next_byte() { return at_ < size_ ? buf_[at_++] : 0; }
    void refill()
    {
        while (rng_ <= (1u << 23)) {
            const int nxt = next_byte();
            const uint32_t sym = (uint32_t)((last_ << 8 | nxt) >> 1) & 0xffu;    // the coder is 7 bits into its first byte
            last_ = nxt;
            val_ = ((val_ << 8) | (sym ^ 0xffu)) & 0x7fffffffu;
            rng_ <<= 8;
            bits_ += 8;
        }
    }
    void take(uint32_t scale, uint32_t low, uint32_t high, uint32_t total)
    {
        val_ -= scale * (total - high);
        rng_ = low ? scale * (high - low) : rng_ - scale * (total - high);
        refill();
    }
    const uint8_t *buf_ = nullptr, *tail_ = nullptr;
    uint32_t size_ = 0, at_ = 0, tail_left_ = 0;
    int last_ = 0;
    uint32_t rng_ = 0, val_ = 0, bits_ = 0, window_ = 0, window_bits_ = 0;
};

// ---------------------------------------------------------------------------------------------
// CELT frame decoder: everything of ff_celt_decode_frame except the per-channel transform loop.
// ---------------------------------------------------------------------------------------------
struct ChannelMemory {
    float energy[kBands];
    float prev_energy[2][kBands];
    uint8_t collapse[kBands];
    int pf_period;
    float pf_gains[3];
};

struct FrameInfo {
    int blocks = 1;
    float imdct_scale = 1.0f;
    int pf_period = 0;
    float pf_gains[3] = { 0, 0, 0 };
};

class CeltDecoder {
public:
    void reset(int output_channels)                 // ff_celt_init + ff_celt_flush (:3733-3757, :3774-3810), host half
    {
        out_channels_ = output_channels;
        for (ChannelMemory &m : ch_) {
            std::memset(&m, 0, sizeof(m));
            for (int j = 0; j < kBands; j++) m.prev_energy[0][j] = m.prev_energy[1][j] = kSilenceEnergy;
        }
        seed_ = 0;
    }
    const float *coeffs(int c) const { return coeffs_[c]; }

    void decode(RangeDecoder &rc, int coded_channels, int frame_size, int endband, FrameInfo &info)
    {
        rc_ = &rc;
        C_ = coded_channels;
        start_ = 0;
        end_ = endband;
        framebits_ = (int)rc.frame_bits();
        LM_ = ilog2_floor((uint32_t)(frame_size / kShortBlock));
        for (ChannelMemory &m : ch_) std::memset(m.collapse, 0, sizeof(m.collapse));

        int consumed = (int)rc.tell();
        bool silence = false;
        if (consumed >= framebits_) silence = true;
        else if (consumed == 1) silence = rc.bit_logp(15) != 0;
        if (silence) {
            consumed = framebits_;
            rc.skip_to_end();
        }
        consumed = postfilter_params(consumed);
        bool transient = false;
        if (LM_ != 0 && consumed + 3 <= framebits_) transient = rc.bit_logp(3) != 0;
        blocks_ = transient ? 1 << LM_ : 1;

        if (C_ == 1)
            for (int i = 0; i < kBands; i++) ch_[0].energy[i] = ch_[0].energy[i] > ch_[1].energy[i] ? ch_[0].energy[i] : ch_[1].energy[i];

        coarse_energy();
        tf_changes(transient);
        allocation();
        fine_energy();
        all_bands();
        const bool anticollapse = anticollapse_rsv_ ? rc.raw(1) != 0 : false;
        final_energy(framebits_ - (int)rc.tell());

        for (int c = 0; c < C_; c++) {
            if (anticollapse) anti_collapse(ch_[c], coeffs_[c]);
            denormalise(ch_[c], coeffs_[c]);
        }
        info.imdct_scale = 1.0f;
        if (out_channels_ < C_) {                    // stereo -> mono downmix (:3663-3666)
            for (int i = 0; i < frame_size; i++) coeffs_[0][i] += coeffs_[1][i] * 1.0f;
            info.imdct_scale = 0.5f;
        } else if (out_channels_ > C_) {
            std::memcpy(coeffs_[1], coeffs_[0], (size_t)frame_size * sizeof(float));
        }
        if (silence) {
            for (ChannelMemory &m : ch_)
                for (int j = 0; j < kBands; j++) m.energy[j] = kSilenceEnergy;
            std::memset(coeffs_, 0, sizeof(coeffs_));
        }
        info.blocks = blocks_;
        info.pf_period = ch_[0].pf_period;           // both channels always hold the same parameters (:3404-3411)
        std::memcpy(info.pf_gains, ch_[0].pf_gains, sizeof(info.pf_gains));

        // memory update (:3704-3729)
        if (C_ == 1) std::memcpy(ch_[1].energy, ch_[0].energy, sizeof(ch_[0].energy));
        for (ChannelMemory &m : ch_) {
            if (!transient) {
                std::memcpy(m.prev_energy[1], m.prev_energy[0], sizeof(m.prev_energy[0]));
                std::memcpy(m.prev_energy[0], m.energy, sizeof(m.prev_energy[0]));
            } else {
                for (int j = 0; j < kBands; j++) m.prev_energy[0][j] = m.prev_energy[0][j] < m.energy[j] ? m.prev_energy[0][j] : m.energy[j];
            }
            for (int j = 0; j < start_; j++) { m.prev_energy[0][j] = kSilenceEnergy; m.energy[j] = 0.0f; }
            for (int j = end_; j < kBands; j++) { m.prev_energy[0][j] = kSilenceEnergy; m.energy[j] = 0.0f; }
        }
        seed_ = rc.range();
    }

private:
    uint32_t noise() { return seed_ = 1664525u * seed_ + 1013904223u; }
    int band_lo(int i) const { return k_celt_freq_bands[i] << LM_; }
    int band_n(int i) const { return k_celt_freq_range[i] << LM_; }

    int postfilter_params(int consumed)             // :3380-3418
    {
        static const float taps[3][3] = { { 0.3066406250f, 0.2170410156f, 0.1296386719f },
                                          { 0.4638671875f, 0.2680664062f, 0.0f },
                                          { 0.7998046875f, 0.1000976562f, 0.0f } };
        RangeDecoder &rc = *rc_;
        for (ChannelMemory &m : ch_) m.pf_gains[0] = m.pf_gains[1] = m.pf_gains[2] = 0.0f;
        if (start_ == 0 && consumed + 16 <= framebits_) {
            if (rc.bit_logp(1)) {
                const int octave = (int)rc.uniform(6);
                const int period = (16 << octave) + (int)rc.raw(4 + (unsigned)octave) - 1;
                const float gain = 0.09375f * (float)(rc.raw(3) + 1);
                const int tapset = (rc.tell() + 2 <= (uint32_t)framebits_) ? (int)rc.cdf(k_celt_model_tapset) : 0;
                for (ChannelMemory &m : ch_) {
                    m.pf_period = imax(period, 15);
                    for (int t = 0; t < 3; t++) m.pf_gains[t] = gain * taps[tapset][t];
                }
            }
            consumed = (int)rc.tell();
        }
        return consumed;
    }

    void coarse_energy()                            // :2128-2176
    {
        RangeDecoder &rc = *rc_;
        float prev[2] = { 0.0f, 0.0f }, alpha, beta;
        const uint8_t *model;
        if (rc.tell() + 3 <= (uint32_t)framebits_ && rc.bit_logp(3)) {          // intra frame
            alpha = 0.0f;
            beta = 1.0f - 4915.0f / 32768.0f;
            model = k_celt_coarse_energy_dist + (LM_ * 2 + 1) * 42;
        } else {
            alpha = table_f32(k_celt_alpha_coef_bits, LM_);
            beta = 1.0f - table_f32(k_celt_beta_coef_bits, LM_);
            model = k_celt_coarse_energy_dist + (LM_ * 2) * 42;
        }
        for (int i = 0; i < kBands; i++) {
            for (int c = 0; c < C_; c++) {
                float &e = ch_[c].energy[i];
                if (i < start_ || i >= end_) { e = 0.0f; continue; }
                const int available = framebits_ - (int)rc.tell();
                float value;
                if (available >= 15) {
                    const int k = imin(i, 20) << 1;
                    value = (float)rc.laplace((uint32_t)model[k] << 7, model[k + 1] << 6);
                } else if (available >= 2) {
                    const int x = (int)rc.cdf(k_celt_model_energy_small);
                    value = (float)((x >> 1) ^ -(x & 1));
                } else if (available >= 1) {
                    value = -(float)rc.bit_logp(1);
                } else {
                    value = -1.0f;
                }
                e = (-9.0f > e ? -9.0f : e) * alpha + prev[c] + value;
                prev[c] += beta * value;
            }
        }
    }

    void fine_energy()                              // :2178-2195
    {
        for (int i = start_; i < end_; i++) {
            if (!fine_bits_[i]) continue;
            for (int c = 0; c < C_; c++) {
                const int q = (int)rc_->raw((unsigned)fine_bits_[i]);
                ch_[c].energy[i] += ((float)q + 0.5f) * (float)(1 << (14 - fine_bits_[i])) / 16384.0f - 0.5f;
            }
        }
    }

    void final_energy(int bits_left)                // :2197-2216
    {
        for (int priority = 0; priority < 2; priority++)
            for (int i = start_; i < end_ && bits_left >= C_; i++) {
                if (fine_priority_[i] != priority || fine_bits_[i] >= kMaxFineBits) continue;
                for (int c = 0; c < C_; c++) {
                    const int q = (int)rc_->raw(1);
                    ch_[c].energy[i] += ((float)q - 0.5f) * (float)(1 << (14 - fine_bits_[i] - 1)) / 16384.0f;
                    bits_left--;
                }
            }
    }

    void tf_changes(bool transient)                 // :2218-2243
    {
        RangeDecoder &rc = *rc_;
        const int t = transient ? 1 : 0;
        int diff = 0, changed = 0, select = 0;
        unsigned bits = transient ? 2 : 4;
        int consumed = (int)rc.tell();
        const int select_bit = (LM_ != 0 && consumed + (int)bits + 1 <= framebits_) ? 1 : 0;
        for (int i = start_; i < end_; i++) {
            if (consumed + (int)bits + select_bit <= framebits_) {
                diff ^= (int)rc.bit_logp(bits);
                consumed = (int)rc.tell();
                changed |= diff;
            }
            tf_change_[i] = diff;
            bits = transient ? 4 : 5;
        }
        const int8_t *sel = k_celt_tf_select + (LM_ * 2 + t) * 4;               // [select][changed]
        if (select_bit && sel[0 + changed] != sel[2 + changed]) select = (int)rc.bit_logp(1);
        for (int i = start_; i < end_; i++) tf_change_[i] = sel[select * 2 + tf_change_[i]];
    }

    void allocation()                               // :2245-2575
    {
        RangeDecoder &rc = *rc_;
        int cap[kBands], boost[kBands], threshold[kBands], lo_bits[kBands], span_bits[kBands], trim_offset[kBands];
        const int C = C_, LM = LM_, stereo_shift = C - 1;

        spread_ = 2;
        if ((int)rc.tell() + 4 <= framebits_) spread_ = (int)rc.cdf(k_celt_model_spread);

        for (int i = 0; i < kBands; i++)
            cap[i] = (k_celt_static_caps[(LM * 2 + stereo_shift) * kBands + i] + 64) * k_celt_freq_range[i] << stereo_shift << LM >> 2;

        // band boosts
        int totalbits = framebits_ << 3, dynalloc = 6;
        int consumed = (int)rc.tell_frac();
        for (int i = start_; i < end_; i++) {
            boost[i] = 0;
            int quanta = k_celt_freq_range[i] << stereo_shift << LM;
            quanta = imin(quanta << 3, imax(6 << 3, quanta));
            int cost = dynalloc;
            while (consumed + (cost << 3) < totalbits && boost[i] < cap[i]) {
                const unsigned add = rc.bit_logp((unsigned)cost);
                consumed = (int)rc.tell_frac();
                if (!add) break;
                boost[i] += quanta;
                totalbits -= quanta;
                cost = 1;
            }
            if (boost[i]) dynalloc = imax(2, dynalloc - 1);
        }

        int alloctrim = 5;
        if (consumed + (6 << 3) <= totalbits) alloctrim = (int)rc.cdf(k_celt_model_alloc_trim);

        // reservations: anti-collapse, skip, intensity, dual stereo
        totalbits = (framebits_ << 3) - (int)rc.tell_frac() - 1;
        anticollapse_rsv_ = (blocks_ > 1 && LM >= 2 && totalbits >= ((LM + 2) << 3)) ? 1 << 3 : 0;
        totalbits -= anticollapse_rsv_;
        const int skip_rsv = totalbits >= 1 << 3 ? 1 << 3 : 0;
        totalbits -= skip_rsv;
        int intensity_rsv = 0, dual_rsv = 0;
        if (C == 2) {
            intensity_rsv = k_celt_log2_frac[end_ - start_];
            if (intensity_rsv <= totalbits) {
                totalbits -= intensity_rsv;
                if (totalbits >= 1 << 3) {
                    dual_rsv = 1 << 3;
                    totalbits -= 1 << 3;
                }
            } else {
                intensity_rsv = 0;
            }
        }

        for (int i = start_; i < end_; i++) {
            const int trim = alloctrim - 5 - LM;
            const int band = k_celt_freq_range[i] * (end_ - i - 1);
            const int duration = LM + 3;
            const int scale = duration + stereo_shift;
            threshold[i] = imax(3 * k_celt_freq_range[i] << duration >> 4, C << 3);
            trim_offset[i] = trim * (band << scale) >> 6;
            if (k_celt_freq_range[i] << LM == 1) trim_offset[i] -= C << 3;
        }

        auto vector_bits = [&](int row, int i) { return k_celt_freq_range[i] * k_celt_static_alloc[row * kBands + i] << stereo_shift << LM >> 2; };

        // coarse search over the static allocation vectors
        int low = 1, high = 10;
        while (low <= high) {
            const int center = (low + high) >> 1;
            int total = 0;
            bool done = false;
            for (int i = end_ - 1; i >= start_; i--) {
                int bandbits = vector_bits(center, i);
                if (bandbits) bandbits = imax(0, bandbits + trim_offset[i]);
                bandbits += boost[i];
                if (bandbits >= threshold[i] || done) {
                    done = true;
                    total += imin(bandbits, cap[i]);
                } else if (bandbits >= C << 3) {
                    total += C << 3;
                }
            }
            if (total > totalbits) high = center - 1;
            else low = center + 1;
        }
        high = low--;

        int skip_start = start_;
        for (int i = start_; i < end_; i++) {
            lo_bits[i] = vector_bits(low, i);
            span_bits[i] = high >= 11 ? cap[i] : vector_bits(high, i);
            if (lo_bits[i]) lo_bits[i] = imax(0, lo_bits[i] + trim_offset[i]);
            if (span_bits[i]) span_bits[i] = imax(0, span_bits[i] + trim_offset[i]);
            if (low) lo_bits[i] += boost[i];
            span_bits[i] += boost[i];
            if (boost[i]) skip_start = i;
            span_bits[i] = imax(0, span_bits[i] - lo_bits[i]);
        }

        // fine search: interpolate between the two vectors in 1/64 steps
        low = 0;
        high = 1 << 6;
        for (int step = 0; step < 6; step++) {
            const int center = (low + high) >> 1;
            int total = 0;
            bool done = false;
            for (int j = end_ - 1; j >= start_; j--) {
                const int bandbits = lo_bits[j] + (center * span_bits[j] >> 6);
                if (bandbits >= threshold[j] || done) {
                    done = true;
                    total += imin(bandbits, cap[j]);
                } else if (bandbits >= C << 3) {
                    total += C << 3;
                }
            }
            if (total > totalbits) high = center;
            else low = center;
        }

        int total = 0;
        {
            bool done = false;
            for (int i = end_ - 1; i >= start_; i--) {
                int bandbits = lo_bits[i] + (low * span_bits[i] >> 6);
                if (bandbits >= threshold[i] || done) done = true;
                else bandbits = (bandbits >= C << 3) ? C << 3 : 0;
                bandbits = imin(bandbits, cap[i]);
                pulses_[i] = bandbits;
                total += bandbits;
            }
        }

        // band skipping
        for (coded_bands_ = end_;; coded_bands_--) {
            const int j = coded_bands_ - 1;
            if (j == skip_start) {
                totalbits += skip_rsv;
                break;
            }
            int remaining = totalbits - total;
            const int width = k_celt_freq_bands[j + 1] - k_celt_freq_bands[start_];
            const int per_bin = remaining / width;
            remaining -= per_bin * width;
            int alloc = pulses_[j] + per_bin * k_celt_freq_range[j] + imax(0, remaining - (k_celt_freq_bands[j] - k_celt_freq_bands[start_]));
            if (alloc >= imax(threshold[j], (C + 1) << 3)) {
                if (rc.bit_logp(1)) break;
                total += 1 << 3;
                alloc -= 1 << 3;
            }
            total -= pulses_[j];
            if (intensity_rsv) {
                total -= intensity_rsv;
                intensity_rsv = k_celt_log2_frac[j - start_];
                total += intensity_rsv;
            }
            pulses_[j] = (alloc >= C << 3) ? C << 3 : 0;
            total += pulses_[j];
        }

        intensity_ = 0;
        dual_stereo_ = 0;
        if (intensity_rsv) intensity_ = start_ + (int)rc.uniform((uint32_t)(coded_bands_ + 1 - start_));
        if (intensity_ <= start_) totalbits += dual_rsv;
        else if (dual_rsv) dual_stereo_ = (int)rc.bit_logp(1);

        // left-over bits go to the coded bands, lowest first
        {
            int remaining = totalbits - total;
            const int width = k_celt_freq_bands[coded_bands_] - k_celt_freq_bands[start_];
            const int per_bin = remaining / width;
            remaining -= per_bin * width;
            for (int i = start_; i < coded_bands_; i++) {
                const int bits = imin(remaining, k_celt_freq_range[i]);
                pulses_[i] += bits + per_bin * k_celt_freq_range[i];
                remaining -= bits;
            }
        }

        // split each band's bits between fine energy and PVQ
        int extrabits = 0, i = start_;
        for (; i < coded_bands_; i++) {
            const int N = k_celt_freq_range[i] << LM;
            const int prev_extra = extrabits;
            pulses_[i] += extrabits;
            if (N > 1) {
                extrabits = imax(0, pulses_[i] - cap[i]);
                pulses_[i] -= extrabits;
                const int dof = N * C + ((C == 2 && N > 2 && !dual_stereo_ && i < intensity_) ? 1 : 0);
                const int temp = dof * (k_celt_log_freq_range[i] + (LM << 3));
                int offset = (temp >> 1) - dof * 21;
                if (N == 2) offset += dof << 1;
                if (pulses_[i] + offset < 2 * (dof << 3)) offset += temp >> 2;
                else if (pulses_[i] + offset < 3 * (dof << 3)) offset += temp >> 3;
                const int fine = (pulses_[i] + offset + (dof << 2)) / (dof << 3);
                const int max_bits = imax(imin((pulses_[i] >> 3) >> stereo_shift, kMaxFineBits), 0);
                fine_bits_[i] = clampi(fine, 0, max_bits);
                fine_priority_[i] = (fine_bits_[i] * (dof << 3) >= pulses_[i] + offset) ? 1 : 0;
                pulses_[i] -= fine_bits_[i] << stereo_shift << 3;
            } else {
                extrabits = imax(0, pulses_[i] - (C << 3));
                pulses_[i] -= extrabits;
                fine_bits_[i] = 0;
                fine_priority_[i] = 1;
            }
            if (extrabits > 0) {
                int fineextra = imin(extrabits >> (C + 2), kMaxFineBits - fine_bits_[i]);
                fine_bits_[i] += fineextra;
                fineextra <<= C + 2;
                fine_priority_[i] = (fineextra >= extrabits - prev_extra) ? 1 : 0;
                extrabits -= fineextra;
            }
        }
        balance_ = extrabits;
        for (; i < end_; i++) {                      // skipped bands: everything to fine energy
            fine_bits_[i] = pulses_[i] >> stereo_shift >> 3;
            pulses_[i] = 0;
            fine_priority_[i] = fine_bits_[i] < 1 ? 1 : 0;
        }
    }

    // ---- PVQ shape decoding (:2577-2915) ----
    static const uint32_t *pvq_row(unsigned r) { return k_celt_pvq_u + k_celt_pvq_u_row[r]; }
    static uint32_t pvq_u(unsigned n, unsigned k) { return n < k ? pvq_row(n)[k] : pvq_row(k)[n]; }

    static uint64_t pulses_from_index(unsigned N, unsigned K, uint32_t idx, int *y)   // celt_cwrsi
    {
        uint64_t norm = 0;
        auto emit = [&](int v) { norm += (uint64_t)(v * v); *y++ = v; };
        while (N > 2) {
            if (K >= N) {                            // many pulses
                const uint32_t *row = pvq_row(N);
                uint32_t p = row[K + 1];
                const int s = idx >= p ? -1 : 0;
                idx -= p & (uint32_t)s;
                const int k0 = (int)K;
                if (row[N] > idx) {
                    K = N;
                    do p = pvq_row(--K)[N];
                    while (p > idx);
                } else {
                    for (p = row[K]; p > idx; p = row[K]) K--;
                }
                idx -= p;
                emit((k0 - (int)K + s) ^ s);
            } else {                                 // many dimensions
                uint32_t p = pvq_row(K)[N];
                const uint32_t q = pvq_row(K + 1)[N];
                if (p <= idx && idx < q) {
                    idx -= p;
                    *y++ = 0;
                } else {
                    const int s = idx >= q ? -1 : 0;
                    idx -= q & (uint32_t)s;
                    const int k0 = (int)K;
                    do p = pvq_row(--K)[N];
                    while (p > idx);
                    idx -= p;
                    emit((k0 - (int)K + s) ^ s);
                }
            }
            N--;
        }
        {                                            // two dimensions left, then one
            const uint32_t p = 2 * K + 1;
            int s = idx >= p ? -1 : 0;
            idx -= p & (uint32_t)s;
            const int k0 = (int)K;
            K = (idx + 1) / 2;
            if (K) idx -= 2 * K - 1;
            emit((k0 - (int)K + s) ^ s);
            s = -(int)idx;
            emit(((int)K + s) ^ s);
        }
        return norm;
    }

    static void rotate_pass(float *X, int len, int stride, float c, float s)     // celt_exp_rotation1
    {
        float *p = X;
        for (int i = 0; i < len - stride; i++, p++) {
            const float x1 = p[0], x2 = p[stride];
            p[stride] = c * x2 + s * x1;
            p[0] = c * x1 - s * x2;
        }
        for (int i = len - 2 * stride - 1; i >= 0; i--) {
            const float x1 = X[i], x2 = X[i + stride];
            X[i + stride] = c * x2 + s * x1;
            X[i] = c * x1 - s * x2;
        }
    }

    static void spread_rotation(float *X, unsigned len, unsigned stride, unsigned K, int spread)   // celt_exp_rotation
    {
        if (2 * K >= len || spread == 0) return;
        const float gain = (float)len / (float)(len + (20u - 5u * (unsigned)spread) * K);
        const float theta = (float)(3.14159265358979323846264338327950288L * gain * gain / 4);
        const float c = (float)std::cos((double)theta), s = (float)std::sin((double)theta);
        unsigned stride2 = 0;
        if (len >= stride << 3) {
            stride2 = 1;
            while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
        }
        len /= stride;
        for (unsigned i = 0; i < stride; i++) {
            if (stride2) rotate_pass(X + i * len, (int)len, (int)stride2, s, c);
            rotate_pass(X + i * len, (int)len, 1, c, s);
        }
    }

    uint32_t shape(float *X, unsigned N, unsigned K, unsigned blocks, float gain)   // celt_alg_unquant
    {
        int y[176];
        const uint32_t idx = rc_->uniform(pvq_u(N, K) + pvq_u(N, K + 1));
        const float g = gain / std::sqrt((float)pulses_from_index(N, K, idx, y));
        for (unsigned i = 0; i < N; i++) X[i] = g * (float)y[i];
        spread_rotation(X, N, blocks, K, spread_);
        if (blocks <= 1) return 1;
        const unsigned per = N / blocks;
        uint32_t mask = 0;
        for (unsigned b = 0; b < blocks; b++)
            for (unsigned j = 0; j < per; j++) mask |= (uint32_t)(y[b * per + j] != 0) << b;
        return mask;
    }

    static void renormalise(float *X, int N, float gain)
    {
        float g = 1e-15f;
        for (int i = 0; i < N; i++) g += X[i] * X[i];
        g = gain / std::sqrt(g);
        for (int i = 0; i < N; i++) X[i] *= g;
    }

    static void merge_stereo(float *X, float *Y, float mid, int N)
    {
        float xp = 0.0f, side = 0.0f;
        for (int i = 0; i < N; i++) {
            xp += X[i] * Y[i];
            side += Y[i] * Y[i];
        }
        xp *= mid;
        const float e0 = mid * mid + side - 2 * xp, e1 = mid * mid + side + 2 * xp;
        if (e0 < 6e-4f || e1 < 6e-4f) {
            for (int i = 0; i < N; i++) Y[i] = X[i];
            return;
        }
        const float g0 = 1.0f / std::sqrt(e0), g1 = 1.0f / std::sqrt(e1);
        for (int i = 0; i < N; i++) {
            const float m = mid * X[i], sd = Y[i];
            X[i] = g0 * (m - sd);
            Y[i] = g1 * (m + sd);
        }
    }

    // time <-> frequency ordering of the blocks of a band (hadamard = the natural-order table for long blocks)
    void reorder(float *X, int N0, int stride, bool hadamard, bool interleave)
    {
        const uint8_t *order = k_celt_hadamard_ordery + stride - 2;
        float *tmp = scratch_;
        for (int i = 0; i < stride; i++) {
            const int src_blk = hadamard ? order[i] : i;
            for (int j = 0; j < N0; j++) {
                if (interleave) tmp[j * stride + i] = X[src_blk * N0 + j];
                else tmp[src_blk * N0 + j] = X[j * stride + i];
            }
        }
        std::memcpy(X, tmp, (size_t)(N0 * stride) * sizeof(float));
    }

    static void haar(float *X, int N0, int stride)
    {
        N0 >>= 1;
        for (int i = 0; i < stride; i++)
            for (int j = 0; j < N0; j++) {
                float &a = X[stride * (2 * j) + i], &b = X[stride * (2 * j + 1) + i];
                const float x0 = a, x1 = b;
                a = (float)((double)(x0 + x1) * 0.70710678118654752440);
                b = (float)((double)(x0 - x1) * 0.70710678118654752440);
            }
    }

    static int cos_q15(int x)                        // celt_cos
    {
        x = (int16_t)((x * x + 4096) >> 13);
        x = (int16_t)((32767 - x) + round_mul16(x, -7651 + round_mul16(x, 8277 + round_mul16(-626, x))));
        return (int16_t)(1 + x);
    }
    static int log2_tan(int isin, int icos)
    {
        const int lc = ilog((uint32_t)icos), ls = ilog((uint32_t)isin);
        icos <<= 15 - lc;
        isin <<= 15 - ls;
        return (ls << 11) - (lc << 11) + round_mul16(isin, round_mul16(isin, -2597) + 7932) - round_mul16(icos, round_mul16(icos, -2597) + 7932);
    }
    static int theta_resolution(int N, int b, int offset, int pulse_cap, bool stereo)    // celt_compute_qn
    {
        int N2 = 2 * N - 1;
        if (stereo && N == 2) N2--;
        const int qb = imin(imin(b - pulse_cap - (4 << 3), (b + N2 * offset) / N2), 8 << 3);
        return qb < 4 ? 1 : ((k_celt_qn_exp2[qb & 7] >> (14 - (qb >> 3))) + 1) >> 1 << 1;
    }
    static int bits_to_pulses(const uint8_t *cache, int bits)
    {
        int low = 0, high = cache[0];
        bits--;
        for (int i = 0; i < 6; i++) {
            const int center = (low + high + 1) >> 1;
            if (cache[center] >= bits) high = center;
            else low = center;
        }
        return (bits - (low == 0 ? -1 : cache[low]) <= cache[high] - bits) ? low : high;
    }
    static int pulses_to_bits(const uint8_t *cache, int pulses) { return pulses == 0 ? 0 : cache[pulses] + 1; }

    // celt_decode_band (:2917-3266).  Y != nullptr: a stereo band (X = mid channel, Y = side channel on entry to the merge).
    uint32_t band(int i, float *X, float *Y, int N, int b, unsigned blocks, float *lowband, int duration, float *lowband_out, int level,
                  float gain, float *lowband_scratch, int fill)
    {
        RangeDecoder &rc = *rc_;
        const bool stereo = Y != nullptr;
        const unsigned N0 = (unsigned)N;
        int n_per_block = N / (int)blocks, n_per_block0 = n_per_block;
        int B0 = (int)blocks, time_divide = 0, recombine = 0, inv = 0;
        const bool longblocks = B0 == 1;
        float mid = 0.0f, side = 0.0f;
        uint32_t cm = 0;
        bool split = stereo;

        if (N == 1) {                                // one line: just a sign
            float *x = X;
            for (int c = 0; c <= (stereo ? 1 : 0); c++) {
                unsigned sign = 0;
                if (budget_ >= 1 << 3) {
                    sign = rc.raw(1);
                    budget_ -= 1 << 3;
                    b -= 1 << 3;
                }
                x[0] = sign ? -1.0f : 1.0f;
                x = Y;
            }
            if (lowband_out) lowband_out[0] = X[0];
            return 1;
        }

        if (!stereo && level == 0) {
            int tf = tf_change_[i];
            if (tf > 0) recombine = tf;
            if (lowband && (recombine || ((n_per_block & 1) == 0 && tf < 0) || B0 > 1)) {
                std::memcpy(lowband_scratch, lowband, (size_t)N * sizeof(float));
                lowband = lowband_scratch;
            }
            for (int k = 0; k < recombine; k++) {    // fewer, longer blocks
                if (lowband) haar(lowband, N >> k, 1 << k);
                fill = k_celt_bit_interleave[fill & 0xF] | k_celt_bit_interleave[fill >> 4] << 2;
            }
            blocks >>= recombine;
            n_per_block <<= recombine;
            while ((n_per_block & 1) == 0 && tf < 0) {   // more, shorter blocks
                if (lowband) haar(lowband, n_per_block, (int)blocks);
                fill |= fill << blocks;
                blocks <<= 1;
                n_per_block >>= 1;
                time_divide++;
                tf++;
            }
            B0 = (int)blocks;
            n_per_block0 = n_per_block;
            if (B0 > 1 && lowband) reorder(lowband, n_per_block >> recombine, B0 << recombine, longblocks, false);
        }

        const uint8_t *cache = k_celt_cache_bits + k_celt_cache_index[(duration + 1) * kBands + i];
        if (!stereo && duration >= 0 && b > cache[cache[0]] + 12 && N > 2) {      // more bits than one codebook holds: halve
            N >>= 1;
            Y = X + N;
            split = true;
            duration -= 1;
            if (blocks == 1) fill = (fill & 1) | (fill << 1);
            blocks = (blocks + 1) >> 1;
        }

        if (split) {
            const int pulse_cap = k_celt_log_freq_range[i] + duration * 8;
            const int offset = (pulse_cap >> 1) - (stereo && N == 2 ? 16 : 4);
            const int qn = (stereo && i >= intensity_) ? 1 : theta_resolution(N, b, offset, pulse_cap, stereo);
            const int tell = (int)rc.tell_frac();
            int itheta = 0;
            if (qn != 1) {
                if (stereo && N > 2) itheta = (int)rc.step(qn / 2);
                else if (stereo || B0 > 1) itheta = (int)rc.uniform((uint32_t)qn + 1);
                else itheta = (int)rc.triangular(qn);
                itheta = itheta * 16384 / qn;
            } else if (stereo) {
                inv = (b > 2 << 3 && budget_ > 2 << 3) ? (int)rc.bit_logp(2) : 0;
                itheta = 0;
            }
            const int qalloc = (int)rc.tell_frac() - tell;
            b -= qalloc;

            const int orig_fill = fill;
            int imid, iside, delta;
            if (itheta == 0) {
                imid = 32767;
                iside = 0;
                fill &= (1 << blocks) - 1;
                delta = -16384;
            } else if (itheta == 16384) {
                imid = 0;
                iside = 32767;
                fill &= ((1 << blocks) - 1) << blocks;
                delta = 16384;
            } else {
                imid = cos_q15(itheta);
                iside = cos_q15(16384 - itheta);
                delta = round_mul16((N - 1) << 7, log2_tan(iside, imid));
            }
            mid = (float)imid / 32768.0f;
            side = (float)iside / 32768.0f;

            if (N == 2 && stereo) {                  // mid and side are orthogonal: one sign bit codes the side
                int mbits = b;
                const int sbits = (itheta != 0 && itheta != 16384) ? 1 << 3 : 0;
                mbits -= sbits;
                const bool swap = itheta > 8192;
                budget_ -= qalloc + sbits;
                float *x2 = swap ? Y : X, *y2 = swap ? X : Y;
                int sign = sbits ? (int)rc.raw(1) : 0;
                sign = 1 - 2 * sign;
                cm = band(i, x2, nullptr, N, mbits, blocks, lowband, duration, lowband_out, level, gain, lowband_scratch, orig_fill);
                y2[0] = (float)-sign * x2[1];
                y2[1] = (float)sign * x2[0];
                X[0] *= mid;
                X[1] *= mid;
                Y[0] *= side;
                Y[1] *= side;
                float t = X[0];
                X[0] = t - Y[0];
                Y[0] = t + Y[0];
                t = X[1];
                X[1] = t - Y[1];
                Y[1] = t + Y[1];
            } else {
                if (B0 > 1 && !stereo && (itheta & 0x3fff)) {
                    if (itheta > 8192) delta -= delta >> (4 - duration);
                    else delta = imin(0, delta + (N << 3 >> (5 - duration)));
                }
                int mbits = clampi((b - delta) / 2, 0, b);
                int sbits = b - mbits;
                budget_ -= qalloc;
                float *side_lowband = (lowband && !stereo) ? lowband + N : nullptr;
                float *mid_out = stereo ? lowband_out : nullptr;
                const int next_level = stereo ? 0 : level + 1;
                const float mid_gain = stereo ? 1.0f : gain * mid;
                const int side_shift = (B0 >> 1) & ((stereo ? 1 : 0) - 1);
                int rebalance = budget_;
                if (mbits >= sbits) {
                    cm = band(i, X, nullptr, N, mbits, blocks, lowband, duration, mid_out, next_level, mid_gain, lowband_scratch, fill);
                    rebalance = mbits - (rebalance - budget_);
                    if (rebalance > 3 << 3 && itheta != 0) sbits += rebalance - (3 << 3);
                    cm |= band(i, Y, nullptr, N, sbits, blocks, side_lowband, duration, nullptr, next_level, gain * side, nullptr, fill >> blocks)
                          << side_shift;
                } else {
                    cm = band(i, Y, nullptr, N, sbits, blocks, side_lowband, duration, nullptr, next_level, gain * side, nullptr, fill >> blocks)
                         << side_shift;
                    rebalance = sbits - (rebalance - budget_);
                    if (rebalance > 3 << 3 && itheta != 16384) mbits += rebalance - (3 << 3);
                    cm |= band(i, X, nullptr, N, mbits, blocks, lowband, duration, mid_out, next_level, mid_gain, lowband_scratch, fill);
                }
            }
        } else {
            // no split: one PVQ codeword (or none)
            unsigned q = (unsigned)bits_to_pulses(cache, b);
            unsigned cost = (unsigned)pulses_to_bits(cache, (int)q);
            budget_ -= (int)cost;
            while (budget_ < 0 && q > 0) {
                budget_ += (int)cost;
                cost = (unsigned)pulses_to_bits(cache, (int)--q);
                budget_ -= (int)cost;
            }
            if (q != 0) {
                cm = shape(X, (unsigned)N, q < 8 ? q : (8 + (q & 7)) << ((q >> 3) - 1), blocks, gain);
            } else {
                const uint32_t cm_mask = (1u << blocks) - 1;
                fill &= (int)cm_mask;
                if (!fill) {
                    std::memset(X, 0, (size_t)N * sizeof(float));
                } else {
                    if (!lowband) {                  // noise
                        for (int j = 0; j < N; j++) X[j] = (float)((int32_t)noise() >> 20);
                        cm = cm_mask;
                    } else {                         // folded spectrum, ~48 dB dither
                        for (int j = 0; j < N; j++) X[j] = lowband[j] + ((noise() & 0x8000) ? 1.0f / 256 : -1.0f / 256);
                        cm = (uint32_t)fill;
                    }
                    renormalise(X, N, gain);
                }
            }
        }

        if (stereo) {
            if (N != 2) merge_stereo(X, Y, mid, N);
            if (inv)
                for (int j = 0; j < N; j++) Y[j] *= -1;
        } else if (level == 0) {
            if (B0 > 1) reorder(X, n_per_block >> recombine, B0 << recombine, longblocks, true);
            n_per_block = n_per_block0;
            blocks = (unsigned)B0;
            for (int k = 0; k < time_divide; k++) {
                blocks >>= 1;
                n_per_block <<= 1;
                cm |= cm >> blocks;
                haar(X, n_per_block, (int)blocks);
            }
            for (int k = 0; k < recombine; k++) {
                cm = k_celt_bit_deinterleave[cm];
                haar(X, (int)(N0 >> k), 1 << k);
            }
            blocks <<= recombine;
            if (lowband_out) {
                const float n = std::sqrt((float)N0);
                for (unsigned j = 0; j < N0; j++) lowband_out[j] = n * X[j];
            }
            cm &= (1u << blocks) - 1;
        }
        return cm;
    }

    void all_bands()                                // celt_decode_bands (:3472-3566)
    {
        RangeDecoder &rc = *rc_;
        float lowband_scratch[8 * 22];
        float *norm = norm_, *norm2 = norm_ + 8 * 100;
        const int totalbits = (framebits_ << 3) - anticollapse_rsv_;
        bool update_lowband = true;
        int lowband_offset = 0;
        std::memset(coeffs_, 0, sizeof(coeffs_));
        for (int i = start_; i < end_; i++) {
            const int lo = band_lo(i), n = band_n(i);
            float *X = coeffs_[0] + lo;
            float *Y = C_ == 2 ? coeffs_[1] + lo : nullptr;
            const int consumed = (int)rc.tell_frac();
            if (i != start_) balance_ -= consumed;
            budget_ = totalbits - consumed - 1;
            int b = 0;
            if (i <= coded_bands_ - 1) {
                const int share = balance_ / imin(3, coded_bands_ - i);
                const int want = imin(budget_ + 1, pulses_[i] + share);
                b = (want & ~16383) ? ((-want >> 31) & 16383) : want;             // av_clip_uintp2(want, 14)
            }
            if (k_celt_freq_bands[i] - k_celt_freq_range[i] >= k_celt_freq_bands[start_] && (update_lowband || lowband_offset == 0))
                lowband_offset = i;

            int fold_from = -1;                       // first line (unscaled) of the lower band folded into this one
            uint32_t cm[2];
            if (lowband_offset != 0 && (spread_ != 3 || blocks_ > 1 || tf_change_[i] < 0)) {
                fold_from = imax(k_celt_freq_bands[start_], k_celt_freq_bands[lowband_offset] - k_celt_freq_range[i]);
                int fold_start = lowband_offset, fold_end = lowband_offset - 1;
                while (k_celt_freq_bands[--fold_start] > fold_from) {}
                while (k_celt_freq_bands[++fold_end] < fold_from + k_celt_freq_range[i]) {}
                cm[0] = cm[1] = 0;
                for (int j = fold_start; j < fold_end; j++) {
                    cm[0] |= ch_[0].collapse[j];
                    cm[1] |= ch_[C_ - 1].collapse[j];
                }
            } else {
                cm[0] = cm[1] = (1u << blocks_) - 1;
            }
            if (dual_stereo_ && i == intensity_) {    // from here on: intensity stereo
                dual_stereo_ = 0;
                for (int j = band_lo(start_); j < lo; j++) norm[j] = (norm[j] + norm2[j]) / 2;
            }
            float *fold = fold_from != -1 ? norm + (fold_from << LM_) : nullptr;
            if (dual_stereo_) {
                float *fold2 = fold_from != -1 ? norm2 + (fold_from << LM_) : nullptr;
                cm[0] = band(i, X, nullptr, n, b / 2, (unsigned)blocks_, fold, LM_, norm + lo, 0, 1.0f, lowband_scratch, (int)cm[0]);
                cm[1] = band(i, Y, nullptr, n, b / 2, (unsigned)blocks_, fold2, LM_, norm2 + lo, 0, 1.0f, lowband_scratch, (int)cm[1]);
            } else {
                cm[0] = band(i, X, Y, n, b, (unsigned)blocks_, fold, LM_, norm + lo, 0, 1.0f, lowband_scratch, (int)(cm[0] | cm[1]));
                cm[1] = cm[0];
            }
            ch_[0].collapse[i] = (uint8_t)cm[0];
            ch_[C_ - 1].collapse[i] = (uint8_t)cm[1];
            balance_ += pulses_[i] + consumed;
            update_lowband = b > n << 3;
        }
    }

    void anti_collapse(const ChannelMemory &m, float *X)    // :3420-3470
    {
        for (int i = start_; i < end_; i++) {
            const int n = band_n(i);
            const int depth = (1 + pulses_[i]) / n;
            const float thresh = (float)std::exp2((double)(float)(-1.0 - (double)(0.125f * (float)depth)));
            const float sqrt_1 = 1.0f / std::sqrt((float)n);
            float p0 = m.prev_energy[0][i], p1 = m.prev_energy[1][i];
            if (C_ == 1) {
                p0 = p0 > ch_[1].prev_energy[0][i] ? p0 : ch_[1].prev_energy[0][i];
                p1 = p1 > ch_[1].prev_energy[1][i] ? p1 : ch_[1].prev_energy[1][i];
            }
            float ediff = m.energy[i] - (p0 < p1 ? p0 : p1);
            ediff = 0.0f > ediff ? 0.0f : ediff;
            float r = (float)std::exp2((double)(1 - ediff));
            if (LM_ == 3) r = (float)((double)r * 1.41421356237309504880);
            r = (thresh < r ? thresh : r) * sqrt_1;
            float *x = X + band_lo(i);
            bool touched = false;
            for (int k = 0; k < 1 << LM_; k++) {
                if (m.collapse[i] & 1 << k) continue;
                for (int j = 0; j < k_celt_freq_range[i]; j++) x[(j << LM_) + k] = (noise() & 0x8000) ? r : -r;
                touched = true;
            }
            if (touched) renormalise(x, n, 1.0f);
        }
    }

    void denormalise(const ChannelMemory &m, float *X)      // :3268-3279
    {
        for (int i = start_; i < end_; i++) {
            const float g = (float)std::exp2((double)(m.energy[i] + table_f32(k_celt_mean_energy_bits, i)));
            float *x = X + band_lo(i);
            for (int j = 0, n = band_n(i); j < n; j++) x[j] *= g;
        }
    }

    RangeDecoder *rc_ = nullptr;
    int out_channels_ = 0;
    ChannelMemory ch_[2];
    uint32_t seed_ = 0;
    // per frame
    int C_ = 0, start_ = 0, end_ = 0, framebits_ = 0, LM_ = 0, blocks_ = 1;
    int spread_ = 2, coded_bands_ = 0, anticollapse_rsv_ = 0, intensity_ = 0, dual_stereo_ = 0, balance_ = 0, budget_ = 0;
    int fine_bits_[kBands], fine_priority_[kBands], pulses_[kBands], tf_change_[kBands];
    float coeffs_[2][kMaxFrame];
    float scratch_[22 * 8];
    float norm_[2 * 8 * 100] = {};
};

// ---------------------------------------------------------------------------------------------
// Opus packet framing (RFC 6716 section 3 as ff_opus_parse_packet reads it, no self-delimiting form)
// ---------------------------------------------------------------------------------------------
struct PacketLayout {
    int config = 0, stereo = 0, frame_count = 0, frame_samples = 0, mode = 0, bandwidth = 0;   // mode: 0 SILK, 1 hybrid, 2 CELT
    int offset[48], size[48];
};

bool read_len16(const uint8_t *&p, const uint8_t *end, int &v)
{
    if (p >= end) return false;
    v = *p++;
    if (v >= 252) {
        if (p >= end) return false;
        v += 4 * *p++;
    }
    return true;
}

bool parse_packet(const uint8_t *buf, int n, PacketLayout &L)
{
    if (n < 1) return false;
    const uint8_t *p = buf, *end = buf + n;
    const int toc = *p++;
    const int code = toc & 3;
    L.stereo = (toc >> 2) & 1;
    L.config = toc >> 3;
    if (code >= 2 && n < 2) return false;
    int padding = 0;
    if (code == 0) {
        L.frame_count = 1;
        if (end - p > 1275) return false;
        L.offset[0] = (int)(p - buf);
        L.size[0] = (int)(end - p);
    } else if (code == 1) {
        const int both = (int)(end - p);
        if ((both & 1) || (both >> 1) > 1275) return false;
        L.frame_count = 2;
        L.offset[0] = (int)(p - buf);
        L.size[0] = L.size[1] = both >> 1;
        L.offset[1] = L.offset[0] + L.size[0];
    } else if (code == 2) {
        int first;
        if (!read_len16(p, end, first)) return false;
        const int second = (int)(end - p) - first;
        if (second < 0 || second > 1275) return false;
        L.frame_count = 2;
        L.offset[0] = (int)(p - buf);
        L.size[0] = first;
        L.offset[1] = L.offset[0] + first;
        L.size[1] = second;
    } else {
        const int b = *p++;
        L.frame_count = b & 0x3F;
        const bool vbr = (b >> 7) & 1;
        if (L.frame_count == 0 || L.frame_count > 48) return false;
        if ((b >> 6) & 1) {
            for (;;) {
                if (p >= end || padding > 0x7fffffff - 254) return false;
                const int next = *p++;
                padding += next;
                if (next < 255) break;
                --padding;
            }
        }
        if (vbr) {
            int total = 0;
            for (int i = 0; i < L.frame_count - 1; i++) {
                if (!read_len16(p, end, L.size[i])) return false;
                total += L.size[i];
            }
            const int left = (int)(end - p) - padding;
            if (total > left) return false;
            L.offset[0] = (int)(p - buf);
            for (int i = 1; i < L.frame_count; i++) L.offset[i] = L.offset[i - 1] + L.size[i - 1];
            L.size[L.frame_count - 1] = left - total;
        } else {
            const int left = (int)(end - p) - padding;
            if (left < 0) return false;              // (the reference goes on with a negative size and asserts, dopus.d:582)
            if (left % L.frame_count || left / L.frame_count > 1275) return false;
            const int each = left / L.frame_count;
            for (int i = 0; i < L.frame_count; i++) {
                L.offset[i] = (int)(p - buf) + i * each;
                L.size[i] = each;
            }
        }
    }
    L.frame_samples = k_opus_frame_duration[L.config];
    if (L.frame_samples * L.frame_count > 5760) return false;
    if (L.config < 12) {
        L.mode = 0;
        L.bandwidth = L.config >> 2;
    } else if (L.config < 16) {
        L.mode = 1;
        L.bandwidth = 3 + (L.config >= 14 ? 1 : 0);
    } else {
        L.mode = 2;
        L.bandwidth = (L.config - 16) >> 2;
        if (L.bandwidth) ++L.bandwidth;              // there is no medium band in CELT
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// Ogg: one logical stream, pages accepted as the reference's parsePageHeader accepts them (capture pattern, version 0,
// flag bits, checksum; dopus.d:7055-7097).  A page that fails ends the stream.
// ---------------------------------------------------------------------------------------------
struct CrcTable {
    uint32_t t[8][256];                                   // slicing-by-8 tables of the Ogg CRC (MSB first, 0x04c11db7, no reflection)
    CrcTable()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t r = i << 24;
            for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : r << 1;
            t[0][i] = r;
        }
        for (int k = 1; k < 8; k++)
            for (uint32_t i = 0; i < 256; i++) t[k][i] = (t[k - 1][i] << 8) ^ t[0][t[k - 1][i] >> 24];
    }
};
const CrcTable g_crc;

// CRC of len bytes continuing from c: eight bytes per step (the batch path checks every page of every file more than
// once -- length scan, framing scan, decode -- and a byte-wise table walk was a visible share of the Opus host stage)
uint32_t ogg_crc(uint32_t c, const uint8_t *p, size_t len)
{
    while (len >= 8) {
        const uint32_t hi = c ^ ((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | (uint32_t)p[3]);
        c = g_crc.t[7][hi >> 24] ^ g_crc.t[6][(hi >> 16) & 0xff] ^ g_crc.t[5][(hi >> 8) & 0xff] ^ g_crc.t[4][hi & 0xff] ^
            g_crc.t[3][p[4]] ^ g_crc.t[2][p[5]] ^ g_crc.t[1][p[6]] ^ g_crc.t[0][p[7]];
        p += 8;
        len -= 8;
    }
    for (size_t i = 0; i < len; i++) c = (c << 8) ^ g_crc.t[0][((c >> 24) ^ p[i]) & 0xff];
    return c;
}

size_t valid_page(const uint8_t *p, size_t n)        // size of the page at p, 0 if there is none
{
    if (n < 27 || std::memcmp(p, "OggS", 4) != 0 || p[4] != 0 || (p[5] & ~7)) return 0;
    const size_t nseg = p[26];
    if (27 + nseg > n) return 0;
    size_t len = 27 + nseg;
    for (size_t i = 0; i < nseg; i++) len += p[27 + i];
    if (len > n) return 0;
    uint8_t head[27];                                      // the checksum field counts as zeros
    std::memcpy(head, p, 27);
    head[22] = head[23] = head[24] = head[25] = 0;
    const uint32_t c = ogg_crc(ogg_crc(0, head, 27), p + 27, len - 27);
    const uint32_t want = (uint32_t)p[22] | (uint32_t)p[23] << 8 | (uint32_t)p[24] << 16 | (uint32_t)p[25] << 24;
    return c == want ? len : 0;
}

uint64_t le64(const uint8_t *p)
{
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = v << 8 | p[i];
    return v;
}
uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

class PacketWalk {
public:
    void start(const uint8_t *d, size_t n) { d_ = d; n_ = n; pos_ = 0; nseg_ = seg_ = 0; lacing_ = body_ = nullptr; }
    int page_flags() const { return flags_; }
    uint64_t page_granule() const { return granule_; }
    // next packet: false at the end of the data.  A packet held by one page is returned in place, one that spans
    // pages is assembled in `hold`.
    bool next(const uint8_t *&pkt, size_t &len, std::vector<uint8_t> &hold)
    {
        hold.clear();
        const uint8_t *first = nullptr;
        size_t got = 0;
        bool spans = false;
        for (;;) {
            if (seg_ >= nseg_) {
                if (!page()) return false;
                if (nseg_ == 0) continue;
                if (first) {                          // continues on a new page: collect
                    if (!spans) hold.assign(first, first + got);
                    spans = true;
                }
            }
            while (seg_ < nseg_) {
                const int l = lacing_[seg_++];
                if (spans) hold.insert(hold.end(), body_, body_ + l);
                else if (!first) first = body_;
                body_ += l;
                got += (size_t)l;
                if (l < 255) {
                    pkt = spans ? hold.data() : first;
                    len = got;
                    return true;
                }
            }
            if (!first) first = body_;                // (a packet of whole 255-byte segments ending with the page)
        }
    }
private:
    bool page()
    {
        const size_t len = valid_page(d_ + pos_, n_ - pos_);
        if (!len) return false;
        const uint8_t *h = d_ + pos_;
        flags_ = h[5];
        granule_ = le64(h + 6);
        nseg_ = h[26];
        seg_ = 0;
        lacing_ = h + 27;
        body_ = lacing_ + nseg_;
        pos_ += len;
        return true;
    }
    const uint8_t *d_ = nullptr, *lacing_ = nullptr, *body_ = nullptr;
    size_t n_ = 0, pos_ = 0;
    int nseg_ = 0, seg_ = 0, flags_ = 0;
    uint64_t granule_ = 0;
};

int track_gain_comment(const uint8_t *c, size_t n)   // OpusFileCtx.getGain (:8011-8059); c points behind "OpusTags"
{
    if (n < 4) return 0;
    uint32_t len = le32(c);
    if (len > n || n - len < 4) return 0;
    size_t at = 4 + (size_t)len;
    if (at >= n || n - at < 4) return 0;
    uint32_t count = le32(c + at);
    at += 4;
    for (; count > 0 && at + 4 <= n; --count) {
        len = le32(c + at);
        at += 4;
        if (at > n || n - at < len) break;
        const uint8_t *s = c + at, *e = s + len;
        at += len;
        while (s < e && *s <= ' ') s++;
        while (e > s && e[-1] <= ' ') e--;
        static const char key[] = "R128_TRACK_GAIN=";
        const size_t kl = sizeof(key) - 1;
        if ((size_t)(e - s) <= kl) continue;
        bool match = true;
        for (size_t i = 0; i < kl && match; i++) {
            int ch = s[i];
            if (ch >= 'a' && ch <= 'z') ch -= 32;
            match = ch == key[i];
        }
        if (!match) continue;
        s += kl;
        bool neg = false;
        if (*s == '-') { neg = true; s++; }
        else if (*s == '+') s++;
        if (s == e) continue;
        int v = 0;
        for (; s < e; s++) {
            if (*s < '0' || *s > '9') { v = -1; break; }
            v = v * 10 + (*s - '0');
            if (v > (neg ? 32768 : 32767)) { v = -1; break; }
        }
        if (v >= 0) return neg ? -v : v;
    }
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// reader
// ---------------------------------------------------------------------------------------------
struct Reader::Impl {
    const uint8_t *data = nullptr;
    size_t size = 0;
    PacketWalk walk;
    std::vector<uint8_t> hold;
    CeltDecoder celt;
    int channels = 0;
    bool failed = false, first_audio = true;
    int preskip = 0;

    Status open(const uint8_t *d, size_t n, File &meta)
    {
        data = d;
        size = n;
        meta = File();
        if (!d || n < 47) return kNotOpus;
        walk.start(d, n);
        const uint8_t *pkt;
        size_t len;
        // opus_header (:7791-7818): a BOS page whose packet is at least 19 bytes with version nibble 0 (the magic itself
        // is not compared there; a foreign first packet fails on the tags check below)
        if (!walk.next(pkt, len, hold) || len < 19 || !(walk.page_flags() & 2) || (pkt[8] & 0xF0)) return kNotOpus;
        const int nch = pkt[9], map_type = pkt[18];
        preskip = pkt[10] | pkt[11] << 8;
        int gain_i = pkt[16] | pkt[17] << 8;          // read unsigned, like AV_RL16 (:516, :1311)
        if (!walk.next(pkt, len, hold) || len < 8 || std::memcmp(pkt, "OpusTags", 8) != 0) return kNotOpus;
        gain_i += len >= 12 ? track_gain_comment(pkt + 8, len - 8) : 0;
        // ff_opus_parse_extradata (:1270-1405) + opusOpen (:8164-8169): one stream of one or two channels
        if (nch < 1 || nch > 2 || map_type != 0) return kNotOpus;
        gain_i = clampi(gain_i, -32768, 32767);
        meta.channels = channels = nch;
        meta.preskip = preskip;
        meta.gain_i = gain_i;
        meta.gain = gain_i ? (float)std::exp2(3.32192809488736234787 * (gain_i / (20.0 * 256))) : 1.0f;    // ff_exp10 (:66-69)
        // stream length: granule position of the last valid page (findLastPage) minus the pre-skip (:8157-8159)
        uint64_t last = 0;
        for (size_t at = 0, l; (l = valid_page(d + at, n - at)) != 0; at += l) last = le64(d + at + 6);
        if (last < (uint64_t)preskip) return kNotOpus;
        meta.declared_frames = (int64_t)(last - (uint64_t)preskip);
        // every audio packet's TOC: refuse what cannot be decoded here, before anything is delivered
        PacketWalk scan = walk;
        std::vector<uint8_t> tmp;
        bool any = false, counting = true;
        while (scan.next(pkt, len, tmp)) {
            if (!any && scan.page_granule() < (uint64_t)preskip) return kNotOpus;     // :8155
            any = true;
            if (len && (pkt[0] >> 3) < 16) return kUnsupported;
            // what `more` will record: every frame up to the first packet it gives up on (same rules as there)
            PacketLayout L;
            if (counting && parse_packet(pkt, (int)len, L) && !(nch == 2 && L.frame_count * L.frame_samples > 2880)) {
                meta.bound_frames += (size_t)L.frame_count;
                meta.bound_coeffs += (size_t)L.frame_count * (size_t)L.frame_samples * (size_t)nch;
            } else {
                counting = false;
            }
        }
        if (!any) return kNotOpus;                   // opusOpen needs one packet behind the tags (:8147)
        celt.reset(nch);
        failed = false;
        return kOpened;
    }

    bool more(File &out, int max_packets)
    {
        out.frames.clear();
        out.coeffs.clear();
        out.n_frames = out.n_coeffs = 0;
        out.pcm_frames = 0;
        out.error = false;
        if (failed) return false;
        const uint8_t *pkt;
        size_t len;
        int taken = 0;
        while (taken < max_packets && walk.next(pkt, len, hold)) {
            taken++;
            PacketLayout L;
            // a packet the reference cannot frame makes its read fail (stream.d:452-456); so does a stereo packet longer
            // than 60 ms, which overruns readFrame's two 2880-float halves there (dopus.d:7961, :8079-8082)
            if (!parse_packet(pkt, (int)len, L) || L.mode != 2 || (channels == 2 && L.frame_count * L.frame_samples > 2880)) {
                out.error = failed = true;
                break;
            }
            for (int f = 0; f < L.frame_count; f++) {
                RangeDecoder rc;
                rc.start(pkt + L.offset[f], L.size[f]);
                FrameInfo info;
                celt.decode(rc, L.stereo + 1, L.frame_samples, k_celt_band_end[L.bandwidth], info);
                afg_celt_frame r;
                std::memset(&r, 0, sizeof(r));
                r.coef_off = out.n_coeffs;
                r.out_off = out.pcm_frames * (uint64_t)channels;
                r.out_stride = (uint32_t)channels;
                r.frame_size = (uint16_t)L.frame_samples;
                r.blocks = (uint8_t)info.blocks;
                r.pf_period_new = info.pf_period;
                std::memcpy(r.pf_gains_new, info.pf_gains, sizeof(r.pf_gains_new));
                r.imdct_scale = info.imdct_scale;
                const size_t nco = (size_t)L.frame_samples * (size_t)channels;
                if (out.ext_frames) {
                    if (out.n_frames + 1 > out.ext_frames_cap || out.n_coeffs + nco > out.ext_coeffs_cap) {
                        out.overflow = failed = true;
                        break;
                    }
                    out.ext_frames[out.n_frames] = r;
                    for (int c = 0; c < channels; c++)
                        std::memcpy(out.ext_coeffs + out.n_coeffs + (size_t)c * L.frame_samples, celt.coeffs(c), (size_t)L.frame_samples * sizeof(float));
                } else {
                    out.frames.push_back(r);
                    for (int c = 0; c < channels; c++) out.coeffs.insert(out.coeffs.end(), celt.coeffs(c), celt.coeffs(c) + L.frame_samples);
                }
                out.n_frames++;
                out.n_coeffs += nco;
                out.pcm_frames += (uint64_t)L.frame_samples;
            }
            if (out.overflow) break;
        }
        return taken > 0;
    }
};

Reader::Reader() : p(new Impl) {}
Reader::~Reader() { delete p; }
Status Reader::open(const uint8_t *data, size_t size, File &meta) { return p->open(data, size, meta); }
bool Reader::more(File &out, int max_packets) { return p->more(out, max_packets); }

Status parse_file_into(const uint8_t *data, size_t size, File &out, afg_celt_frame *frames, size_t frames_cap, float *coeffs, size_t coeffs_cap)
{
    Reader r;
    const Status st = r.open(data, size, out);
    if (st != kOpened) return st;
    // the whole stream in one pass of the chunked reader, straight into the caller's storage
    out.ext_frames = frames;
    out.ext_coeffs = coeffs;
    out.ext_frames_cap = frames_cap;
    out.ext_coeffs_cap = coeffs_cap;
    r.more(out, 0x7fffffff);
    return kOpened;
}

Status parse_file(const uint8_t *data, size_t size, File &out)
{
    Reader r;
    const Status st = r.open(data, size, out);
    if (st != kOpened) return st;
    File chunk;
    while (r.more(chunk, 256)) {
        const uint64_t coef0 = out.coeffs.size(), out0 = out.pcm_frames * (uint64_t)out.channels;
        for (afg_celt_frame f : chunk.frames) {
            f.coef_off += coef0;
            f.out_off += out0;
            out.frames.push_back(f);
        }
        out.coeffs.insert(out.coeffs.end(), chunk.coeffs.begin(), chunk.coeffs.end());
        out.pcm_frames += chunk.pcm_frames;
        if (chunk.error) { out.error = true; break; }
    }
    out.n_frames = out.frames.size();
    out.n_coeffs = out.coeffs.size();
    return kOpened;
}

}  // namespace afg_opus
