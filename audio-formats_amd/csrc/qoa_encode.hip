// qoa_encode.hip -- QOA encoder on gfx950 (the "output side" of the transcode path).
//
// Restates qoa_encode_frame (reference qoa.d:295-399) with the file framing of QOAEncoder (:538-700): file header
// (:413-453 layout), per frame the header, the LMS state of every channel, then the slices with the channels
// interleaved slice by slice.  The encoder's LMS state runs through the whole stream, so a channel of a stream is
// strictly serial -- but the brute-force search over the 16 scalefactors of every slice (:330-377) is 16 independent
// trials: lane = (channel slot, scalefactor).  A wavefront owns two streams: side by side when both are mono / stereo
// (two 16-lane groups each), else one after the other with four channels at a time; each lane
// encodes the slice's 20 samples with its scalefactor, a 4-step butterfly over the 16 lanes of a channel finds the
// smallest squared error (ties: the smallest scalefactor, as the reference's strict `<` keeps the first), and the
// winner's LMS state and slice bits are broadcast.  The early `break` of the reference (:361) only skips work on
// trials that can no longer win; results are identical.  Optional float input is converted as
// QOAEncoder.writeSamples does (:632-636).  Integer arithmetic wraps exactly as in D: output bytes are the
// reference's.
#include "afg_common.h"

namespace {

constexpr int kSliceLen = 20;
constexpr int kFrameLen = 256 * kSliceLen;               // 5120 samples per channel

__device__ const int k_reciprocal[16] = { 65536, 9363, 3121, 1457, 781, 475, 311, 216, 156, 117, 90, 71, 57, 47, 39, 32 };
// magnitudes of qoa_dequant_tab[scalefactor][2k] (odd entries are their negatives), qoa.d:197-215
__device__ const int k_dequant_mag[16 * 4] = {
    1, 3, 5, 7,  5, 18, 32, 49,  16, 53, 95, 147,  34, 113, 203, 315,  63, 210, 378, 588,  104, 345, 621, 966,
    158, 528, 950, 1477,  228, 760, 1368, 2128,  316, 1053, 1895, 2947,  422, 1405, 2529, 3934,
    548, 1828, 3290, 5117,  696, 2320, 4176, 6496,  868, 2893, 5207, 8099,  1064, 3548, 6386, 9933,
    1286, 4288, 7718, 12005,  1536, 5120, 9216, 14336 };
// qoa_quant_tab[17] = {7,7,7,5,5,3,3,1,0,0,2,2,4,4,6,6,6} packed 3 bits per entry (entry 0 in the low bits)
constexpr uint64_t kQuantPacked =
    7ull | 7ull << 3 | 7ull << 6 | 5ull << 9 | 5ull << 12 | 3ull << 15 | 3ull << 18 | 1ull << 21 | 0ull << 24 | 0ull << 27 |
    2ull << 30 | 2ull << 33 | 4ull << 36 | 4ull << 39 | 6ull << 42 | 6ull << 45 | 6ull << 48;

__device__ __forceinline__ void store_be64(uint8_t *p, uint64_t v)      // p is 8-byte aligned
{
    *(uint64_t *)p = ((uint64_t)__builtin_bswap32((uint32_t)v) << 32) | __builtin_bswap32((uint32_t)(v >> 32));
}

__device__ __forceinline__ int clamp_s16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

__device__ __forceinline__ int load_sample(const int16_t *__restrict__ pi, const float *__restrict__ pf, uint64_t at)
{
    if (pi) return pi[at];
    const double x = pf[at];                               // QOAEncoder.writeSamples, qoa.d:632-636
    return (int)(32768.5 + x * 32767.0) - 32768;
}

// One pass of the wavefront: lane group g (16 lanes) encodes channel `c` of stream `st` when `act`; the groups of a
// wavefront may belong to two different streams (stereo / mono streams are packed two to a wavefront), so every
// loop runs to the longest group and the shorter ones idle.
__device__ __forceinline__ void encode_pass(const afg_qoa_enc_stream &st, int c, bool act, bool writes_headers,
                                            const int16_t *__restrict__ pcm_i16, const float *__restrict__ pcm_f32,
                                            uint8_t *__restrict__ out, int sf, int recip, const int (&mag)[4])
{
    const int lane = threadIdx.x;
    const int C = st.channels;
    const uint32_t samples = act ? st.samples : 0;
    uint8_t *file = out + st.out_off;
    const uint64_t full_frame = 8 + 16ull * C + 8ull * 256 * C;
    if (act && writes_headers && sf == 0) store_be64(file, ((uint64_t)0x716f6166u << 32) | st.samples);    // 'qoaf', qoa.d:413-453

    int w[4] = { 0, 0, -(1 << 13), 1 << 14 }, h[4] = { 0, 0, 0, 0 };                 // qoa.d:566-581
    for (uint32_t s0 = 0; __any(s0 < samples); s0 += kFrameLen) {
        const bool live = s0 < samples;
        const uint32_t frame_len = !live ? 0 : (samples - s0 < (uint32_t)kFrameLen ? samples - s0 : (uint32_t)kFrameLen);
        const uint32_t slices = (frame_len + kSliceLen - 1) / kSliceLen;
        uint8_t *fb = file + 8 + (uint64_t)(s0 / kFrameLen) * full_frame;
        if (live && writes_headers && sf == 0) {
            const uint64_t frame_size = 8 + 16ull * C + 8ull * slices * C;
            store_be64(fb, (uint64_t)C << 56 | (uint64_t)st.samplerate << 32 | (uint64_t)frame_len << 16 | frame_size);
        }
        if (live && sf == 0) {                               // the LMS state the frame starts from, 16 bits per entry
            uint64_t weights = 0, history = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                history = (history << 16) | (uint64_t)(h[i] & 0xffff);
                weights = (weights << 16) | (uint64_t)(w[i] & 0xffff);
            }
            store_be64(fb + 8 + 16 * c, history);
            store_be64(fb + 16 + 16 * c, weights);
        }
        uint8_t *slice_base = fb + 8 + 16 * C;
        const uint64_t pcm0 = st.pcm_off + (uint64_t)s0 * C + c;

        int x[kSliceLen], nx[kSliceLen];
#pragma unroll
        for (int j = 0; j < kSliceLen; j++) {
            x[j] = 0;
            if (live) x[j] = load_sample(pcm_i16, pcm_f32, pcm0 + (uint64_t)min((uint32_t)j, frame_len - 1) * C);
        }
        uint32_t max_slices = slices;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) max_slices = max(max_slices, (uint32_t)__shfl_xor((int)max_slices, off));

        for (uint32_t k = 0; k < max_slices; k++) {
            const bool slive = k < slices;
            const uint32_t sample_index = k * kSliceLen;
            const int slice_len = slive ? (int)min((uint32_t)kSliceLen, frame_len - sample_index) : 0;
            // the next slice's samples (clamped inside the frame) travel while this one is searched
#pragma unroll
            for (int j = 0; j < kSliceLen; j++) {
                nx[j] = 0;
                if (slive) nx[j] = load_sample(pcm_i16, pcm_f32, pcm0 + (uint64_t)min(sample_index + kSliceLen + j, frame_len - 1) * C);
            }
            // this lane's trial: scalefactor sf on the slice (qoa.d:336-370)
            int tw[4] = { w[0], w[1], w[2], w[3] }, th[4] = { h[0], h[1], h[2], h[3] };
            uint64_t slice = (uint64_t)sf, err = 0;
#pragma unroll
            for (int j = 0; j < kSliceLen; j++) {
                if (j < slice_len) {
                    const int sample = x[j];
                    const int predicted = (tw[0] * th[0] + tw[1] * th[1] + tw[2] * th[2] + tw[3] * th[3]) >> 13;
                    const int residual = sample - predicted;
                    int n = (int)((unsigned)residual * (unsigned)recip + (1u << 15)) >> 16;        // qoa_div, :263-269
                    n = n + ((residual > 0) - (residual < 0)) - ((n > 0) - (n < 0));
                    const int clamped = n < -8 ? -8 : (n > 8 ? 8 : n);
                    const int quantized = (int)((kQuantPacked >> (3 * (clamped + 8))) & 7);
                    const int m = (quantized & 4) ? ((quantized & 2) ? mag[3] : mag[2]) : ((quantized & 2) ? mag[1] : mag[0]);
                    const int dequantized = (quantized & 1) ? -m : m;
                    const int reconstructed = clamp_s16(predicted + dequantized);
                    const long long e = sample - reconstructed;
                    err += (uint64_t)(e * e);
                    const int delta = dequantized >> 4;                                          // qoa_lms_update, :241-254
#pragma unroll
                    for (int i = 0; i < 4; i++) tw[i] += th[i] < 0 ? -delta : delta;
                    th[0] = th[1]; th[1] = th[2]; th[2] = th[3]; th[3] = reconstructed;
                    slice = (slice << 3) | (uint64_t)quantized;
                }
            }
            // smallest error among the 16 trials of the channel; ties go to the smallest scalefactor
            uint64_t best_err = err;
            int best_sf = sf;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const uint32_t olo = __shfl_xor((uint32_t)best_err, off, 16), ohi = __shfl_xor((uint32_t)(best_err >> 32), off, 16);
                const int osf = __shfl_xor(best_sf, off, 16);
                const uint64_t oerr = ((uint64_t)ohi << 32) | olo;
                if (oerr < best_err || (oerr == best_err && osf < best_sf)) { best_err = oerr; best_sf = osf; }
            }
            const int src = (lane & 48) | best_sf;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int nw = __shfl(tw[i], src), nh = __shfl(th[i], src);
                if (slive) { w[i] = nw; h[i] = nh; }
            }
            const uint32_t slo = __shfl((uint32_t)slice, src), shi = __shfl((uint32_t)(slice >> 32), src);
            if (slive && sf == 0) {
                uint64_t best_slice = ((uint64_t)shi << 32) | slo;
                best_slice <<= (kSliceLen - slice_len) * 3;                                        // :379-383
                store_be64(slice_base + ((uint64_t)k * C + c) * 8, best_slice);
            }
#pragma unroll
            for (int j = 0; j < kSliceLen; j++) x[j] = nx[j];
        }
    }
}

__global__ __launch_bounds__(64) void qoa_encode_kernel(
    const afg_qoa_enc_stream *__restrict__ streams, uint32_t n_streams, const int16_t *__restrict__ pcm_i16,
    const float *__restrict__ pcm_f32, uint8_t *__restrict__ out)
{
    const int lane = threadIdx.x, grp = lane >> 4, sf = lane & 15;
    const int recip = k_reciprocal[sf];
    int mag[4];
#pragma unroll
    for (int k = 0; k < 4; k++) mag[k] = k_dequant_mag[sf * 4 + k];

    // a wavefront takes streams 2w and 2w+1: side by side (two lane groups each) when both have at most two
    // channels, else one after the other with four channels at a time
    const uint32_t sa = 2 * blockIdx.x, sb = sa + 1;
    const afg_qoa_enc_stream A = streams[sa];
    const bool have_b = sb < n_streams;
    const afg_qoa_enc_stream B = have_b ? streams[sb] : A;
    if (have_b && A.channels <= 2 && B.channels <= 2) {
        const bool second = grp >= 2;
        const afg_qoa_enc_stream &st = second ? B : A;
        const int c = grp & 1;
        encode_pass(st, c, c < (int)st.channels, c == 0, pcm_i16, pcm_f32, out, sf, recip, mag);
        return;
    }
    for (int cb = 0; cb < (int)A.channels; cb += 4)
        encode_pass(A, cb + grp, cb + grp < (int)A.channels, cb + grp == 0, pcm_i16, pcm_f32, out, sf, recip, mag);
    if (have_b)
        for (int cb = 0; cb < (int)B.channels; cb += 4)
            encode_pass(B, cb + grp, cb + grp < (int)B.channels, cb + grp == 0, pcm_i16, pcm_f32, out, sf, recip, mag);
}

}  // namespace

extern "C" {

uint64_t afg_qoa_encoded_size(uint32_t samples, uint32_t channels)
{
    const uint64_t frames = ((uint64_t)samples + kFrameLen - 1) / kFrameLen;
    if (frames == 0) return 8;
    const uint64_t last = samples - (frames - 1) * kFrameLen, last_slices = (last + kSliceLen - 1) / kSliceLen;
    return 8 + (frames - 1) * (8 + 16ull * channels + 8ull * 256 * channels) + (8 + 16ull * channels + 8 * last_slices * channels);
}

int afg_qoa_encode_hip(uint32_t n_streams, const afg_qoa_enc_stream *d_streams, const int16_t *d_pcm_i16,
                       const float *d_pcm_f32, uint8_t *d_out, void *hip_stream)
{
    if (n_streams == 0) return AFG_OK;
    if (!d_streams || !d_out || (!d_pcm_i16 == !d_pcm_f32)) {
        afg::set_error("afg_qoa_encode_hip: need the stream table, the output plane and exactly one of the int16 / float inputs");
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    hipLaunchKernelGGL(qoa_encode_kernel, dim3((n_streams + 1) / 2), dim3(64), 0, (hipStream_t)hip_stream, d_streams, n_streams,
                       d_pcm_i16, d_pcm_f32, d_out);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}

}  // extern "C"
