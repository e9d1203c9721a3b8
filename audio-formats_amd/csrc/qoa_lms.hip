// qoa_lms.hip -- QOA frame decode on gfx950.
//
// Replaces the slice loop of qoa_decode_frame (reference qoa.d:489-530 with qoa_lms_predict /
// qoa_lms_update, :231-254) and the float conversion of QOADecoder.readSamples (:831-838).
// The sign-sign LMS adapts after every sample, so a frame-channel is strictly serial; frames carry
// their full LMS state (:489-503) and are independent, so lane = (frame, channel):
//   * a wavefront owns 32 frames x 2 channel slots and walks the 256 slices in steps of 8;
//   * the big-endian 64-bit slices are fetched 8 steps at a time as coalesced 128-byte rows
//     (8-byte loads: frame offsets are only 8-byte aligned) into an LDS tile with 17-word rows;
//   * each lane decodes one slice (20 samples) per step in registers, the 32 x 20 x 2 samples of a
//     step are staged in LDS and leave as 16-byte stores of 160-byte rows.
// Integer results are bit-exact (int arithmetic wraps exactly as in D).
#include "afg_common.h"

#ifndef AFG_QOA_STEP
#define AFG_QOA_STEP 8
#endif

namespace {

constexpr int kSliceLen = 20;
constexpr int kFramesPerWave = 32;
constexpr int kStepSlices = AFG_QOA_STEP;                       // slices fetched per channel per refill
constexpr int kInRow = kStepSlices * 2 + 1;          // 64-bit words per tile row (+1 pad)
constexpr int kFlushSlices = 4;                      // all-stereo path: slices staged per row before a flush (4 x 160 B = five full 128-byte lines)
constexpr int kRowPairs = kFlushSlices * kSliceLen;  // (L,R) pairs per staged row
constexpr int kRowPitch = kRowPairs + 2;             // dwords; 82: rows 8-byte aligned, 2-way bank conflicts at most
constexpr int kFlushIters = kFramesPerWave * kRowPairs / 2 / 64;   // two pairs (8 staged bytes, 16 output bytes) per lane and iteration

__device__ const short k_dequant[16 * 8] = {
    1, -1, 3, -3, 5, -5, 7, -7,  5, -5, 18, -18, 32, -32, 49, -49,
    16, -16, 53, -53, 95, -95, 147, -147,  34, -34, 113, -113, 203, -203, 315, -315,
    63, -63, 210, -210, 378, -378, 588, -588,  104, -104, 345, -345, 621, -621, 966, -966,
    158, -158, 528, -528, 950, -950, 1477, -1477,  228, -228, 760, -760, 1368, -1368, 2128, -2128,
    316, -316, 1053, -1053, 1895, -1895, 2947, -2947,  422, -422, 1405, -1405, 2529, -2529, 3934, -3934,
    548, -548, 1828, -1828, 3290, -3290, 5117, -5117,  696, -696, 2320, -2320, 4176, -4176, 6496, -6496,
    868, -868, 2893, -2893, 5207, -5207, 8099, -8099,  1064, -1064, 3548, -3548, 6386, -6386, 9933, -9933,
    1286, -1286, 4288, -4288, 7718, -7718, 12005, -12005,  1536, -1536, 5120, -5120, 9216, -9216, 14336, -14336 };

__device__ __forceinline__ uint64_t bswap64(uint64_t v)
{
    return ((uint64_t)__builtin_bswap32((uint32_t)v) << 32) | __builtin_bswap32((uint32_t)(v >> 32));
}

struct RowInfo {
    uint64_t byte_off;
    uint64_t out_off;
    uint32_t samples;
    uint32_t channels;
};

__global__ __launch_bounds__(64) void qoa_decode_kernel(
    const afg_qoa_frame *__restrict__ frames, const uint8_t *__restrict__ bytes,
    int16_t *__restrict__ out_i16, float *__restrict__ out_f32, uint64_t n_frames)
{
    __shared__ uint64_t tile[kFramesPerWave * kInRow];
    __shared__ __attribute__((aligned(16))) uint32_t stage16[kFramesPerWave * kRowPitch];      // all-stereo path: (L,R) int16 pairs
    static_assert(sizeof(uint32_t) * kFramesPerWave * kRowPitch >= sizeof(float) * kFramesPerWave * kSliceLen * 2, "staging overlay");
    float *const stage = (float *)stage16;                                                     // general path: one slice of floats
    __shared__ RowInfo rows[kFramesPerWave];
    __shared__ short dq[16 * 8];

    const int lane = threadIdx.x;
    const int fr = lane >> 1, slot = lane & 1;
    for (int i = lane; i < 128; i += 64) dq[i] = k_dequant[i];
    if (lane < kFramesPerWave) {
        const uint64_t f = (uint64_t)blockIdx.x * kFramesPerWave + lane;
        RowInfo r = { 0, 0, 0, 0 };
        if (f < n_frames) {
            const afg_qoa_frame q = frames[f];
            r.byte_off = q.byte_off; r.out_off = q.out_off; r.samples = q.samples; r.channels = q.channels;
        }
        rows[lane] = r;
    }
    __syncthreads();
    const RowInfo me = rows[fr];
    const int C = (int)me.channels;
    int max_pairs = (C + 1) >> 1, max_samples = (int)me.samples;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int a = __shfl_xor(max_pairs, o), b = __shfl_xor(max_samples, o);
        max_pairs = a > max_pairs ? a : max_pairs;
        max_samples = b > max_samples ? b : max_samples;
    }
    const int max_slices = (max_samples + kSliceLen - 1) / kSliceLen;

    // all-stereo wavefronts (every frame two channels, rows 16-byte aligned) stage four slices per row as int16 and
    // flush them as whole 128-byte lines
    const bool all_stereo = __ballot((C != 2 || (me.out_off & 3)) && me.samples != 0) == 0;

    for (int pair = 0; pair < max_pairs; pair++) {
        const int ch = 2 * pair + slot;
        const bool active = ch < C;
        // LMS state from the frame header (qoa.d:489-503): history then weights, 4 x int16 big-endian
        int h0 = 0, h1 = 0, h2 = 0, h3 = 0, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        if (active) {
            const uint64_t *st = (const uint64_t *)(bytes + me.byte_off + 8 + 16 * ch);
            const uint64_t hh = bswap64(st[0]), ww = bswap64(st[1]);
            h0 = (short)(hh >> 48); h1 = (short)(hh >> 32); h2 = (short)(hh >> 16); h3 = (short)hh;
            w0 = (short)(ww >> 48); w1 = (short)(ww >> 32); w2 = (short)(ww >> 16); w3 = (short)ww;
        }

        for (int s0 = 0; s0 < max_slices; s0 += kStepSlices) {
            // refill: kStepSlices slices of both channel slots of every frame; 2 * kStepSlices lanes x 8 B per frame row
            __syncthreads();
#pragma unroll
            for (int i = 0; i < kStepSlices; i++) {
                constexpr int W = 2 * kStepSlices;                      // 64-bit words per frame row and refill
                const int r = (64 / W) * i + lane / W, wd = lane % W;   // word = (slice j, slot) = (wd >> 1, wd & 1)
                const RowInfo m = rows[r];
                const int cidx = 2 * pair + (wd & 1), sl = s0 + (wd >> 1);
                const int nsl = ((int)m.samples + kSliceLen - 1) / kSliceLen;
                uint64_t v = 0;
                if (cidx < (int)m.channels && sl < nsl)
                    v = *(const uint64_t *)(bytes + m.byte_off + 8 + 16 * m.channels + ((uint64_t)sl * m.channels + cidx) * 8);
                tile[r * kInRow + wd] = v;
            }
            __syncthreads();

            for (int j = 0; j < kStepSlices && s0 + j < max_slices; j++) {
                const int sidx = s0 + j;
                uint64_t slice = bswap64(tile[fr * kInRow + 2 * j + slot]);
                const int sf = (int)((slice >> 60) & 0xf);
                const short *dqs = dq + sf * 8;
                float outv[kSliceLen];
#pragma unroll
                for (int k = 0; k < kSliceLen; k++) {
                    // qoa_lms_predict (:231-239): wrapping int sum, arithmetic shift
                    const int pred = (int)((unsigned)w0 * (unsigned)h0 + (unsigned)w1 * (unsigned)h1 +
                                           (unsigned)w2 * (unsigned)h2 + (unsigned)w3 * (unsigned)h3) >> 13;
                    const int q = (int)((slice >> 57) & 0x7);
                    const int deq = dqs[q];
                    int rec = pred + deq;
                    rec = rec < -32768 ? -32768 : (rec > 32767 ? 32767 : rec);       // qoa_clamp_s16 (:278-286)
                    slice <<= 3;
                    // qoa_lms_update (:241-254)
                    const int delta = deq >> 4;
                    w0 += h0 < 0 ? -delta : delta;
                    w1 += h1 < 0 ? -delta : delta;
                    w2 += h2 < 0 ? -delta : delta;
                    w3 += h3 < 0 ? -delta : delta;
                    h0 = h1; h1 = h2; h2 = h3; h3 = rec;
                    outv[k] = (float)rec;
                }
                if (all_stereo) {
                    // (one wavefront per workgroup: its LDS accesses complete in order, no barrier needed)
                    const int jj = sidx % kFlushSlices;
                    short *row16 = (short *)(stage16 + fr * kRowPitch + jj * kSliceLen) + slot;
#pragma unroll
                    for (int k = 0; k < kSliceLen; k++) row16[2 * k] = (short)outv[k];
                    if (jj != kFlushSlices - 1 && sidx != max_slices - 1) continue;
                    __builtin_amdgcn_wave_barrier();
                    const int first = (sidx - jj) * kSliceLen;               // first sample of the staged rows
                    const int have = (jj + 1) * kSliceLen;                   // samples staged per row
#pragma unroll 4
                    for (int it = 0; it < kFlushIters; it++) {
                        const int idx = lane + 64 * it, r = idx / (kRowPairs / 2), k = 2 * (idx - r * (kRowPairs / 2));
                        const RowInfo m = rows[r];
                        const uint2 v = *(const uint2 *)(stage16 + r * kRowPitch + k);        // pairs k, k+1 of row r
                        const short l0 = (short)(v.x & 0xffff), r0 = (short)(v.x >> 16), l1 = (short)(v.y & 0xffff), r1 = (short)(v.y >> 16);
                        const int left = min((int)m.samples - first, have) - k;               // pairs of this row still to write from k
                        const uint64_t o = m.out_off + (uint64_t)(first + k) * 2;
                        if (left >= 2) {
                            if (out_i16) *(short4 *)(out_i16 + o) = make_short4(l0, r0, l1, r1);
                            if (out_f32) {
                                typedef float f32x4nt __attribute__((ext_vector_type(4)));
                                __builtin_nontemporal_store(f32x4nt{ (float)l0 * (1.0f / 32767), (float)r0 * (1.0f / 32767),
                                                                     (float)l1 * (1.0f / 32767), (float)r1 * (1.0f / 32767) },
                                                            (f32x4nt *)(out_f32 + o));
                            }
                        } else if (left == 1) {
                            if (out_i16) *(short2 *)(out_i16 + o) = make_short2(l0, r0);
                            if (out_f32) *(float2 *)(out_f32 + o) = make_float2((float)l0 * (1.0f / 32767), (float)r0 * (1.0f / 32767));
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    continue;
                }
                // stage [frame][sample][slot]
#pragma unroll
                for (int k = 0; k < kSliceLen; k++) stage[(fr * kSliceLen + k) * 2 + slot] = outv[k];
                __syncthreads();
                // store: frame rows of 20 samples x (1|2) slots
                for (int idx = lane; idx < kFramesPerWave * kSliceLen * 2; idx += 64) {
                    const int r = idx / (kSliceLen * 2), rem = idx - r * (kSliceLen * 2);
                    const int k = rem >> 1, sl2 = rem & 1;
                    const RowInfo m = rows[r];
                    const int cidx = 2 * pair + sl2, smp = sidx * kSliceLen + k;
                    if (cidx < (int)m.channels && smp < (int)m.samples) {
                        const uint64_t o = m.out_off + (uint64_t)smp * m.channels + cidx;
                        const float v = stage[idx];
                        if (out_i16) out_i16[o] = (short)v;
                        if (out_f32) out_f32[o] = v * (1.0f / 32767);                 // qoa.d:831-838
                    }
                }
                __syncthreads();
            }
        }
    }
}

}  // namespace

extern "C" int afg_qoa_transform_hip(uint64_t n_frames, const afg_qoa_frame *d_frames, const uint8_t *d_bytes,
                                     int16_t *d_out_i16, float *d_out_f32, void *hip_stream)
{
    if (n_frames == 0) return AFG_OK;
    if (!d_frames || !d_bytes || (!d_out_i16 && !d_out_f32)) {
        afg::set_error("afg_qoa_transform_hip: NULL device pointer");
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    const uint64_t groups = (n_frames + kFramesPerWave - 1) / kFramesPerWave;
    if (groups > 0x7fffffffull) {
        afg::set_error("afg_qoa_transform_hip: too many frames in one call");
        return AFG_ERR_INVALID;
    }
    hipLaunchKernelGGL(qoa_decode_kernel, dim3((uint32_t)groups), dim3(64), 0, (hipStream_t)hip_stream,
                       d_frames, d_bytes, d_out_i16, d_out_f32, n_frames);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}
