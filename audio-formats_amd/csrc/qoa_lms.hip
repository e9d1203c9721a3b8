// qoa_lms.hip -- QOA frame decode on gfx950.
//
// Replaces the slice loop of qoa_decode_frame (reference qoa.d:489-530 with qoa_lms_predict /
// qoa_lms_update, :231-254) and the float conversion of QOADecoder.readSamples (:831-838).
// The sign-sign LMS adapts after every sample, so a frame-channel is strictly serial; frames carry
// their full LMS state (:489-503) and are independent, so lane = (frame, channel):
//   * a wavefront owns 32 frames x 2 channel slots and walks the 256 slices in steps of 8;
//   * all-stereo wavefronts (the common case): a lane fetches its frame's big-endian 64-bit slices 16 bytes at a time (both
//     channels of one slice index) and trades halves with the neighbouring lane; each lane decodes one slice (20 samples) at a
//     time in registers, four slices of (L,R) int16 pairs are staged per frame row in LDS and leave as five whole 128-byte
//     lines per row; the one wait for memory in a step sits where the loads are a step old and the last stores half a step;
//   * any other layout goes slice by slice through an LDS tile of coalesced 128-byte rows and indexed stores.
// Integer results are bit-exact (int arithmetic wraps exactly as in D).
#include "afg_common.h"

namespace {

constexpr int kSliceLen = 20;
constexpr int kFramesPerWave = 32;
constexpr int kStepSlices = 8;                       // slices fetched per channel per step
constexpr int kInRow = kStepSlices * 2 + 1;          // 64-bit words per tile row (+1 pad)
constexpr int kFlushSlices = 4;                      // all-stereo path: slices staged per row before a flush (4 x 160 B = five full 128-byte lines)
constexpr int kRowPairs = kFlushSlices * kSliceLen;  // (L,R) pairs per staged row
constexpr int kRowPitch = kRowPairs + 2;             // dwords; 82: rows 8-byte aligned, 2-way bank conflicts at most
static_assert(kStepSlices % kFlushSlices == 0 && kRowPairs % 16 == 0, "a step is whole flushes, a row whole 128-byte lines");

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

// One slice (20 samples) of one channel: qoa_lms_predict (:231-239: wrapping int sum, arithmetic shift), dequantisation,
// qoa_clamp_s16 (:278-286), qoa_lms_update (:241-254).  I24: every factor fits 24 signed bits -- histories are clamped int16,
// a weight starts as an int16 and moves by at most 896 (= 14336 >> 4) per sample, i.e. stays below 2^23 for 9325 samples -- so
// the products come from v_mad_i32_i24 (full rate; v_mul_lo_u32 is quarter rate), the low 32 bits of which are the wrapped
// int product; the update  w += h < 0 ? -delta : delta  is one more such mad with the sign (+-1) kept beside each history value.
__device__ __forceinline__ int mad24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <bool I24>
__device__ __forceinline__ void decode_slice(uint64_t slice, const short *dq, int (&h)[4], int (&w)[4], int (&sg)[4], int (&out)[kSliceLen])
{
    const short *dqs = dq + (int)(slice >> 60) * 8;
#pragma unroll
    for (int k = 0; k < kSliceLen; k++) {
        const int deq = dqs[(int)(slice >> (57 - 3 * k)) & 7];
        const int delta = deq >> 4;
        int pred;
        if (I24) {
            // (a chain of mads ending on the newest sample, as written: the compiler's tree of three multiplies, one mad and
            // an add3 is one instruction more, and two instead of one behind the sample just reconstructed)
            int acc = __mul24(w[0], h[0]);
            acc = mad24(w[1], h[1], acc);
            acc = mad24(w[2], h[2], acc);
            pred = mad24(w[3], h[3], acc) >> 13;
#pragma unroll
            for (int i = 0; i < 4; i++) w[i] = (int)((unsigned)__mul24(sg[i], delta) + (unsigned)w[i]);
        } else {
            pred = (int)((unsigned)w[0] * (unsigned)h[0] + (unsigned)w[1] * (unsigned)h[1] + (unsigned)w[2] * (unsigned)h[2] +
                         (unsigned)w[3] * (unsigned)h[3]) >> 13;
#pragma unroll
            for (int i = 0; i < 4; i++) w[i] += h[i] < 0 ? -delta : delta;
        }
        int rec = pred + deq;
        rec = rec < -32768 ? -32768 : (rec > 32767 ? 32767 : rec);
        h[0] = h[1]; h[1] = h[2]; h[2] = h[3]; h[3] = rec;
        if (I24) { sg[0] = sg[1]; sg[1] = sg[2]; sg[2] = sg[3]; sg[3] = (rec >> 31) | 1; }
        out[k] = rec;
    }
}

constexpr int kI24MaxSamples = 9000;                 // see decode_slice

// lgkmcnt(0) alone: a one-wavefront workgroup orders its LDS traffic without waiting for its global stores (which
// __syncthreads' vmcnt(0) does: with HBM saturated by writes that wait is the kernel's time)
__device__ __forceinline__ void lds_fence()
{
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}

// All-stereo wavefronts: (L,R) int16 pairs of four slices are staged per frame row and leave as five whole 128-byte lines per
// row.  The next step's slices are fetched into registers at the top of a step and unpacked between the second half's compute
// and its stores: the one vmcnt wait of a step covers loads a step old and stores half a step old (loads and stores share
// one in-order counter, so any wait for a load also waits for every older store).
template <bool I24>
__device__ __forceinline__ void stereo_walk(const RowInfo *rows, uint32_t *stage16, const short *dq, const uint8_t *__restrict__ bytes,
                                            int16_t *__restrict__ out_i16, float *__restrict__ out_f32, const RowInfo &me, int max_slices)
{
    const int lane = threadIdx.x;
    const int fr = lane >> 1, slot = lane & 1;
    int h[4] = { 0, 0, 0, 0 }, w[4] = { 0, 0, 0, 0 }, sg[4] = { 1, 1, 1, 1 };
    if (me.samples) {
        // LMS state from the frame header (qoa.d:489-503): history then weights, 4 x int16 big-endian
        const uint64_t *st = (const uint64_t *)(bytes + me.byte_off + 8 + 16 * slot);
        const uint64_t hh = bswap64(st[0]), ww = bswap64(st[1]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            h[i] = (short)(hh >> (48 - 16 * i));
            w[i] = (short)(ww >> (48 - 16 * i));
            sg[i] = (h[i] >> 31) | 1;
        }
    }
    // no tile: a lane fetches 16 bytes = the two channels' slices sl = s0 + 2 m + slot and trades one of the two with its
    // neighbour (the frame's other channel), which holds the slices of the other parity
    struct __attribute__((aligned(8))) Two { uint64_t lo, hi; };
    Two Q[kStepSlices / 2];
    auto fetch2 = [&](int s0) {
#pragma unroll
        for (int m = 0; m < kStepSlices / 2; m++) {
            const int sl = s0 + 2 * m + slot;
            // (always a load, on every path: the compiler's wait-count bookkeeping is not path sensitive, and a "maybe pending"
            // load at the loop head costs a vmcnt(0) behind the stores just issued.  Past the frame's last slice the frame
            // header is read instead: in the plane, and decoded into samples that are never stored)
            const uint64_t at = sl * kSliceLen < (int)me.samples ? 8 + 32 + (uint64_t)sl * 16 : 0;
            Q[m] = *(const Two *)(bytes + me.byte_off + at);
        }
    };
    auto swap1 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); };   // quad_perm [1,0,3,2]
    auto unpack = [&](uint64_t *S) {
#pragma unroll
        for (int m = 0; m < kStepSlices / 2; m++) {
            const uint64_t send = slot ? Q[m].lo : Q[m].hi;
            const uint64_t recv = ((uint64_t)swap1((uint32_t)(send >> 32)) << 32) | swap1((uint32_t)send);
            S[2 * m] = slot ? recv : Q[m].lo;
            S[2 * m + 1] = slot ? Q[m].hi : recv;
            asm volatile("" : "+v"(S[2 * m]), "+v"(S[2 * m + 1]) :: "memory");          // (here, not after the stores that follow)
        }
    };
    // four slices (S) -> staged rows -> global; `Snext`: the fetched slices are unpacked between the compute and the stores
    auto half_step = [&](const uint64_t *S, int base, uint64_t *Snext) {
#pragma unroll
        for (int j = 0; j < kFlushSlices; j++) {
            if (base + j >= max_slices) break;
            int rec[kSliceLen];
            decode_slice<I24>(bswap64(S[j]), dq, h, w, sg, rec);
            short *row16 = (short *)(stage16 + fr * kRowPitch + j * kSliceLen) + slot;
#pragma unroll
            for (int k = 0; k < kSliceLen; k++) row16[2 * k] = (short)rec[k];
        }
        if (Snext) unpack(Snext);               // the step's one vmcnt wait: loads a step old, stores half a step old
        lds_fence();
        // flush: lane = (row r of a group of 8, 16 output bytes u of a 128-byte line); five lines per row
        const int first = base * kSliceLen;                                     // first sample of the staged rows
        const int have = min(max_slices - base, kFlushSlices) * kSliceLen;      // samples staged per row
#pragma unroll
        for (int g = 0; g < kFramesPerWave / 8; g++) {
            const int r = 8 * g + (lane >> 3);
            const RowInfo m = rows[r];
            const int lim = min((int)m.samples - first, have);                  // pairs of this row to write
            const uint64_t o = m.out_off + (uint64_t)first * 2 + 4 * (lane & 7);
#pragma unroll
            for (int t = 0; t < kRowPairs / 16; t++) {
                const int k = 2 * (lane & 7) + 16 * t;
                const uint2 v = *(const uint2 *)(stage16 + r * kRowPitch + k);  // pairs k, k+1 of row r
                const short l0 = (short)(v.x & 0xffff), r0 = (short)(v.x >> 16), l1 = (short)(v.y & 0xffff), r1 = (short)(v.y >> 16);
                if (k + 2 <= lim) {
                    if (out_i16) *(short4 *)(out_i16 + o + 32 * t) = make_short4(l0, r0, l1, r1);
                    if (out_f32) {
                        typedef float f32x4nt __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(f32x4nt{ (float)l0 * (1.0f / 32767), (float)r0 * (1.0f / 32767),
                                                             (float)l1 * (1.0f / 32767), (float)r1 * (1.0f / 32767) },
                                                    (f32x4nt *)(out_f32 + o + 32 * t));
                    }
                } else if (k + 1 == lim) {
                    if (out_i16) *(short2 *)(out_i16 + o + 32 * t) = make_short2(l0, r0);
                    if (out_f32) *(float2 *)(out_f32 + o + 32 * t) = make_float2((float)l0 * (1.0f / 32767), (float)r0 * (1.0f / 32767));
                }
            }
        }
        lds_fence();
    };
    static_assert(kStepSlices == 2 * kFlushSlices, "a step is two flushes");
    uint64_t S[kStepSlices], Sn[kStepSlices];
    fetch2(0);
    unpack(S);
    for (int s0 = 0; s0 < max_slices; s0 += kStepSlices) {
        fetch2(s0 + kStepSlices);
        half_step(S, s0, nullptr);
        half_step(S + kFlushSlices, s0 + kFlushSlices, Sn);
#pragma unroll
        for (int j = 0; j < kStepSlices; j++) S[j] = Sn[j];
    }
}

__global__ __launch_bounds__(64) void qoa_decode_kernel(
    const afg_qoa_frame *__restrict__ frames, const uint8_t *__restrict__ bytes,
    int16_t *__restrict__ out_i16, float *__restrict__ out_f32, uint64_t n_frames)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage16[kFramesPerWave * kRowPitch];      // all-stereo path: (L,R) int16 pairs
    float *const stage = (float *)stage16;                                                     // general path: one slice of floats ...
    uint64_t *const tile = (uint64_t *)(stage16 + kFramesPerWave * kSliceLen * 2);             // ... and behind it the slice tile
    static_assert(sizeof(uint32_t) * kFramesPerWave * kRowPitch >= sizeof(float) * kFramesPerWave * kSliceLen * 2 + 8 * kFramesPerWave * kInRow, "staging overlay");
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

    // all-stereo wavefronts (every frame two channels, rows 16-byte aligned)
    if (__ballot((C != 2 || (me.out_off & 3)) && me.samples != 0) == 0) {
        if (max_samples <= kI24MaxSamples) stereo_walk<true>(rows, stage16, dq, bytes, out_i16, out_f32, me, max_slices);
        else stereo_walk<false>(rows, stage16, dq, bytes, out_i16, out_f32, me, max_slices);
        return;
    }

    // any other layout: one slice per step through a float staging tile, indexed stores
    for (int pair = 0; pair < max_pairs; pair++) {
        const int ch = 2 * pair + slot;
        const bool active = ch < C;
        int h[4] = { 0, 0, 0, 0 }, w[4] = { 0, 0, 0, 0 }, sg[4] = { 1, 1, 1, 1 };
        if (active) {
            const uint64_t *st = (const uint64_t *)(bytes + me.byte_off + 8 + 16 * ch);
            const uint64_t hh = bswap64(st[0]), ww = bswap64(st[1]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                h[i] = (short)(hh >> (48 - 16 * i));
                w[i] = (short)(ww >> (48 - 16 * i));
            }
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
                int rec[kSliceLen];
                decode_slice<false>(bswap64(tile[fr * kInRow + 2 * j + slot]), dq, h, w, sg, rec);
                // stage [frame][sample][slot]
#pragma unroll
                for (int k = 0; k < kSliceLen; k++) stage[(fr * kSliceLen + k) * 2 + slot] = (float)rec[k];
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
