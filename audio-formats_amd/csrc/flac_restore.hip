// flac_restore.hip -- FLAC LPC sample restore + channel decorrelation on gfx950.
//
// Replaces, for whole batches of frames, the prediction half of the fused
// Rice+predict loop (reference drflac.d:1235 with drflac__calculate_prediction_32
// / _64, drflac.d:1060-1140) and the decorrelate / shift / interleave of
// drflac_read_s32 (drflac.d:2885-2941), optionally followed by the int32 -> float
// conversion of stream.d:505-511.  Integer results are bit-exact by construction:
//
//   * the LPC recurrence is serial inside a subframe (floor shift: not a scan),
//     so the parallel axis is frames: one lane owns one frame and runs the
//     recurrence of its (up to two at a time) channels with the last `order`
//     samples and the coefficients in registers;
//   * both reference accumulators come out of ONE int64 multiply-add chain: the
//     low 32 bits of the 64-bit sum are exactly the wrapping int32 sum of
//     drflac__calculate_prediction_32, so `use64` only selects which bits are
//     shifted (drflac.d:1098 vs :1139);
//   * residual planes are subframe-major in HBM (what the Rice decoder writes),
//     so a wavefront moves 64 frames x 2 channels x 32 samples through an LDS
//     tile per step: coalesced 128-byte row reads in, one-row-per-lane in the
//     recurrence (row stride 65 words: conflict-free), coalesced 256-byte
//     interleaved rows out with the decorrelation done on the way out.
#include "afg_common.h"

namespace {

constexpr int kT = 32;                 // samples per tile step
constexpr int kRow = 2 * kT + 1;       // LDS row: [ch0 | ch1] + 1 pad word

struct RowMeta {                       // what the load/store phases need to know about a lane's frame
    uint64_t in_off;
    uint64_t out_off;
    uint32_t bs;
    uint32_t info;                     // channels | assignment << 8 | bps << 16
};

__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int other = __shfl_xor(v, o);
        v = other > v ? other : v;
    }
    return v;
}

// One tile of one channel of this lane's frame: kT steps of
//   s[t] = r[t] + (sum_k coef[k]*s[t-1-k]) >> shift        (drflac.d:1235)
// for t >= order, verbatim warm-up below (drflac.d:1406-1410, :1419-1423).
template <int MAXORD>
__device__ __forceinline__ void restore_tile(int32_t *row, int t0, int bs, int order, int shift, bool use64,
                                             const int32_t (&c)[MAXORD], int32_t (&h)[MAXORD])
{
#pragma unroll
    for (int j = 0; j < kT; j++) {
        const int t = t0 + j;
        int32_t r = row[j];
        // taps on the older samples first: they do not wait for the newest output
        int64_t acc = 0;
#pragma unroll
        for (int k = MAXORD - 1; k >= 1; k--) acc += (int64_t)c[k] * (int64_t)h[k];
        acc += (int64_t)c[0] * (int64_t)h[0];
        const int32_t p32 = (int32_t)(uint32_t)(uint64_t)acc >> shift;   // prediction_32: wrapped int32 sum, arithmetic shift
        const int32_t p64 = (int32_t)(uint32_t)(uint64_t)(acc >> shift); // prediction_64: shift in 64 bits, then truncate
        const int32_t pred = use64 ? p64 : p32;
        const int32_t s = (t >= order) ? (int32_t)((uint32_t)r + (uint32_t)pred) : r;
        if (t < bs) {
            row[j] = s;
#pragma unroll
            for (int k = MAXORD - 1; k >= 1; k--) h[k] = h[k - 1];
            h[0] = s;
        }
    }
}

__device__ __forceinline__ int32_t shl32(int32_t v, unsigned sh) { return (int32_t)((uint32_t)v << (sh & 31u)); }

template <int MAXORD>
__device__ __forceinline__ void run_frames(int32_t *tile, const RowMeta *meta, const RowMeta &me, bool valid,
                                           const afg_flac_subframe *__restrict__ subframes, uint32_t sf_index,
                                           const int32_t *__restrict__ res, int32_t *__restrict__ out_i32,
                                           float *__restrict__ out_f32, int max_bs, int max_pairs,
                                           const uint32_t *row_shift /* LDS [64][8] packed shifts */)
{
    const int lane = threadIdx.x;
    const int my_ch = valid ? (int)(me.info & 0xff) : 0;
    const int half = lane >> 5;       // channel slot in load phase
    const int w = lane & 31;

    for (int pair = 0; pair < max_pairs; pair++) {
        // this lane's two subframes of the pair
        int32_t c0[MAXORD], c1[MAXORD];
        int32_t h0[MAXORD], h1[MAXORD];
        int order0 = 0, order1 = 0, shift0 = 0, shift1 = 0;
        bool u0 = false, u1 = false;
        const int chA = 2 * pair, chB = 2 * pair + 1;
#pragma unroll
        for (int k = 0; k < MAXORD; k++) { c0[k] = c1[k] = 0; h0[k] = h1[k] = 0; }
        if (chA < my_ch) {
            const afg_flac_subframe *sf = subframes + sf_index + chA;
            order0 = sf->order; shift0 = sf->shift; u0 = sf->use64 != 0;
#pragma unroll
            for (int k = 0; k < MAXORD; k++) c0[k] = (k < order0) ? (int32_t)sf->coef[k] : 0;
        }
        if (chB < my_ch) {
            const afg_flac_subframe *sf = subframes + sf_index + chB;
            order1 = sf->order; shift1 = sf->shift; u1 = sf->use64 != 0;
#pragma unroll
            for (int k = 0; k < MAXORD; k++) c1[k] = (k < order1) ? (int32_t)sf->coef[k] : 0;
        }

        for (int t0 = 0; t0 < max_bs; t0 += kT) {
            // ---- load: row r <- frame r, lanes 0..31 channel A, lanes 32..63 channel B (128 B each)
            for (int r = 0; r < 64; r++) {
                const RowMeta m = meta[r];
                const int C = (int)(m.info & 0xff);
                const int cidx = 2 * pair + half;
                const int t = t0 + w;
                int32_t v = 0;
                if (cidx < C && t < (int)m.bs) v = res[m.in_off + (uint64_t)cidx * m.bs + (uint64_t)t];
                tile[r * kRow + half * kT + w] = v;
            }
            __syncthreads();

            // ---- recurrence: lane = frame
            if (t0 < (int)me.bs) {
                if (chA < my_ch) restore_tile<MAXORD>(tile + lane * kRow, t0, (int)me.bs, order0, shift0, u0, c0, h0);
                if (chB < my_ch) restore_tile<MAXORD>(tile + lane * kRow + kT, t0, (int)me.bs, order1, shift1, u1, c1, h1);
            }
            __syncthreads();

            // ---- store: decorrelate (drflac.d:2885-2941), shift, interleave; row by row
            for (int r = 0; r < 64; r++) {
                const RowMeta m = meta[r];
                const int C = (int)(m.info & 0xff);
                const int asg = (int)((m.info >> 8) & 0xff);
                if (2 * pair >= C) continue;
                const int npair = (C - 2 * pair) >= 2 ? 2 : 1;        // channels of this pair present
                // lane -> (sample j, channel slot s)
                const int j = (npair == 2) ? (lane >> 1) : lane;
                const int s = (npair == 2) ? (lane & 1) : 0;
                const int t = t0 + j;
                if (j >= kT || t >= (int)m.bs) continue;
                const int32_t a = tile[r * kRow + j];
                const int32_t b = tile[r * kRow + kT + j];
                int32_t v;
                if (asg == AFG_FLAC_LEFT_SIDE) {                      // :2886-2897
                    v = s ? (int32_t)((uint32_t)a - (uint32_t)b) : a;
                } else if (asg == AFG_FLAC_RIGHT_SIDE) {              // :2899-2909
                    v = s ? b : (int32_t)((uint32_t)b + (uint32_t)a);
                } else if (asg == AFG_FLAC_MID_SIDE) {                // :2911-2920
                    const int32_t mid = (int32_t)(((uint32_t)a << 1) | (uint32_t)(b & 1));
                    v = s ? ((int32_t)((uint32_t)mid - (uint32_t)b) >> 1)
                          : ((int32_t)((uint32_t)mid + (uint32_t)b) >> 1);
                } else {                                              // :2922-2940
                    v = s ? b : a;
                }
                const int cidx = 2 * pair + s;
                v = shl32(v, row_shift[r * 8 + cidx]);
                const uint64_t o = m.out_off + (uint64_t)t * C + cidx;
                if (out_i32) out_i32[o] = v;
                if (out_f32) out_f32[o] = (float)((double)v * (1.0 / 2147483647.0));   // stream.d:507-510
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(64) void flac_restore_kernel(
    const afg_flac_frame *__restrict__ frames, const afg_flac_subframe *__restrict__ subframes,
    const int32_t *__restrict__ res, int32_t *__restrict__ out_i32, float *__restrict__ out_f32,
    uint64_t n_frames)
{
    __shared__ int32_t tile[64 * kRow];
    __shared__ RowMeta meta[64];
    __shared__ uint32_t row_shift[64 * 8];

    const int lane = threadIdx.x;
    const uint64_t f = (uint64_t)blockIdx.x * 64 + lane;
    const bool valid = f < n_frames;

    RowMeta me;
    me.in_off = 0; me.out_off = 0; me.bs = 0; me.info = 0;
    uint32_t sf_index = 0;
    int my_order = 0;
    if (valid) {
        const afg_flac_frame fr = frames[f];
        me.in_off = fr.in_off;
        me.out_off = fr.out_off;
        me.bs = fr.block_size;
        me.info = (uint32_t)fr.channels | ((uint32_t)fr.assignment << 8) | ((uint32_t)fr.bps << 16);
        sf_index = fr.sf_index;
        for (int c = 0; c < 8; c++) {
            uint32_t sh = 0;
            if (c < (int)fr.channels) {
                const afg_flac_subframe *sf = subframes + sf_index + c;
                sh = (32u - fr.bps) + sf->wasted;                     // drflac.d:2883, :2894
                my_order = sf->order > my_order ? sf->order : my_order;
            }
            row_shift[lane * 8 + c] = sh;
        }
    } else {
        for (int c = 0; c < 8; c++) row_shift[lane * 8 + c] = 0;
    }
    meta[lane] = me;
    __syncthreads();

    const int max_bs = wave_max((int)me.bs);
    const int max_pairs = wave_max(((int)(me.info & 0xff) + 1) >> 1);
    const int max_order = wave_max(my_order);

#define AFG_FLAC_RUN(N) run_frames<N>(tile, meta, me, valid, subframes, sf_index, res, out_i32, out_f32, \
                                      max_bs, max_pairs, row_shift)
    if (max_order <= 4) AFG_FLAC_RUN(4);
    else if (max_order <= 8) AFG_FLAC_RUN(8);
    else if (max_order <= 12) AFG_FLAC_RUN(12);
    else if (max_order <= 16) AFG_FLAC_RUN(16);
    else AFG_FLAC_RUN(32);
#undef AFG_FLAC_RUN
}

}  // namespace

extern "C" int afg_flac_transform_hip(uint64_t n_frames, const afg_flac_frame *d_frames,
                                      const afg_flac_subframe *d_subframes, const int32_t *d_res,
                                      int32_t *d_out_i32, float *d_out_f32, void *hip_stream)
{
    if (n_frames == 0) return AFG_OK;
    if (!d_frames || !d_subframes || !d_res || (!d_out_i32 && !d_out_f32)) {
        afg::set_error("afg_flac_transform_hip: NULL device pointer");
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    const uint64_t groups = (n_frames + 63) / 64;
    if (groups > 0x7fffffffull) {
        afg::set_error("afg_flac_transform_hip: too many frames in one call");
        return AFG_ERR_INVALID;
    }
    hipLaunchKernelGGL(flac_restore_kernel, dim3((uint32_t)groups), dim3(64), 0, (hipStream_t)hip_stream,
                       d_frames, d_subframes, d_res, d_out_i32, d_out_f32, n_frames);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}
