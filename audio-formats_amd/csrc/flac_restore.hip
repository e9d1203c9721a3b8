// flac_restore.hip -- FLAC LPC sample restore + channel decorrelation on gfx950.
//
// Replaces, for whole batches of frames, the prediction half of the fused
// Rice+predict loop (reference drflac.d:1235 with drflac__calculate_prediction_32
// / _64, drflac.d:1060-1140) and the decorrelate / shift / interleave of
// drflac_read_s32 (drflac.d:2885-2941), optionally followed by the int32 -> float
// conversion of stream.d:505-511.  Integer results are bit-exact by construction:
//
//   * the LPC recurrence is serial inside a subframe (floor shift: not a scan), so the parallel axis is subframes: a
//     wavefront takes 32 consecutive frames and a lane ONE channel of a frame, with the last `order` samples and the
//     coefficients in registers;
//   * prediction_32 is a wrapping int32 sum: v_mul_lo_u32 + add per tap.  When a wavefront holds `use64` subframes the
//     sum is kept in int64 instead; its low 32 bits are exactly the wrapping sum, so the flag only selects which bits
//     are shifted (drflac.d:1098 vs :1139);
//   * residual planes are subframe-major in HBM (what the Rice decoder writes, as int32 or -- 16-bit material -- int16
//     rows), so a wavefront moves 32 frames x 2 channels x 32 samples per step through an 8 KB LDS tile.  The next step's
//     rows are already in flight (16-byte loads parked in registers) while the current step runs its recurrence; the
//     tile is stored as 16-byte pieces rotated by the row, which makes the one-row-per-lane 16-byte accesses of the
//     recurrence conflict-free; outputs leave as interleaved 16-byte stores with the decorrelation done on the way.
#include "afg_common.h"

#include <mutex>

namespace {

#ifndef AFG_FLAC_TILE
#define AFG_FLAC_TILE 32
#endif
constexpr int kT = AFG_FLAC_TILE;      // samples per tile step (16 or 32)
constexpr int kRowWords = 2 * kT;      // LDS row: [channel A | channel B], 2*kT/4 pieces of 4 words
constexpr int kPieces = kT / 4;        // 16-byte pieces per channel chunk
constexpr int kLoads = 2 * kPieces;    // 16-byte load instructions per tile step

struct RowMeta {                       // what the load/store phases need to know about a lane's frame
    uint64_t in_off;
    uint64_t out_off;
    uint32_t bs;
    uint32_t info;                     // channels | assignment << 8 | bps << 16 | res16 << 24
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

// word offset of 16-byte piece `piece` (0..7) of tile row `row`
__device__ __forceinline__ int piece_off(int row, int piece)
{
    // rotation making the one-row-per-lane 16-byte accesses conflict-free: row/2 for 8 pieces per
    // row (two rows share a 256-byte bank row), row for 16 pieces per row
    const int rot = (kT == 16) ? (row >> 1) : row;
    return row * kRowWords + (((piece + rot) & (2 * kPieces - 1)) << 2);
}

__device__ __forceinline__ int32_t shl32(int32_t v, unsigned sh) { return (int32_t)((uint32_t)v << (sh & 31u)); }

// One tile of one channel of this lane's frame: kT steps of
//   s[t] = r[t] + (sum_k coef[k]*s[t-1-k]) >> shift        (drflac.d:1235)
// for t >= order, verbatim warm-up below (drflac.d:1406-1410, :1419-1423).
// Coefficients past `order` are zero, values past the end of the block are never stored:
// no branches, so the unrolled history shift is pure register renaming.
// prediction of one sample from the history h (h[k] = s[t-1-k]); two independent partial sums
// halve the dependent multiply-add chain (integer sums are associative: same bits).
template <int MAXORD, bool WIDE>
__device__ __forceinline__ int32_t predict(const int32_t (&c)[MAXORD], const int32_t (&h)[MAXORD], int shift, bool use64)
{
    int64_t a0 = 0, a1 = 0;
#pragma unroll
    for (int k = MAXORD - 1; k >= 1; k -= 2) {            // older taps first: they do not wait for the newest output
        a1 += (int64_t)c[k] * (int64_t)h[k];
        a0 += (int64_t)c[k - 1] * (int64_t)h[k - 1];
    }
    if (WIDE) {
        const int64_t acc = a0 + a1;
        const int32_t p32 = (int32_t)(uint32_t)(uint64_t)acc >> shift;     // prediction_32 (:1098)
        const int32_t p64 = (int32_t)(uint32_t)(uint64_t)(acc >> shift);   // prediction_64 (:1139)
        return use64 ? p64 : p32;
    }
    return (int32_t)((uint32_t)(uint64_t)a0 + (uint32_t)(uint64_t)a1) >> shift;
}

// ---------------------------------------------------------------------------------------------------------------
// Lane = subframe (round 3).  Round 2 gave a lane a whole frame and interleaved its two channels' recurrences: 48
// coefficient / history registers and a 16 KB tile per wavefront -- two wavefronts per SIMD, each parked at a memory
// wait a third of the time (profiles/r02_pmc_flac_restore_kernel.json; that walk is in the history of this file).  Here a
// wavefront takes 32 frames and a lane ONE channel of a frame (lane = 2 * frame row + channel slot): half the registers, an 8 KB tile, so four wavefronts per
// SIMD share the same rows-per-step memory pattern (64-byte residual pieces in, 256-byte interleaved pieces out) with
// twice the bytes in flight.  The tile keeps its layout -- row = frame, [slot A | slot B], 16-byte pieces rotated by the
// row -- which is conflict-free for this lane mapping as well (a 16-lane LDS group holds 8 rows x 2 slots).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFpw = 32;                                   // frames per wavefront

template <int MAXORD, bool WIDE>
__device__ __forceinline__ void restore_tile1(int32_t *tile, int row, int slot, int t0, int order, int shift, bool u64,
                                              const int32_t (&c)[MAXORD], int32_t (&h)[MAXORD])
{
    // the LDS reads run two pieces ahead of the recurrence (they may not pass a write the compiler cannot tell apart)
    int4 v[kT / 4];
    v[0] = *(const int4 *)(tile + piece_off(row, slot * kPieces + 0));
    v[1] = *(const int4 *)(tile + piece_off(row, slot * kPieces + 1));
#pragma unroll
    for (int q = 0; q < kT / 4; q++) {
        if (q + 2 < kT / 4) v[q + 2] = *(const int4 *)(tile + piece_off(row, slot * kPieces + q + 2));
        int32_t r[4] = { v[q].x, v[q].y, v[q].z, v[q].w };
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int t = t0 + 4 * q + e;
            const int32_t p = predict<MAXORD, WIDE>(c, h, shift, u64);
            const int32_t sv = (t >= order) ? (int32_t)((uint32_t)r[e] + (uint32_t)p) : r[e];
            r[e] = sv;
#pragma unroll
            for (int k = MAXORD - 1; k >= 1; k--) h[k] = h[k - 1];
            h[0] = sv;
        }
        *(int4 *)(tile + piece_off(row, slot * kPieces + q)) = make_int4(r[0], r[1], r[2], r[3]);
    }
}

template <int MODE> struct Loads1 { static constexpr int n = MODE == 1 ? kLoads / 4 : kLoads / 2; };

template <int MODE>
__device__ __forceinline__ void load_tile1(int4 (&nxt)[Loads1<MODE>::n], const RowMeta *meta, const int32_t *__restrict__ res,
                                           int pair, int t0)
{
    const int lane = threadIdx.x;
    if (MODE == 1) {
        constexpr int P16 = kPieces / 2;                     // lanes per row-chunk, 8 samples (16 bytes) each
#pragma unroll
        for (int i = 0; i < Loads1<1>::n; i++) {
            const int rc = (64 / P16) * i + lane / P16;
            const int row = rc >> 1, slot = rc & 1, p = lane % P16;
            const RowMeta m = meta[row];
            const int C = (int)(m.info & 0xff);
            const int cidx = 2 * pair + slot;
            const int t = t0 + 8 * p;
            int4 v = make_int4(0, 0, 0, 0);
            if (cidx < C && t < (int)m.bs)
                v = *(const int4 *)((const int16_t *)res + m.in_off + (uint64_t)cidx * (((uint64_t)m.bs + 7u) & ~(uint64_t)7u) + (uint64_t)t);
            nxt[i] = v;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < Loads1<MODE>::n; i++) {
        const int rc = (64 / kPieces) * i + lane / kPieces;
        const int row = rc >> 1, slot = rc & 1, p = lane % kPieces;
        const RowMeta m = meta[row];
        const int C = (int)(m.info & 0xff);
        const int cidx = 2 * pair + slot;
        const int t = t0 + 4 * p;
        int4 v = make_int4(0, 0, 0, 0);
        if (MODE == 2 && cidx < C && t < (int)m.bs && (m.info >> 24)) {
            const int16_t *src = (const int16_t *)res + m.in_off + (uint64_t)cidx * (((uint64_t)m.bs + 7u) & ~(uint64_t)7u) + (uint64_t)t;
            const int2 w = *(const int2 *)src;
            v.x = w.x;
            v.y = w.y;
        } else if (cidx < C && t < (int)m.bs) {
            const int32_t *src = res + m.in_off + (uint64_t)cidx * m.bs + (uint64_t)t;
            if (t + 3 < (int)m.bs) {
                v = *(const int4 *)src;                              // may be 4-byte aligned only (odd block sizes)
            } else {
                v.x = src[0];
                if (t + 1 < (int)m.bs) v.y = src[1];
                if (t + 2 < (int)m.bs) v.z = src[2];
            }
        }
        nxt[i] = v;
    }
}

template <int MODE>
__device__ __forceinline__ void park_tile1(int32_t *tile, const RowMeta *meta, const int4 (&nxt)[Loads1<MODE>::n])
{
    const int lane = threadIdx.x;
    if (MODE == 1) {
        constexpr int P16 = kPieces / 2;
#pragma unroll
        for (int i = 0; i < Loads1<1>::n; i++) {
            const int rc = (64 / P16) * i + lane / P16;
            const int piece = (rc & 1) * kPieces + 2 * (lane % P16);
            const int4 v = nxt[i];
            *(int4 *)(tile + piece_off(rc >> 1, piece)) = make_int4((int)(int16_t)v.x, v.x >> 16, (int)(int16_t)v.y, v.y >> 16);
            *(int4 *)(tile + piece_off(rc >> 1, piece + 1)) = make_int4((int)(int16_t)v.z, v.z >> 16, (int)(int16_t)v.w, v.w >> 16);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < Loads1<MODE>::n; i++) {
        const int rc = (64 / kPieces) * i + lane / kPieces;
        int4 v = nxt[i];
        if (MODE == 2 && (meta[rc >> 1].info >> 24))                 // int16 row: four samples in two dwords
            v = make_int4((int)(int16_t)v.x, v.x >> 16, (int)(int16_t)v.y, v.y >> 16);
        *(int4 *)(tile + piece_off(rc >> 1, (rc & 1) * kPieces + lane % kPieces)) = v;
    }
}

// decorrelate (drflac.d:2885-2941), shift, interleave, convert; 4 rows per instruction
__device__ __forceinline__ void store_tile1(const int32_t *tile, const RowMeta *meta, const uint32_t *row_shift,
                                            int32_t *__restrict__ out_i32, float *__restrict__ out_f32, int pair, int t0)
{
    const int lane = threadIdx.x;
#pragma unroll 2
    for (int i = 0; i < kFpw / (128 / kT); i++) {
        const int row = (128 / kT) * i + lane / (kT / 2);
        const int q = lane % (kT / 2);                               // samples 2q, 2q+1 of the tile
        const RowMeta m = meta[row];
        const int C = (int)(m.info & 0xff);
        const int asg = (int)((m.info >> 8) & 0xff);
        const int t = t0 + 2 * q;
        if (2 * pair >= C || t >= (int)m.bs) continue;
        const int2 a = *(const int2 *)(tile + piece_off(row, q >> 1) + 2 * (q & 1));
        const int2 b = *(const int2 *)(tile + piece_off(row, kPieces + (q >> 1)) + 2 * (q & 1));
        const bool two = (C - 2 * pair) >= 2;
        int32_t l0, r0, l1, r1;
        if (asg == AFG_FLAC_LEFT_SIDE) {                             // :2886-2897
            l0 = a.x; r0 = (int32_t)((uint32_t)a.x - (uint32_t)b.x);
            l1 = a.y; r1 = (int32_t)((uint32_t)a.y - (uint32_t)b.y);
        } else if (asg == AFG_FLAC_RIGHT_SIDE) {                     // :2899-2909
            l0 = (int32_t)((uint32_t)b.x + (uint32_t)a.x); r0 = b.x;
            l1 = (int32_t)((uint32_t)b.y + (uint32_t)a.y); r1 = b.y;
        } else if (asg == AFG_FLAC_MID_SIDE) {                       // :2911-2920
            const int32_t m0 = (int32_t)(((uint32_t)a.x << 1) | (uint32_t)(b.x & 1));
            const int32_t m1 = (int32_t)(((uint32_t)a.y << 1) | (uint32_t)(b.y & 1));
            l0 = (int32_t)((uint32_t)m0 + (uint32_t)b.x) >> 1; r0 = (int32_t)((uint32_t)m0 - (uint32_t)b.x) >> 1;
            l1 = (int32_t)((uint32_t)m1 + (uint32_t)b.y) >> 1; r1 = (int32_t)((uint32_t)m1 - (uint32_t)b.y) >> 1;
        } else {                                                     // :2922-2940
            l0 = a.x; r0 = b.x; l1 = a.y; r1 = b.y;
        }
        const unsigned shA = row_shift[row * 8 + 2 * pair];
        const unsigned shB = row_shift[row * 8 + ((2 * pair + 1) & 7)];
        l0 = shl32(l0, shA); l1 = shl32(l1, shA);
        r0 = shl32(r0, shB); r1 = shl32(r1, shB);
        const bool second = (t + 1 < (int)m.bs);
        const double factor = 1.0 / 2147483647.0;                    // stream.d:507
        if (C == 2 && second) {
            const uint64_t o = m.out_off + (uint64_t)t * 2;
            if (out_i32) *(int4 *)(out_i32 + o) = make_int4(l0, r0, l1, r1);
            if (out_f32)
                *(float4 *)(out_f32 + o) = make_float4((float)((double)l0 * factor), (float)((double)r0 * factor),
                                                       (float)((double)l1 * factor), (float)((double)r1 * factor));
        } else {
            const int32_t vals[4] = { l0, r0, l1, r1 };
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int smp = e >> 1, slot = e & 1;
                if ((slot && !two) || (smp && !second)) continue;
                const uint64_t o = m.out_off + (uint64_t)(t + smp) * C + (2 * pair + slot);
                if (out_i32) out_i32[o] = vals[e];
                if (out_f32) out_f32[o] = (float)((double)vals[e] * factor);
            }
        }
    }
}

template <int MAXORD, bool WIDE, int MODE>
__device__ __forceinline__ void run_frames1(int32_t *tile, const RowMeta *meta, const RowMeta &me, bool valid,
                                            const afg_flac_subframe *__restrict__ subframes, uint32_t sf_index,
                                            const int32_t *__restrict__ res, int32_t *__restrict__ out_i32,
                                            float *__restrict__ out_f32, int max_bs, int max_pairs, const uint32_t *row_shift)
{
    const int lane = threadIdx.x, row = lane >> 1, slot = lane & 1;
    const int my_ch = valid ? (int)(me.info & 0xff) : 0;
    for (int pair = 0; pair < max_pairs; pair++) {
        int32_t c[MAXORD], h[MAXORD];
        int order = 0, shift = 0;
        bool u64 = false;
        const int ch = 2 * pair + slot;
#pragma unroll
        for (int k = 0; k < MAXORD; k++) { c[k] = 0; h[k] = 0; }
        if (ch < my_ch) {
            const afg_flac_subframe *sf = subframes + sf_index + ch;
            order = sf->order; shift = sf->shift; u64 = sf->use64 != 0;
#pragma unroll
            for (int k = 0; k < MAXORD; k++) c[k] = (k < order) ? (int32_t)sf->coef[k] : 0;
        }
        int4 nxt[Loads1<MODE>::n];
        load_tile1<MODE>(nxt, meta, res, pair, 0);
        park_tile1<MODE>(tile, meta, nxt);
        __syncthreads();
        for (int t0 = 0; t0 < max_bs; t0 += kT) {
            if (t0 + kT < max_bs) load_tile1<MODE>(nxt, meta, res, pair, t0 + kT);   //  2 no row loads, 3 no recurrence, 4 no store phase at all)
            if (t0 < (int)me.bs) restore_tile1<MAXORD, WIDE>(tile, row, slot, t0, order, shift, u64, c, h);
            __syncthreads();
            // make the prefetched residuals resident here: loads and stores share one in-order counter
#pragma unroll
            for (int i = 0; i < Loads1<MODE>::n; i++)
                asm volatile("" : "+v"(nxt[i].x), "+v"(nxt[i].y), "+v"(nxt[i].z), "+v"(nxt[i].w) : : "memory");
            store_tile1(tile, meta, row_shift, out_i32, out_f32, pair, t0);
            __syncthreads();
            if (t0 + kT < max_bs) park_tile1<MODE>(tile, meta, nxt);
            __syncthreads();
        }
    }
}

// One kernel per (order bucket, accumulator width): a wavefront runs only in the instantiation that matches the largest
// LPC order / widest accumulator among the subframes of its 32 frames and leaves the others at once.
#ifndef AFG_FLAC_WAVES
#define AFG_FLAC_WAVES 4               // wavefronts per SIMD the orders <= 12 instantiations are compiled for (register budget)
#endif
template <int LO, int MAXORD, bool WIDE>
__global__ __launch_bounds__(64, (MAXORD <= 12 ? AFG_FLAC_WAVES : 2)) void flac_restore1_kernel(
    const afg_flac_frame *__restrict__ frames, const afg_flac_subframe *__restrict__ subframes,
    const int32_t *__restrict__ res, int32_t *__restrict__ out_i32, float *__restrict__ out_f32, uint64_t n_frames)
{
    static_assert(kT == 32, "the lane-per-subframe walk is written for 32-sample tile steps");
    __shared__ __attribute__((aligned(16))) int32_t tile[kFpw * kRowWords];
    __shared__ RowMeta meta[kFpw];
    __shared__ uint32_t row_shift[kFpw * 8];

    const int lane = threadIdx.x;
    const uint64_t f = (uint64_t)blockIdx.x * kFpw + (lane >> 1);
    const bool valid = f < n_frames;

    RowMeta me;
    me.in_off = 0; me.out_off = 0; me.bs = 0; me.info = 0;
    uint32_t sf_index = 0;
    int my_order = 0, my_wide = 0;
    afg_flac_frame fr;
    if (valid) {
        fr = frames[f];
        sf_index = fr.sf_index;
        for (int c = 0; c < (int)fr.channels && c < 8; c++) {
            const afg_flac_subframe *sf = subframes + sf_index + c;
            my_order = sf->order > my_order ? sf->order : my_order;
            my_wide |= sf->use64;
        }
    }
    const int max_order = wave_max(my_order);
    const bool wide = wave_max(my_wide) != 0;
    if (!(max_order > LO && max_order <= MAXORD && wide == WIDE)) return;

    if (valid) {
        me.in_off = fr.in_off;
        me.out_off = fr.out_off;
        me.bs = fr.block_size;
        me.info = (uint32_t)fr.channels | ((uint32_t)fr.assignment << 8) | ((uint32_t)fr.bps << 16) | ((uint32_t)(fr.res16 != 0) << 24);
    }
    if ((lane & 1) == 0) {
        for (int c = 0; c < 8; c++) {
            uint32_t sh = 0;
            if (valid && c < (int)fr.channels) sh = (32u - fr.bps) + subframes[sf_index + c].wasted;   // drflac.d:2883, :2894
            row_shift[(lane >> 1) * 8 + c] = sh;
        }
        meta[lane >> 1] = me;
    }
    __syncthreads();

    const int max_bs = wave_max((int)me.bs);
    const int max_pairs = wave_max(((int)(me.info & 0xff) + 1) >> 1);
    const bool any16 = __any(valid && (me.info >> 24) != 0), any32 = __any(valid && (me.info >> 24) == 0);
    if (!any16)
        run_frames1<MAXORD, WIDE, 0>(tile, meta, me, valid, subframes, sf_index, res, out_i32, out_f32, max_bs, max_pairs, row_shift);
    else if (!any32)
        run_frames1<MAXORD, WIDE, 1>(tile, meta, me, valid, subframes, sf_index, res, out_i32, out_f32, max_bs, max_pairs, row_shift);
    else
        run_frames1<MAXORD, WIDE, 2>(tile, meta, me, valid, subframes, sf_index, res, out_i32, out_f32, max_bs, max_pairs, row_shift);
}

}  // namespace

namespace {

// second stream + fork / join events of a device: populated instantiations of one call run side by side (each is a few
// waves of long-lived wavefronts per SIMD with its own tail; together they share one).  One enqueue sequence at a time.
struct SideLane {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
std::mutex g_side_mu;
SideLane g_side[AFG_MAX_DEVICES];

int launch_variants(uint64_t n_frames, const afg_flac_frame *d_frames, const afg_flac_subframe *d_subframes, const int32_t *d_res,
                    int32_t *d_out_i32, float *d_out_f32, uint32_t variants, hipStream_t stream)
{
    if (n_frames == 0) return AFG_OK;
    if (!d_frames || !d_subframes || !d_res || (!d_out_i32 && !d_out_f32)) {
        afg::set_error("afg_flac_transform_hip: NULL device pointer");
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    const uint64_t groups1 = (n_frames + kFpw - 1) / kFpw;
    if (groups1 > 0x7fffffffull) {
        afg::set_error("afg_flac_transform_hip: too many frames in one call");
        return AFG_ERR_INVALID;
    }
    const uint32_t all = 0xffu;                                      // eight instantiations: (order bucket, accumulator width)
    variants &= all;
    if (!variants) return AFG_OK;
    // a known set of two or more: alternate between the caller's stream and the device's side stream
    const bool two = variants != all && (variants & (variants - 1)) != 0;
    std::unique_lock<std::mutex> lk(g_side_mu, std::defer_lock);
    SideLane *side = nullptr;
    if (two) {
        int dev = 0;
        if (int rc = afg::device_slot(&dev, "afg_flac_transform_variants_hip")) return rc;
        {
            lk.lock();
            side = &g_side[dev];
            if (!side->stream) {
                AFG_HIP_CHECK(hipStreamCreateWithFlags(&side->stream, hipStreamNonBlocking));
                AFG_HIP_CHECK(hipEventCreateWithFlags(&side->fork, hipEventDisableTiming));
                AFG_HIP_CHECK(hipEventCreateWithFlags(&side->join, hipEventDisableTiming));
            }
            AFG_HIP_CHECK(hipEventRecord(side->fork, stream));
            AFG_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
        }
    }
    int idx = 0, used = 0;
#define AFG_FLAC_LAUNCH2(LO, HI, W)                                                                                    \
    if (variants & (1u << idx)) {                                                                                      \
        hipLaunchKernelGGL((flac_restore1_kernel<LO, HI, W>), dim3((uint32_t)groups1), dim3(64), 0,                    \
                           (side && (used & 1)) ? side->stream : stream, d_frames, d_subframes, d_res, d_out_i32,      \
                           d_out_f32, n_frames);                                                                       \
        used++;                                                                                                        \
    }                                                                                                                  \
    idx++
    AFG_FLAC_LAUNCH2(-1, 4, false);  AFG_FLAC_LAUNCH2(-1, 4, true);
    AFG_FLAC_LAUNCH2(4, 8, false);   AFG_FLAC_LAUNCH2(4, 8, true);
    AFG_FLAC_LAUNCH2(8, 12, false);  AFG_FLAC_LAUNCH2(8, 12, true);
    AFG_FLAC_LAUNCH2(12, 32, false); AFG_FLAC_LAUNCH2(12, 32, true);
#undef AFG_FLAC_LAUNCH2
    if (side) {
        AFG_HIP_CHECK(hipEventRecord(side->join, side->stream));
        AFG_HIP_CHECK(hipStreamWaitEvent(stream, side->join, 0));
    }
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}

}  // namespace

// The instantiations a batch populates, exactly as the kernels decide it per wavefront (32 frames): bucket of the largest LPC
// order (<= 4, <= 8, <= 12, <= 32) and whether any subframe needs the 64-bit accumulator -- bit 2 * bucket + wide.
extern "C" uint32_t afg_flac_variants(uint64_t n_frames, const afg_flac_frame *frames, const afg_flac_subframe *subframes)
{
    uint32_t mask = 0;
    if (!frames || !subframes) return 0xffu;
    for (uint64_t g = 0; g < n_frames; g += kFpw) {
        int order = 0, wide = 0;
        for (uint64_t f = g; f < n_frames && f < g + kFpw; f++)
            for (int c = 0; c < (int)frames[f].channels && c < 8; c++) {
                const afg_flac_subframe &sf = subframes[frames[f].sf_index + c];
                if (sf.order > order) order = sf.order;
                wide |= sf.use64 != 0;
            }
        const int bucket = order <= 4 ? 0 : order <= 8 ? 1 : order <= 12 ? 2 : 3;
        mask |= 1u << (bucket * 2 + wide);
    }
    return mask;
}

// Stream-ordered: the records are read on the device when `hip_stream` gets there, never on the host at enqueue time (they
// may be the product of earlier work on the stream), so every instantiation is launched and the unpopulated ones leave at
// once.  A caller that still holds the records in host memory names the populated ones: afg_flac_variants +
// afg_flac_transform_variants_hip.
extern "C" int afg_flac_transform_hip(uint64_t n_frames, const afg_flac_frame *d_frames,
                                      const afg_flac_subframe *d_subframes, const int32_t *d_res,
                                      int32_t *d_out_i32, float *d_out_f32, void *hip_stream)
{
    return launch_variants(n_frames, d_frames, d_subframes, d_res, d_out_i32, d_out_f32, 0xffu, (hipStream_t)hip_stream);
}

extern "C" int afg_flac_transform_variants_hip(uint64_t n_frames, const afg_flac_frame *d_frames,
                                               const afg_flac_subframe *d_subframes, const int32_t *d_res,
                                               int32_t *d_out_i32, float *d_out_f32, uint32_t variants, void *hip_stream)
{
    return launch_variants(n_frames, d_frames, d_subframes, d_res, d_out_i32, d_out_f32, variants, (hipStream_t)hip_stream);
}
