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
//     rows), so a wavefront moves 32 frames x 2 channels x 64 samples per step through a 16 KB LDS tile (round 5; 32
//     samples and 8 KB until then).  The next step's rows are already in flight (16-byte loads parked in registers)
//     while the current step runs its recurrence; the tile is stored as 16-byte pieces rotated by the row, which makes
//     the one-row-per-lane 16-byte accesses of the recurrence conflict-free; outputs leave as interleaved 16-byte
//     stores -- 512 bytes per row and step, nontemporal -- with the decorrelation done on the way.  Two tile steps: the
//     branch-free COMMON step (stereo or mono wavefronts, complete tiles, one output; below) and the GENERAL one.
#include "afg_common.h"

#include <mutex>

namespace {

constexpr int kT = 64;                 // samples per tile step (round 5; 32 until then: piece_off keeps that layout too)
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
    // rotation by the row: the one-row-per-lane 16-byte accesses of the recurrence are conflict-free
    if (kT == 32) return row * kRowWords + (((piece + row) & (2 * kPieces - 1)) << 2);
    // 64-sample tiles: a channel chunk is a whole 256-byte bank row; rotate inside it, slot B half a row further
    const int chunk = piece / kPieces;
    return row * kRowWords + ((chunk * kPieces + ((piece + row + 8 * chunk) & (kPieces - 1))) << 2);
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
__device__ __forceinline__ void store_tile1(const int32_t *tile, const RowMeta *meta, const uint8_t *row_shift,
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
            // (nontemporal, like the common step's: the row pieces are written once)
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            if (out_i32) __builtin_nontemporal_store(i32x4{ l0, r0, l1, r1 }, (i32x4 *)(out_i32 + o));
            if (out_f32)
                __builtin_nontemporal_store(f32x4{ (float)((double)l0 * factor), (float)((double)r0 * factor), (float)((double)l1 * factor),
                                                   (float)((double)r1 * factor) }, (f32x4 *)(out_f32 + o));
        } else if (two && !(C & 1)) {
            // an even channel count above two: this pair's 8-byte column of the interleaved samples (4-byte pieces until round 6:
            // 61.8 ms per C4-sized batch of six channels)
            typedef int i32x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const uint64_t o = m.out_off + (uint64_t)t * C + 2 * pair;
            if (out_i32) {
                *(i32x2 *)(out_i32 + o) = i32x2{ l0, r0 };
                if (second) *(i32x2 *)(out_i32 + o + C) = i32x2{ l1, r1 };
            }
            if (out_f32) {
                *(f32x2 *)(out_f32 + o) = f32x2{ (float)((double)l0 * factor), (float)((double)r0 * factor) };
                if (second) *(f32x2 *)(out_f32 + o + C) = f32x2{ (float)((double)l1 * factor), (float)((double)r1 * factor) };
            }
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


// ---------------------------------------------------------------------------------------------------------------
// The common tile step (round 5).  A wavefront whose 32 frames are all stereo (or all mono: slot B stays empty), hold their residual rows in one width and
// lie within 1 GiB of each other takes every tile that is complete in all of its rows through a step without a branch:
//   * rows and PCM go through buffer instructions -- a wave-uniform base in a descriptor, the lane's part fixed for the
//     whole walk in one register per load, the tile's position in the scalar offset -- so neither side spends vector
//     instructions on addresses;
//   * the store phase is straight-line: the frame's decorrelation (drflac.d:2885-2941) is selected by masks from a
//     32-byte row record in LDS, not by divergent branches (four rows with four assignments share a store instruction),
//     and a store instruction takes rows i, i+8, i+16, i+24: the two rows a 32-lane LDS group reads sit in opposite
//     halves of the bank row;
//   * because the phase has no branch the compiler can count its eight stores, so the NEXT tile's rows -- issued before
//     the recurrence -- are waited for after the stores (vmcnt(8)), not before them: the fetch has the recurrence and
//     the store phase to land, where the general step (conditional stores: a counted wait is impossible, and an
//     uncounted one would drain the stores just issued) must have it resident before its first store.
// Everything else (other channel counts, mixed row widths, partial tiles, both outputs at once) takes the general step
// below; the two leave the tile and the parked rows in the same state, so a walk switches between them tile by tile.
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// cache policy bits of the buffer instructions (gfx940+: sc0 = 1, nt = 2, sc1 = 16).  PCM pieces are written once and never
// read, residual rows are read once: nontemporal both ways, the stores also sc1 (A/B on one box, C4: plain 13.5 ms, nt stores
// 12.06, + sc1 11.96-12.17, + nt loads 11.69-11.95)
constexpr int kStoreNt = 2 | 16;
constexpr int kLoadNt = 2;

struct RowFast {                       // 32 bytes per tile row
    uint32_t voff;                     // byte offset of the frame's PCM from the wavefront's base
    uint32_t m_ls, m_rs, m_ms;         // 0 / ~0: the frame's channel assignment
    uint32_t sh_a, sh_b;               // output shifts of the two channels (drflac.d:2883, :2894)
    uint32_t in_a, in_b;               // byte offsets of the two residual rows from the wavefront's base
};

__device__ __forceinline__ void wave_sync()
{
    // the workgroup is one wavefront and its LDS instructions execute in order: only the compiler has to be held
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr uint32_t kFastReach = 0x80000000u;          // bytes a wavefront's descriptors cover; an offset past it is "no row" (reads 0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void *p)
{
    const uint64_t a = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)(((uint64_t)hi << 32) | lo), 0, (int)kFastReach, 0x00020000);
}

template <int MODE>
__device__ __forceinline__ void load_fast(int4 (&nxt)[Loads1<MODE>::n], __amdgpu_buffer_rsrc_t rin, const RowFast *rf, int t)
{
    constexpr int L = MODE == 1 ? kPieces / 2 : kPieces;                     // lanes per row chunk of a fetch
    const int lane = threadIdx.x;
    // row chunk (64 / L) * i + lane / L: one lane-constant LDS address, the step an immediate
    const uint32_t *mine = &rf[(lane / L) >> 1].in_a + ((lane / L) & 1);
    const uint32_t piece = 16u * (uint32_t)(lane % L);
#pragma unroll
    for (int i = 0; i < Loads1<MODE>::n; i++) {
        const uint32_t voff = mine[(32 / L) * i * (sizeof(RowFast) / 4)] + piece;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, (int)voff, t * (MODE == 1 ? 2 : 4), kLoadNt);
        nxt[i] = make_int4((int)v.x, (int)v.y, (int)v.z, (int)v.w);
    }
}

// mono frames: slot A only, 64 samples = 256 bytes per row, four rows to a store instruction
template <bool F32>
__device__ __forceinline__ void store_fast_mono(const int32_t *tile, const RowFast *rf, __amdgpu_buffer_rsrc_t rout, int t0)
{
    static_assert(kT == 64, "the mono store is written for 64-sample tiles");
    const int lane = threadIdx.x, q = lane & 15, g = lane >> 4;
    uint32_t c16 = 16u * (uint32_t)(q + (kFpw / 4) * g);          // piece q of row 8g + i sits at slot (q + row) mod 16
    asm volatile("" : "+v"(c16));
    const char *tb = (const char *)tile + (kFpw / 4) * g * (kRowWords * 4);
    const RowFast *rfl = rf + (kFpw / 4) * g;
    const uint32_t lane_off = 16u * (uint32_t)q + 4u * (uint32_t)t0;
#pragma unroll
    for (int i = 0; i < kFpw / 4; i++) {
        const uint32_t voff = rfl[i].voff, sh = rfl[i].sh_a;
        const int4 a = *(const int4 *)(tb + i * (kRowWords * 4) + ((c16 + 16u * (uint32_t)i) & 0xf0u));
        u32x4 v = { (uint32_t)a.x << (sh & 31u), (uint32_t)a.y << (sh & 31u), (uint32_t)a.z << (sh & 31u), (uint32_t)a.w << (sh & 31u) };
        if (F32) {
            const double factor = 1.0 / 2147483647.0;                           // stream.d:507
            v.x = __float_as_uint((float)((double)(int32_t)v.x * factor));
            v.y = __float_as_uint((float)((double)(int32_t)v.y * factor));
            v.z = __float_as_uint((float)((double)(int32_t)v.z * factor));
            v.w = __float_as_uint((float)((double)(int32_t)v.w * factor));
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, rout, (int)(voff + lane_off), 0, kStoreNt);
    }
}

template <bool F32>
__device__ __forceinline__ void store_fast(const int32_t *tile, const RowFast *rf, __amdgpu_buffer_rsrc_t rout, int t0)
{
    constexpr int LPR = kT / 2, NI = kFpw / (64 / LPR);                      // lanes per row, store instructions per tile
    const int lane = threadIdx.x, q = lane % LPR, g = lane / LPR;
    // row NI * g + i, piece (q >> 1) + row (mod 16): one lane constant; kept opaque so that the addresses it gives are
    // worked out again in every step (two instructions each) instead of being held in -- spilled -- registers
    uint32_t c16 = 16u * (uint32_t)((q >> 1) + NI * g);
    asm volatile("" : "+v"(c16));
    const char *tb = (const char *)tile + NI * g * (kRowWords * 4) + 8 * (q & 1);
    const RowFast *rfl = rf + NI * g;
    // The tile's position goes into the vector offset, not the scalar one: a 16-byte store with a REGISTER scalar offset
    // is taken by the compiler to have read its data when it issues, and the next instruction may overwrite the data
    // registers; on this part it had not always (int32 rows at full size: the following step's address arithmetic showed
    // up in lanes 12-15 of a row piece, a few thousand samples per launch, only under load).  Without the register the
    // compiler spaces such a write from the store itself.
    const uint32_t lane_off = 16u * (uint32_t)q + 8u * (uint32_t)t0;
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const uint4 f = *(const uint4 *)&rfl[i];
        const uint2 sh = *(const uint2 *)&rfl[i].sh_a;
        const uint32_t s16 = (c16 + 16u * (uint32_t)i) & 0xf0u;
        const char *pa = tb + i * (kRowWords * 4) + s16;
        const int2 a = *(const int2 *)pa;
        // slot B: eight pieces on within the (first) bank row; with 64-sample tiles in the row's second bank row
        const int2 b = *(const int2 *)(tb + i * (kRowWords * 4) + (kT == 64 ? 256 : 0) + (s16 ^ 0x80u));
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const uint32_t av = (uint32_t)(e ? a.y : a.x), bv = (uint32_t)(e ? b.y : b.x);
            const uint32_t x = av + bv, d = av - bv;                            // right/side :2899, left/side :2886
            const uint32_t l0 = (f.z & x) | (~f.z & av);
            const uint32_t r0 = (f.y & d) | (~f.y & bv);
            const uint32_t m = (av << 1) | (bv & 1u);                           // mid/side :2911-2920
            const uint32_t ml = (uint32_t)((int32_t)(m + bv) >> 1), mr = (uint32_t)((int32_t)(m - bv) >> 1);
            o[2 * e] = ((f.w & ml) | (~f.w & l0)) << (sh.x & 31u);
            o[2 * e + 1] = ((f.w & mr) | (~f.w & r0)) << (sh.y & 31u);
        }
        u32x4 v;
        if (F32) {
            const double factor = 1.0 / 2147483647.0;                           // stream.d:507
            v.x = __float_as_uint((float)((double)(int32_t)o[0] * factor));
            v.y = __float_as_uint((float)((double)(int32_t)o[1] * factor));
            v.z = __float_as_uint((float)((double)(int32_t)o[2] * factor));
            v.w = __float_as_uint((float)((double)(int32_t)o[3] * factor));
        } else {
            v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, rout, (int)(f.x + lane_off), 0, kStoreNt);
    }
}

template <int MAXORD, bool WIDE, int MODE>
__device__ __forceinline__ void run_frames1(int32_t *tile, const RowMeta *meta, const RowMeta &me, bool valid,
                                            const afg_flac_subframe *__restrict__ subframes, uint32_t sf_index,
                                            const int32_t *__restrict__ res, int32_t *__restrict__ out_i32,
                                            float *__restrict__ out_f32, int max_bs, int max_pairs, const uint8_t *row_shift,
                                            const RowFast *rf, bool fastw, bool mono, int min_bs, uint64_t in_base, uint64_t out_base)
{
    const int lane = threadIdx.x, row = lane >> 1, slot = lane & 1;
    const int my_ch = valid ? (int)(me.info & 0xff) : 0;
    // the common step's descriptors and the lane's fixed part of every row fetch (MODE 2 -- mixed row widths -- has none)
    // (int32 rows under orders above 12 keep the general step: 64 row registers beside 64 of taps and history spill inside the loop)
    constexpr bool kFastMode = MODE != 2 && !(MODE == 0 && MAXORD > 12);
    // (bases made scalar word by word: a descriptor the compiler takes for divergent is applied in a loop over its values)
    const __amdgpu_buffer_rsrc_t rin = uniform_rsrc((const char *)res + in_base * (MODE == 1 ? 2 : 4));
    const __amdgpu_buffer_rsrc_t rout = uniform_rsrc(out_i32 ? (const void *)(out_i32 + out_base) : (const void *)(out_f32 + out_base));
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
        int t0 = 0;
        // the common step: tiles that are complete, with a complete successor, in every row (its own loop: what the general
        // step keeps in registers for the whole walk does not crowd this one)
        if (kFastMode && fastw) {
            // (one loop per output type: with the choice inside, the compiler no longer counts the stores ahead of the wait)
#define AFG_FLAC_FAST_WALK(F32, STORE)                                                                                        \
            for (; t0 + 2 * kT <= min_bs; t0 += kT) {                                                                  \
                load_fast<MODE>(nxt, rin, rf, t0 + kT);                                                                \
                restore_tile1<MAXORD, WIDE>(tile, row, slot, t0, order, shift, u64, c, h);                             \
                wave_sync();                                                                                           \
                STORE<F32>(tile, rf, rout, t0);                                                                        \
                wave_sync();                                                                                           \
                park_tile1<MODE>(tile, meta, nxt);                                                                     \
                wave_sync();                                                                                           \
            }
            if (mono) {
                if (out_f32) { AFG_FLAC_FAST_WALK(true, store_fast_mono) } else { AFG_FLAC_FAST_WALK(false, store_fast_mono) }
            } else {
                if (out_f32) { AFG_FLAC_FAST_WALK(true, store_fast) } else { AFG_FLAC_FAST_WALK(false, store_fast) }
            }
#undef AFG_FLAC_FAST_WALK
        }
        for (; t0 < max_bs; t0 += kT) {
            if (t0 + kT < max_bs) load_tile1<MODE>(nxt, meta, res, pair, t0 + kT);
            if (t0 < (int)me.bs) restore_tile1<MAXORD, WIDE>(tile, row, slot, t0, order, shift, u64, c, h);
            // (the workgroup is one wavefront: ordering points, not __syncthreads(), which also waits for every load and store in
            // flight.  It is not what makes six channels slow -- 58 ms per C4-sized batch either way: the pair loop is the outer
            // one, so a 128-byte line of interleaved samples gets its 8-byte pieces from three passes a whole frame apart)
            wave_sync();
            // make the prefetched residuals resident here: loads and stores share one in-order counter
#pragma unroll
            for (int i = 0; i < Loads1<MODE>::n; i++)
                asm volatile("" : "+v"(nxt[i].x), "+v"(nxt[i].y), "+v"(nxt[i].z), "+v"(nxt[i].w) : : "memory");
            store_tile1(tile, meta, row_shift, out_i32, out_f32, pair, t0);
            wave_sync();
            if (t0 + kT < max_bs) park_tile1<MODE>(tile, meta, nxt);
            wave_sync();
        }
    }
}

// (a wavefront of one multi-channel shape: flac_restore_mc_kernel, below, takes it)
__device__ __forceinline__ bool mc_wavefront(bool valid, const afg_flac_frame &fr)
{
    const int c0 = __builtin_amdgcn_readfirstlane(valid ? (int)fr.channels : 0);      // lane 0's frame is the block's first
    const int r0 = __builtin_amdgcn_readfirstlane(valid ? (int)(fr.res16 != 0) : 0);
    const bool ok = !valid || ((int)fr.channels == c0 && fr.assignment == AFG_FLAC_INDEPENDENT && (int)(fr.res16 != 0) == r0);
    return c0 > 2 && c0 <= 8 && __all(ok);
}

// One kernel per (order bucket, accumulator width): a wavefront runs only in the instantiation that matches the largest
// LPC order / widest accumulator among the subframes of its 32 frames and leaves the others at once.
template <int LO, int MAXORD, bool WIDE>
__global__ __launch_bounds__(64, 2) void flac_restore1_kernel(
    const afg_flac_frame *__restrict__ frames, const afg_flac_subframe *__restrict__ subframes,
    const int32_t *__restrict__ res, int32_t *__restrict__ out_i32, float *__restrict__ out_f32, uint64_t n_frames)
{
    // 16 KB tile + 768 B + 1 KB + 256 B = 18 KB: eight wavefronts per CU, two per SIMD
    __shared__ __attribute__((aligned(16))) int32_t tile[kFpw * kRowWords];
    __shared__ RowMeta meta[kFpw];
    __shared__ __attribute__((aligned(16))) RowFast rowfast[kFpw];
    __shared__ uint8_t row_shift[kFpw * 8];

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
    if (mc_wavefront(valid, fr)) return;                 // more than two channels throughout: flac_restore_mc_kernel's

    if (valid) {
        me.in_off = fr.in_off;
        me.out_off = fr.out_off;
        me.bs = fr.block_size;
        me.info = (uint32_t)fr.channels | ((uint32_t)fr.assignment << 8) | ((uint32_t)fr.bps << 16) | ((uint32_t)(fr.res16 != 0) << 24);
    }
    // the common step (above) needs every row stereo and one base per side within reach of a 32-bit byte offset
    const uint64_t in_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(me.in_off >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)me.in_off);
    const uint64_t out_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(me.out_off >> 32)) << 32) |
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)me.out_off);
    const int wave_ch = __builtin_amdgcn_readfirstlane(valid ? (int)fr.channels : 0);       // mono or stereo throughout
    const bool mono = wave_ch == 1;
    const bool near = valid && (int)fr.channels == wave_ch && (wave_ch == 1 || wave_ch == 2) && me.in_off >= in_base &&
                      me.in_off - in_base < (1ull << 27) && me.out_off >= out_base && me.out_off - out_base < (1ull << 27);
    const bool fastw = __all(near) && ((out_i32 != nullptr) != (out_f32 != nullptr));
    if ((lane & 1) == 0) {
        uint32_t sh2[2] = { 0, 0 };
        for (int c = 0; c < 8; c++) {
            uint32_t sh = 0;
            if (valid && c < (int)fr.channels) sh = (32u - fr.bps) + subframes[sf_index + c].wasted;   // drflac.d:2883, :2894
            row_shift[(lane >> 1) * 8 + c] = (uint8_t)(sh & 31u);
            if (c < 2) sh2[c] = sh & 31u;
        }
        meta[lane >> 1] = me;
        RowFast rfv;
        const int asg = valid ? (int)fr.assignment : 0;
        rfv.voff = fastw ? (uint32_t)((me.out_off - out_base) * 4) : 0u;
        rfv.m_ls = asg == AFG_FLAC_LEFT_SIDE ? ~0u : 0u;
        rfv.m_rs = asg == AFG_FLAC_RIGHT_SIDE ? ~0u : 0u;
        rfv.m_ms = asg == AFG_FLAC_MID_SIDE ? ~0u : 0u;
        rfv.sh_a = sh2[0]; rfv.sh_b = sh2[1];
        // residual rows: int16 rows padded to 8 samples (AFG_FLAC_ROW16), int32 rows back to back
        const uint32_t esz = fr.res16 ? 2u : 4u;
        const uint32_t chunk = fr.res16 ? ((me.bs + 7u) & ~7u) : me.bs;
        rfv.in_a = fastw ? (uint32_t)(me.in_off - in_base) * esz : 0u;
        rfv.in_b = mono ? kFastReach : rfv.in_a + chunk * esz;         // a mono frame has no second row: past the descriptor's reach, reads 0
        rowfast[lane >> 1] = rfv;
    }
    __syncthreads();

    // (scalar for the compiler too: a loop bound it takes for divergent makes the tile position a vector register)
    const int max_bs = __builtin_amdgcn_readfirstlane(wave_max((int)me.bs));
    const int min_bs = __builtin_amdgcn_readfirstlane(-wave_max(valid ? -(int)me.bs : -0x7fffffff));
    const int max_pairs = __builtin_amdgcn_readfirstlane(wave_max(((int)(me.info & 0xff) + 1) >> 1));
    const bool any16 = __any(valid && (me.info >> 24) != 0), any32 = __any(valid && (me.info >> 24) == 0);
    if (!any16)
        run_frames1<MAXORD, WIDE, 0>(tile, meta, me, valid, subframes, sf_index, res, out_i32, out_f32, max_bs, max_pairs, row_shift, rowfast, fastw, mono, min_bs, in_base, out_base);
    else if (!any32)
        run_frames1<MAXORD, WIDE, 1>(tile, meta, me, valid, subframes, sf_index, res, out_i32, out_f32, max_bs, max_pairs, row_shift, rowfast, fastw, mono, min_bs, in_base, out_base);
    else
        run_frames1<MAXORD, WIDE, 2>(tile, meta, me, valid, subframes, sf_index, res, out_i32, out_f32, max_bs, max_pairs, row_shift, rowfast, fastw, mono, min_bs, in_base, out_base);
}


// ---------------------------------------------------------------------------------------------------------------
// More than two channels (round 6).  The general step above walks a frame once per channel PAIR, so a 128-byte line of
// interleaved samples gets its 8-byte pieces from passes a whole frame apart (six channels: 58 ms per C4-sized batch, 0.14
// of peak).  A wavefront whose frames all have the same channel count C in 3 .. 8 (independent channels: the only
// assignment FLAC has above two) and one row width belongs to flac_restore_mc_kernel instead: lane = subframe for
// floor(64 / C) whole frames at a time, the same 64-sample tile and recurrence, and the tile leaves as whole interleaved
// frames -- 64 C consecutive samples per frame and step, 256 contiguous bytes per store instruction.
// ---------------------------------------------------------------------------------------------------------------
// e / C for e < 4096, C in 3 .. 8: multiply by ceil(2^16 / C) (exact: the error stays below 1 / 16, fractions are at most 7 / 8)
__device__ __forceinline__ int div_small(int e, int inv) { return (int)(((uint32_t)e * (uint32_t)inv) >> 16); }

template <int LO, int MAXORD, bool WIDE>
__global__ __launch_bounds__(64, 2) void flac_restore_mc_kernel(
    const afg_flac_frame *__restrict__ frames, const afg_flac_subframe *__restrict__ subframes,
    const int32_t *__restrict__ res, int32_t *__restrict__ out_i32, float *__restrict__ out_f32, uint64_t n_frames)
{
    __shared__ __attribute__((aligned(16))) int32_t tile[kFpw * kRowWords];      // 64 subframe rows x 64 samples, rows in pairs as above
    __shared__ RowMeta meta[kFpw];
    __shared__ uint8_t row_shift[kFpw * 8];
    const int lane = threadIdx.x;
    const uint64_t f = (uint64_t)blockIdx.x * kFpw + (lane >> 1);
    const bool valid = f < n_frames;
    afg_flac_frame fr;
    fr.channels = 0; fr.assignment = 0; fr.res16 = 0; fr.sf_index = 0; fr.block_size = 0; fr.bps = 0; fr.in_off = 0; fr.out_off = 0;
    int my_order = 0, my_wide = 0;
    if (valid) {
        fr = frames[f];
        for (int c = 0; c < (int)fr.channels && c < 8; c++) {
            const afg_flac_subframe *sf = subframes + fr.sf_index + c;
            my_order = sf->order > my_order ? sf->order : my_order;
            my_wide |= sf->use64;
        }
    }
    const int max_order = wave_max(my_order);
    const bool wide = wave_max(my_wide) != 0;
    if (!(max_order > LO && max_order <= MAXORD && wide == WIDE)) return;
    if (!mc_wavefront(valid, fr)) return;
    const int C = __builtin_amdgcn_readfirstlane((int)fr.channels);
    const bool r16 = __builtin_amdgcn_readfirstlane((int)(fr.res16 != 0)) != 0;
    const int inv = (65536 + C - 1) / C;
    if ((lane & 1) == 0) {
        RowMeta me;
        me.in_off = fr.in_off; me.out_off = fr.out_off; me.bs = valid ? fr.block_size : 0u;
        me.info = (uint32_t)fr.channels | ((uint32_t)fr.sf_index << 8);          // (this kernel's own use of the word: the subframe index)
        meta[lane >> 1] = me;
        for (int c = 0; c < 8; c++) {
            uint32_t sh = 0;
            if (valid && c < (int)fr.channels) sh = (32u - fr.bps) + subframes[fr.sf_index + c].wasted;       // drflac.d:2883, :2894
            row_shift[(lane >> 1) * 8 + c] = (uint8_t)(sh & 31u);
        }
    }
    __syncthreads();
    const uint64_t left = n_frames - (uint64_t)blockIdx.x * kFpw;
    const int nfr = left < (uint64_t)kFpw ? (int)left : kFpw;                      // frames of this block
    const int G = 64 / C;                                                          // frames per pass
    const int g = div_small(lane, inv), ch = lane - g * C;                         // this lane's subframe of a pass
    const double factor = 1.0 / 2147483647.0;                                      // stream.d:507

    for (int g0 = 0; g0 < nfr; g0 += G) {
        const int ng = nfr - g0 < G ? nfr - g0 : G;                                // frames of this pass
        const bool act = lane < ng * C;
        const RowMeta mine = meta[g0 + (act ? g : 0)];
        int32_t c[MAXORD], h[MAXORD];
        int order = 0, shift = 0;
        bool u64 = false;
#pragma unroll
        for (int k = 0; k < MAXORD; k++) { c[k] = 0; h[k] = 0; }
        if (act) {
            const afg_flac_subframe *sf = subframes + (mine.info >> 8) + ch;
            order = sf->order; shift = sf->shift; u64 = sf->use64 != 0;
#pragma unroll
            for (int k = 0; k < MAXORD; k++) c[k] = (k < order) ? (int32_t)sf->coef[k] : 0;
        }
        const int max_bs = __builtin_amdgcn_readfirstlane(wave_max(act ? (int)mine.bs : 0));
        // rows in: row rc = (frame, channel) of the pass; 8 (int16 rows: 8 samples) or 16 (int32 rows: 4 samples) lanes per row piece of
        // a fetch.  int16 rows are fetched a tile ahead into registers (8 per lane) and parked after the stores; int32 rows
        // (16 registers per lane beside up to 64 of taps and history) go straight to the tile.
        constexpr int P16 = kPieces / 2;
        int4 nxt[kLoads / 4];
        auto fetch16 = [&](int t0) {
#pragma unroll
            for (int i = 0; i < kLoads / 4; i++) {
                const int rc = (64 / P16) * i + lane / P16, p = lane % P16;
                const int gg = div_small(rc, inv), cc = rc - gg * C, t = t0 + 8 * p;
                int4 v = make_int4(0, 0, 0, 0);
                if (rc < ng * C) {
                    const RowMeta m = meta[g0 + gg];
                    if (t < (int)m.bs)
                        v = *(const int4 *)((const int16_t *)res + m.in_off + (uint64_t)cc * (((uint64_t)m.bs + 7u) & ~(uint64_t)7u) + (uint64_t)t);
                }
                nxt[i] = v;
            }
        };
        auto park16 = [&]() {
#pragma unroll
            for (int i = 0; i < kLoads / 4; i++) {
                const int rc = (64 / P16) * i + lane / P16, p = lane % P16;
                const int piece = (rc & 1) * kPieces + 2 * p;
                const int4 v = nxt[i];
                *(int4 *)(tile + piece_off(rc >> 1, piece)) = make_int4((int)(int16_t)v.x, v.x >> 16, (int)(int16_t)v.y, v.y >> 16);
                *(int4 *)(tile + piece_off(rc >> 1, piece + 1)) = make_int4((int)(int16_t)v.z, v.z >> 16, (int)(int16_t)v.w, v.w >> 16);
            }
        };
        auto fill32 = [&](int t0) {
#pragma unroll
            for (int i = 0; i < kLoads / 2; i++) {
                const int rc = (64 / kPieces) * i + lane / kPieces, p = lane % kPieces;
                const int gg = div_small(rc, inv), cc = rc - gg * C, t = t0 + 4 * p;
                int4 v = make_int4(0, 0, 0, 0);
                if (rc < ng * C) {
                    const RowMeta m = meta[g0 + gg];
                    if (t < (int)m.bs) {
                        const int32_t *src = res + m.in_off + (uint64_t)cc * m.bs + (uint64_t)t;
                        if (t + 3 < (int)m.bs) {
                            v = *(const int4 *)src;                              // may be 4-byte aligned only (odd block sizes)
                        } else {
                            v.x = src[0];
                            if (t + 1 < (int)m.bs) v.y = src[1];
                            if (t + 2 < (int)m.bs) v.z = src[2];
                        }
                    }
                }
                *(int4 *)(tile + piece_off(rc >> 1, (rc & 1) * kPieces + p)) = v;
            }
        };
        if (r16) { fetch16(0); park16(); }
        for (int t0 = 0; t0 < max_bs; t0 += kT) {
            if (!r16) fill32(t0);
            else if (t0 + kT < max_bs) fetch16(t0 + kT);
            wave_sync();
            restore_tile1<MAXORD, WIDE>(tile, lane >> 1, lane & 1, t0, order, shift, u64, c, h);
            wave_sync();
            if (r16) {                                       // make the prefetched rows resident before the stores enter the queue
#pragma unroll
                for (int i = 0; i < kLoads / 4; i++)
                    asm volatile("" : "+v"(nxt[i].x), "+v"(nxt[i].y), "+v"(nxt[i].z), "+v"(nxt[i].w) : : "memory");
            }
            // frames out: element e of a frame's 64 C samples of this step is sample e / C of channel e % C -- 256 contiguous
            // bytes per store instruction
            for (int gg = 0; gg < ng; gg++) {
                const RowMeta m = meta[g0 + gg];
                const int left_in_frame = (int)m.bs - t0;                         // samples of this frame the step still holds
                int32_t *const oi = out_i32 ? out_i32 + m.out_off + (uint64_t)t0 * C : nullptr;
                float *const of = out_f32 ? out_f32 + m.out_off + (uint64_t)t0 * C : nullptr;
                const uint8_t *const shr = row_shift + (g0 + gg) * 8;
#pragma unroll 2
                for (int k = 0; k < C; k++) {
                    const int e = 64 * k + lane;
                    const int sm = div_small(e, inv), cc = e - sm * C;
                    if (sm < left_in_frame) {
                        const int rc = gg * C + cc;
                        const int32_t v = shl32(tile[piece_off(rc >> 1, (rc & 1) * kPieces + (sm >> 2)) + (sm & 3)], shr[cc]);
                        if (oi) __builtin_nontemporal_store(v, oi + e);
                        if (of) __builtin_nontemporal_store((float)((double)v * factor), of + e);
                    }
                }
            }
            wave_sync();
            if (r16 && t0 + kT < max_bs) park16();
            wave_sync();
        }
    }
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
    const uint32_t all = 0xfffu;                                     // eight instantiations: (order bucket, accumulator width) + four of the multi-channel kernel
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
    // wavefronts of more than two channels throughout: order <= 12 / <= 32 x accumulator width (bits 8 .. 11)
#define AFG_FLAC_LAUNCH_MC(LO, HI, W)                                                                                  \
    if (variants & (1u << idx)) {                                                                                      \
        hipLaunchKernelGGL((flac_restore_mc_kernel<LO, HI, W>), dim3((uint32_t)groups1), dim3(64), 0,                  \
                           (side && (used & 1)) ? side->stream : stream, d_frames, d_subframes, d_res, d_out_i32,      \
                           d_out_f32, n_frames);                                                                       \
        used++;                                                                                                        \
    }                                                                                                                  \
    idx++
    AFG_FLAC_LAUNCH_MC(-1, 12, false); AFG_FLAC_LAUNCH_MC(-1, 12, true);
    AFG_FLAC_LAUNCH_MC(12, 32, false); AFG_FLAC_LAUNCH_MC(12, 32, true);
#undef AFG_FLAC_LAUNCH_MC
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
    if (!frames || !subframes) return 0xfffu;
    for (uint64_t g = 0; g < n_frames; g += kFpw) {
        int order = 0, wide = 0;
        // (mc_wavefront's test: one channel count in 3 .. 8, independent channels, one row width, in every frame of the group)
        const int c0 = (int)frames[g].channels, r0 = frames[g].res16 != 0;
        bool mc = c0 > 2 && c0 <= 8;
        for (uint64_t f = g; f < n_frames && f < g + kFpw; f++) {
            mc = mc && (int)frames[f].channels == c0 && frames[f].assignment == AFG_FLAC_INDEPENDENT && (frames[f].res16 != 0) == r0;
            for (int c = 0; c < (int)frames[f].channels && c < 8; c++) {
                const afg_flac_subframe &sf = subframes[frames[f].sf_index + c];
                if (sf.order > order) order = sf.order;
                wide |= sf.use64 != 0;
            }
        }
        const int bucket = order <= 4 ? 0 : order <= 8 ? 1 : order <= 12 ? 2 : 3;
        if (mc) mask |= 1u << (8 + (order <= 12 ? 0 : 2) + wide);
        else mask |= 1u << (bucket * 2 + wide);
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
    return launch_variants(n_frames, d_frames, d_subframes, d_res, d_out_i32, d_out_f32, 0xfffu, (hipStream_t)hip_stream);
}

extern "C" int afg_flac_transform_variants_hip(uint64_t n_frames, const afg_flac_frame *d_frames,
                                               const afg_flac_subframe *d_subframes, const int32_t *d_res,
                                               int32_t *d_out_i32, float *d_out_f32, uint32_t variants, void *hip_stream)
{
    return launch_variants(n_frames, d_frames, d_subframes, d_res, d_out_i32, d_out_f32, variants, (hipStream_t)hip_stream);
}
