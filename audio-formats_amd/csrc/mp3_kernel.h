// mp3_kernel.h -- device code of the MP3 Layer III transform stage, compiled twice:
//   mp3_transform.hip   AFG_MP3_FMA 0, -ffp-contract=off: every float32 result by the reference's own expression tree
//                       (AFG_NUMERIC_EXACT; bit-identical to the oracle)
//   mp3_tolerance.hip   AFG_MP3_FMA 1, -ffp-contract=fast: multiply-adds fuse and the polyphase window accumulates its
//                       sixteen products per output in one fma chain (AFG_NUMERIC_TOLERANCE: 1e-5 RMS; measured 1.3e-6)
// Same walk, same LDS layout, same carry-state blob in both.  The text below is mp3_transform.hip's of rounds 1-3.
#pragma once
#include "afg_common.h"
#ifndef AFG_MP3_FMA
#error "define AFG_MP3_FMA (0 / 1) and AFG_MP3_KERNEL before including mp3_kernel.h"
#endif

#ifndef AFG_MP3_NT_LOAD
#define AFG_MP3_NT_LOAD 0      // nontemporal spectrum loads (A/B builds)
#endif
#if AFG_MP3_NT_LOAD
typedef float afg_f32x2 __attribute__((ext_vector_type(2)));
#define AFG_MP3_LD(p) ([&] { const afg_f32x2 t_ = __builtin_nontemporal_load((const afg_f32x2 *)(p)); return f2{ t_.x, t_.y }; }())
#else
#define AFG_MP3_LD(p) (*(p))
#endif
#ifndef AFG_MP3_NT_STORE
#define AFG_MP3_NT_STORE 1     // nontemporal PCM stores (0: plain stores -- A/B builds; measured 10.08 -> 9.95 ms on C2)
#endif

namespace {

struct Mp3Seg {
    uint32_t stream;
    uint32_t g0;
    uint32_t count;
    uint32_t last;      // 1 if this segment ends the stream
};

struct Mp3Stream {
    uint64_t blk_base;  // first gr-ch block of the stream
    uint32_t ngr;
    uint32_t nch;
};

constexpr int kStateOverlap = 64 * 9; // floats of overlap in the opaque state blob

// -- tables (values: minimp3.d:1004-1007, :1065-1067, :1113, :1154-1157, :1234-1236, :1336-1352) --
__device__ const float k_aa_cs[8] = {
    0.85749293f, 0.88174200f, 0.94962865f, 0.98331459f, 0.99551782f, 0.99916056f, 0.99989920f, 0.99999316f };
__device__ const float k_aa_ca[8] = {
    0.51449576f, 0.47173197f, 0.31337745f, 0.18191320f, 0.09457419f, 0.04096558f, 0.01419856f, 0.00369997f };

__device__ const float k_win[15 * 16] = {
    -1, 26, -31, 208, 218, 401, -519, 2063, 2000, 4788, -5517, 7134, 5959, 35640, -39336, 74992,
    -1, 24, -35, 202, 222, 347, -581, 2080, 1952, 4425, -5879, 7640, 5288, 33791, -41176, 74856,
    -1, 21, -38, 196, 225, 294, -645, 2087, 1893, 4063, -6237, 8092, 4561, 31947, -43006, 74630,
    -1, 19, -41, 190, 227, 244, -711, 2085, 1822, 3705, -6589, 8492, 3776, 30112, -44821, 74313,
    -1, 17, -45, 183, 228, 197, -779, 2075, 1739, 3351, -6935, 8840, 2935, 28289, -46617, 73908,
    -1, 16, -49, 176, 228, 153, -848, 2057, 1644, 3004, -7271, 9139, 2037, 26482, -48390, 73415,
    -2, 14, -53, 169, 227, 111, -919, 2032, 1535, 2663, -7597, 9389, 1082, 24694, -50137, 72835,
    -2, 13, -58, 161, 224, 72, -991, 2001, 1414, 2330, -7910, 9592, 70, 22929, -51853, 72169,
    -2, 11, -63, 154, 221, 36, -1064, 1962, 1280, 2006, -8209, 9750, -998, 21189, -53534, 71420,
    -2, 10, -68, 147, 215, 2, -1137, 1919, 1131, 1692, -8491, 9863, -2122, 19478, -55178, 70590,
    -3, 9, -73, 139, 208, -29, -1210, 1870, 970, 1388, -8755, 9935, -3300, 17799, -56778, 69679,
    -3, 8, -79, 132, 200, -57, -1283, 1817, 794, 1095, -8998, 9966, -4533, 16155, -58333, 68692,
    -4, 7, -85, 125, 189, -83, -1356, 1759, 605, 814, -9219, 9959, -5818, 14548, -59838, 67629,
    -4, 7, -91, 117, 177, -106, -1428, 1698, 402, 545, -9416, 9916, -7154, 12980, -61289, 66494,
    -5, 6, -97, 111, 163, -127, -1498, 1634, 185, 288, -9585, 9838, -8540, 11455, -62684, 65290 };

// 9-point DCT-III; operation order of minimp3.d:1022-1060.
__device__ __forceinline__ void dct3_9(float (&y)[9])
{
    float e0 = y[0], e2 = y[2], e4 = y[4], e6 = y[6], e8 = y[8];
    float m0 = e0 + e6 * 0.5f;
    e0 = e0 - e6;
    float m4 = (e4 + e2) * 0.93969262f;
    float m2 = (e8 + e2) * 0.76604444f;
    e6 = (e4 - e8) * 0.17364818f;
    e4 = e4 + (e8 - e2);

    e2 = e0 - e4 * 0.5f;
    y[4] = e4 + e0;
    e8 = m0 - m2 + e6;
    e0 = m0 - m4 + m2;
    e4 = m0 + m4 - e6;

    float o1 = y[1], o3 = y[3], o5 = y[5], o7 = y[7];
    o3 = o3 * 0.86602540f;
    m0 = (o5 + o1) * 0.98480775f;
    m4 = (o5 - o7) * 0.34202014f;
    m2 = (o1 + o7) * 0.64278761f;
    o1 = (o1 - o5 - o7) * 0.86602540f;

    o5 = m0 - o3 - m2;
    o7 = m4 - o3 - m0;
    o3 = m4 + o3 - m2;

    y[0] = e4 - o7;
    y[1] = e2 + o1;
    y[2] = e0 - o3;
    y[3] = e8 + o5;
    y[5] = e8 - o5;
    y[6] = e0 + o3;
    y[7] = e2 - o1;
    y[8] = e4 + o7;
}

// Long-block IMDCT of one subband held in registers; minimp3.d:1062-1100.
// `stop` selects window row 1 (minimp3.d:1167).
__device__ __forceinline__ void imdct36_lane(float (&x)[18], float (&ov)[9], bool stop)
{
    constexpr float tw[18] = {
        0.73727734f, 0.79335334f, 0.84339145f, 0.88701083f, 0.92387953f, 0.95371695f, 0.97629601f, 0.99144486f, 0.99904822f,
        0.67559021f, 0.60876143f, 0.53729961f, 0.46174861f, 0.38268343f, 0.30070580f, 0.21643961f, 0.13052619f, 0.04361938f };
    constexpr float w0[18] = {
        0.99904822f, 0.99144486f, 0.97629601f, 0.95371695f, 0.92387953f, 0.88701083f, 0.84339145f, 0.79335334f, 0.73727734f,
        0.04361938f, 0.13052619f, 0.21643961f, 0.30070580f, 0.38268343f, 0.46174861f, 0.53729961f, 0.60876143f, 0.67559021f };
    constexpr float w1[18] = {
        1, 1, 1, 1, 1, 1, 0.99144486f, 0.92387953f, 0.79335334f,
        0, 0, 0, 0, 0, 0, 0.13052619f, 0.38268343f, 0.60876143f };

    float co[9], si[9];
    co[0] = -x[0];
    si[0] = x[17];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        si[8 - 2 * i] = x[4 * i + 1] - x[4 * i + 2];
        co[1 + 2 * i] = x[4 * i + 1] + x[4 * i + 2];
        si[7 - 2 * i] = x[4 * i + 4] - x[4 * i + 3];
        co[2 + 2 * i] = -(x[4 * i + 3] + x[4 * i + 4]);
    }
    dct3_9(co);
    dct3_9(si);
    si[1] = -si[1];
    si[3] = -si[3];
    si[5] = -si[5];
    si[7] = -si[7];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        float wa = stop ? w1[i] : w0[i];
        float wb = stop ? w1[9 + i] : w0[9 + i];
        float ovl = ov[i];
        float sum = co[i] * tw[9 + i] + si[i] * tw[i];
        ov[i] = co[i] * tw[i] - si[i] * tw[9 + i];
        x[i] = ovl * wa - sum * wb;
        x[17 - i] = ovl * wb + sum * wa;
    }
}

// minimp3.d:1102-1109
__device__ __forceinline__ void idct3(float x0, float x1, float x2, float (&dst)[3])
{
    float m1 = x1 * 0.86602540f;
    float a1 = x0 - x2 * 0.5f;
    dst[1] = x0 + x2;
    dst[0] = a1 + m1;
    dst[2] = a1 - m1;
}

// minimp3.d:1111-1129; x = 6 lines at stride 3 starting at `off`, dst = 6 outputs, ov = overlap[6..8]
__device__ __forceinline__ void imdct12(const float (&t)[18], int off, float (&dst)[6], float (&ov)[3])
{
    constexpr float tw3[6] = { 0.79335334f, 0.92387953f, 0.99144486f, 0.60876143f, 0.38268343f, 0.13052619f };
    float co[3], si[3];
    idct3(-t[off + 0], t[off + 6] + t[off + 3], t[off + 12] + t[off + 9], co);
    idct3(t[off + 15], t[off + 12] - t[off + 9], t[off + 6] - t[off + 3], si);
    si[1] = -si[1];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float ovl = ov[i];
        float sum = co[i] * tw3[3 + i] + si[i] * tw3[i];
        ov[i] = co[i] * tw3[i] - si[i] * tw3[3 + i];
        dst[i] = ovl * tw3[2 - i] - sum * tw3[5 - i];
        dst[5 - i] = ovl * tw3[5 - i] + sum * tw3[2 - i];
    }
}

// Short-block subband; minimp3.d:1131-1142.
__device__ __forceinline__ void imdct_short_lane(float (&x)[18], float (&ov)[9])
{
    float t[18];
#pragma unroll
    for (int i = 0; i < 18; i++) t[i] = x[i];
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = ov[i];
    float tail[3] = { ov[6], ov[7], ov[8] };
    float d[6];
    imdct12(t, 0, d, tail);
#pragma unroll
    for (int i = 0; i < 6; i++) x[6 + i] = d[i];
    imdct12(t, 1, d, tail);
#pragma unroll
    for (int i = 0; i < 6; i++) x[12 + i] = d[i];
    imdct12(t, 2, d, tail);
#pragma unroll
    for (int i = 0; i < 6; i++) ov[i] = d[i];
    ov[6] = tail[0];
    ov[7] = tail[1];
    ov[8] = tail[2];
}

// 32-point DCT-II of one time slot; minimp3.d:1232-1298.  in[b] = subband b, out[q].
__device__ __forceinline__ void dct2_32(const float (&in)[32], float (&out)[32])
{
    constexpr float sec[24] = {
        10.19000816f, 0.50060302f, 0.50241929f, 3.40760851f, 0.50547093f, 0.52249861f, 2.05778098f, 0.51544732f,
        0.56694406f, 1.48416460f, 0.53104258f, 0.64682180f, 1.16943991f, 0.55310392f, 0.78815460f, 0.97256821f,
        0.58293498f, 1.06067765f, 0.83934963f, 0.62250412f, 1.72244716f, 0.74453628f, 0.67480832f, 5.10114861f };
    float t[4][8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float x0 = in[i];
        float x1 = in[15 - i];
        float x2 = in[16 + i];
        float x3 = in[31 - i];
        float t0 = x0 + x3;
        float t1 = x1 + x2;
        float t2 = (x1 - x2) * sec[3 * i + 0];
        float t3 = (x0 - x3) * sec[3 * i + 1];
        t[0][i] = t0 + t1;
        t[1][i] = (t0 - t1) * sec[3 * i + 2];
        t[2][i] = t3 + t2;
        t[3][i] = (t3 - t2) * sec[3 * i + 2];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float x0 = t[r][0], x1 = t[r][1], x2 = t[r][2], x3 = t[r][3];
        float x4 = t[r][4], x5 = t[r][5], x6 = t[r][6], x7 = t[r][7], xt;
        xt = x0 - x7; x0 = x0 + x7;
        x7 = x1 - x6; x1 = x1 + x6;
        x6 = x2 - x5; x2 = x2 + x5;
        x5 = x3 - x4; x3 = x3 + x4;
        x4 = x0 - x3; x0 = x0 + x3;
        x3 = x1 - x2; x1 = x1 + x2;
        t[r][0] = x0 + x1;
        t[r][4] = (x0 - x1) * 0.70710677f;
        x5 = x5 + x6;
        x6 = (x6 + x7) * 0.70710677f;
        x7 = x7 + xt;
        x3 = (x3 + x4) * 0.70710677f;
        x5 = x5 - x7 * 0.198912367f;
        x7 = x7 + x5 * 0.382683432f;
        x5 = x5 - x7 * 0.198912367f;
        x0 = xt - x6; xt = xt + x6;
        t[r][1] = (xt + x7) * 0.50979561f;
        t[r][2] = (x4 + x3) * 0.54119611f;
        t[r][3] = (x0 - x5) * 0.60134488f;
        t[r][5] = (x0 + x5) * 0.89997619f;
        t[r][6] = (x4 - x3) * 1.30656302f;
        t[r][7] = (xt - x7) * 2.56291556f;
    }
#pragma unroll
    for (int i = 0; i < 7; i++) {
        out[4 * i + 0] = t[0][i];
        out[4 * i + 1] = t[2][i] + t[3][i] + t[3][i + 1];
        out[4 * i + 2] = t[1][i] + t[1][i + 1];
        out[4 * i + 3] = t[2][i + 1] + t[3][i] + t[3][i + 1];
    }
    out[28] = t[0][7];
    out[29] = t[2][7] + t[3][7];
    out[30] = t[1][7];
    out[31] = t[3][7];
}

// ---------------------------------------------------------------------------------------
// LDS layout of one wavefront (floats).  H holds the polyphase history as 33 rows of
// kHS floats: row r = time slot (r - 15) relative to the current granule, rows 0..14 are
// the 15 slots carried from the previous granule (minimp3.d:1416), rows 15..32 the 18
// slots of this granule.  Inside a row the 32 DCT outputs V[q] of a channel are stored
// as 16 pairs  P[i] = (V[31-i], V[1+i]) (i < 15),  P[15] = (V[16], V[0])  -- the two
// columns every window tap pair needs -- at  c*32 + 2*i.
// Rows 15..32 double as the (channel, subband) -> (channel, slot) transposition buffer;
// rows 0..8 are reused, slot by slot as the window leaves them behind, as the PCM staging
// area of slots 0..8 (slots 9..17 stage in P1).  Program order inside the wavefront keeps
// the uses apart.
constexpr int kHS = 66;                      // row stride: even (8-byte pair reads), 66 mod 32 = 2
constexpr int kHistRows = 15;
constexpr int kRegion = kHistRows * kHS;     // float offset of row 15
constexpr int kLdsFloats = 33 * kHS;
constexpr int kWinStride = 18;               // floats between the rows of the window table in LDS (9 eight-byte slots: odd)

struct alignas(8) f2 { float x, y; };

// The workgroup is ONE wavefront: its LDS accesses execute in program order, so phases only need a
// compiler-level ordering point.  (__syncthreads() would also drain vmcnt, i.e. wait for the
// spectrum loads that were issued a whole granule ahead precisely to stay in flight.)
#define WAVE_SYNC() __builtin_amdgcn_wave_barrier()

// value of the lane below / above (wave_shr:1 / wave_shl:1); lane 0 / 63 get 0
__device__ __forceinline__ float from_lane_below(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_lane_above(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

#ifndef AFG_MP3_MIN_WAVES
#define AFG_MP3_MIN_WAVES 4
#endif

// The walk of one segment; the channel count is a compile-time constant (a stream is mono or stereo throughout), so
// the lane-role predicates and the interleave strides fold.
template <int NCH>
__device__ __forceinline__ void mp3_segment(
    const Mp3Seg &seg, const Mp3Stream &st, const float *__restrict__ coef, const uint32_t *__restrict__ flags,
    float *__restrict__ pcm, float *__restrict__ state, float *const H, float *const Wt)
{
    float *const R = H + kRegion;
    const int lane = threadIdx.x;
    constexpr int nch = NCH;
    constexpr int nval = nch * 576;             // floats per granule

    // role A: lane = (channel, subband): antialias / IMDCT
    const int ch = lane >> 5;
    const int band = lane & 31;
    // role C: lane = (slot parity, channel, column pair i): polyphase window
    const int s2 = lane >> 5;
    const int sc = (lane >> 4) & 1;
    const int si = lane & 15;

    // window taps of this lane's column pair: row 14-i of g_win (minimp3.d:1336-1352, :1388-1395)
    // (rows kWinStride = 18 floats apart: with 16, the eight 8-byte reads of a granule hit four bank groups from fifteen lanes)
    const f2 *const wrow = (const f2 *)(Wt + (14 - (si < 15 ? si : 14)) * kWinStride);

    // ---- carry state ------------------------------------------------------------
    float ov[9];
    const int n_warm = seg.g0 < 2u ? (int)seg.g0 : 2;
    const int g_first = (int)seg.g0 - n_warm;
    const bool from_state = (g_first == 0) && (state != nullptr);
    float *st_blob = state ? state + (size_t)seg.stream * AFG_MP3_STATE_FLOATS : nullptr;
#pragma unroll
    for (int i = 0; i < 9; i++) ov[i] = from_state ? st_blob[lane * 9 + i] : 0.0f;
    for (int r = 0; r < kHistRows; r++)
        H[r * kHS + lane] = from_state ? st_blob[kStateOverlap + r * 64 + lane] : 0.0f;

    // ---- first granule's spectrum: 18 consecutive lines of this lane's subband --------
    const int g_end = (int)(seg.g0 + seg.count);
    const bool ch_on = ch < nch;
    // flag words of 64 granules at a time sit in lane registers (lane i: granule fbase + i, one register per
    // channel) and are read with v_readlane: no memory access on the per-granule path
    int fbase = g_first;
    uint32_t fl_a = 0, fl_b = 0;
    auto refill = [&](int gb) {
        fbase = gb;
        const int gi = gb + lane;
        fl_a = fl_b = 0;
        if (gi < g_end) {
            fl_a = flags[st.blk_base + (uint64_t)gi * nch];
            if (nch == 2) fl_b = flags[st.blk_base + (uint64_t)gi * nch + 1];
        }
        asm volatile("" : "+v"(fl_a), "+v"(fl_b) : : "memory");     // waited for here, never inside the granule loop
    };
    refill(g_first);
    // Subbands a block declares empty (AFG_MP3_NZ_BANDS in flag bits 24..29) are taken as +0.0 and never fetched: a 128 kbit/s stream is silent above ~16 kHz, a quarter of
    // the plane.  Lanes of such subbands, and the idle half of a mono stream, issue no load.
    auto loads_of = [&](int g) -> bool {
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)fl_a, g - fbase);
        const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)fl_b, g - fbase);
        const uint32_t nz = ((ch ? w1 : w0) >> 24) & 63u;        // AFG_MP3_NZ_BANDS: 0 = not declared, else count + 1
        return ch_on && band < (int)(nz ? nz - 1u : 32u);
    };
    f2 pre[9];
    {
        const f2 *src = (const f2 *)(coef + (st.blk_base + (uint64_t)g_first * nch) * 576) + lane * 9;
        const bool on = loads_of(g_first);
#pragma unroll
        for (int q = 0; q < 9; q++) pre[q] = f2{ 0.0f, 0.0f };
        if (on) {                                            // one predicated region for the nine loads, not nine
#pragma unroll
            for (int q = 0; q < 9; q++) pre[q] = AFG_MP3_LD(src + q);
        }
    }
#pragma unroll
    for (int q = 0; q < 9; q++) asm volatile("" : "+v"(pre[q].x), "+v"(pre[q].y) : : "memory");
    WAVE_SYNC();

    for (int g = g_first; g < g_end; g++) {
        const bool do_synth = g >= (int)seg.g0;
        const bool do_dct = g >= (int)seg.g0 - 1;

        // A. this granule's lines; B. issue the next granule's loads (consumed one iteration later)
        float x[18];
#pragma unroll
        for (int m = 0; m < 9; m++) {
            x[2 * m] = pre[m].x;
            x[2 * m + 1] = pre[m].y;
        }
        if (g + 1 - fbase >= 64) refill(g);                // the window covers this granule and the next
        const uint32_t fl0 = (uint32_t)__builtin_amdgcn_readlane((int)fl_a, g - fbase);
        const uint32_t fl1 = (uint32_t)__builtin_amdgcn_readlane((int)fl_b, g - fbase);
        const uint32_t fl = ch_on ? (ch ? fl1 : fl0) : 0u;
        if (g + 1 < g_end) {
            const f2 *src = (const f2 *)(coef + (st.blk_base + (uint64_t)(g + 1) * nch) * 576) + lane * 9;
            const bool on = loads_of(g + 1);
#pragma unroll
            for (int q = 0; q < 9; q++) pre[q] = f2{ 0.0f, 0.0f };
            if (on) {
#pragma unroll
                for (int q = 0; q < 9; q++) pre[q] = AFG_MP3_LD(src + q);
            }
        }

        // C. alias reduction (minimp3.d:1002-1020), IMDCT (:1152-1168), frequency inversion (:1144-1150).  A block of subband
        //    samples (AFG_MP3_SUBBAND: Layer I / II, :1563-1566) skips all three; the flag is the same for every block of a
        //    stream and comes out of v_readlane, so the branch is scalar.
        if (!(fl0 >> 31)) {
            const int block_type = (int)(fl & 3u);
            const int n_long = (int)((fl >> 8) & 0xffu);
            const int aa = (int)((fl >> 16) & 0xffu) - 1;
            const bool lower = (band >= 1) && (band - 1 < aa);   // boundary (band-1 | band)
            const bool upper = (band < aa);                        // boundary (band | band+1)
            float nl[8], nh[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float d_prev = from_lane_below(x[17 - i]);         // line 17-i of subband band-1
                float u_next = from_lane_above(x[i]);              // line i of subband band+1
                nl[i] = x[i] * k_aa_cs[i] - d_prev * k_aa_ca[i];
                nh[i] = u_next * k_aa_ca[i] + x[17 - i] * k_aa_cs[i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                x[i] = lower ? nl[i] : x[i];
                x[17 - i] = upper ? nh[i] : x[17 - i];
            }
            const bool is_short = (block_type == 2) && (band >= n_long);
            if (is_short) {
                imdct_short_lane(x, ov);
            } else {
                const bool stop = (block_type == 3) && (band >= n_long);
                imdct36_lane(x, ov, stop);
            }
            if (band & 1) {
#pragma unroll
                for (int i = 1; i < 18; i += 2) x[i] = -x[i];
            }
        }

        if (do_dct) {
            // D. transposition buffer <- x
#pragma unroll
            for (int m = 0; m < 9; m++) ((f2 *)R)[lane * 9 + m] = f2{ x[2 * m], x[2 * m + 1] };
            WAVE_SYNC();

            // E. 32-point DCT-II, lane = (channel, slot) (minimp3.d:1232-1298); all reads of the
            //    transposition buffer precede the row writes in program order
            float out[32];
            const bool dct_lane = (band < 18) && ch_on;
            {
                float in[32];
                const float *col = R + ch * 576 + (dct_lane ? band : 0);
#pragma unroll
                for (int b = 0; b < 32; b++) in[b] = col[b * 18];
                dct2_32(in, out);
            }
            WAVE_SYNC();
            if (dct_lane) {
                f2 *row = (f2 *)(R + band * kHS + ch * 32);
#pragma unroll
                for (int i = 0; i < 15; i++) row[i] = f2{ out[31 - i], out[1 + i] };
                row[15] = f2{ out[16], out[0] };
            }
            WAVE_SYNC();
        }

        if (do_synth) {
            // G. samples 0 and 16 of every slot (mp3d_synth_pair, :1305-1328), lane = (channel, slot)
            float op0 = 0.0f, op16 = 0.0f;
            const bool pair_lane = (band < 18) && ch_on;
            if (pair_lane) {
                const f2 *base = (const f2 *)(H + band * kHS + ch * 32) + 15;
                f2 z[15];
#pragma unroll
                for (int m = 0; m < 15; m++) z[m] = base[m * (kHS / 2)];     // slot t-15+m: (V[16], V[0])
                float a;
                a  = (z[14].x - z[0].x) * 29;
                a += (z[1].x + z[13].x) * 213;
                a += (z[12].x - z[2].x) * 459;
                a += (z[3].x + z[11].x) * 2037;
                a += (z[10].x - z[4].x) * 5153;
                a += (z[5].x + z[9].x) * 6574;
                a += (z[8].x - z[6].x) * 37489;
                a += z[7].x * 75038;
                op0 = a * (1.0f / 32768.0f);                                  // mp3d_scale_pcm, :1300
                a  = z[14].y * 104;
                a += z[12].y * 1567;
                a += z[10].y * 9727;
                a += z[8].y * 64019;
                a += z[6].y * -9975;
                a += z[4].y * -45;
                a += z[2].y * 146;
                a += z[0].y * -5;
                op16 = a * (1.0f / 32768.0f);
            }

            // F. 512-tap window (minimp3.d:1371-1405), streamed: lane (s2, c, i) walks slots
            //    t' = 9*s2 + n, n = 0..8, over a sliding window of 16 rows.  For slot t' and tap k:
            //        vz = row(15+t'-k) element (k odd ? lo : hi),  vy = row(t'+k) element (k odd ? hi : lo)
            //    i.e. with rw[m] = row(t0 + m):  vz = rw[15+n-k],  vy = rw[n+k]   (24 distinct rows).
            //    PCM of slot n (s2 = 0) lands in H floats [64n, 64n+64) = rows <= n, dead by then.
            f2 w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = wrow[k];                       // (w0, w1) of tap k
            const bool main_lane = (si < 15) && (sc < nch);
            const f2 *base = (const f2 *)(H + s2 * kHS + sc * 32) + (si < 15 ? si : 14);
            // Only one element of each row pair is ever needed by a lane: relative row m (absolute
            // row s2 + 2n + m) feeds tap vy_k (m = k < 8) or vz_k (m = 15 - k), and in both cases the
            // element is hi for odd m, lo for even m.  rw[m] = that element of row(s2 + m).
            const float *rbase = (const float *)base;
            float rw[32];
#pragma unroll
            for (int m = 0; m < 18; m++) rw[m] = rbase[m * kHS + (((m & 1) ^ 0) ? 0 : 1)];
#pragma unroll
            for (int n = 0; n < 9; n++) {                   // slot t = 2n + s2
                if (n + 1 < 9 && n > 0) {                   // rows of the NEXT step: in flight during this one
                    rw[16 + 2 * n] = rbase[(16 + 2 * n) * kHS + 1];
                    rw[17 + 2 * n] = rbase[(17 + 2 * n) * kHS + 0];
                }
                float a = 0.0f, b = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float vz = rw[15 + 2 * n - k];
                    const float vy = rw[2 * n + k];
#if AFG_MP3_FMA
                    // tolerance mode: the sixteen products of an output go into ONE chain of fused multiply-adds (the reference sums
                    // eight two-product terms, minimp3.d:1371-1405): half the instructions of this, the largest, phase
                    b = __builtin_fmaf(vz, w[k].y, b);
                    b = __builtin_fmaf(vy, w[k].x, b);
                    if (k & 1) {
                        a = __builtin_fmaf(vy, w[k].y, a);
                        a = __builtin_fmaf(-vz, w[k].x, a);
                    } else {
                        a = __builtin_fmaf(vz, w[k].x, a);
                        a = __builtin_fmaf(-vy, w[k].y, a);
                    }
#else
                    const float tb = vz * w[k].y + vy * w[k].x;
                    const float ta = (k & 1) ? (vy * w[k].y - vz * w[k].x) : (vz * w[k].x - vy * w[k].y);
                    b = (k == 0) ? tb : (b + tb);
                    a = (k == 0) ? ta : (a + ta);
#endif
                }
                // PCM of slot t goes to floats [64t, 64t+64) (stereo) = rows <= t, which no later step reads
                if (main_lane) {
                    float *pdst = H + ((2 * n + s2) * 32) * nch + sc;
                    pdst[(15 - si) * nch] = a * (1.0f / 32768.0f);
                    pdst[(17 + si) * nch] = b * (1.0f / 32768.0f);
                }
            }
            if (pair_lane) {
                float *pp = H + (band * 32) * nch + ch;
                pp[0] = op0;
                pp[16 * nch] = op16;
            }
            WAVE_SYNC();
            // Make the prefetched spectrum resident *here*: loads and stores share one in-order counter on this
            // hardware, so a wait placed after the stores below would also wait for them to drain.
#pragma unroll
            for (int q = 0; q < 9; q++) asm volatile("" : "+v"(pre[q].x), "+v"(pre[q].y) : : "memory");
            // I. 16-byte coalesced PCM stores
            float4 *dst = (float4 *)(pcm + (st.blk_base + (uint64_t)g * nch) * 576);
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const int idx = lane + 64 * q;
#if AFG_MP3_NT_STORE
                if (4 * idx < nval) {
                    typedef float f32x4nt __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(((const f32x4nt *)H)[idx], (f32x4nt *)dst + idx);
                }
#else
                if (4 * idx < nval) dst[idx] = ((const float4 *)H)[idx];
#endif
            }
        }
        WAVE_SYNC();

        // H. the last 15 slots become the history of the next granule (:1432)
        if (do_dct) {
            constexpr int n2 = kHistRows * kHS / 2;         // 495 float pairs, one contiguous block
            const f2 *srcp = (const f2 *)(H + 18 * kHS);
            f2 hcp[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int idx = lane + 64 * q;
                hcp[q] = (idx < n2) ? srcp[idx] : f2{ 0.0f, 0.0f };
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int idx = lane + 64 * q;
                if (idx < n2) ((f2 *)H)[idx] = hcp[q];
            }
        }
        WAVE_SYNC();
    }

    // ---- hand the carry state on (chunked decoding) ------------------------------
    if (st_blob && seg.last) {
#pragma unroll
        for (int i = 0; i < 9; i++) st_blob[lane * 9 + i] = ov[i];
        for (int r = 0; r < kHistRows; r++) st_blob[kStateOverlap + r * 64 + lane] = H[r * kHS + lane];
    }
}


__global__ __launch_bounds__(64, AFG_MP3_MIN_WAVES) void AFG_MP3_KERNEL(
    const Mp3Seg *__restrict__ segs, const Mp3Stream *__restrict__ streams,
    const float *__restrict__ coef, const uint32_t *__restrict__ flags,
    float *__restrict__ pcm, float *__restrict__ state)
{
    __shared__ __attribute__((aligned(16))) float H[kLdsFloats];
    __shared__ __attribute__((aligned(16))) float Wt[15 * kWinStride];      // g_win, re-read every granule (saves 16 VGPRs)
    const int lane = threadIdx.x;
    for (int i = lane; i < 15 * 16; i += 64) Wt[(i >> 4) * kWinStride + (i & 15)] = k_win[i];
    const Mp3Seg seg = segs[blockIdx.x];
    const Mp3Stream st = streams[seg.stream];
    if (st.nch == 2) mp3_segment<2>(seg, st, coef, flags, pcm, state, H, Wt);
    else mp3_segment<1>(seg, st, coef, flags, pcm, state, H, Wt);
}

}  // namespace

