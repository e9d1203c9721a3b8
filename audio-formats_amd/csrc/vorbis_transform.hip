// vorbis_transform.hip -- Vorbis inverse MDCT + window/overlap-add on gfx950.
//
// Replaces, for whole batches of streams, reference stb_vorbis2.d:2526-2527
// (inverse_mdct per channel, :1941-2242), vorbis_finish_frame (:2606-2657) and
// the interleave of stb_vorbis_get_samples_float_interleaved (:3927-3952).
// Every output sample is produced by the same float32 expression tree as the
// reference (library built with -ffp-contract=off); only the schedule differs:
//
//   * a thread group (one wavefront, or a 256-thread workgroup for long blocks on the general path) walks `seg_packets`
//     consecutive packets of ONE CHANNEL of one stream (the fast path: both channels of a stereo stream) with its working
//     set resident in LDS; a segment that does not start at packet 0 first redoes the IMDCT of the preceding packet to
//     get its right half (the only carried state, stb_vorbis2.d:2641-2643);
//   * every pass of the reference's in-place algorithm is a set of independent
//     butterflies; passes are spread over the group's threads, ordered by the hardware's in-order LDS queue within a
//     wavefront and by a barrier across wavefronts.  Step 3's iter0 / inner_r / inner_s loops (:1720-1864)
//     are one formula: stage l, group i < 2^(l+1), butterfly b < n >> (l+4):
//         p = n/2-1 - (n >> (l+2))*i - 2b,  q = p - (n >> (l+3)),  twiddle A[b << (l+3)]
//   * spectra are read with coalesced row loads, PCM leaves as interleaved
//     rows (a channel's column of them on the general path); twiddle/window tables (host-computed exactly as
//     :851-881) are shared by all workgroups and served from L2 or staged in LDS.
// Kernels: vorbis_wave_kernel (blocksize_1 = 2048, one or two channels: register passes), vorbis_channel_kernel (every
// other stream).  The default numeric mode sends most streams to vorbis_walk.hip instead (afg_vorbis_transform_hip below).
#include "afg_common.h"
#ifndef AFG_VORBIS_NT_LOAD
#define AFG_VORBIS_NT_LOAD 1   // nontemporal spectrum loads (0: plain -- A/B builds)
#endif
#ifndef AFG_VORBIS_NT_STORE
#define AFG_VORBIS_NT_STORE 1  // nontemporal PCM stores in the wave kernel (0: plain stores -- A/B builds)
#endif
#if AFG_VORBIS_NT_STORE
#define AFG_VORBIS_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define AFG_VORBIS_ST(ptr, val) (*(ptr) = (val))
#endif
#include "afg_pk.h"
#include "vorbis_core.h"
#include "vorbis_walk.h"

#include <atomic>
#include <cmath>
#include <map>
#include <vector>

namespace {

#ifndef AFG_VORBIS_CHAN_WIDE
#define AFG_VORBIS_CHAN_WIDE 2048    // long blocks from this size up: a 256-thread workgroup per channel
#endif

using namespace afg_vorbis;

// General path (round 5; round 1's version gave a 256-thread workgroup to a segment and walked its channels one after
// the other with block-wide barriers between the reference's steps: 54-109 ms per C3-sized batch): ONE WAVEFRONT walks one
// channel of a segment -- the channels of a stream meet only in the interleave of the output, a strided store -- with the
// reference's own transform over LDS (vorbis_core.h: inverse_mdct_lds, program-ordered LDS accesses instead of barriers),
// any block sizes, any number of channels.  Per wavefront: the channel buffer (n floats), inverse_mdct's scratch (n/2) and
// previous_window (n/2) of the stream's long block size.  Arithmetic per output is the reference's, operation for operation.
// T threads walk one channel of a segment, CPG channels per workgroup: (64, up to 8) -- wavefronts that never meet, for block
// sizes up to 1024 -- or (256, 1), a workgroup per channel with block-wide barriers between the passes, for the long
// blocks whose passes have work for four wavefronts (one wavefront alone: 4096-sample blocks 75 ms per C3-sized batch,
// 32 KB of LDS each; four: see HISTORY.md 3.2).
template <int T, int CPG>
__global__ __launch_bounds__(T * CPG) void vorbis_channel_kernel(
    const VorbisSeg *__restrict__ segs, uint32_t n_segs, uint32_t chan_floats, const VorbisStream *__restrict__ streams,
    const uint8_t *__restrict__ pflags, const uint64_t *__restrict__ spec_off,
    const uint64_t *__restrict__ out_off, const float *__restrict__ tables,
    const float *__restrict__ spec, float *__restrict__ out)
{
    static_assert(T == 64 || CPG == 1, "block-wide barriers: one channel per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x & (T - 1);
    const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / T));
    const uint32_t sidx = blockIdx.x * CPG + slot;
    if (sidx >= n_segs) return;
    const VorbisSeg seg = segs[sidx];
    const VorbisStream st = streams[seg.stream];
    const int C = (int)st.nch, c = (int)seg.pad;
    const int bs0 = (int)st.bs[0], bs1 = (int)st.bs[1];
    const uint32_t tab0 = st.tab[0], tab1 = st.tab[1];

    float *chan = smem + (size_t)slot * chan_floats;      // bs1       (channel_buffers[c])
    float *buf2 = chan + bs1;                             // bs1 / 2   (temp buffer of inverse_mdct)
    float *prevw = buf2 + bs1 / 2;                        // bs1 / 2   (previous_window[c])

    int previous_length = 0;
    const int p_first = seg.p0 > 0 ? (int)seg.p0 - 1 : 0;
    const int p_end = (int)(seg.p0 + seg.count);

    for (int p = p_first; p < p_end; p++) {
        const uint64_t gp = st.pkt_base + (uint64_t)p;
        const unsigned fl = pflags[gp];
        int n, left, right, right_end;
        window_bounds(bs0, bs1, fl, n, left, right, right_end);
        const int n2 = n >> 1;
        const int which = (fl & AFG_VORBIS_LONG) ? 1 : 0;
        const int ld = 31 - __clz(n);
        const float *Tb = tables + (which ? tab1 : tab0);           // (a runtime index into the struct would put it in scratch memory)
        const float *A = Tb, *B = Tb + n2, *Ct = Tb + n;

        // spectrum -> LDS (coalesced rows)
        const float *src = spec + spec_off[gp] + (size_t)c * (size_t)n2;
        for (int k = tid; k < n2; k += T) chan[k] = src[k];
        pass_sync<T>();

        inverse_mdct_lds<T>(chan, buf2, n, ld, A, B, Ct);           // :2526-2527

        // vorbis_finish_frame, :2606-2657
        const bool emit = (p >= (int)seg.p0) && previous_length > 0;
        if (emit) {
            const int pn = previous_length;
            const float *w = tables + ((pn * 2 == bs1) ? tab1 : tab0) + (pn * 2) + (pn * 2 / 4);   // window of size 2*pn (:2245-2251)
            float *o = out + out_off[gp] + c;
            const int nout = right - left;
            for (int jj = tid; jj < nout; jj += T) {
                float vcur = chan[left + jj];
                if (jj < pn) vcur = vcur * w[jj] + prevw[jj] * w[pn - 1 - jj];   // :2624-2626
                o[(size_t)jj * (size_t)C] = vcur;                                  // :3927-3952
            }
        }
        pass_sync<T>();
        // last half of this data becomes previous window, :2633-2643
        previous_length = right_end - right;
        for (int k = tid; k < previous_length; k += T) prevw[k] = chan[right + k];
        pass_sync<T>();
    }
}

// the general path's launch shape for a stream set whose longest block has n samples
struct ChannelShape { int threads, per_group; };
inline ChannelShape channel_shape(uint32_t nmax) { return nmax >= AFG_VORBIS_CHAN_WIDE ? ChannelShape{ 256, 1 } : nmax > 1024 ? ChannelShape{ 64, 5 } : ChannelShape{ 64, 8 }; }

template <int T, int CPG>
int channel_launch(bool set_attr, uint32_t n_segs, uint32_t nmax, hipStream_t stream, const VorbisSeg *segs, const VorbisStream *streams,
                   const uint8_t *pflags, const uint64_t *spec_off, const uint64_t *out_off, const float *tables, const float *spec,
                   float *out)
{
    const size_t lds = (size_t)CPG * 2 * nmax * sizeof(float);
    if (set_attr) {
        const hipError_t e = hipFuncSetAttribute((const void *)vorbis_channel_kernel<T, CPG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            afg::set_error("hipFuncSetAttribute(%zu) failed: %s", lds, hipGetErrorString(e));
            return AFG_ERR_HIP;
        }
        return AFG_OK;
    }
    hipLaunchKernelGGL((vorbis_channel_kernel<T, CPG>), dim3((n_segs + CPG - 1) / CPG), dim3(T * CPG), lds, stream, segs, n_segs, 2 * nmax,
                       streams, pflags, spec_off, out_off, tables, spec, out);
    return AFG_OK;
}

// set_attr: the dynamic-LDS attribute of the instantiation this stream set uses (plan creation); else the launch
int channel_dispatch(bool set_attr, uint32_t n_segs, uint32_t nmax, hipStream_t stream, const VorbisSeg *segs, const VorbisStream *streams,
                     const uint8_t *pflags, const uint64_t *spec_off, const uint64_t *out_off, const float *tables, const float *spec, float *out)
{
    const ChannelShape sh = channel_shape(nmax);
    if (sh.threads == 256) return channel_launch<256, 1>(set_attr, n_segs, nmax, stream, segs, streams, pflags, spec_off, out_off, tables, spec, out);
    if (sh.per_group == 5) return channel_launch<64, 5>(set_attr, n_segs, nmax, stream, segs, streams, pflags, spec_off, out_off, tables, spec, out);
    return channel_launch<64, 8>(set_attr, n_segs, nmax, stream, segs, streams, pflags, spec_off, out_off, tables, spec, out);
}


// =========================================================================================
// Fast path: one wavefront per (stream, packet segment), blocksize_1 == 2048, <= 2 channels.
//
// The whole transform of a channel lives in 8.7 KB of LDS: the n/4 = 512 complex points
// E[m] = (u[n/2-1-2m], u[n/2-2-2m]) of the reference's step 3 are kept as float2 at index
// m + (m >> 3) (one pad per 8 points: every access pattern below is conflict-free or 2-way),
// buf2 as 512 linear float2, and the n output samples alias both once they are dead.
// The 8 radix-2 stages of :2053-2090 run as three register passes (distances 128/64,
// 32/16/8 and the reference's own fused 4/2/1 pass), steps 7 and 8 are fused (each step-7
// butterfly feeds exactly two step-8 items), the spectrum is consumed straight from HBM
// with 16-byte loads issued one transform ahead, the previous window half lives in
// registers, and stereo PCM leaves as interleaved 8-byte stores.
// Arithmetic per output is the reference's, operation for operation.
// =========================================================================================
constexpr int kNL = 2048;
constexpr int kUFloats = kNL / 2 + kNL / 16;       // 1152: padded complex buffer
constexpr int kWaveLds = kUFloats + kNL / 2;        // + buf2 = 2176 floats >= n

// f2 and the packed-arithmetic helpers: afg_pk.h

__device__ __forceinline__ int pad_e(int m) { return m + (m >> 3); }

// Opaque copy of a (wave-uniform) table pointer.  Without it the compiler hoists the 64-bit
// address of every table access out of the packet loop (two VGPRs each, ~120 in total);
// laundering the base per pass keeps addresses as SGPR base + 32-bit lane offset, live
// only inside the pass.
// Opaque copy of the lane id: address arithmetic derived from it cannot be hoisted out of the
// packet loop (where it would sit in VGPRs for the whole kernel), it is recomputed per pass.
__device__ __forceinline__ int fresh_lane()
{
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    return l;
}

template <typename T>
__device__ __forceinline__ const T *fresh(const T *p)
{
    asm volatile("" : "+v"(p));
    return p;
}

__device__ __forceinline__ void bfly2(f2 &p, f2 &q, f2 c)
{
    const f2 d = p - q;
    p = p + q;
    q = pk_mul_ll_hl(d, c) + pk_mul_xneg(d, c);       // (d.x*c.x - d.y*c.y, d.y*c.x + d.x*c.y)
}

// issue the loads of one channel spectrum (n = 2048): 4 x 16 bytes per lane
__device__ __forceinline__ void load_spectrum(float4 (&x)[4], const float *__restrict__ X)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        typedef float v4f __attribute__((ext_vector_type(4)));
#if AFG_VORBIS_NT_LOAD
        const v4f t = __builtin_nontemporal_load((const v4f *)X + lane + 64 * r);   // read once: keep L1 for the tables
#else
        const v4f t = *((const v4f *)X + lane + 64 * r);
#endif
        x[r] = make_float4(t.x, t.y, t.z, t.w);
    }
}

// inverse_mdct for n = 2048 (stb_vorbis2.d:1941-2242); result in smem[0..2048)
// Twiddles of the passes whose lane -> table index mapping is strided (they would hit a few LDS
// banks only): they do not depend on the packet, so each lane keeps its own in registers.
#ifndef AFG_VORBIS_TW_REGS
#define AFG_VORBIS_TW_REGS 1
#endif
// Step 2 and the first two butterfly stages without the LDS round trip between them: half-iteration `it` of step 2 makes the
// points n8-1-it and n4-1-it; with it = 63 - lane + 64 r these are j + 64 (3 - r) and n4/2 + j + 64 (3 - r) for j = lane, i.e.
// exactly the eight points lane j combines in stages 0 and 1 -- they stay in registers.
#ifndef AFG_VORBIS_FUSE12
#define AFG_VORBIS_FUSE12 1
#endif
// Stages 0-1 and stages 2-4 without the LDS round trip between them (stereo walk): after stages 0, 1 slot s of lane j holds
// point 64 s + j; stages 2..4 want slot k of lane 8 g + jp to hold point 64 g + jp + 8 k -- an 8 x 8 transpose between the slot
// index and lane bits [5:3], done in place with v_permlane32_swap (bit 5), v_permlane16_swap (bit 4) and a row_ror:8 DPP
// move with two selects (bit 3): tools/ubench_lanetr.hip checks the exchange on its own.
#ifndef AFG_VORBIS_FUSE23
#define AFG_VORBIS_FUSE23 AFG_VORBIS_FUSE12
#endif
constexpr int kLaneTw = 14;                          // lane-constant twiddles per lane
constexpr int kTabBase = kNL / 2 + kNL / 2 + kNL / 4 + kNL / 2;      // A B C window of n = 2048 (floats)
constexpr int kTabFloats = kTabBase + (AFG_VORBIS_TW_REGS ? 0 : kLaneTw * 64 * 2);   // + the lane-major twiddle copy
// which A entry lane `lane` needs as its i-th lane-constant twiddle (complex index into A)
__device__ __forceinline__ int lane_twiddle_index(int i, int lane)
{
    constexpr int n4 = kNL / 4;
    const int jp = lane & 7;
#if AFG_VORBIS_FUSE12
    if (i < 4) return n4 - 2 - 2 * ((63 - lane) + 64 * i);      // step 2:      A[n2-4-2o], o = 2 it, it = 63 - lane + 64 r (below)
#else
    if (i < 4) return n4 - 2 - 2 * (lane + 64 * i);             // step 2:      A[n2-4-2o], o = 2 (lane + 64 r)
#endif
    if (i == 4) return 4 * lane;                                // stages 0, 1: A[j << 3]
    if (i == 5) return 4 * (lane + 64);                         //              A[(j+64) << 3]
    if (i == 6) return 8 * lane;                                //              A[j << 4]
    if (i < 11) return 16 * (jp + 8 * (i - 7));                 // stages 2..4: A[(jp+8k) << 5], k < 4
    if (i < 13) return 32 * (jp + 8 * (i - 11));                //              A[(jp+8k) << 6], k < 2
    return 64 * jp;                                             //              A[jp << 7]
}

#if AFG_VORBIS_TW_REGS
struct LaneTwiddles {                                // kept in registers
    f2 v[kLaneTw];
    __device__ __forceinline__ f2 p1(int r) const { return v[r]; }
    __device__ __forceinline__ f2 pa(int i) const { return v[4 + i]; }
    __device__ __forceinline__ f2 pb(int k) const { return v[7 + k]; }
};
__device__ __forceinline__ void load_lane_twiddles(LaneTwiddles &t, const float *A, const float *)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < kLaneTw; i++) t.v[i] = ((const f2 *)A)[lane_twiddle_index(i, lane)];
}
#else
struct LaneTwiddles {                                // lane-major copy in LDS: [i][lane], conflict-free 8-byte reads
    const f2 *t;
    __device__ __forceinline__ f2 p1(int r) const { return t[r * 64]; }
    __device__ __forceinline__ f2 pa(int i) const { return t[(4 + i) * 64]; }
    __device__ __forceinline__ f2 pb(int k) const { return t[(7 + k) * 64]; }
};
__device__ __forceinline__ void load_lane_twiddles(LaneTwiddles &t, const float *, const float *lane_major)
{
    t.t = (const f2 *)lane_major + (threadIdx.x & 63);
}
#endif

__device__ __forceinline__ void lane_swap32(float &x, float &y)      // x of lanes 32..63 <-> y of lanes 0..31
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void lane_swap16(float &x, float &y)      // x of the odd rows of 16 <-> y of the even rows
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void lane_swap8(float &x, float &y, bool hi)   // x of lanes with bit 3 set <-> y of their partners (lane ^ 8)
{
    const float send = hi ? x : y;
    const float got = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(send), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    x = hi ? got : x;
    y = hi ? y : got;
}
// slot s of lane 8 k + jp  <->  slot k of lane 8 s + jp
__device__ __forceinline__ void transpose_slots_hi(f2 (&e)[8], bool hi8)
{
    float x[8], y[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { x[k] = e[k].x; y[k] = e[k].y; }
#pragma unroll
    for (int k = 0; k < 4; k++) { lane_swap32(x[k], x[k + 4]); lane_swap32(y[k], y[k + 4]); }
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (!(k & 2)) { lane_swap16(x[k], x[k + 2]); lane_swap16(y[k], y[k + 2]); }
#pragma unroll
    for (int k = 0; k < 8; k += 2) { lane_swap8(x[k], x[k + 1], hi8); lane_swap8(y[k], y[k + 1], hi8); }
#pragma unroll
    for (int k = 0; k < 8; k++) e[k] = f2{ x[k], y[k] };
}

template <typename AfterStep0>
__device__ __forceinline__ void imdct_2048_wave(const float4 (&xin)[4], float *smem, const LaneTwiddles &tw,
                                                const float *A, const float *B, const float *C, AfterStep0 after_step0)
{
    constexpr int n = kNL, n2 = n / 2, n4 = n / 4, n8 = n / 8;
    int lane = fresh_lane();
    f2 *const U = (f2 *)smem;
    f2 *const V = (f2 *)(smem + kUFloats);
    const f2 *A2p = (const f2 *)A;

    // step 0 (:1972-1994): item q and the mirrored item n8-1-q share one 16-byte load
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int q = lane + 64 * r, qm = n8 - 1 - q;
        const f2 xa = f2{ xin[r].x, xin[r].y }, xb = f2{ xin[r].z, xin[r].w };
        const f2 a0 = A2p[q];                       // A[2q], A[2q+1]
        const f2 a1 = A2p[n8 + qm];                 // A[n4+2q'], A[n4+2q'+1]
        // d.x = x.x*a0.y + x.z*a0.x, d.y = x.x*a0.x - x.z*a0.y
        const f2 d = pk_mul_lh_ll(xa, a0) + pk_mul_ll_lnh(xb, a0);
        V[n4 - 1 - q] = d;                          // buf2[n2-2-2q], [n2-1-2q]
        // g.x = -x.w*a1.y + -x.y*a1.x, g.y = -x.w*a1.x - -x.y*a1.y
        const f2 g = pk_mul_nhh_nhl(xb, a1) + pk_mul_nhl_hh(xa, a1);
        V[q] = g;                                   // buf2[n4-2-2q'] = buf2[2q]
    }
    __builtin_amdgcn_wave_barrier();
    after_step0();                                  // the spectrum registers are free from here on
    lane = fresh_lane();

    // step 2 (:2006-2040): half-iteration `it` makes points n4-1-it and n8-1-it
#if AFG_VORBIS_FUSE12
    f2 s2[2][4];                                    // [half][r']: the points base + 64 r' of stages 0, 1
#endif
#pragma unroll
    for (int r = 0; r < 4; r++) {
#if AFG_VORBIS_FUSE12
        const int it = 63 - lane + 64 * r;
#else
        const int it = lane + 64 * r;
#endif
        const f2 e0 = V[n8 + it];                   // v[n4+o], v[n4+o+1], o = 2 it
        const f2 e1 = V[it];
        const f2 aa = tw.p1(r);                     // A[n2-4-2o], A[n2-3-2o]
        const f2 df = e0 - e1;                      // (v40_20, v41_21)
        const f2 hi = pk_add_swap(e0, e1);          // (d0[1], d0[0])
        // lo.x = v41_21*aa.x - v40_20*aa.y (d1[1]), lo.y = v40_20*aa.x + v41_21*aa.y (d1[0])
        const f2 lo = pk_mul_hl_ll(df, aa) + pk_mul_nlh_hh(df, aa);
#if AFG_VORBIS_FUSE12
        s2[0][3 - r] = hi;
        s2[1][3 - r] = lo;
#else
        U[pad_e(n8 - 1 - it)] = hi;
        U[pad_e(n4 - 1 - it)] = lo;
#endif
    }
#if !AFG_VORBIS_FUSE12
    __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();
#endif

    // stages l = 0, 1 (:2053-2060): point sets {base + j + 64 r}, lane j, both halves
    {
            const int j = lane;
        const f2 w00 = tw.pa(0);                    // A[(j) << 3]
        const f2 w01 = tw.pa(1);                    // A[(j+64) << 3]
        const f2 w1 = tw.pa(2);                     // A[j << 4]
#pragma unroll
        for (int hb = 0; hb < 2; hb++) {
            const int base = hb * (n4 / 2) + j;
            f2 e[4];
#pragma unroll
#if AFG_VORBIS_FUSE12
            for (int r = 0; r < 4; r++) e[r] = s2[hb][r];
#else
            for (int r = 0; r < 4; r++) e[r] = U[pad_e(base + 64 * r)];
#endif
            bfly2(e[0], e[2], w00);
            bfly2(e[1], e[3], w01);
            bfly2(e[0], e[1], w1);
            bfly2(e[2], e[3], w1);
#pragma unroll
            for (int r = 0; r < 4; r++) U[pad_e(base + 64 * r)] = e[r];
        }
    }
    __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();

    // stages l = 2, 3, 4 (:2062-2083): point sets {64 g + j' + 8 e}
    {
            const int g = lane >> 3, jp = lane & 7;
        const int base = 64 * g + jp;
        f2 e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) e[k] = U[pad_e(base + 8 * k)];
#pragma unroll
        for (int k = 0; k < 4; k++) bfly2(e[k], e[k + 4], tw.pb(k));                    // A[b << 5]
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const f2 w = tw.pb(4 + k);                                                    // A[b << 6]
            bfly2(e[k], e[k + 2], w);
            bfly2(e[k + 4], e[k + 6], w);
        }
        {
            const f2 w = tw.pb(6);                                                        // A[b << 7]
#pragma unroll
            for (int k = 0; k < 8; k += 2) bfly2(e[k], e[k + 1], w);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) U[pad_e(base + 8 * k)] = e[k];
    }
    __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();

    // last three stages, the reference's fused loop (:1898-1939) on points 8 it .. 8 it + 7;
    // zz[i] = z[-i]
    {
        const float A2 = A[n >> 3];
        const f2 A2p2 = f2{ A2, A2 };
        const int b9 = 9 * lane;                    // pad_e(8 lane)
        f2 t[8];                                    // t[k] = (zz[2k], zz[2k+1]), zz[i] = z[-i]
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = U[b9 + k];
        {
            const f2 K = t[0] - t[4];               // (k00, k11)
            const f2 L = t[1] - t[5];               // (l00, l11)
            t[0] = t[0] + t[4];
            t[1] = t[1] + t[5];
            t[4] = K;                                                   // zz[8] = k00, zz[9] = k11
            t[5] = pk_add_lh_hnl(L, L) * A2p2;                          // ((l00+l11)*A2, (l11-l00)*A2)
        }
        {
            const f2 L = t[3] - t[7];               // (l00, l11)
            const f2 k6 = pk_add_hnh_nll(t[2], t[6]);                   // (k11, -k00) = (zz5-zz13, zz12-zz4)
            t[2] = t[2] + t[6];
            t[3] = t[3] + t[7];
            t[6] = k6;
            t[7] = pk_add_hnl_lh(L, L) * f2{ A2, -A2 };                 // ((l11-l00)*A2, (l00+l11)*-A2)
        }
#pragma unroll
        for (int h = 0; h < 8; h += 4) {            // iter_54 (:1866-1896) on z and z-8
            const f2 y02 = t[h] + t[h + 2];         // (y0, y1)
            const f2 i01 = t[h] - t[h + 2];         // (i00, i11)
            const f2 y23 = t[h + 1] + t[h + 3];     // (y2, y3)
            const f2 i23 = t[h + 1] - t[h + 3];     // (i22, i33)
            t[h] = y02 + y23;
            t[h + 1] = y02 - y23;
            t[h + 2] = pk_add_lh_hnl(i01, i23);     // (i00 + i33, i11 - i22)
            t[h + 3] = pk_add_lnh_hl(i01, i23);     // (i00 - i33, i11 + i22)
        }
        // steps 4-6 (:2096-2124) folded into the store: entry e of the bit-reverse table takes points
        // 511-2*brev8(e) -> v2[511-e] and 510-2*brev8(e) -> v2[255-e].  Point 8*lane+k is 511-(8J+kk) with
        // J = 63-lane, kk = 7-k, i.e. e = brev2(kk>>1)*64 + brev6(J): a per-k constant minus a per-lane one.
        f2 *const Vr = V - (int)(__brev((unsigned)(63 - lane)) >> 26);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            constexpr int rev2[4] = { 0, 2, 1, 3 };
            const int kk = 7 - k;
            Vr[((kk & 1) ? n8 - 1 : n4 - 1) - 64 * rev2[kk >> 1]] = t[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();

    // step 7 (:2133-2175) fused with step 8 (:2187-2238).  Item s works on pairs v2[s] and
    // v2[n4-1-s]; the results feed step-8 items x = n4-1-s and x = s.  All of buf2 is read
    // before the first output is written (the output aliases it).
    {
        const f2 *const C2 = (const f2 *)C;
        const f2 *const B2 = (const f2 *)B;
        f2 dn[4], en[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int sidx = lane + 64 * r;
            dn[r] = V[sidx];
            en[r] = V[n4 - 1 - sidx];
        }
        __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int sidx = lane + 64 * r;
            const f2 cc = C2[sidx];
            const f2 aa = pk_add_lnl_hh(dn[r], en[r]);                  // (a02, a11)
            const f2 bs = pk_add_ll_hnh(dn[r], en[r]);                  // (b2, b3)
            // b0 = cc.y*a02 + cc.x*a11, b1 = cc.y*a11 - cc.x*a02
            const f2 bq = pk_mul_lh_hh(aa, cc) + pk_mul_hl_nll(aa, cc);
            const f2 dnew = bs + bq;                                    // (b2 + b0, b3 + b1)
            const f2 enew = pk_add_lnl_nhh(bs, bq);                     // (b2 - b0, b1 - b3)
            // step 8 for x = sidx (pair v2[n4-1-x] = enew) and x = n4-1-sidx (pair v2[sidx] = dnew):
            //   pa = e.x*bb.y - e.y*bb.x, pb = -e.x*bb.x - e.y*bb.y
            {
                const int x = sidx;
                const f2 bb = B2[n4 - 1 - x];
                const f2 pp = pk_mul_lh_nll(enew, bb) + pk_mul_nhl_nhh(enew, bb);
                smem[x] = pp.x;
                smem[n2 - 1 - x] = -pp.x;
                smem[n2 + x] = pp.y;
                smem[n - 1 - x] = pp.y;
            }
            {
                const int x = n4 - 1 - sidx;
                const f2 bb = B2[sidx];
                const f2 pp = pk_mul_lh_nll(dnew, bb) + pk_mul_nhl_nhh(dnew, bb);
                smem[x] = pp.x;
                smem[n2 - 1 - x] = -pp.x;
                smem[n2 + x] = pp.y;
                smem[n - 1 - x] = pp.y;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();
}

#ifndef AFG_VORBIS_MIN_WAVES
#define AFG_VORBIS_MIN_WAVES 2
#endif
// Make the prefetched spectrum resident *here*: the wait this forces only covers loads that were
// issued a whole transform ago.  (Loads and stores share one in-order counter on gfx9-class hardware;
// waiting for a load that was issued after a batch of PCM stores would wait for those stores too.)
__device__ __forceinline__ void settle(float4 (&x)[4])
{
    asm volatile("" : "+v"(x[0].x), "+v"(x[0].y), "+v"(x[0].z), "+v"(x[0].w), "+v"(x[1].x), "+v"(x[1].y), "+v"(x[1].z),
                 "+v"(x[1].w), "+v"(x[2].x), "+v"(x[2].y), "+v"(x[2].z), "+v"(x[2].w), "+v"(x[3].x), "+v"(x[3].y),
                 "+v"(x[3].z), "+v"(x[3].w) : : "memory");
}

// One wavefront walks ONE channel of a segment (the channels of a stream are independent up to the
// interleave of the output, which is a strided store): half the registers of a two-channel walk.
__device__ __forceinline__ void vorbis_wave_body(
    float *smem, const float *ltab, const VorbisSeg &seg, const VorbisStream &st,
    const uint8_t *__restrict__ pflags, const uint64_t *__restrict__ spec_off,
    const uint64_t *__restrict__ out_off, const float *tables,
    const float *__restrict__ spec, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int bs0 = (int)st.bs[0], bs1 = (int)st.bs[1];
    const uint32_t tab0 = st.tab[0], tab1 = st.tab[1];
    const int C = (int)st.nch, c = (int)seg.pad;      // channels of the stream, channel of this wavefront

    LaneTwiddles tw;
    load_lane_twiddles(tw, ltab, ltab + kTabBase);
    int previous_length = 0;
    const int p_first = seg.p0 > 0 ? (int)seg.p0 - 1 : 0;
    const int p_end = (int)(seg.p0 + seg.count);
    const float *const lwin = ltab + kNL + kNL / 4;   // window of n = 2048 (LDS)

    // previous_window (:2641-2643) lives in registers: sample lane + 64 i of channel c in pv[i]
    // (its length is 64, 128 or 1024 on this path: right_end - right of a 256- or 2048-sample block)
    float pv[16];
#pragma unroll
    for (int i = 0; i < 16; i++) pv[i] = 0.0f;

    // flags of packets [fbase, fbase + 64), one per lane: the packet loop reads them with a lane read
    // instead of a (vector-memory) byte load per packet
    int fbase = 0;
    unsigned fl_reg = 0;
    uint64_t so_reg = 0, oo_reg = 0;                  // spectrum / output offsets of the same packets
    auto refill = [&](int from) {
        fbase = from;
        const int q = from + lane;
        const bool in = q < p_end;
        fl_reg = in ? (unsigned)pflags[st.pkt_base + (uint64_t)q] : 0u;
        so_reg = in ? spec_off[st.pkt_base + (uint64_t)q] : 0;
        oo_reg = in ? out_off[st.pkt_base + (uint64_t)q] : 0;
        // waited for here, inside the rarely taken refill: a wait at the first readlane would sit on the
        // per-packet path and drain the previous packet's PCM stores every time
        uint32_t s0 = (uint32_t)so_reg, s1 = (uint32_t)(so_reg >> 32), o0 = (uint32_t)oo_reg, o1 = (uint32_t)(oo_reg >> 32);
        asm volatile("" : "+v"(fl_reg), "+v"(s0), "+v"(s1), "+v"(o0), "+v"(o1) : : "memory");
        so_reg = ((uint64_t)s1 << 32) | s0;
        oo_reg = ((uint64_t)o1 << 32) | o0;
    };
    auto flags_of = [&](int p) -> unsigned { return (unsigned)__builtin_amdgcn_readlane((int)fl_reg, p - fbase); };
    auto lane64 = [&](uint64_t v, int p) -> uint64_t {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, p - fbase);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), p - fbase);
        return ((uint64_t)hi << 32) | lo;
    };
    refill(p_first);

    // One spectrum is in flight: step 0 of transform k empties xin, the loads of transform k+1 follow at once
    // and are waited for (settle) just before the PCM stores of transform k enter the queue.
    float4 xin[4];
    auto issue = [&](int p) {
        if (p < p_end && (flags_of(p) & AFG_VORBIS_LONG))
            load_spectrum(xin, spec + lane64(so_reg, p) + c * (kNL / 2));
    };
    issue(p_first);
    settle(xin);                                     // waited for here: a wait at the loop top would be executed by every
                                                     // iteration and drain the previous packet's PCM stores each time

    for (int p = p_first; p < p_end; p++) {
        if (p + 1 - fbase >= 64) refill(p);
        const unsigned fl = flags_of(p);
        int n, left, right, right_end;
        window_bounds(bs0, bs1, fl, n, left, right, right_end);
        const int n2 = n >> 1;
        const int which = (fl & AFG_VORBIS_LONG) ? 1 : 0;
        const float *T = tables + (which ? tab1 : tab0);            // (a runtime index into the struct would put it in scratch memory)
        const float *A = T, *B = T + n2, *Ct = T + n;
        const float *src = spec + lane64(so_reg, p);
        const bool emit = (p >= (int)seg.p0) && previous_length > 0;
        const int pn = previous_length;
        const int nout = right - left, plen = right_end - right;
        float *o = out + lane64(oo_reg, p);

        {
            auto next = [&]() { issue(p + 1); };
            if (which) {
                imdct_2048_wave(xin, smem, tw, ltab, ltab + kNL / 2, ltab + kNL, next);   // :2526-2527, tables in LDS
            } else {
                for (int k = lane; k < n2; k += 64) smem[k] = src[c * n2 + k];
                __builtin_amdgcn_wave_barrier();
                inverse_mdct_lds<64>(smem, smem + n, n, 31 - __clz(n), A, B, Ct);
                next();
            }
            settle(xin);
            // vorbis_finish_frame (:2606-2657) + interleave (:3927-3952) for this channel: each
            // channel stores its own 4-byte column of the interleaved frames (merged in L2)
            if (emit) {
                const int nwin = pn < nout ? pn : nout;
                if (pn * 2 == kNL && nout == kNL / 2) {            // long after long: all 1024 outputs are windowed, no conditions
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int jj = lane + 64 * i;
                        o[jj * C + c] = smem[left + jj] * lwin[jj] + pv[i] * lwin[1023 - jj];                  // :2624-2626
                    }
                } else if (pn * 2 == kNL) {                        // get_window(pn), :2245-2251: the long window (LDS)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int jj = lane + 64 * i;
                        if (jj < nwin) o[jj * C + c] = smem[left + jj] * lwin[jj] + pv[i] * lwin[1023 - jj];   // :2624-2626
                    }
                } else {
                    const float *wt = tables + ((pn * 2 == bs1) ? tab1 : tab0) + (pn * 2) + (pn * 2 / 4);
#pragma unroll
                    for (int i = 0; i < 8; i++) {                  // pn is 64 .. 512 here (blocksize_0 <= 1024)
                        const int jj = lane + 64 * i;
                        if (jj < nwin) o[jj * C + c] = smem[left + jj] * wt[jj] + pv[i] * wt[pn - 1 - jj];
                    }
                }
                for (int jj = nwin + lane; jj < nout; jj += 64) o[jj * C + c] = smem[left + jj];
            }
            if (plen == kNL / 2) {                                                     // :2641-2643; the long-long case:
#pragma unroll
                for (int i = 0; i < 16; i++) pv[i] = smem[right + lane + 64 * i];      // sixteen reads, no conditions
            } else {                                                                   // 64 .. 512 samples
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (64 * i < plen) pv[i] = smem[right + lane + 64 * i];
            }
            __builtin_amdgcn_wave_barrier();
        }
        previous_length = plen;
    }
}

// =========================================================================================
// Two channels per wavefront (stereo streams): the transforms of the left and the right channel of a packet are
// independent, so every pass issues the LDS reads of both, computes both, writes both.  The second channel's reads
// are in flight while the first is being computed -- the single-channel walk spent half its life in s_waitcnt
// (profiles/r02_pmc_vorbis_wave_kernel.json) -- the tables are read once for both, and the PCM leaves as
// interleaved (L, R) 8-byte stores, half as many instructions for the same bytes.
// =========================================================================================
constexpr int kDualStride = kWaveLds;               // channel 1's transform area follows channel 0's

template <typename AfterStep0>
__device__ __forceinline__ void imdct_2048_wave2(const float4 (&xin)[2][4], float *smem, const LaneTwiddles &tw,
                                                 const float *A, const float *B, const float *C, AfterStep0 after_step0)
{
    constexpr int n = kNL, n2 = n / 2, n4 = n / 4, n8 = n / 8;
    int lane = fresh_lane();
    f2 *const U0 = (f2 *)smem;
    f2 *const V0 = (f2 *)(smem + kUFloats);
    constexpr int CS = kDualStride / 2;             // channel stride in f2
    const f2 *A2p = (const f2 *)A;

    // step 0 (:1972-1994)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int q = lane + 64 * r, qm = n8 - 1 - q;
        const f2 a0 = A2p[q];
        const f2 a1 = A2p[n8 + qm];
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            const f2 xa = f2{ xin[ch][r].x, xin[ch][r].y }, xb = f2{ xin[ch][r].z, xin[ch][r].w };
            const f2 d = pk_mul_lh_ll(xa, a0) + pk_mul_ll_lnh(xb, a0);
            V0[ch * CS + n4 - 1 - q] = d;
            const f2 g = pk_mul_nhh_nhl(xb, a1) + pk_mul_nhl_hh(xa, a1);
            V0[ch * CS + q] = g;
        }
    }
    __builtin_amdgcn_wave_barrier();
    after_step0();                                  // the spectrum registers are free from here on
    lane = fresh_lane();

    // step 2 (:2006-2040)
    {
        f2 e0[2][4], e1[2][4];
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
#if AFG_VORBIS_FUSE12
                const int it = 63 - lane + 64 * r;
#else
                const int it = lane + 64 * r;
#endif
                e0[ch][r] = V0[ch * CS + n8 + it];
                e1[ch][r] = V0[ch * CS + it];
            }
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const f2 aa = tw.p1(r);
                const f2 df = e0[ch][r] - e1[ch][r];
                const f2 hi = pk_add_swap(e0[ch][r], e1[ch][r]);
                const f2 lo = pk_mul_hl_ll(df, aa) + pk_mul_nlh_hh(df, aa);
#if AFG_VORBIS_FUSE12
                e0[ch][r] = hi;                                 // point j + 64 (3 - r) of the lower half (j = lane)
                e1[ch][r] = lo;                                 // the same point of the upper half
#else
                const int it = lane + 64 * r;
                U0[ch * CS + pad_e(n8 - 1 - it)] = hi;
                U0[ch * CS + pad_e(n4 - 1 - it)] = lo;
#endif
            }
#if !AFG_VORBIS_FUSE12
    }
    __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();

    // stages l = 0, 1 (:2053-2060)
    {
#else
        // stages l = 0, 1 (:2053-2060) on the registers step 2 left
#endif
        const int j = lane;
        const f2 w00 = tw.pa(0), w01 = tw.pa(1), w1 = tw.pa(2);
#if AFG_VORBIS_FUSE23
        f2 e[2][8];                                             // slot 4 hb + r: point hb n4/2 + j + 64 r
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int r = 0; r < 4; r++) { e[ch][r] = e0[ch][3 - r]; e[ch][4 + r] = e1[ch][3 - r]; }
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int hb = 0; hb < 8; hb += 4) {
                bfly2(e[ch][hb + 0], e[ch][hb + 2], w00);
                bfly2(e[ch][hb + 1], e[ch][hb + 3], w01);
                bfly2(e[ch][hb + 0], e[ch][hb + 1], w1);
                bfly2(e[ch][hb + 2], e[ch][hb + 3], w1);
            }
        // stages l = 2, 3, 4 (:2062-2083) want slot k of lane 8 g + jp to hold point 64 g + jp + 8 k
        transpose_slots_hi(e[0], (j & 8) != 0);
        transpose_slots_hi(e[1], (j & 8) != 0);
        const int g = lane >> 3, jp = lane & 7;
        const int base = 64 * g + jp;
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
#else
#pragma unroll
        for (int hb = 0; hb < 2; hb++) {
            const int base = hb * (n4 / 2) + j;
            f2 e[2][4];
#pragma unroll
            for (int ch = 0; ch < 2; ch++)
#pragma unroll
#if AFG_VORBIS_FUSE12
                for (int r = 0; r < 4; r++) e[ch][r] = hb ? e1[ch][3 - r] : e0[ch][3 - r];
#else
                for (int r = 0; r < 4; r++) e[ch][r] = U0[ch * CS + pad_e(base + 64 * r)];
#endif
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {
                bfly2(e[ch][0], e[ch][2], w00);
                bfly2(e[ch][1], e[ch][3], w01);
                bfly2(e[ch][0], e[ch][1], w1);
                bfly2(e[ch][2], e[ch][3], w1);
#pragma unroll
                for (int r = 0; r < 4; r++) U0[ch * CS + pad_e(base + 64 * r)] = e[ch][r];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();

    // stages l = 2, 3, 4 (:2062-2083)
    {
        const int g = lane >> 3, jp = lane & 7;
        const int base = 64 * g + jp;
        f2 e[2][8];
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int k = 0; k < 8; k++) e[ch][k] = U0[ch * CS + pad_e(base + 8 * k)];
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
#endif
#pragma unroll
            for (int k = 0; k < 4; k++) bfly2(e[ch][k], e[ch][k + 4], tw.pb(k));
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const f2 w = tw.pb(4 + k);
                bfly2(e[ch][k], e[ch][k + 2], w);
                bfly2(e[ch][k + 4], e[ch][k + 6], w);
            }
            {
                const f2 w = tw.pb(6);
#pragma unroll
                for (int k = 0; k < 8; k += 2) bfly2(e[ch][k], e[ch][k + 1], w);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) U0[ch * CS + pad_e(base + 8 * k)] = e[ch][k];
        }
    }
    __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();

    // last three stages (:1898-1939) + steps 4-6 folded into the store (:2096-2124)
    {
        const float A2 = A[n >> 3];
        const f2 A2p2 = f2{ A2, A2 };
        const int b9 = 9 * lane;
        f2 t[2][8];
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int k = 0; k < 8; k++) t[ch][k] = U0[ch * CS + b9 + k];
        f2 *const Vr = V0 - (int)(__brev((unsigned)(63 - lane)) >> 26);
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            f2 (&tt)[8] = t[ch];
            {
                const f2 K = tt[0] - tt[4];
                const f2 L = tt[1] - tt[5];
                tt[0] = tt[0] + tt[4];
                tt[1] = tt[1] + tt[5];
                tt[4] = K;
                tt[5] = pk_add_lh_hnl(L, L) * A2p2;
            }
            {
                const f2 L = tt[3] - tt[7];
                const f2 k6 = pk_add_hnh_nll(tt[2], tt[6]);
                tt[2] = tt[2] + tt[6];
                tt[3] = tt[3] + tt[7];
                tt[6] = k6;
                tt[7] = pk_add_hnl_lh(L, L) * f2{ A2, -A2 };
            }
#pragma unroll
            for (int h = 0; h < 8; h += 4) {
                const f2 y02 = tt[h] + tt[h + 2];
                const f2 i01 = tt[h] - tt[h + 2];
                const f2 y23 = tt[h + 1] + tt[h + 3];
                const f2 i23 = tt[h + 1] - tt[h + 3];
                tt[h] = y02 + y23;
                tt[h + 1] = y02 - y23;
                tt[h + 2] = pk_add_lh_hnl(i01, i23);
                tt[h + 3] = pk_add_lnh_hl(i01, i23);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                constexpr int rev2[4] = { 0, 2, 1, 3 };
                const int kk = 7 - k;
                Vr[ch * CS + ((kk & 1) ? n8 - 1 : n4 - 1) - 64 * rev2[kk >> 1]] = tt[k];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();

    // step 7 (:2133-2175) fused with step 8 (:2187-2238); all of buf2 is read before the first output is written
    {
        const f2 *const C2 = (const f2 *)C;
        const f2 *const B2 = (const f2 *)B;
        f2 dn[2][4], en[2][4];
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int sidx = lane + 64 * r;
                dn[ch][r] = V0[ch * CS + sidx];
                en[ch][r] = V0[ch * CS + n4 - 1 - sidx];
            }
        __builtin_amdgcn_wave_barrier();
        lane = fresh_lane();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int sidx = lane + 64 * r;
            const f2 cc = C2[sidx];
            const f2 bb0 = B2[n4 - 1 - sidx], bb1 = B2[sidx];
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {
                float *const sm = smem + ch * kDualStride;
                const f2 aa = pk_add_lnl_hh(dn[ch][r], en[ch][r]);
                const f2 bs = pk_add_ll_hnh(dn[ch][r], en[ch][r]);
                const f2 bq = pk_mul_lh_hh(aa, cc) + pk_mul_hl_nll(aa, cc);
                const f2 dnew = bs + bq;
                const f2 enew = pk_add_lnl_nhh(bs, bq);
                {
                    const int x = sidx;
                    const f2 pp = pk_mul_lh_nll(enew, bb0) + pk_mul_nhl_nhh(enew, bb0);
                    sm[x] = pp.x;
                    sm[n2 - 1 - x] = -pp.x;
                    sm[n2 + x] = pp.y;
                    sm[n - 1 - x] = pp.y;
                }
                {
                    const int x = n4 - 1 - sidx;
                    const f2 pp = pk_mul_lh_nll(dnew, bb1) + pk_mul_nhl_nhh(dnew, bb1);
                    sm[x] = pp.x;
                    sm[n2 - 1 - x] = -pp.x;
                    sm[n2 + x] = pp.y;
                    sm[n - 1 - x] = pp.y;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    lane = fresh_lane();
}

__device__ __forceinline__ void settle2(float4 (&x)[2][4])
{
    settle(x[0]);
    settle(x[1]);
}

// One wavefront walks BOTH channels of a stereo segment.
__device__ __forceinline__ void vorbis_wave2_body(
    float *smem, const float *ltab, const VorbisSeg &seg, const VorbisStream &st,
    const uint8_t *__restrict__ pflags, const uint64_t *__restrict__ spec_off,
    const uint64_t *__restrict__ out_off, const float *tables,
    const float *__restrict__ spec, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int bs0 = (int)st.bs[0], bs1 = (int)st.bs[1];
    const uint32_t tab0 = st.tab[0], tab1 = st.tab[1];

    LaneTwiddles tw;
    load_lane_twiddles(tw, ltab, ltab + kTabBase);
    int previous_length = 0;
    const int p_first = seg.p0 > 0 ? (int)seg.p0 - 1 : 0;
    const int p_end = (int)(seg.p0 + seg.count);
    const float *const lwin = ltab + kNL + kNL / 4;   // window of n = 2048 (LDS)

    float pv[2][16];                                   // previous_window (:2641-2643) of both channels, in registers
#pragma unroll
    for (int i = 0; i < 16; i++) pv[0][i] = pv[1][i] = 0.0f;

    int fbase = 0;
    unsigned fl_reg = 0;
    uint64_t so_reg = 0, oo_reg = 0;
    auto refill = [&](int from) {
        fbase = from;
        const int q = from + lane;
        const bool in = q < p_end;
        fl_reg = in ? (unsigned)pflags[st.pkt_base + (uint64_t)q] : 0u;
        so_reg = in ? spec_off[st.pkt_base + (uint64_t)q] : 0;
        oo_reg = in ? out_off[st.pkt_base + (uint64_t)q] : 0;
        uint32_t s0 = (uint32_t)so_reg, s1 = (uint32_t)(so_reg >> 32), o0 = (uint32_t)oo_reg, o1 = (uint32_t)(oo_reg >> 32);
        asm volatile("" : "+v"(fl_reg), "+v"(s0), "+v"(s1), "+v"(o0), "+v"(o1) : : "memory");
        so_reg = ((uint64_t)s1 << 32) | s0;
        oo_reg = ((uint64_t)o1 << 32) | o0;
    };
    auto flags_of = [&](int p) -> unsigned { return (unsigned)__builtin_amdgcn_readlane((int)fl_reg, p - fbase); };
    auto lane64 = [&](uint64_t v, int p) -> uint64_t {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, p - fbase);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), p - fbase);
        return ((uint64_t)hi << 32) | lo;
    };
    refill(p_first);

    float4 xin[2][4];
    auto issue = [&](int p) {
        if (p < p_end && (flags_of(p) & AFG_VORBIS_LONG)) {
            const float *src = spec + lane64(so_reg, p);
            load_spectrum(xin[0], src);
            load_spectrum(xin[1], src + kNL / 2);
        }
    };
    issue(p_first);
    settle2(xin);

    for (int p = p_first; p < p_end; p++) {
        if (p + 1 - fbase >= 64) refill(p);
        const unsigned fl = flags_of(p);
        int n, left, right, right_end;
        window_bounds(bs0, bs1, fl, n, left, right, right_end);
        const int n2 = n >> 1;
        const int which = (fl & AFG_VORBIS_LONG) ? 1 : 0;
        const float *T = tables + (which ? tab1 : tab0);            // (a runtime index into the struct would put it in scratch memory)
        const float *A = T, *B = T + n2, *Ct = T + n;
        const float *src = spec + lane64(so_reg, p);
        const bool emit = (p >= (int)seg.p0) && previous_length > 0;
        const int pn = previous_length;
        const int nout = right - left, plen = right_end - right;
        f2 *o = (f2 *)(out + lane64(oo_reg, p));           // interleaved frames: one (L, R) pair per frame, 8-byte aligned
        float *const sm0 = smem, *const sm1 = smem + kDualStride;

        auto next = [&]() { issue(p + 1); };
        if (which) {
            imdct_2048_wave2(xin, smem, tw, ltab, ltab + kNL / 2, ltab + kNL, next);   // :2526-2527, tables in LDS
        } else {
            for (int ch = 0; ch < 2; ch++) {
                float *sm = smem + ch * kDualStride;
                for (int k = lane; k < n2; k += 64) sm[k] = src[ch * n2 + k];
                __builtin_amdgcn_wave_barrier();
                inverse_mdct_lds<64>(sm, sm + n, n, 31 - __clz(n), A, B, Ct);
            }
            next();
        }
        settle2(xin);
        // vorbis_finish_frame (:2606-2657) + interleave (:3927-3952)
        if (emit) {
            const int nwin = pn < nout ? pn : nout;
            if (pn * 2 == kNL && nout == kNL / 2) {            // long after long: all 1024 frames are windowed
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int jj = lane + 64 * i;
                    const float w0 = lwin[jj], w1 = lwin[1023 - jj];
                    AFG_VORBIS_ST(o + jj, (f2{ sm0[left + jj] * w0 + pv[0][i] * w1, sm1[left + jj] * w0 + pv[1][i] * w1 }));   // :2624-2626
                }
            } else if (pn * 2 == kNL) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int jj = lane + 64 * i;
                    if (jj < nwin) {
                        const float w0 = lwin[jj], w1 = lwin[1023 - jj];
                        AFG_VORBIS_ST(o + jj, (f2{ sm0[left + jj] * w0 + pv[0][i] * w1, sm1[left + jj] * w0 + pv[1][i] * w1 }));
                    }
                }
            } else {
                const float *wt = tables + ((pn * 2 == bs1) ? tab1 : tab0) + (pn * 2) + (pn * 2 / 4);
#pragma unroll
                for (int i = 0; i < 8; i++) {                  // pn is 64 .. 512 here (blocksize_0 <= 1024)
                    const int jj = lane + 64 * i;
                    if (jj < nwin) {
                        const float w0 = wt[jj], w1 = wt[pn - 1 - jj];
                        AFG_VORBIS_ST(o + jj, (f2{ sm0[left + jj] * w0 + pv[0][i] * w1, sm1[left + jj] * w0 + pv[1][i] * w1 }));
                    }
                }
            }
            for (int jj = nwin + lane; jj < nout; jj += 64) AFG_VORBIS_ST(o + jj, (f2{ sm0[left + jj], sm1[left + jj] }));
        }
        if (plen == kNL / 2) {                                 // :2641-2643
#pragma unroll
            for (int i = 0; i < 16; i++) {
                pv[0][i] = sm0[right + lane + 64 * i];
                pv[1][i] = sm1[right + lane + 64 * i];
            }
        } else {                                               // 64 .. 512 samples
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (64 * i < plen) {
                    pv[0][i] = sm0[right + lane + 64 * i];
                    pv[1][i] = sm1[right + lane + 64 * i];
                }
        }
        __builtin_amdgcn_wave_barrier();
        previous_length = plen;
    }
}

#ifndef AFG_VORBIS_GROUP_WAVES
#define AFG_VORBIS_GROUP_WAVES 8
#endif
constexpr int kWavesPerGroup = AFG_VORBIS_GROUP_WAVES;
constexpr int kWaveStride = 2 * kWaveLds;             // transform areas of two channels: previous_window is in registers
constexpr uint32_t kMcGroupsMax = 24;          // runs of one (shape, channel count) among the streams with more than two channels
constexpr uint32_t kCounterSets = 32, kCountersPerLaunch = 1 + kWalkShapes + kMcGroupsMax;
constexpr uint32_t kBothChannels = 0xffffffffu;       // VorbisSeg.pad of a wavefront that walks both channels of a stereo stream

#ifndef AFG_VORBIS_WAVES_PER_EU
#define AFG_VORBIS_WAVES_PER_EU 2
#endif
__global__ __launch_bounds__(64 * kWavesPerGroup) __attribute__((amdgpu_waves_per_eu(AFG_VORBIS_WAVES_PER_EU, AFG_VORBIS_WAVES_PER_EU)))
void vorbis_wave_kernel(
    const VorbisSeg *__restrict__ segs, uint32_t n_segs, const VorbisStream *__restrict__ streams,
    const uint8_t *__restrict__ pflags, const uint64_t *__restrict__ spec_off,
    const uint64_t *__restrict__ out_off, const float *tables, uint32_t tab2048,
    const float *__restrict__ spec, float *__restrict__ out, uint32_t *__restrict__ next_seg)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ltab = lds;                                                   // one copy per workgroup
    {   // 14 KB of tables: every 16-byte piece in flight at once (a dependent load -> LDS-store loop of 28 rounds cost
        // 15 % of a 16-packet wavefront's life)
        typedef float v4f __attribute__((ext_vector_type(4)));
        constexpr int kQuads = kTabBase / 4, kPer = (kQuads + 64 * kWavesPerGroup - 1) / (64 * kWavesPerGroup);
        static_assert(kTabBase % 4 == 0, "table quads");
        const v4f *src = (const v4f *)(tables + tab2048);                // tab2048 is a multiple of 4 floats (16-byte aligned)
        v4f t[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const int i = (int)threadIdx.x + k * 64 * kWavesPerGroup;
            t[k] = src[i < kQuads ? i : kQuads - 1];
        }
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const int i = (int)threadIdx.x + k * 64 * kWavesPerGroup;
            if (i < kQuads) ((v4f *)ltab)[i] = t[k];
        }
    }
    for (int i = threadIdx.x; i < (AFG_VORBIS_TW_REGS ? 0 : kLaneTw * 64); i += 64 * kWavesPerGroup)
        ((f2 *)(ltab + kTabBase))[i] = ((const f2 *)(tables + tab2048))[lane_twiddle_index(i >> 6, i & 63)];
    __syncthreads();                                                     // the only block-level barrier
    // Persistent wavefronts: the grid is one workgroup per CU (the LDS holds exactly one) and every wavefront draws
    // segments from a counter until none are left.  Launching a workgroup per 8 segments instead left each CU idle
    // between workgroups -- a new one cannot start before all 8 wavefronts of the old one have ended, and then pays
    // the launch and the table staging again: 2 of 11.5 ms on the C3 batch (profiles/r02_pmc_vorbis_*.json).
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *smem = lds + kTabFloats + wave * kWaveStride;
    for (;;) {
        uint32_t sidx = 0;
        if ((threadIdx.x & 63) == 0) sidx = atomicAdd(next_seg, 1u);
        sidx = (uint32_t)__builtin_amdgcn_readfirstlane((int)sidx);                             // scalar: segment and stream
        if (sidx >= n_segs) return;                                                            // records load as SMEM
        const VorbisSeg seg = segs[sidx];
        const VorbisStream st = streams[seg.stream];
        if (seg.pad == kBothChannels) vorbis_wave2_body(smem, ltab, seg, st, pflags, spec_off, out_off, tables, spec, out);
        else vorbis_wave_body(smem, ltab, seg, st, pflags, spec_off, out_off, tables, spec, out);
    }
}

int ilog_host(int n)       // stb_vorbis2.d:634-650
{
    int r = 0;
    while (n > 0) { r++; n >>= 1; }
    return r;
}

// Table set of one blocksize, exactly as stb_vorbis2.d:851-873 (float angle, double cos/sin).
void build_tables(int n, std::vector<float> &t)
{
    const float pi_f = 3.14159265358979323846264f;   // :652 (float enum)
    const int n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
    const size_t base = t.size();
    t.resize(base + (size_t)n2 + n2 + n4 + n2);
    float *A = t.data() + base, *B = A + n2, *C = B + n2, *W = C + n4;
    for (int k = 0, k2 = 0; k < n4; ++k, k2 += 2) {
        float a0 = (float)(4 * k) * pi_f / (float)n;
        float a1 = (float)(k2 + 1) * pi_f / (float)n / (float)2;
        A[k2] = (float)std::cos((double)a0);
        A[k2 + 1] = (float)-std::sin((double)a0);
        B[k2] = (float)std::cos((double)a1) * 0.5f;
        B[k2 + 1] = (float)std::sin((double)a1) * 0.5f;
    }
    for (int k = 0, k2 = 0; k < n8; ++k, k2 += 2) {
        float a2 = (float)(2 * (k2 + 1)) * pi_f / (float)n;
        C[k2] = (float)std::cos((double)a2);
        C[k2 + 1] = (float)-std::sin((double)a2);
    }
    for (int i = 0; i < n2; ++i) {
        double inner = std::sin((i - 0 + 0.5) / n2 * 0.5 * (double)pi_f);
        float sq = (float)inner;
        sq = sq * sq;
        W[i] = (float)std::sin(0.5 * (double)pi_f * (double)sq);
    }
}

}  // namespace

struct afg_vorbis_plan {
    uint32_t n_streams = 0;
    uint32_t n_segs = 0;
    uint64_t n_packets = 0;
    uint64_t spec_floats = 0;
    uint64_t out_floats = 0;
    uint32_t chan_nmax = 0;         // longest block among the streams of the general path (d_segs)
    std::vector<uint64_t> h_spec_off, h_out_off;
    uint32_t n_wave_segs = 0;      // segments of streams on the wave-level fast path
    // AFG_NUMERIC_TOLERANCE (vorbis_walk.hip): the segments of the streams the walk takes, one run per shape in
    // d_walk_segs.  The same streams' segments for the bit-exact kernels are the first n_wave_walk of d_wave_segs and the
    // last n_segs_walk of d_segs: a launch in tolerance mode with aligned planes skips those.
    uint32_t walk_first[kWalkShapes + 1] = {};
    // the shapes with more than two channels (6 ..): a workgroup per segment, as many wavefronts as the stream has channels
    // or pairs of them -- one launch per run of equal channel counts inside the shape's run
    struct McGroup { int shape; uint32_t first, count, nch; };
    std::vector<McGroup> mc_groups;
    uint32_t n_wave_walk = 0, n_segs_walk = 0;
    afg::DeviceArray d_walk_segs, d_walk_tables[kWalkShapes];
    uint32_t tab2048 = 0;          // float offset of the n = 2048 table set
    afg::DeviceArray d_segs, d_wave_segs, d_streams, d_pflags, d_spec_off, d_out_off, d_tables;
    afg::DeviceArray d_walk_pflags;   // d_pflags as the walk reads them (equal block sizes: every packet a long block between long blocks); empty: d_pflags
    // work counters of the persistent kernels: launch k uses (and first clears, on its stream) set k % kCounterSets -- one
    // counter for the wave kernel, one per walk shape -- so launches of one plan that overlap on different streams do not
    // share one
    afg::DeviceArray d_counters;
    mutable std::atomic<uint32_t> launches{ 0 };
    uint32_t wave_groups = 0;      // workgroups the persistent kernel is launched with
};

extern "C" int afg_vorbis_plan_create(afg_vorbis_plan **plan, uint32_t n_streams, const uint32_t *packets,
                                      const uint8_t *channels, const uint16_t *blocksize0,
                                      const uint16_t *blocksize1, const uint8_t *pflags, uint32_t seg_packets)
{
    return afg::vorbis_plan_create_at(plan, n_streams, packets, channels, blocksize0, blocksize1, pflags, nullptr, seg_packets);
}

// Library-internal variant: the spectra of stream s start at float spec_base[s] of the input plane (gaps between
// streams allowed; NULL packs them).  The host pipeline runs the kernel on its staging layout as is.
int afg::vorbis_plan_create_at(afg_vorbis_plan **plan, uint32_t n_streams, const uint32_t *packets,
                               const uint8_t *channels, const uint16_t *blocksize0, const uint16_t *blocksize1,
                               const uint8_t *pflags, const uint64_t *spec_base, uint32_t seg_packets)
{
    if (!plan) return AFG_ERR_INVALID;
    *plan = nullptr;
    if (n_streams && (!packets || !channels || !blocksize0 || !blocksize1)) {
        afg::set_error("afg_vorbis_plan_create: NULL stream description");
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    uint64_t total_packets = 0;
    for (uint32_t s = 0; s < n_streams; s++) total_packets += packets[s];
    if (seg_packets == 0) {
        // A walk item re-reads (and re-transforms) the packet in front of it to rebuild the carried half block: 1 / seg_packets of
        // the spectra are fetched twice.  Longer items do not pay that back on C3 (one box, 10 launches each: 16 packets 6.94 ms,
        // 32 7.03, 64 7.11, 128 7.22, 256 7.46 -- the tail of a persistent launch is one item long): 16 stays.
        const long v = afg::dev_option(afg::kDevVorbisSegPackets);
        seg_packets = (v > 0 && v <= (1 << 20)) ? (uint32_t)v : 16u;
    }
    const bool single_only = afg::dev_option(afg::kDevVorbisSingle) > 0;      // tests: the one-channel-per-wavefront walk

    std::vector<VorbisStream> streams(n_streams);
    std::vector<VorbisSeg> segs, segs_walk, wave_segs, wave_walk, walk_segs[kWalkShapes];
    std::vector<uint8_t> walk_pflags;
    std::vector<float> tables;
    std::map<int, uint32_t> tab_of;
    auto p = new (std::nothrow) afg_vorbis_plan;
    if (!p) return AFG_ERR_OOM;

    uint64_t pkt = 0, so = 0, oo = 0, so_extent = 0;
    uint32_t chan_nmax = 0;
    for (uint32_t s = 0; s < n_streams; s++) {
        if (spec_base) so = spec_base[s];
        const int bs[2] = { blocksize0[s], blocksize1[s] };
        for (int b = 0; b < 2; b++) {
            const int n = bs[b];
            // 64/128 are legal Vorbis sizes but the reference transform is wrong for them
            // (stb_vorbis2.d:2053-2090 applies a butterfly stage twice when n < 256): rejected.
            if (n < 256 || n > 8192 || (n & (n - 1))) {
                afg::set_error("afg_vorbis_plan_create: stream %u blocksize %d unsupported (256..8192, power of two)", s, n);
                delete p;
                return AFG_ERR_UNSUPPORTED;
            }
            if (!tab_of.count(n)) {
                tab_of[n] = (uint32_t)tables.size();
                build_tables(n, tables);
            }
        }
        if (bs[0] > bs[1] || channels[s] < 1 || channels[s] > 16) {
            afg::set_error("afg_vorbis_plan_create: stream %u: bad block sizes/channels", s);
            delete p;
            return AFG_ERR_INVALID;
        }
        if (packets[s] && !pflags) {
            delete p;
            return AFG_ERR_INVALID;
        }
        VorbisStream &st = streams[s];
        st.pkt_base = pkt;
        st.npkt = packets[s];
        st.nch = channels[s];
        st.bs[0] = bs[0];
        st.bs[1] = bs[1];
        st.tab[0] = tab_of[bs[0]];
        st.tab[1] = tab_of[bs[1]];
        const bool fast = (bs[1] == kNL) && (bs[0] <= kNL / 2) && channels[s] <= 2;
        if (!fast) chan_nmax = std::max<uint32_t>(chan_nmax, (uint32_t)bs[1]);

        int prev_len = 0;
        for (uint32_t q = 0; q < packets[s]; q++, pkt++) {
            const unsigned fl = pflags[pkt];
            const bool lng = fl & AFG_VORBIS_LONG;
            const bool prevf = lng && (fl & AFG_VORBIS_PREV), nextf = lng && (fl & AFG_VORBIS_NEXT);
            const int n = lng ? bs[1] : bs[0];
            const int left = (lng && !prevf) ? ((n - bs[0]) >> 2) : 0;                 // :2336-2342
            const int right = (lng && !nextf) ? ((n * 3 - bs[0]) >> 2) : (n >> 1);     // :2343-2349
            const int right_end = (lng && !nextf) ? ((n * 3 + bs[0]) >> 2) : n;
            const int left_end = (lng && !prevf) ? ((n + bs[0]) >> 2) : (n >> 1);
            if (prev_len && prev_len != left_end - left) {
                afg::set_error("afg_vorbis_plan_create: stream %u packet %u: window flags inconsistent with the previous packet", s, q);
                delete p;
                return AFG_ERR_INVALID;
            }
            p->h_spec_off.push_back(so);
            p->h_out_off.push_back(oo);
            so += (uint64_t)(n / 2) * channels[s];
            so_extent = so > so_extent ? so : so_extent;
            if (prev_len) oo += (uint64_t)(right - left) * channels[s];                // :2645-2656
            prev_len = right_end - right;
        }
        const int shape = single_only ? -1 : walk_shape((int)channels[s], bs[0], bs[1]);
        if (shape >= 0 && bs[0] == bs[1]) {
            // one block size: the packets' blockflag says "short" or "long" as the encoder pleased, the two windows are the
            // same -- for the walk every one of them is a long block between long blocks (the same bounds, vorbis_core.h)
            if (walk_pflags.empty()) walk_pflags.assign(pflags, pflags + total_packets);
            for (uint64_t q = st.pkt_base; q < st.pkt_base + packets[s]; q++)
                walk_pflags[q] |= (uint8_t)(AFG_VORBIS_LONG | AFG_VORBIS_PREV | AFG_VORBIS_NEXT);
        }
        for (uint32_t p0 = 0; p0 < packets[s]; p0 += seg_packets) {
            uint32_t cnt = packets[s] - p0 < seg_packets ? packets[s] - p0 : seg_packets;
            if (shape >= 0)      // (more than two channels: the workgroup's wavefronts share the item; pad = the channel count, the sort key below)
                walk_segs[shape].push_back(VorbisSeg{ s, p0, cnt, shape >= 6 ? (uint32_t)channels[s] : 0u });
            auto &wave_list = shape >= 0 ? wave_walk : wave_segs;
            if (fast && channels[s] == 2 && !single_only)
                wave_list.push_back(VorbisSeg{ s, p0, cnt, kBothChannels });                                  // both channels, interleaved
            else if (fast)
                for (uint32_t c = 0; c < channels[s]; c++) wave_list.push_back(VorbisSeg{ s, p0, cnt, c });   // one per channel
            else
                for (uint32_t c = 0; c < channels[s]; c++) (shape >= 0 ? segs_walk : segs).push_back(VorbisSeg{ s, p0, cnt, c });   // one per channel
        }
    }
    p->n_wave_walk = (uint32_t)wave_walk.size();
    wave_segs.insert(wave_segs.begin(), wave_walk.begin(), wave_walk.end());
    p->n_segs_walk = (uint32_t)segs_walk.size();
    segs.insert(segs.end(), segs_walk.begin(), segs_walk.end());
    std::vector<VorbisSeg> all_walk;
    for (int k = 0; k < kWalkShapes; k++) {
        p->walk_first[k] = (uint32_t)all_walk.size();
        if (k >= 6 && !walk_segs[k].empty()) {
            std::stable_sort(walk_segs[k].begin(), walk_segs[k].end(), [](const VorbisSeg &a, const VorbisSeg &b) { return a.pad < b.pad; });
            for (size_t i = 0; i < walk_segs[k].size();) {
                size_t j = i;
                while (j < walk_segs[k].size() && walk_segs[k][j].pad == walk_segs[k][i].pad) j++;
                p->mc_groups.push_back({ k, (uint32_t)(all_walk.size() + i), (uint32_t)(j - i), walk_segs[k][i].pad });
                i = j;
            }
        }
        all_walk.insert(all_walk.end(), walk_segs[k].begin(), walk_segs[k].end());
    }
    p->walk_first[kWalkShapes] = (uint32_t)all_walk.size();
    if (p->mc_groups.size() > kMcGroupsMax) {
        afg::set_error("afg_vorbis_plan_create: more than %u different (block size, channel count) groups above two channels", kMcGroupsMax);
        delete p;
        return AFG_ERR_UNSUPPORTED;
    }
    p->n_streams = n_streams;
    p->n_segs = (uint32_t)segs.size();
    p->n_wave_segs = (uint32_t)wave_segs.size();
    p->tab2048 = tab_of.count(kNL) ? tab_of[kNL] : 0;
    p->n_packets = pkt;
    p->spec_floats = so_extent;
    p->out_floats = oo;
    p->chan_nmax = chan_nmax;
    int rc = p->d_segs.upload(segs.data(), segs.size() * sizeof(VorbisSeg));
    if (!rc) rc = p->d_wave_segs.upload(wave_segs.data(), wave_segs.size() * sizeof(VorbisSeg));
    if (!rc) rc = p->d_streams.upload(streams.data(), streams.size() * sizeof(VorbisStream));
    if (!rc) rc = p->d_pflags.upload(pflags, (size_t)pkt);
    if (!rc && !walk_pflags.empty()) rc = p->d_walk_pflags.upload(walk_pflags.data(), walk_pflags.size());
    if (!rc) rc = p->d_spec_off.upload(p->h_spec_off.data(), p->h_spec_off.size() * sizeof(uint64_t));
    if (!rc) rc = p->d_out_off.upload(p->h_out_off.data(), p->h_out_off.size() * sizeof(uint64_t));
    if (!rc) rc = p->d_tables.upload(tables.data(), tables.size() * sizeof(float));
    if (!rc && (p->n_wave_segs || !all_walk.empty())) {
        const std::vector<uint32_t> zeros(kCounterSets * kCountersPerLaunch, 0u);
        rc = p->d_counters.upload(zeros.data(), zeros.size() * sizeof(uint32_t));
    }
    if (!rc && p->n_wave_segs) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const uint32_t need = (p->n_wave_segs + kWavesPerGroup - 1) / kWavesPerGroup;
        p->wave_groups = need < (uint32_t)cus ? need : (uint32_t)cus;
    }
    if (!rc) rc = p->d_walk_segs.upload(all_walk.data(), all_walk.size() * sizeof(VorbisSeg));
    for (int k = 0; k < kWalkShapes && !rc; k++) {
        if (p->walk_first[k + 1] == p->walk_first[k]) continue;
        const int n = walk_shape_blocksize(k);
        std::vector<float> wt(walk_table_floats(k));
        walk_build_tables(k, wt.data(), tables.data() + tab_of[n] + n + n / 4);
        rc = p->d_walk_tables[k].upload(wt.data(), wt.size() * sizeof(float));
    }
    if (!rc && p->n_wave_segs) {
        hipError_t e = hipFuncSetAttribute((const void *)vorbis_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(sizeof(float) * (kTabFloats + kWavesPerGroup * kWaveStride)));
        if (e != hipSuccess) {
            afg::set_error("hipFuncSetAttribute(wave kernel) failed: %s", hipGetErrorString(e));
            rc = AFG_ERR_HIP;
        }
    }
    if (!rc && p->n_segs) rc = channel_dispatch(true, 0, chan_nmax, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) {
        afg_vorbis_plan_destroy(p);
        return rc;
    }
    *plan = p;
    return AFG_OK;
}

extern "C" {

void afg_vorbis_plan_destroy(afg_vorbis_plan *plan)
{
    if (!plan) return;
    plan->d_segs.release();
    plan->d_wave_segs.release();
    plan->d_streams.release();
    plan->d_pflags.release();
    plan->d_walk_pflags.release();
    plan->d_spec_off.release();
    plan->d_out_off.release();
    plan->d_tables.release();
    plan->d_counters.release();
    plan->d_walk_segs.release();
    for (auto &t : plan->d_walk_tables) t.release();
    delete plan;
}

uint64_t afg_vorbis_plan_packets(const afg_vorbis_plan *plan) { return plan ? plan->n_packets : 0; }
uint64_t afg_vorbis_plan_spec_floats(const afg_vorbis_plan *plan) { return plan ? plan->spec_floats : 0; }
uint64_t afg_vorbis_plan_out_floats(const afg_vorbis_plan *plan) { return plan ? plan->out_floats : 0; }

int afg_vorbis_plan_offsets(const afg_vorbis_plan *plan, uint64_t *spec_off, uint64_t *out_off)
{
    if (!plan) return AFG_ERR_INVALID;
    if (spec_off) std::memcpy(spec_off, plan->h_spec_off.data(), plan->h_spec_off.size() * sizeof(uint64_t));
    if (out_off) std::memcpy(out_off, plan->h_out_off.data(), plan->h_out_off.size() * sizeof(uint64_t));
    return AFG_OK;
}

int afg_vorbis_transform_hip(const afg_vorbis_plan *plan, const float *d_spec, float *d_out, void *hip_stream)
{
    if (!plan) return AFG_ERR_INVALID;
    if (plan->n_segs == 0 && plan->n_wave_segs == 0) return AFG_OK;
    if (!d_spec || (!d_out && plan->out_floats)) {
        afg::set_error("afg_vorbis_transform_hip: NULL device pointer");
        return AFG_ERR_INVALID;
    }
    // AFG_NUMERIC_TOLERANCE: the streams vorbis_walk.hip has a shape for take the re-factored walk (16-byte PCM stores,
    // 8-byte spectrum loads -- hence the alignment test); everything else, and everything in AFG_NUMERIC_EXACT, the bit-exact kernels
    const uint32_t n_walk = plan->walk_first[kWalkShapes];
    const bool walk = n_walk && afg::numeric_mode() == AFG_NUMERIC_TOLERANCE && ((uintptr_t)d_out & 15) == 0 && ((uintptr_t)d_spec & 7) == 0;
    const uint32_t wave_skip = walk ? plan->n_wave_walk : 0, n_segs = plan->n_segs - (walk ? plan->n_segs_walk : 0);
    if (walk || plan->n_wave_segs > wave_skip) {
        uint32_t *counter = (uint32_t *)plan->d_counters.ptr + kCountersPerLaunch * (plan->launches.fetch_add(1) % kCounterSets);
        AFG_HIP_CHECK(hipMemsetAsync(counter, 0, kCountersPerLaunch * sizeof(uint32_t), (hipStream_t)hip_stream));
        for (int k = 0; walk && k < 6; k++) {
            const uint32_t first = plan->walk_first[k], count = plan->walk_first[k + 1] - first;
            if (!count) continue;
            if (int rc = walk_launch(k, (const VorbisSeg *)plan->d_walk_segs.ptr + first, count, 0, (const VorbisStream *)plan->d_streams.ptr,
                                     (const uint8_t *)(plan->d_walk_pflags.ptr ? plan->d_walk_pflags.ptr : plan->d_pflags.ptr), (const uint64_t *)plan->d_spec_off.ptr,
                                     (const uint64_t *)plan->d_out_off.ptr, (const float *)plan->d_tables.ptr,
                                     (const float *)plan->d_walk_tables[k].ptr, d_spec, d_out, counter + 1 + k, (hipStream_t)hip_stream))
                return rc;
        }
        for (size_t g = 0; walk && g < plan->mc_groups.size(); g++) {
            const auto &mg = plan->mc_groups[g];
            if (int rc = walk_launch(mg.shape, (const VorbisSeg *)plan->d_walk_segs.ptr + mg.first, mg.count, (int)mg.nch, (const VorbisStream *)plan->d_streams.ptr,
                                     (const uint8_t *)(plan->d_walk_pflags.ptr ? plan->d_walk_pflags.ptr : plan->d_pflags.ptr), (const uint64_t *)plan->d_spec_off.ptr,
                                     (const uint64_t *)plan->d_out_off.ptr, (const float *)plan->d_tables.ptr,
                                     (const float *)plan->d_walk_tables[mg.shape].ptr, d_spec, d_out, counter + 1 + kWalkShapes + g, (hipStream_t)hip_stream))
                return rc;
        }
        if (plan->n_wave_segs > wave_skip) {
            const uint32_t rest = plan->n_wave_segs - wave_skip, need = (rest + kWavesPerGroup - 1) / kWavesPerGroup;
            hipLaunchKernelGGL(vorbis_wave_kernel, dim3(need < plan->wave_groups ? need : plan->wave_groups),
                               dim3(64 * kWavesPerGroup), sizeof(float) * (kTabFloats + kWavesPerGroup * kWaveStride),
                               (hipStream_t)hip_stream, (const VorbisSeg *)plan->d_wave_segs.ptr + wave_skip, rest,
                               (const VorbisStream *)plan->d_streams.ptr, (const uint8_t *)plan->d_pflags.ptr,
                               (const uint64_t *)plan->d_spec_off.ptr, (const uint64_t *)plan->d_out_off.ptr,
                               (const float *)plan->d_tables.ptr, plan->tab2048, d_spec, d_out, counter);
        }
    }
    if (n_segs)
        if (int rc = channel_dispatch(false, n_segs, plan->chan_nmax, (hipStream_t)hip_stream, (const VorbisSeg *)plan->d_segs.ptr,
                                      (const VorbisStream *)plan->d_streams.ptr, (const uint8_t *)plan->d_pflags.ptr,
                                      (const uint64_t *)plan->d_spec_off.ptr, (const uint64_t *)plan->d_out_off.ptr,
                                      (const float *)plan->d_tables.ptr, d_spec, d_out))
            return rc;
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}

}  // extern "C"
