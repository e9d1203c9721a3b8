// mp3_transform.hip -- MP3 Layer III transform stage on gfx950.
//
// Replaces, for whole batches of streams, the tail of L3_decode
// (reference minimp3.d:1226-1228: L3_antialias -> L3_imdct_gr ->
// L3_change_sign) and mp3d_synth_granule (minimp3.d:1408-1434, called at
// :1553).  Every output sample is produced by the same float32 expression tree
// as the reference (the library is compiled with -ffp-contract=off), only the
// data movement is different:
//
//   * one wavefront (a 64-thread workgroup) walks `seg_granules` consecutive
//     granules of one stream; a segment that does not start at granule 0
//     first re-derives its carry state from the two preceding granules
//     (PCM(g) depends on X[g], X[g-1], X[g-2] only: the IMDCT overlap written
//     by a granule depends on that granule's spectrum alone, minimp3.d:1095,
//     :1137-1140, and the polyphase window spans 15 earlier slots < 18);
//   * antialias + IMDCT36/12 + frequency inversion run with lane = (channel,
//     subband), 18 spectral lines and the 9 overlap values in registers,
//     neighbour subbands exchanged with wave shuffles;
//   * the 32-point DCT-II runs with lane = (channel, time slot) out of an LDS
//     transposition buffer and writes 32-value rows of the polyphase history;
//   * the history is a 36-row LDS ring (two granules), the 512-tap window runs
//     with lane = (channel, output sample) and stores fully coalesced
//     interleaved PCM.
//
// Coefficients are read with fully coalesced 256-byte wave loads one granule
// ahead of their use and staged through LDS into the (channel, subband) layout.
#define AFG_MP3_FMA 0
#define AFG_MP3_KERNEL mp3_transform_kernel
#include "mp3_kernel.h"

#include <vector>

struct afg_mp3_plan {
    uint32_t n_streams = 0;
    uint32_t n_segs = 0;
    uint64_t blocks = 0;
    afg::DeviceArray d_segs, d_streams;
};

extern "C" {

int afg_mp3_plan_create(afg_mp3_plan **plan, uint32_t n_streams, const uint32_t *granules,
                        const uint8_t *channels, uint32_t seg_granules)
{
    return afg::mp3_plan_create_at(plan, n_streams, granules, channels, nullptr, seg_granules);
}

}  // extern "C"

// Library-internal variant: stream s starts at block blk_base[s] of the planes (gaps between streams allowed);
// NULL packs the streams back to back.  The host pipeline uses it to run the kernel on the staging layout as is.
int afg::mp3_plan_create_at(afg_mp3_plan **plan, uint32_t n_streams, const uint32_t *granules,
                            const uint8_t *channels, const uint64_t *blk_base, uint32_t seg_granules, afg::PlanArena *arena)
{
    if (!plan) return AFG_ERR_INVALID;
    *plan = nullptr;
    if (n_streams && (!granules || !channels)) {
        afg::set_error("afg_mp3_plan_create: NULL stream description");
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    if (seg_granules == 0) seg_granules = 48;

    std::vector<Mp3Stream> streams(n_streams);
    std::vector<Mp3Seg> segs;
    uint64_t blk = 0, extent = 0;
    for (uint32_t s = 0; s < n_streams; s++) {
        if (channels[s] != 1 && channels[s] != 2) {
            afg::set_error("afg_mp3_plan_create: stream %u has %u channels (1 or 2 expected)", s, channels[s]);
            return AFG_ERR_INVALID;
        }
        if (blk_base) blk = blk_base[s];
        streams[s] = Mp3Stream{ blk, granules[s], channels[s] };
        for (uint32_t g0 = 0; g0 < granules[s]; g0 += seg_granules) {
            uint32_t cnt = granules[s] - g0 < seg_granules ? granules[s] - g0 : seg_granules;
            segs.push_back(Mp3Seg{ s, g0, cnt, (g0 + cnt == granules[s]) ? 1u : 0u });
        }
        blk += (uint64_t)granules[s] * channels[s];
        extent = blk > extent ? blk : extent;
    }
    if (segs.size() > 0x7fffffffu) {
        afg::set_error("afg_mp3_plan_create: too many segments");
        return AFG_ERR_INVALID;
    }
    afg_mp3_plan *p = new (std::nothrow) afg_mp3_plan;
    if (!p) return AFG_ERR_OOM;
    p->n_streams = n_streams;
    p->n_segs = (uint32_t)segs.size();
    p->blocks = extent;
    int rc = AFG_OK;
    const size_t seg_bytes = segs.size() * sizeof(Mp3Seg), st_bytes = streams.size() * sizeof(Mp3Stream);
    if (arena && arena->used + seg_bytes + st_bytes <= arena->cap) {
        uint8_t *h = arena->host + arena->used, *d = arena->dev + arena->used;
        std::memcpy(h, segs.data(), seg_bytes);
        std::memcpy(h + seg_bytes, streams.data(), st_bytes);
        hipError_t e = hipMemcpyAsync(d, h, seg_bytes + st_bytes, hipMemcpyHostToDevice, arena->stream);
        if (e != hipSuccess) {
            afg::set_error("plan table upload failed: %s", hipGetErrorString(e));
            rc = AFG_ERR_HIP;
        }
        p->d_segs.ptr = d; p->d_segs.bytes = seg_bytes; p->d_segs.owned = false;
        p->d_streams.ptr = d + seg_bytes; p->d_streams.bytes = st_bytes; p->d_streams.owned = false;
        arena->used += (seg_bytes + st_bytes + 15) & ~(size_t)15;
    } else {
        rc = p->d_segs.upload(segs.data(), seg_bytes);
        if (!rc) rc = p->d_streams.upload(streams.data(), st_bytes);
    }
    if (rc) {
        afg_mp3_plan_destroy(p);
        return rc;
    }
    *plan = p;
    return AFG_OK;
}

extern "C" {

void afg_mp3_plan_destroy(afg_mp3_plan *plan)
{
    if (!plan) return;
    plan->d_segs.release();
    plan->d_streams.release();
    delete plan;
}

uint64_t afg_mp3_plan_blocks(const afg_mp3_plan *plan) { return plan ? plan->blocks : 0; }
uint32_t afg_mp3_plan_segments(const afg_mp3_plan *plan) { return plan ? plan->n_segs : 0; }

int afg_mp3_transform_hip(const afg_mp3_plan *plan, const float *d_coef, const uint32_t *d_flags,
                          float *d_pcm, float *d_state, void *hip_stream)
{
    if (!plan) return AFG_ERR_INVALID;
    if (plan->n_segs == 0) return AFG_OK;
    if (!d_coef || !d_flags || !d_pcm) {
        afg::set_error("afg_mp3_transform_hip: NULL device pointer");
        return AFG_ERR_INVALID;
    }
    // AFG_NUMERIC_TOLERANCE: the same walk with fused multiply-adds (mp3_tolerance.hip); the carry-state blob is common
    if (afg::numeric_mode() == AFG_NUMERIC_TOLERANCE)
        afg::mp3_launch_tolerance(plan->n_segs, plan->d_segs.ptr, plan->d_streams.ptr, d_coef, d_flags, d_pcm, d_state, (hipStream_t)hip_stream);
    else
        hipLaunchKernelGGL(mp3_transform_kernel, dim3(plan->n_segs), dim3(64), 0, (hipStream_t)hip_stream,
                           (const Mp3Seg *)plan->d_segs.ptr, (const Mp3Stream *)plan->d_streams.ptr,
                           d_coef, d_flags, d_pcm, d_state);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}

}  // extern "C"
