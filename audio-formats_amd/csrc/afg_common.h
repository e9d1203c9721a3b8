// afg_common.h -- shared plumbing of the C-ABI library (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/afg.h"

namespace afg {

// thread-local detail string behind afg_last_error()
void set_error(const char *fmt, ...);

// Verifies a gfx950 device is present and selected; loud failure otherwise.
int require_device();

// Per-device state of the library (table copies, side streams, work counters) lives in arrays of this many slots --
// 8 MI355X in CPX partition mode show up as 64 devices.  device_slot() returns the calling thread's current device
// in *dev, or AFG_ERR_INVALID with a message naming the cap when its index does not fit.
#define AFG_MAX_DEVICES 64
int device_slot(int *dev, const char *who);

// Numeric mode of the float transform stages (afg_set_numeric_mode; env AFG_NUMERIC=exact|tolerance until it is called).
int numeric_mode();

// Alternative code paths the test-suite runs side by side with the default ones (afg.h: afg_dev_option).  Set through
// that call only -- the library reads no environment variable for them -- and -1 while unset.
enum DevOption { kDevCeltPath, kDevCeltDeSeq, kDevCeltDeDuo, kDevCeltSegRecs, kDevCeltWholeFrames, kDevVorbisSingle,
                 kDevMp3Chunks, kDevMp3FloatUpload, kDevVorbisHostFloor, kDevFlacHostRes32, kDevVorbisSegPackets, kDevBatchGroups, kDevCount };
long dev_option(DevOption which);

#define AFG_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::afg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                             __FILE__, __LINE__);                                        \
            return AFG_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

// Caller-owned room for plan tables: `host` is page-locked, `dev` device memory of the same size; tables of plans
// created with an arena are written to `host`, copied on `stream` and never allocated or waited for (the host
// pipeline creates plans while earlier chunks are in flight: a synchronous upload would queue behind their copies).
struct PlanArena {
    uint8_t *host = nullptr, *dev = nullptr;
    size_t cap = 0, used = 0;
    hipStream_t stream = nullptr;
};

// afg_mp3_plan_create with explicit block offsets per stream (NULL: packed); see mp3_transform.hip
int mp3_plan_create_at(afg_mp3_plan **plan, uint32_t n_streams, const uint32_t *granules, const uint8_t *channels,
                       const uint64_t *blk_base, uint32_t seg_granules, PlanArena *arena = nullptr);

// the MP3 transform in AFG_NUMERIC_TOLERANCE (mp3_tolerance.hip); d_segs / d_streams are the plan's device tables
void mp3_launch_tolerance(uint32_t n_segs, const void *d_segs, const void *d_streams, const float *d_coef,
                          const uint32_t *d_flags, float *d_pcm, float *d_state, hipStream_t stream);

// afg_vorbis_plan_create with an explicit input offset per stream (NULL: packed); see vorbis_transform.hip
int vorbis_plan_create_at(afg_vorbis_plan **plan, uint32_t n_streams, const uint32_t *packets, const uint8_t *channels,
                          const uint16_t *blocksize0, const uint16_t *blocksize1, const uint8_t *pflags,
                          const uint64_t *spec_base, uint32_t seg_packets);

// Owns a device buffer filled from a host array at plan creation.
struct DeviceArray {
    void *ptr = nullptr;
    size_t bytes = 0;
    bool owned = true;                  // false: points into a PlanArena
    int upload(const void *host, size_t nbytes);
    void release();
};

}  // namespace afg
