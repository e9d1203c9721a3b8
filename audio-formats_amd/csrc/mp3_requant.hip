// mp3_requant.hip -- MP3 requantisation, stereo processing and reorder on gfx950.
//
// Replaces, between the host's Huffman decoder and the transform stage (mp3_transform.hip):
//   minimp3.d:722-746    L3_pow_43, and the `* sf` of L3_huffman (:835-858 big values, :868-879 count1)
//   minimp3.d:885-982    L3_midside_stereo / L3_intensity_stereo / L3_stereo_process
//   minimp3.d:984-1000   L3_reorder
// so that 2 bytes per spectral line cross PCIe instead of 4.  Every float is produced by the reference's expression
// (built with -ffp-contract=off, correctly rounded division): the output equals what L3_decode holds in grbuf at :1226.
//
// One wavefront per granule; lane l takes the lines l, l + 64, ... of both channels (stereo processing pairs the
// channels line by line, before the reorder), 2-byte coalesced loads, 4-byte coalesced stores of the long part, the
// short part scattered by the table.  A pure streaming pass: 2 + 4 bytes per line of HBM traffic.
#include "afg_common.h"

#include <mutex>

#include "../host/afg_mp3_front.h"

namespace {

__device__ uint8_t d_band_of_line[24][576];
__device__ uint16_t d_dst_of_src[24][576];
__device__ float d_pow43[145];

// L3_pow_43 (minimp3.d:737-746) for x >= 129; below, the table
__device__ __forceinline__ float pow43_big(int x)
{
    int mult = 256;
    if (x < 1024) {
        mult = 16;
        x <<= 3;
    }
    const int sign = 2 * x & 64;
    const float frac = (float)((x & 63) - sign) / (float)((x & ~63) + sign);
    return d_pow43[16 + ((x + sign) >> 6)] * (1.0f + frac * ((4.0f / 3) + frac * (2.0f / 9))) * (float)mult;
}

// the requantised line: sign * (scale * |v|^(4/3)) -- the same bits as `g_pow43[16 + lsb - 16*sign] * one` (:856),
// `one * L3_pow_43(lsb) * (sign ? -1 : 1)` (:851) and `+-one` (:875-878)
__device__ __forceinline__ float requant(int v, float one)
{
    if (v == 0) return 0.0f;
    const int a = v < 0 ? -v : v;
    const float p = a < 129 ? d_pow43[16 + a] : pow43_big(a);
    const float r = one * p;
    return v < 0 ? -r : r;
}

__global__ __launch_bounds__(64) void mp3_requant_kernel(const afg_mp3_qgranule *__restrict__ grs, const int16_t *__restrict__ q,
                                                         const afg_mp3_sdesc *__restrict__ sdesc, float *__restrict__ coef, uint64_t n)
{
    const uint64_t gi = blockIdx.x;
    if (gi >= n) return;
    const afg_mp3_qgranule *g = grs + gi;
    const int lane = threadIdx.x;
    const int nch = g->nch;
    if (nch == 0) return;                                // an unused record slot (the host pipeline keeps one slot per block)
    const unsigned stereo = nch == 2 ? g->stereo : 0;
    const int t0 = g->table[0] & 0x1f, t1 = g->table[1] & 0x1f;
    const bool ro0 = (g->table[0] & 0x80) != 0, ro1 = (g->table[1] & 0x80) != 0;
    const int16_t *q0 = q + g->q_off, *q1 = q0 + 576;
    float *c0 = coef + g->coef_off, *c1 = c0 + 576;
    const afg_mp3_sdesc *sd = stereo == 2 ? sdesc + g->sdesc : nullptr;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int i = lane + 64 * k;
        const int b0 = d_band_of_line[t0][i];
        float x0 = requant(q0[i], g->scale[0][b0]), x1 = 0.0f;
        if (nch == 2) x1 = requant(q1[i], g->scale[1][d_band_of_line[t1][i]]);
        unsigned mode = stereo;
        float fl = 0.0f, fr = 0.0f;
        if (sd) {
            mode = sd->type[b0];                         // bands of the left channel's table (minimp3.d:903-936)
            fl = sd->fl[b0];
            fr = sd->fr[b0];
        }
        if (mode == 1) {                                 // L3_midside_stereo
            const float a = x0, b = x1;
            x0 = a + b;
            x1 = a - b;
        } else if (mode == 2) {                          // L3_intensity_stereo_band: right first, then left
            x1 = x0 * fr;
            x0 = x0 * fl;
        }
        c0[ro0 ? d_dst_of_src[t0][i] : i] = x0;
        if (nch == 2) c1[ro1 ? d_dst_of_src[t1][i] : i] = x1;
    }
}

std::mutex g_mu;
bool g_ready[AFG_MAX_DEVICES] = {};

int ensure_tables()
{
    int dev = 0;
    if (int rc = afg::device_slot(&dev, "afg_mp3_requant_hip")) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ready[dev]) {
        const afg_mp3::QTables &t = afg_mp3::qtables();
        AFG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_band_of_line), t.band_of_line, sizeof(t.band_of_line)));
        AFG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_dst_of_src), t.dst_of_src, sizeof(t.dst_of_src)));
        AFG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_pow43), t.pow43, sizeof(t.pow43)));
        g_ready[dev] = true;
    }
    return AFG_OK;
}

}  // namespace

extern "C" int afg_mp3_requant_hip(uint64_t n_granules, const afg_mp3_qgranule *d_granules, const int16_t *d_q,
                                   const afg_mp3_sdesc *d_sdesc, float *d_coef, void *hip_stream)
{
    static_assert(sizeof(afg_mp3_qgranule) == 344 && sizeof(afg_mp3_sdesc) == 360, "record layout");
    if (n_granules == 0) return AFG_OK;
    if (!d_granules || !d_q || !d_coef) {
        afg::set_error("afg_mp3_requant_hip: NULL device pointer");
        return AFG_ERR_INVALID;
    }
    if (n_granules > 0x7fffffffull) {
        afg::set_error("afg_mp3_requant_hip: at most 2^31 granules per call");
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    if (int rc = ensure_tables()) return rc;
    hipLaunchKernelGGL(mp3_requant_kernel, dim3((uint32_t)n_granules), dim3(64), 0, (hipStream_t)hip_stream, d_granules, d_q,
                       d_sdesc, d_coef, n_granules);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}
