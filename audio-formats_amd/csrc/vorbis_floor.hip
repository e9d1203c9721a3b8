// vorbis_floor.hip -- Vorbis inverse coupling and floor-1 curve multiplication on gfx950.
//
// Replaces, between the host's residue decode and the transform stage (vorbis_transform.hip):
//   stb_vorbis2.d:2493-2514   inverse coupling of the residue vectors
//   stb_vorbis2.d:2516-2523   silent channels (really_zero_channel) / do_floor per channel
//   stb_vorbis2.d:2255-2284   do_floor: the line segments between the floor points that survive step 2, the flat tail
//   stb_vorbis2.d:1534-1563   draw_line: integer line with an error accumulator, target[x] *= inverse_db_table[y & 255]
//
// Every bin is independent once draw_line is written in closed form: after k steps from (x0, y0) the accumulator has
// overflowed floor(k * ady' / adx) times (ady' = ady - |base| * adx < adx, one overflow at most per step), so
//   y(x0 + k) = y0 + k * base + (sy - base) * floor(k * ady' / adx).
// A workgroup takes one packet; a thread takes four consecutive bins of every channel: coupling steps in place (same
// thread, same addresses: program order), then per channel the segment holding the first bin by binary search over the
// curve's points and a linear walk for the other three.  One float multiply per bin, the reference's: bit-identical.
// A streaming pass over the spectrum plane (4 B in + 4 B out per bin); it rides in front of the transform of a chunk
// whose PCIe transfers take an order of magnitude longer.
#include "afg_common.h"

#include <mutex>

#include "../host/vorbis_front_tables.h"

namespace {

__device__ uint32_t d_inverse_db[256];

struct f4 { float v[4]; };

__device__ __forceinline__ f4 ld4(const float *p) { const float4 t = *(const float4 *)p; return f4{ { t.x, t.y, t.z, t.w } }; }
__device__ __forceinline__ void st4(float *p, const f4 &a) { *(float4 *)p = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]); }

// the multiplier of bin j on the curve pts[0 .. np): *seg is the last point with x <= j (kept between calls: j ascends)
__device__ __forceinline__ float floor_value(const int32_t *__restrict__ pts, int np, int j, int *seg)
{
    int s = *seg;
    while (s + 1 < np && pts[2 * (s + 1)] <= j) s++;
    *seg = s;
    const int x0 = pts[2 * s], y0 = pts[2 * s + 1];
    int y = y0;
    if (s + 1 < np) {
        const int x1 = pts[2 * s + 2], y1 = pts[2 * s + 3];
        const int dy = y1 - y0, adx = x1 - x0;                   // adx > 0: x1 > j >= x0
        const int base = dy / adx;
        const int sy = dy < 0 ? base - 1 : base + 1;
        const int ady = (dy < 0 ? -dy : dy) - (base < 0 ? -base : base) * adx;
        const int k = j - x0;
        y = y0 + k * base + (sy - base) * (int)(((int64_t)k * ady) / adx);
    }
    return __uint_as_float(d_inverse_db[y & 255]);
}

__global__ __launch_bounds__(256) void vorbis_floor_kernel(const afg_vorbis_floor_packet *__restrict__ pk,
                                                           const afg_vorbis_floor_curve *__restrict__ curves,
                                                           const int32_t *__restrict__ points, const uint8_t *__restrict__ steps,
                                                           float *__restrict__ spec)
{
    const afg_vorbis_floor_packet p = pk[blockIdx.x];
    const int n2 = (int)p.n2, C = (int)p.channels;
    float *base = spec + p.spec_off;
    for (int j0 = 4 * (int)threadIdx.x; j0 < n2; j0 += 4 * 256) {
        for (uint32_t s = 0; s < p.n_steps; s++) {                  // :2493-2514, steps already in the order applied
            float *mp = base + (size_t)steps[2 * (p.step_off + s)] * n2 + j0;
            float *ap = base + (size_t)steps[2 * (p.step_off + s) + 1] * n2 + j0;
            f4 m = ld4(mp), a = ld4(ap);
            for (int i = 0; i < 4; i++) {
                const float mv = m.v[i], av = a.v[i];
                float m2, a2;
                if (mv > 0) {
                    if (av > 0) { m2 = mv; a2 = mv - av; }
                    else { a2 = mv; m2 = mv + av; }
                } else {
                    if (av > 0) { m2 = mv; a2 = mv + av; }
                    else { a2 = mv; m2 = mv - av; }
                }
                m.v[i] = m2;
                a.v[i] = a2;
            }
            st4(mp, m);
            st4(ap, a);
        }
        for (int c = 0; c < C; c++) {                                  // :2516-2523
            float *t = base + (size_t)c * n2 + j0;
            const afg_vorbis_floor_curve cv = curves[p.curve_index + c];
            f4 v;
            if (cv.n_points == 0) {
                v = f4{ { 0.0f, 0.0f, 0.0f, 0.0f } };                  // really_zero_channel: memset
            } else {
                const int32_t *pts = points + 2 * (size_t)cv.point_off;
                const int np = (int)cv.n_points;
                int lo = 0, hi = np - 1;                               // last point with x <= j0 (pts[0].x == 0 <= j0)
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (pts[2 * mid] <= j0) lo = mid; else hi = mid - 1;
                }
                v = ld4(t);
                for (int i = 0; i < 4; i++) v.v[i] *= floor_value(pts, np, j0 + i, &lo);
            }
            st4(t, v);
        }
    }
}

std::mutex g_mu;
bool g_ready[AFG_MAX_DEVICES] = {};

int ensure_table()
{
    int dev = 0;
    if (int rc = afg::device_slot(&dev, "afg_vorbis_floor_hip")) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ready[dev]) {
        AFG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_inverse_db), k_inverse_db_bits, sizeof(k_inverse_db_bits)));
        g_ready[dev] = true;
    }
    return AFG_OK;
}

}  // namespace

extern "C" int afg_vorbis_floor_hip(uint64_t n_packets, const afg_vorbis_floor_packet *d_packets, const afg_vorbis_floor_curve *d_curves,
                                    const int32_t *d_points, const uint8_t *d_steps, float *d_spec, void *hip_stream)
{
    static_assert(sizeof(afg_vorbis_floor_packet) == 32 && sizeof(afg_vorbis_floor_curve) == 8, "record layout");
    if (n_packets == 0) return AFG_OK;
    if (!d_packets || !d_curves || !d_points || !d_spec) {
        afg::set_error("afg_vorbis_floor_hip: NULL device pointer");
        return AFG_ERR_INVALID;
    }
    if (n_packets > 0x7fffffffull) {
        afg::set_error("afg_vorbis_floor_hip: at most 2^31 packets per call");
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    if (int rc = ensure_table()) return rc;
    hipLaunchKernelGGL(vorbis_floor_kernel, dim3((uint32_t)n_packets), dim3(256), 0, (hipStream_t)hip_stream, d_packets, d_curves,
                       d_points, d_steps, d_spec);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}
