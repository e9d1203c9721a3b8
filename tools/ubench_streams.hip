// Micro-benchmark (round 4): what the "many independent streams" access pattern of the persistent transform kernels costs.
//
// Every headline kernel is a set of resident wavefronts that each read their own contiguous run of the input plane and
// write their own contiguous run of the output plane, one unit (4.6 - 8 KB) per step, and all of them stop at 5.0 - 5.3
// TB/s where a plain copy reaches 6.2 - 6.5.  This program moves 16 GB in + 16 GB out with
//     S   resident wavefronts (S / 256 per CU, 64 lanes each),
//     B   bytes per wavefront and step (16-byte accesses, B / 1024 instructions each way),
//     K   steps per segment: a wavefront owns K consecutive units, then takes the next free segment (atomic counter),
//     pad cycles of s_sleep per step (a stand-in for the transform),
// in three placements of the units:
//     own     segment s = units [s K, (s + 1) K): what the kernels do (the plane is contiguous per file)
//     dense   unit (step k of segment s) at position k * n_segments_in_flight + ... i.e. wavefronts that run together touch
//             neighbouring units: reads, writes or both
// and prints GB/s.  Usage: ubench_streams [mode ...]; no arguments runs the whole table.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

// units of B bytes; n_units total; segments of K units; placement: bit 0 = dense reads, bit 1 = dense writes
template <int NT>
__global__ __launch_bounds__(64) void streams(const f4 *__restrict__ in, f4 *__restrict__ out, uint32_t n_segs, int K, int quads_per_unit,
                                              int placement, int pad, uint32_t *next, int do_read, int do_write)
{
    const int lane = threadIdx.x;
    for (;;) {
        uint32_t s = 0;
        if (lane == 0) s = atomicAdd(next, 1u);
        s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
        if (s >= n_segs) return;
        f4 acc = { 0, 0, 0, 0 };
        for (int k = 0; k < K; k++) {
            const uint64_t own = (uint64_t)s * K + k;
            // dense: unit k of segment s sits at k * n_segs + s -- segments drawn together are neighbours at every step
            const uint64_t dense = (uint64_t)k * n_segs + s;
            const uint64_t ru = (placement & 1) ? dense : own, wu = (placement & 2) ? dense : own;
            const f4 *src = in + ru * quads_per_unit;
            f4 *dst = out + wu * quads_per_unit;
            for (int q = lane; q < quads_per_unit; q += 64 * 8) {
                f4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    v[u] = f4{ 1, 2, 3, 4 };
                    if (do_read && q + 64 * u < quads_per_unit) v[u] = NT ? __builtin_nontemporal_load(src + q + 64 * u) : src[q + 64 * u];
                }
                for (int z = 0; z < pad; z++) __builtin_amdgcn_s_sleep(8);        // ~ 8 x 64 cycles each
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (!do_write) acc += v[u];
                    else if (q + 64 * u < quads_per_unit) {
                        if (NT) __builtin_nontemporal_store(v[u], dst + q + 64 * u);
                        else dst[q + 64 * u] = v[u];
                    }
                }
            }
        }
        if (!do_write && acc.x == 12345.0f) out[0] = acc;
    }
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)16 << 30;
    f4 *a, *b;
    uint32_t *ctr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&ctr, 4) != hipSuccess) return 1;
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    struct Case { const char *name; int waves_per_cu, B, K, placement, pad, rd, wr, nt; };
    std::vector<Case> cases;
    for (int B : { 4608, 8192, 32768, 131072 })
        for (int wpc : { 8, 16 })
            for (int K : { 16, 64 }) {
                cases.push_back({ "copy own", wpc, B, K, 0, 0, 1, 1, 1 });
                cases.push_back({ "copy dense-w", wpc, B, K, 2, 0, 1, 1, 1 });
                cases.push_back({ "copy dense-rw", wpc, B, K, 3, 0, 1, 1, 1 });
            }
    for (int B : { 8192 })
        for (int wpc : { 8, 16 }) {
            cases.push_back({ "write own", wpc, B, 16, 0, 0, 0, 1, 1 });
            cases.push_back({ "write dense", wpc, B, 16, 2, 0, 0, 1, 1 });
            cases.push_back({ "read own", wpc, B, 16, 0, 0, 1, 0, 1 });
            cases.push_back({ "read dense", wpc, B, 16, 1, 0, 1, 0, 1 });
            cases.push_back({ "copy own pad", wpc, B, 16, 0, 40, 1, 1, 1 });
            cases.push_back({ "copy dense-w pad", wpc, B, 16, 2, 40, 1, 1, 1 });
            cases.push_back({ "copy own plain", wpc, B, 16, 0, 0, 1, 1, 0 });
        }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Case &c : cases) {
        const int qpu = c.B / 16;
        const uint64_t n_units = bytes / c.B;
        const uint32_t n_segs = (uint32_t)(n_units / c.K);
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipMemsetAsync(ctr, 0, 4, 0);
            hipEventRecord(e0);
            if (c.nt) hipLaunchKernelGGL(streams<1>, dim3(256 * c.waves_per_cu), dim3(64), 0, 0, a, b, n_segs, c.K, qpu, c.placement, c.pad, ctr, c.rd, c.wr);
            else hipLaunchKernelGGL(streams<0>, dim3(256 * c.waves_per_cu), dim3(64), 0, 0, a, b, n_segs, c.K, qpu, c.placement, c.pad, ctr, c.rd, c.wr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double moved = (double)n_segs * c.K * c.B * (c.rd + c.wr);
        printf("%-18s B=%-6d waves/CU=%-2d K=%-3d %.3f ms  %.0f GB/s\n", c.name, c.B, c.waves_per_cu, c.K, best, moved / best / 1e6);
        fflush(stdout);
    }
    return 0;
}
