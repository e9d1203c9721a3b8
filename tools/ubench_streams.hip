// Micro-benchmark (round 4): what the "many independent streams" access pattern of the persistent transform kernels costs.
//
// Every headline kernel is a set of resident wavefronts that each read their own contiguous run of the input plane and
// write their own contiguous run of the output plane, one unit (4.6 - 8 KB) per step, and all of them stop at 5.0 - 5.3
// TB/s where a plain copy reaches 6.2 - 6.5 (HISTORY.md 8).  This program moves 16 GB in + 16 GB out with
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
// store flavours (MI355X_MICROARCH.md, "stores of each flavour"): 0 plain, 1 nt, 2 sc1 (write-through, dropped from L2), 3 sc0 sc1, 4 sc1 nt
template <int FL>
__device__ __forceinline__ void store16(f4 *p, f4 v)
{
    if (FL == 0) *p = v;
    else if (FL == 1) __builtin_nontemporal_store(v, p);
    else if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    else if (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

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
                    if (do_read && q + 64 * u < quads_per_unit) v[u] = (NT == 1) ? __builtin_nontemporal_load(src + q + 64 * u) : src[q + 64 * u];
                }
                for (int z = 0; z < pad; z++) __builtin_amdgcn_s_sleep(8);        // ~ 8 x 64 cycles each
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (!do_write) acc += v[u];
                    else if (q + 64 * u < quads_per_unit) {
                        if (NT == 1) store16<1>(dst + q + 64 * u, v[u]);
                        else if (NT == 0) store16<0>(dst + q + 64 * u, v[u]);
                        else if (NT == 2) store16<2>(dst + q + 64 * u, v[u]);
                        else if (NT == 3) store16<3>(dst + q + 64 * u, v[u]);
                        else store16<4>(dst + q + 64 * u, v[u]);
                    }
                }
            }
        }
        if (!do_write && acc.x == 12345.0f) out[0] = acc;
    }
}

// W wavefronts of a workgroup walk ONE stream together: unit (k * W + w) of the group's segment -- a contiguous W x B piece per step
template <int W>
__global__ __launch_bounds__(64 * W) void coop(const f4 *__restrict__ in, f4 *__restrict__ out, uint32_t n_segs, int K, int quads_per_unit,
                                               uint32_t *next)
{
    __shared__ uint32_t seg_lds[2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t it = 0;; it++) {
        if (threadIdx.x == 0) seg_lds[it & 1] = atomicAdd(next, 1u);
        __syncthreads();
        const uint32_t s = seg_lds[it & 1];
        if (s >= n_segs) return;
        for (int k = 0; k < K; k++) {
            const uint64_t unit = ((uint64_t)s * K + k) * W + w;
            const f4 *src = in + unit * quads_per_unit;
            f4 *dst = out + unit * quads_per_unit;
            for (int q = lane; q < quads_per_unit; q += 64 * 8) {
                f4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(src + q + 64 * u);
#pragma unroll
                for (int u = 0; u < 8; u++) __builtin_nontemporal_store(v[u], dst + q + 64 * u);
            }
        }
    }
}

template <int E>
__global__ __launch_bounds__(256) void oneshot(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * E;
    f4 v[E];
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = __builtin_nontemporal_load(in + i + e);
#pragma unroll
    for (int e = 0; e < E; e++) __builtin_nontemporal_store(v[e], out + i + e);
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
    if (argc < 2)
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
            cases.push_back({ "copy own st=sc1", wpc, B, 16, 0, 0, 1, 1, 2 });
            cases.push_back({ "copy own st=sc0sc1", wpc, B, 16, 0, 0, 1, 1, 3 });
            cases.push_back({ "copy own st=sc1nt", wpc, B, 16, 0, 0, 1, 1, 4 });
            cases.push_back({ "write own st=sc1", wpc, B, 16, 0, 0, 0, 1, 2 });
            cases.push_back({ "write own st=sc1nt", wpc, B, 16, 0, 0, 0, 1, 4 });
        }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto launch, double moved) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipMemsetAsync(ctr, 0, 4, 0);
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-40s %.3f ms  %.0f GB/s\n", name, best, moved / best / 1e6);
        fflush(stdout);
    };
    if (argc > 1 && argv[1][0] == 'w') {
        // how many streams, how fat: own-region copies with 1 .. 16 wavefronts per CU, one-shot copies, cooperative groups
        timeit("one-shot copy, 16 B per thread", [&] { hipLaunchKernelGGL(oneshot<1>, dim3((unsigned)(bytes / 16 / 256)), dim3(256), 0, 0, a, b); }, 2.0 * bytes);
        timeit("one-shot copy, 64 B per thread", [&] { hipLaunchKernelGGL(oneshot<4>, dim3((unsigned)(bytes / 64 / 256)), dim3(256), 0, 0, a, b); }, 2.0 * bytes);
        for (int wpc : { 1, 2, 4, 8, 16 })
            for (int B : { 8192, 32768 }) {
                const int K = 16, qpu = B / 16;
                const uint32_t n_segs = (uint32_t)(bytes / B / K);
                char name[96];
                snprintf(name, sizeof name, "copy own B=%d waves/CU=%d K=16", B, wpc);
                timeit(name, [&] { hipLaunchKernelGGL(streams<1>, dim3(256 * wpc), dim3(64), 0, 0, a, b, n_segs, K, qpu, 0, 0, ctr, 1, 1); }, 2.0 * n_segs * K * B);
            }
        for (int B : { 8192, 32768 })
            for (int K : { 4, 64 }) {
                const int qpu = B / 16;
                const uint32_t n8 = (uint32_t)(bytes / B / K / 8), n16 = (uint32_t)(bytes / B / K / 16);
                char name[96];
                snprintf(name, sizeof name, "coop 8 waves share a stream B=%d K=%d", B, K);
                timeit(name, [&] { hipLaunchKernelGGL(coop<8>, dim3(256), dim3(512), 0, 0, a, b, n8, K, qpu, ctr); }, 2.0 * n8 * K * 8 * B);
                snprintf(name, sizeof name, "coop 16 waves share a stream B=%d K=%d", B, K);
                timeit(name, [&] { hipLaunchKernelGGL(coop<16>, dim3(256), dim3(1024), 0, 0, a, b, n16, K, qpu, ctr); }, 2.0 * n16 * K * 16 * B);
                snprintf(name, sizeof name, "coop 2 x 8 waves per CU B=%d K=%d", B, K);
                timeit(name, [&] { hipLaunchKernelGGL(coop<8>, dim3(512), dim3(512), 0, 0, a, b, n8, K, qpu, ctr); }, 2.0 * n8 * K * 8 * B);
            }
        return 0;
    }
    for (const Case &c : cases) {
        const int qpu = c.B / 16;
        const uint64_t n_units = bytes / c.B;
        const uint32_t n_segs = (uint32_t)(n_units / c.K);
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipMemsetAsync(ctr, 0, 4, 0);
            hipEventRecord(e0);
#define LAUNCH(F) hipLaunchKernelGGL(streams<F>, dim3(256 * c.waves_per_cu), dim3(64), 0, 0, a, b, n_segs, c.K, qpu, c.placement, c.pad, ctr, c.rd, c.wr)
            switch (c.nt) { case 0: LAUNCH(0); break; case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); }
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
