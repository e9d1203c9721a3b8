// Micro-benchmark: the FLAC kernel's memory access pattern with the compute stripped.  A wavefront owns ROWS frames of
// 4096 samples x 2 channels (planar int32 in: channel c of frame f at (2f + c) * 4096; interleaved out: frame f at
// f * 8192) and walks them in steps of T samples: per step ROWS*2 reads of 4T bytes and ROWS writes of 8T bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
template <int ROWS, int T, int TILED = 0> __global__ __launch_bounds__(64) void k(const int *__restrict__ in, int *__restrict__ out)
{
    constexpr int PIECES = T / 4;                       // 16-byte pieces per row-channel chunk
    constexpr int LOADS = ROWS * 2 * PIECES / 64;       // 16-byte loads per lane per step
    __shared__ __attribute__((aligned(16))) int tile[ROWS * 2 * T];
    const int lane = threadIdx.x;
    const size_t f0 = (size_t)blockIdx.x * ROWS;
    i4 nxt[LOADS];
    auto load = [&](int t0) {
        for (int i = 0; i < LOADS; i++) {
            const int e = lane + 64 * i, rc = e / PIECES, p = e % PIECES;          // rc = row * 2 + channel
            // TILED: the residual plane is written tile by tile (a step's ROWS x 2 chunks are contiguous)
            nxt[i] = TILED ? *(const i4 *)(in + f0 * 8192 + (size_t)(t0 / T) * (ROWS * 2 * T) + 4 * e)
                           : *(const i4 *)(in + (2 * f0 + rc) * 4096 + t0 + 4 * p);
        }
    };
    load(0);
    for (int t0 = 0; t0 < 4096; t0 += T) {
        for (int i = 0; i < LOADS; i++) { const int e = lane + 64 * i; ((i4 *)tile)[e] = nxt[i]; }
        if (t0 + T < 4096) load(t0 + T);
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < LOADS; i++) asm volatile("" : "+v"(nxt[i].x), "+v"(nxt[i].y), "+v"(nxt[i].z), "+v"(nxt[i].w) : : "memory");
        for (int i = 0; i < LOADS; i++) {                                          // row r, 4 interleaved ints (2 samples x 2 ch)
            const int e = lane + 64 * i, r = e / (2 * PIECES), q = e % (2 * PIECES);
            const i4 v = ((const i4 *)tile)[e];
            if (TILED == 2) *(i4 *)(out + f0 * 8192 + (size_t)(t0 / T) * (ROWS * 2 * T) + 4 * e) = v;
            else *(i4 *)(out + (f0 + r) * 8192 + 2 * t0 + 4 * q) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// Variant: residuals as int16: per step ROWS*2 reads of 2T bytes (16-byte loads of 8 samples), widened into the
// same LDS tile, writes unchanged (8T bytes per row).
template <int ROWS, int T, int NT = 0> __global__ __launch_bounds__(64) void k16(const short *__restrict__ in, int *__restrict__ out)
{
    constexpr int PIECES = T / 8;                       // 16-byte pieces (8 samples) per row-channel chunk
    constexpr int LOADS = ROWS * 2 * PIECES / 64;
    constexpr int STORES = ROWS * 2 * (T / 4) / 64;
    __shared__ __attribute__((aligned(16))) int tile[ROWS * 2 * T];
    const int lane = threadIdx.x;
    const size_t f0 = (size_t)blockIdx.x * ROWS;
    i4 nxt[LOADS];
    auto load = [&](int t0) {
        for (int i = 0; i < LOADS; i++) {
            const int e = lane + 64 * i, rc = e / PIECES, p = e % PIECES;
            nxt[i] = *(const i4 *)(in + (2 * f0 + rc) * 4096 + t0 + 8 * p);
        }
    };
    load(0);
    for (int t0 = 0; t0 < 4096; t0 += T) {
        for (int i = 0; i < LOADS; i++) {
            const int e = lane + 64 * i;
            const i4 v = nxt[i];
            ((i4 *)tile)[2 * e] = i4{ (v.x << 16) >> 16, v.x >> 16, (v.y << 16) >> 16, v.y >> 16 };
            ((i4 *)tile)[2 * e + 1] = i4{ (v.z << 16) >> 16, v.z >> 16, (v.w << 16) >> 16, v.w >> 16 };
        }
        if (t0 + T < 4096) load(t0 + T);
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < LOADS; i++) asm volatile("" : "+v"(nxt[i].x), "+v"(nxt[i].y), "+v"(nxt[i].z), "+v"(nxt[i].w) : : "memory");
        for (int i = 0; i < STORES; i++) {
            const int e = lane + 64 * i, r = e / (2 * (T / 4)), q = e % (2 * (T / 4));
            const i4 v = ((const i4 *)tile)[e];
            *(i4 *)(out + (f0 + r) * 8192 + 2 * t0 + 4 * q) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
template <int ROWS, int T, int NT = 0> void run16(const int *in, int *out, size_t frames, const char *name)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k16<ROWS, T, NT>), dim3(frames / ROWS), dim3(64), 0, 0, (const short *)in, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-28s %.3f ms  (%.2f TB/s of 6 B per sample)\n", name, ms, 1.5 * frames * 8192 * 4 / ms / 1e9);
    }
}
template <int ROWS, int T, int TILED = 0> void run(const int *in, int *out, size_t frames, const char *name)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<ROWS, T, TILED>), dim3(frames / ROWS), dim3(64), 0, 0, in, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, 2.0 * frames * 8192 * 4 / ms / 1e9);
    }
}
int main(int argc, char **argv)
{
    const size_t frames = 4096 * 323 / 64 * 64;          // the C4 batch: 1.32M frames, 43 GB each way
    int *in, *out; hipMalloc(&in, frames * 8192 * 4); hipMalloc(&out, frames * 8192 * 4);
    hipMemset(in, 0, frames * 8192 * 4);
    if (argc > 1) {                                      // calibration of FETCH_SIZE / WRITE_SIZE on the restore kernel's pattern:
        // round 5's kernel: 32 frames x 2 channels per wavefront, 64-sample tiles, nontemporal 512-byte row pieces out
        run16<32, 64, 1>(in, out, frames, "int16 in: 64 x 128 B reads, 32 x 512 B nontemporal writes");   // known bytes: 2 B read + 4 B written per sample
        printf("known bytes per launch: read %zu write %zu\n", frames * 8192 * 2, frames * 8192 * 4);
        return 0;
    }
    run<64, 16>(in, out, frames, "64 rows x 64 B (kernel)");
    run<64, 32>(in, out, frames, "64 rows x 128 B");
    run<32, 32>(in, out, frames, "32 rows x 128 B");
    run<32, 64>(in, out, frames, "32 rows x 256 B");
    run<16, 64>(in, out, frames, "16 rows x 256 B");
    run<16, 128>(in, out, frames, "16 rows x 512 B");
    run<8, 256>(in, out, frames, "8 rows x 1 KB");
    run<64, 32, 1>(in, out, frames, "64 x 128 B, tiled reads");
    run<64, 16, 1>(in, out, frames, "64 x 64 B, tiled reads");
    run<64, 32, 2>(in, out, frames, "64 x 128 B, tiled both");
    run16<64, 32>(in, out, frames, "int16 in: 64 x 64 B reads");
    run16<64, 64>(in, out, frames, "int16 in: 64 x 128 B reads");
    run16<32, 64, 0>(in, out, frames, "int16 in: 32 frames, 128 B reads, 512 B writes (round 5's tile)");
    run16<32, 64, 1>(in, out, frames, "int16 in: 32 frames, 128 B reads, 512 B NONTEMPORAL writes (round 5's kernel)");
    run16<16, 128, 1>(in, out, frames, "int16 in: 16 frames, 256 B reads, 1 KB nontemporal writes");
    return 0;
}
