// Micro-benchmark (round 6): does the power-of-two row stride of the FLAC planes cost bandwidth?  The restore kernel's access pattern
// with the compute stripped (32 frames x 2 channels per wavefront, 64-sample steps: 64 int16 row pieces of 128 B in, 32 interleaved
// row pieces of 512 B out, nontemporal) with the channel rows `in_stride` int16 elements apart (4096 = 8 KB in the C4 batch) and the
// frames' PCM rows `out_stride` ints apart (8192 = 32 KB).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_rowpad.hip -o tools/ubench_rowpad.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k16(const short *__restrict__ in, int *__restrict__ out, size_t in_stride, size_t out_stride)
{
    constexpr int ROWS = 32, T = 64, PIECES = T / 8, LOADS = ROWS * 2 * PIECES / 64, STORES = ROWS * 2 * (T / 4) / 64;
    __shared__ __attribute__((aligned(16))) int tile[ROWS * 2 * T];
    const int lane = threadIdx.x;
    const size_t f0 = (size_t)blockIdx.x * ROWS;
    i4 nxt[LOADS];
    auto load = [&](int t0) {
        for (int i = 0; i < LOADS; i++) {
            const int e = lane + 64 * i, rc = e / PIECES, p = e % PIECES;
            nxt[i] = __builtin_nontemporal_load((const i4 *)(in + (2 * f0 + rc) * in_stride + t0 + 8 * p));
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
            __builtin_nontemporal_store(v, (i4 *)(out + (f0 + r) * out_stride + 2 * t0 + 4 * q));
        }
        __builtin_amdgcn_wave_barrier();
    }
}
int main()
{
    const size_t frames = 4096 * 323 / 64 * 64;          // the C4 batch
    const size_t in_max = 4096 + 1024, out_max = 8192 + 1024;
    short *in; int *out;
    if (hipMalloc(&in, frames * 2 * in_max * 2) != hipSuccess || hipMalloc(&out, frames * out_max * 4) != hipSuccess) return 1;
    hipMemset(in, 0, frames * 2 * in_max * 2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const struct { const char *name; size_t ip, op; } cases[] = {
        { "rows 8 KB / 32 KB apart (the C4 planes)", 0, 0 }, { "in rows + 64 B", 32, 0 }, { "in rows + 128 B", 64, 0 }, { "in rows + 256 B", 128, 0 },
        { "in rows + 1 KB", 512, 0 }, { "out rows + 128 B", 0, 32 }, { "out rows + 512 B", 0, 128 }, { "out rows + 2 KB", 0, 512 },
        { "in + 128 B, out + 512 B", 64, 128 }, { "in + 256 B, out + 2 KB", 128, 512 }, { "rows 8 KB / 32 KB apart (again)", 0, 0 } };
    for (const auto &c : cases) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k16, dim3(frames / 32), dim3(64), 0, 0, in, out, 4096 + c.ip, 8192 + c.op);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-44s %.3f ms  %.2f TB/s of 6 B per sample\n", c.name, best, 1.5 * frames * 8192 * 4 / best / 1e9);
    }
    return 0;
}
