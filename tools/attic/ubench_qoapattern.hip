// Micro-benchmark: the QOA decode kernel's store pattern alone.  A wavefront owns 32 frame rows (40960 bytes apart:
// 5120 stereo float samples) and walks them in steps of PIECE bytes per row: 32 scattered PIECE-byte runs per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PIECE> __global__ __launch_bounds__(64) void k(float *__restrict__ out)
{
    constexpr int PAIRS = PIECE / 8;                       // float2 per row per step
    const int lane = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * 32;
    for (int t = 0; t < 40960; t += PIECE) {
        for (int it = 0; it < 32 * PAIRS / 64; it++) {
            const int e = lane + 64 * it, r = e / PAIRS, q = e % PAIRS;
            *(f2 *)((char *)out + (row0 + r) * 40960 + t + 8 * q) = f2{ (float)e, (float)t };
        }
    }
}
template <int PIECE> void run(float *out, size_t rows)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<PIECE>, dim3(rows / 32), dim3(64), 0, 0, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("32 rows x %4d B per step: %.3f ms  %.2f TB/s\n", PIECE, ms, rows * 40960.0 / ms / 1e9);
    }
}
int main()
{
    const size_t rows = 4480 * 32;                         // the bench batch: 143360 frames, 5.9 GB of float PCM
    float *out; hipMalloc(&out, rows * 40960);
    run<160>(out, rows); run<320>(out, rows); run<640>(out, rows); run<1280>(out, rows);
    return 0;
}
