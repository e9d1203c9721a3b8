// Micro-benchmark: issue cost of the integer multiply forms the FLAC recurrence can be built from.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
template <int OP> __global__ void k(int32_t *out, int32_t a0, int32_t b0)
{
    int32_t a = a0 + threadIdx.x, b = b0;
    int64_t x0 = 1, x1 = 2, x2 = 3, x3 = 4;
    int32_t y0 = 1, y1 = 2, y2 = 3, y3 = 4;
    float f0 = 1, f1 = 2, f2 = 3, f3 = 4;
    long long t0 = clock64();
    for (int i = 0; i < N; i++) {
        if (OP == 0) { x0 += (int64_t)a * y0; x1 += (int64_t)a * y1; x2 += (int64_t)a * y2; x3 += (int64_t)a * y3; y0 ^= (int32_t)x0; y1 ^= (int32_t)x1; y2 ^= (int32_t)x2; y3 ^= (int32_t)x3; }
        if (OP == 1) { y0 = y0 * a + b; y1 = y1 * a + b; y2 = y2 * a + b; y3 = y3 * a + b; }
        if (OP == 2) { y0 = __mul24(y0, a) + b; y1 = __mul24(y1, a) + b; y2 = __mul24(y2, a) + b; y3 = __mul24(y3, a) + b; }
        if (OP == 3) { f0 = fmaf(f0, (float)a, 1.0f); f1 = fmaf(f1, (float)a, 1.0f); f2 = fmaf(f2, (float)a, 1.0f); f3 = fmaf(f3, (float)a, 1.0f); }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int32_t)(x0 + x1 + x2 + x3) + y0 + y1 + y2 + y3 + (int32_t)(f0 + f1 + f2 + f3);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (int32_t)(t1 - t0);
}
int main()
{
    int32_t *d; hipMalloc(&d, 256 * 4 * 256 * 4 * 8);
    const char *names[] = {"mad_i64_i32 (+xor)", "mul_lo_u32+add", "mad_i32_i24", "fma_f32"};
    for (int waves = 1; waves <= 8; waves *= 2)
    for (int op = 0; op < 4; op++) {
        int32_t h = 0;
        dim3 g(256 * 4), b(64 * waves);   // `waves` waves per SIMD if one block per CU-SIMD... (256 thr = 1 wave per SIMD)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves > 1024 ? 1024 : 256 * waves), 0, 0, d, 3, 5);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves > 1024 ? 1024 : 256 * waves), 0, 0, d, 3, 5);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * waves > 1024 ? 1024 : 256 * waves), 0, 0, d, 3, 5);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256 * waves > 1024 ? 1024 : 256 * waves), 0, 0, d, 3, 5);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        int wps = (256 * waves > 1024 ? 1024 : 256 * waves) / 256;
        printf("%-22s waves/SIMD=%d  clock64 delta=%d  => %.2f ticks per (4 ops) iteration; wall %.3f ms => %.2f ns/iter/wave-set\n",
               names[op], wps, h, (double)h / N, ms, ms * 1e6 / N);
    }
    return 0;
}
