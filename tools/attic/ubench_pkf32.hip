// ubench_pkf32.hip -- issue cost of packed fp32 (v_pk_mul_f32 / v_pk_add_f32) against scalar fp32 (development aid).
// 8 independent accumulator pairs per lane; one wavefront per SIMD and four.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(const float *in, float *out, int iters)
{
    const int lane = threadIdx.x;
    f2 a[8], b[8], c[8];
    for (int i = 0; i < 8; i++) { a[i] = f2{ in[(lane + i) & 63], in[(lane + 2 * i) & 63] }; b[i] = f2{ in[(lane * 3 + i) & 63], 1.0001f }; c[i] = f2{ 0.5f, in[i] }; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) { a[i] = a[i] * b[i]; a[i] = a[i] + c[i]; }                       // packed: v_pk_mul_f32, v_pk_add_f32
            if (OP == 1) { a[i].x = a[i].x * b[i].x; a[i].y = a[i].y * b[i].y; a[i].x = a[i].x + c[i].x; a[i].y = a[i].y + c[i].y; }   // scalar x4
            if (OP == 2) { a[i] = a[i] * b[i].x; a[i] = a[i] + c[i]; }                     // packed, one factor broadcast
        }
    }
    f2 acc = { 0, 0 };
    for (int i = 0; i < 8; i++) acc += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
}
template <int OP> void run(const char *name, const float *in, float *out)
{
    const int iters = 4000;
    for (int threads : { 256, 1024 }) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, in, out, iters);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, in, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop_pairs = (double)iters * 8 * (threads / 256);       // (mul + add) of two floats, per lane, per SIMD
        printf("%-34s %4d threads/block: %.2f ns per (2 mul + 2 add) per wavefront per SIMD = %.1f cycles at 2.4 GHz\n", name, threads,
               ms * 1e6 / flop_pairs, ms * 1e6 / flop_pairs * 2.4);
    }
}
int main()
{
    float *in, *out;
    (void)hipMalloc(&in, 256); (void)hipMalloc(&out, 1 << 22);
    std::vector<float> h(64);
    for (int i = 0; i < 64; i++) h[i] = 1.0f + i * 1e-3f;
    (void)hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice);
    run<0>("v_pk_mul_f32 + v_pk_add_f32", in, out);
    run<1>("2 v_mul_f32 + 2 v_add_f32", in, out);
    run<2>("pk with a broadcast factor", in, out);
    return 0;
}
