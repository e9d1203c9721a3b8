// ubench_imul.hip -- issue cost of the integer multiply flavours the FLAC recurrence can use (development aid).
// 16 independent accumulators per lane, so that the figure is issue throughput, not latency; one wavefront per SIMD
// (256 threads per block, one block per CU) and four.  Prints wall nanoseconds per wavefront-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void imul_kernel(const int *in, int *out, int iters)
{
    const int lane = threadIdx.x;
    int a[16], b[16];
    long long w[16];
    for (int k = 0; k < 16; k++) { a[k] = in[(lane + k) & 63]; b[k] = in[(lane * 3 + k) & 63]; w[k] = a[k]; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (OP == 0) a[k] = (int)((unsigned)a[k] * (unsigned)b[k]) + 1;                                 // v_mul_lo_u32 + add
            if (OP == 1) a[k] = __mul24(a[k], b[k]) + 1;                                     // v_mad_i32_i24 when fused
            if (OP == 2) w[k] = (long long)a[k] * (long long)b[k] + w[k];                                     // v_mad_i64_i32
            if (OP == 3) a[k] = (int)__builtin_fmaf(__int_as_float(a[k]), 1.0001f, __int_as_float(b[k]));     // reference: one v_fma_f32 (+cvt)
            if (OP == 4) a[k] = a[k] + b[k];                                                                  // v_add
        }
    }
    int acc = 0;
    for (int k = 0; k < 16; k++) acc += a[k] + (int)w[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
void run(const char *name, const int *in, int *out, int per_iter_ops)
{
    const int iters = 4000;
    for (int threads : { 256, 1024 }) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(imul_kernel<OP>, dim3(256), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(imul_kernel<OP>, dim3(256), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double wave_ops_per_simd = (double)iters * 16 * (threads / 256);
        printf("%-28s %4d threads/block: %.2f ns per wavefront-op per SIMD (%.1f cycles at 2.4 GHz), %d instr/op expected\n", name, threads,
               ms * 1e6 / wave_ops_per_simd, ms * 1e6 / wave_ops_per_simd * 2.4, per_iter_ops);
    }
}

int main()
{
    int *in, *out;
    hipMalloc(&in, 64 * 4); hipMalloc(&out, 1 << 22);
    std::vector<int> h(64);
    for (int i = 0; i < 64; i++) h[i] = i * 37 + 5;
    hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice);
    run<0>("mul_lo_u32 + add", in, out, 2);
    run<1>("mul_i24 + add (mad_i32_i24)", in, out, 1);
    run<2>("mad_i64_i32", in, out, 1);
    run<3>("fma_f32", in, out, 1);
    run<4>("add_u32", in, out, 1);
    return 0;
}
