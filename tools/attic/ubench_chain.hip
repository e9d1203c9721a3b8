// ubench_chain.hip -- what a serial float chain costs on one wavefront (development aid).
// t = x + m; m = t * c   (the CELT de-emphasis recurrence, dopus.d:3695-3701): two dependent VALU operations per sample.
// Prints cycles per sample (s_memtime) for 1 wave per block, with 2 or 64 active lanes, and for 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void chain_kernel(const float *in, float *out, unsigned long long *cycles, int iters, int active)
{
    const int lane = threadIdx.x & 63;
    float x[16];
    for (int k = 0; k < 16; k++) x[k] = in[(lane + k) & 63];
    float m = in[lane];
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (lane < active) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float t = x[k] + m;
                m = t * 0.85000610f;
                x[k] = t;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = m;
    for (int k = 0; k < 16; k++) acc += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    float *in, *out;
    unsigned long long *cyc;
    hipMalloc(&in, 64 * 4); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8 * 65536);
    std::vector<float> h(64, 0.001f);
    hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int active : { 2, 64 }) {
        for (int threads : { 64, 256, 512, 1024 }) {
            for (int blocks : { 1, 256, 1024 }) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters, active);
                hipEventRecord(e0);
                hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters, active);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                const double samples = (double)iters * 16;
                printf("active %2d lanes, %4d threads/block, %4d blocks: %.2f counter ticks/sample, %.2f ns/sample (wall)\n", active, threads, blocks,
                       c / samples, ms * 1e6 / samples);
            }
        }
    }
    return 0;
}
