// Micro-benchmark: achievable HBM bandwidth of plain streaming kernels on this device (read, write, copy;
// normal vs nontemporal accesses; persistent grid vs one block per tile).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE, int NT> __global__ __launch_bounds__(256) void k(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 acc = { 0, 0, 0, 0 };
    for (; i < n; i += stride) {
        f4 v = { 1, 2, 3, 4 };
        if (MODE != 1) v = NT ? __builtin_nontemporal_load(in + i) : in[i];
        if (MODE == 0) acc += v;
        if (MODE != 0) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
    }
    if (MODE == 0 && acc.x == 12345.0f) out[0] = acc;
}
int main()
{
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    f4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    const char *modes[] = {"read", "write", "copy"};
    for (int mode = 0; mode < 3; mode++)
    for (int nt = 0; nt < 2; nt++)
    for (int blocks : { 256 * 4, 256 * 8, 256 * 16, 256 * 64, 0 }) {
        const int g = blocks ? blocks : (int)((n + 255) / 256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
#define L(M, T) hipLaunchKernelGGL((k<M, T>), dim3(g), dim3(256), 0, 0, a, b, n)
            if (mode == 0) { if (nt) L(0, 1); else L(0, 0); }
            if (mode == 1) { if (nt) L(1, 1); else L(1, 0); }
            if (mode == 2) { if (nt) L(2, 1); else L(2, 0); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double moved = (mode == 2 ? 2.0 : 1.0) * bytes;
        printf("%-5s nt=%d blocks=%-8d %.3f ms  %.0f GB/s\n", modes[mode], nt, g, best, moved / best / 1e6);
    }
    hipEventRecord(0);
    float ms = 0; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventRecord(e0); hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpy d2d %.3f ms  %.0f GB/s\n", ms, 2.0 * bytes / ms / 1e6);
    return 0;
}
