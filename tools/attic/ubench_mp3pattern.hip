// Micro-benchmark: the MP3 kernel's memory access pattern with the compute stripped.  One wavefront per segment of
// 48 granules (+2 warm-up granules read only), per granule 9 x 8-byte loads per lane at a 72-byte lane stride
// (4608 contiguous bytes per wavefront) and 5 x 16-byte coalesced stores, loads prefetched one granule ahead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(64) void k(const float *__restrict__ in, float *__restrict__ out, int seg, int warm)
{
    __shared__ __attribute__((aligned(16))) float H[1152];
    const int lane = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * seg;
    const size_t gfirst = g0 >= (size_t)warm ? g0 - warm : 0;
    f2 pre[9];
    const f2 *src = (const f2 *)(in + gfirst * 1152) + lane * 9;
    for (int q = 0; q < 9; q++) pre[q] = src[q];
    for (size_t g = gfirst; g < g0 + seg; g++) {
        f2 x[9];
        for (int q = 0; q < 9; q++) x[q] = pre[q];
        if (g + 1 < g0 + seg) {
            const f2 *s2 = (const f2 *)(in + (g + 1) * 1152) + lane * 9;
            for (int q = 0; q < 9; q++) pre[q] = s2[q];
        }
        for (int q = 0; q < 9; q++) ((f2 *)H)[lane * 9 + q] = x[q];
        __builtin_amdgcn_wave_barrier();
        if (MODE == 1) for (int q = 0; q < 9; q++) asm volatile("" : "+v"(pre[q].x), "+v"(pre[q].y) : : "memory");
        if (g >= g0) {
            f4 *dst = (f4 *)(out + g * 1152);
            for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; if (idx < 288) dst[idx] = ((const f4 *)H)[idx]; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// Variant: G granules per iteration (G x 4608-byte read burst, then G x 4608-byte write burst)
template <int G> __global__ __launch_bounds__(64) void kg(const float *__restrict__ in, float *__restrict__ out, int seg)
{
    __shared__ __attribute__((aligned(16))) float H[1152 * G];
    const int lane = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * seg;
    f4 pre[5 * G];
    for (int q = 0; q < 5 * G; q++) { const int idx = lane + 64 * q; pre[q] = ((const f4 *)(in + g0 * 1152))[idx < 288 * G ? idx : 288 * G - 1]; }
    for (size_t g = g0; g < g0 + seg; g += G) {
        for (int q = 0; q < 5 * G; q++) { const int idx = lane + 64 * q; if (idx < 288 * G) ((f4 *)H)[idx] = pre[q]; }
        if (g + G < g0 + seg)
            for (int q = 0; q < 5 * G; q++) { const int idx = lane + 64 * q; pre[q] = ((const f4 *)(in + (g + G) * 1152))[idx < 288 * G ? idx : 288 * G - 1]; }
        __builtin_amdgcn_wave_barrier();
        for (int q = 0; q < 5 * G; q++) asm volatile("" : "+v"(pre[q].x), "+v"(pre[q].y), "+v"(pre[q].z), "+v"(pre[q].w) : : "memory");
        f4 *dst = (f4 *)(out + g * 1152);
        for (int q = 0; q < 5 * G; q++) { const int idx = lane + 64 * q; if (idx < 288 * G) dst[idx] = ((const f4 *)H)[idx]; }
        __builtin_amdgcn_wave_barrier();
    }
}

// Variant: non-temporal loads and / or stores (NT & 1: loads, NT & 2: stores)
template <int NT> __global__ __launch_bounds__(64) void knt(const float *__restrict__ in, float *__restrict__ out, int seg)
{
    __shared__ __attribute__((aligned(16))) float H[1152];
    const int lane = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * seg;
    f4 pre[5];
    auto ld = [&](const f4 *p) { return (NT & 1) ? __builtin_nontemporal_load(p) : *p; };
    for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; pre[q] = ld((const f4 *)(in + g0 * 1152) + (idx < 288 ? idx : 287)); }
    for (size_t g = g0; g < g0 + seg; g++) {
        for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; if (idx < 288) ((f4 *)H)[idx] = pre[q]; }
        if (g + 1 < g0 + seg)
            for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; pre[q] = ld((const f4 *)(in + (g + 1) * 1152) + (idx < 288 ? idx : 287)); }
        __builtin_amdgcn_wave_barrier();
        for (int q = 0; q < 5; q++) asm volatile("" : "+v"(pre[q].x), "+v"(pre[q].y), "+v"(pre[q].z), "+v"(pre[q].w) : : "memory");
        f4 *dst = (f4 *)(out + g * 1152);
        for (int q = 0; q < 5; q++) {
            const int idx = lane + 64 * q;
            if (idx < 288) { if (NT & 2) __builtin_nontemporal_store(((const f4 *)H)[idx], dst + idx); else dst[idx] = ((const f4 *)H)[idx]; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Variants: the read half / the write half of the pattern alone (coalesced form)
template <int WHICH> __global__ __launch_bounds__(64) void khalf(const float *__restrict__ in, float *__restrict__ out, int seg)
{
    const int lane = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * seg;
    f4 acc = { 0, 0, 0, 0 };
    for (size_t g = g0; g < g0 + seg; g++) {
        if (WHICH == 0) {
            const f4 *src = (const f4 *)(in + g * 1152);
            for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; const f4 v = src[idx < 288 ? idx : 287]; acc += v; }
        } else {
            f4 *dst = (f4 *)(out + g * 1152);
            for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; if (idx < 288) dst[idx] = acc; }
        }
    }
    if (WHICH == 0 && acc.x == 12345.0f) out[0] = acc.y;
}

// Variant: the same bytes fetched as 16-byte coalesced loads (5 per lane), parked in LDS, then read back in the
// kernel's (channel, subband) layout; stores unchanged.
__global__ __launch_bounds__(64) void kc(const float *__restrict__ in, float *__restrict__ out, int seg, int warm)
{
    __shared__ __attribute__((aligned(16))) float H[1152 + 1280];
    float *const X = H + 1152;
    const int lane = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * seg;
    const size_t gfirst = g0 >= (size_t)warm ? g0 - warm : 0;
    f4 pre[5];
    {
        const f4 *src = (const f4 *)(in + gfirst * 1152);
        for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; pre[q] = src[idx < 288 ? idx : 287]; }
    }
    float acc = 0;
    for (size_t g = gfirst; g < g0 + seg; g++) {
        for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; if (idx < 288) ((f4 *)X)[idx] = pre[q]; }
        if (g + 1 < g0 + seg) {
            const f4 *src = (const f4 *)(in + (g + 1) * 1152);
            for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; pre[q] = src[idx < 288 ? idx : 287]; }
        }
        __builtin_amdgcn_wave_barrier();
        f2 x[9];
        for (int q = 0; q < 9; q++) x[q] = ((const f2 *)X)[lane * 9 + q];
        for (int q = 0; q < 9; q++) ((f2 *)H)[lane * 9 + q] = x[q];
        __builtin_amdgcn_wave_barrier();
        for (int q = 0; q < 5; q++) asm volatile("" : "+v"(pre[q].x), "+v"(pre[q].y), "+v"(pre[q].z), "+v"(pre[q].w) : : "memory");
        if (g >= g0) {
            f4 *dst = (f4 *)(out + g * 1152);
            for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; if (idx < 288) dst[idx] = ((const f4 *)H)[idx]; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// Variant: does it matter WHERE the resident wavefronts read and write at a given moment?  IL & 1: granule g of segment s is read
// from block (g * nseg + s) instead of (s * seg + g) -- the wavefronts that run together read neighbouring 4.6 KB blocks;
// IL & 2: the same for the stores.  (Same bytes, same per-wavefront pattern; only the global address window differs.)
template <int IL> __global__ __launch_bounds__(64) void kil(const float *__restrict__ in, float *__restrict__ out, int seg, unsigned nseg)
{
    __shared__ __attribute__((aligned(16))) float H[1152];
    const int lane = threadIdx.x;
    const size_t s = blockIdx.x;
    auto rblk = [&](int g) -> size_t { return (IL & 1) ? (size_t)g * nseg + s : s * seg + g; };
    auto wblk = [&](int g) -> size_t { return (IL & 2) ? (size_t)g * nseg + s : s * seg + g; };
    f4 pre[5];
    for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; pre[q] = ((const f4 *)(in + rblk(0) * 1152))[idx < 288 ? idx : 287]; }
    for (int g = 0; g < seg; g++) {
        for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; if (idx < 288) ((f4 *)H)[idx] = pre[q]; }
        if (g + 1 < seg)
            for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; pre[q] = ((const f4 *)(in + rblk(g + 1) * 1152))[idx < 288 ? idx : 287]; }
        __builtin_amdgcn_wave_barrier();
        for (int q = 0; q < 5; q++) asm volatile("" : "+v"(pre[q].x), "+v"(pre[q].y), "+v"(pre[q].z), "+v"(pre[q].w) : : "memory");
        f4 *dst = (f4 *)(out + wblk(g) * 1152);
        for (int q = 0; q < 5; q++) { const int idx = lane + 64 * q; if (idx < 288) dst[idx] = ((const f4 *)H)[idx]; }
        __builtin_amdgcn_wave_barrier();
    }
}

int main(int argc, char **argv)
{
    if (argc > 1) {                                           // `./ubench_mp3pattern il`: only the layout experiment
        const int seg = 48; const unsigned nseg = 98304; const size_t ngr = (size_t)nseg * seg;
        float *in, *out; hipMalloc(&in, ngr * 1152 * 4); hipMalloc(&out, ngr * 1152 * 4);
        hipMemset(in, 0, ngr * 1152 * 4);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int il = 0; il < 4; il++) for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            if (il == 0) hipLaunchKernelGGL(kil<0>, dim3(nseg), dim3(64), 0, 0, in, out, seg, nseg);
            if (il == 1) hipLaunchKernelGGL(kil<1>, dim3(nseg), dim3(64), 0, 0, in, out, seg, nseg);
            if (il == 2) hipLaunchKernelGGL(kil<2>, dim3(nseg), dim3(64), 0, 0, in, out, seg, nseg);
            if (il == 3) hipLaunchKernelGGL(kil<3>, dim3(nseg), dim3(64), 0, 0, in, out, seg, nseg);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("interleaved reads=%d writes=%d: %.3f ms  %.2f TB/s\n", il & 1, (il >> 1) & 1, ms, 2.0 * ngr * 4608 / ms / 1e9);
        }
        return 0;
    }
    const int seg = 48; const size_t nseg = 98304, ngr = nseg * seg;      // 4.7M granule pairs = 21.7 GB each way
    float *in, *out; hipMalloc(&in, ngr * 1152 * 4); hipMalloc(&out, ngr * 1152 * 4);
    hipMemset(in, 0, ngr * 1152 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; mode++) for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nseg), dim3(64), 0, 0, in, out, seg, 2);
        else hipLaunchKernelGGL(k<1>, dim3(nseg), dim3(64), 0, 0, in, out, seg, 2);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("mode %d (settle=%d): %.3f ms  %.2f TB/s (algorithmic 2 x %.1f GB)\n", mode, mode, ms, 2.0 * ngr * 4608 / ms / 1e9, ngr * 4608 / 1e9);
    }
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kc, dim3(nseg), dim3(64), 0, 0, in, out, seg, 2);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("coalesced loads via LDS: %.3f ms  %.2f TB/s\n", ms, 2.0 * ngr * 4608 / ms / 1e9);
    }
    for (int w = 0; w < 2; w++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        if (w == 0) hipLaunchKernelGGL(khalf<0>, dim3(nseg), dim3(64), 0, 0, in, out, seg);
        else hipLaunchKernelGGL(khalf<1>, dim3(nseg), dim3(64), 0, 0, in, out, seg);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s only: %.3f ms  %.2f TB/s\n", w ? "write" : "read", ms, 1.0 * ngr * 4608 / ms / 1e9);
    }
    for (int G = 1; G <= 4; G *= 2) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        if (G == 1) hipLaunchKernelGGL(kg<1>, dim3(nseg), dim3(64), 0, 0, in, out, seg);
        if (G == 2) hipLaunchKernelGGL(kg<2>, dim3(nseg), dim3(64), 0, 0, in, out, seg);
        if (G == 4) hipLaunchKernelGGL(kg<4>, dim3(nseg), dim3(64), 0, 0, in, out, seg);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("G=%d granules per burst: %.3f ms  %.2f TB/s\n", G, ms, 2.0 * ngr * 4608 / ms / 1e9);
    }
    for (int nt = 0; nt < 4; nt++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        if (nt == 0) hipLaunchKernelGGL(knt<0>, dim3(nseg), dim3(64), 0, 0, in, out, seg);
        if (nt == 1) hipLaunchKernelGGL(knt<1>, dim3(nseg), dim3(64), 0, 0, in, out, seg);
        if (nt == 2) hipLaunchKernelGGL(knt<2>, dim3(nseg), dim3(64), 0, 0, in, out, seg);
        if (nt == 3) hipLaunchKernelGGL(knt<3>, dim3(nseg), dim3(64), 0, 0, in, out, seg);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("non-temporal loads=%d stores=%d: %.3f ms  %.2f TB/s\n", nt & 1, (nt >> 1) & 1, ms, 2.0 * ngr * 4608 / ms / 1e9);
    }
    for (int waves = 2; waves <= 32; waves *= 2) {                 // resident wavefronts per CU capped through dynamic LDS
        const size_t dyn = 160 * 1024 / waves - 4608 - 256;
        hipFuncSetAttribute((const void *)knt<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipEventRecord(a);
        hipLaunchKernelGGL(knt<3>, dim3(nseg), dim3(64), dyn, 0, in, out, seg);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("<= %2d wavefronts per CU: %.3f ms  %.2f TB/s\n", waves, ms, 2.0 * ngr * 4608 / ms / 1e9);
    }
    for (int seg2 = 6; seg2 <= 192; seg2 *= 2) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kc, dim3(ngr / seg2), dim3(64), 0, 0, in, out, seg2, 2);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("coalesced, seg %d: %.3f ms\n", seg2, ms);
    }
    return 0;
}
