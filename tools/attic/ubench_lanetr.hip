// Development check: 8 x 8 transposes between a lane's 8 register slots and lane bits [5:3] (permlane32_swap, permlane16_swap,
// row_ror:8) and lane bits [2:0] (DPP quad_perm / row_ror:4,12) -- the exchanges that could replace LDS round trips between the
// butterfly passes of vorbis_wave_kernel.  Prints whether slot k of lane l ends up holding what the pass after it reads.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void swap32(unsigned &x, unsigned &y) { auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); x = r[0]; y = r[1]; }
__device__ __forceinline__ void swap16(unsigned &x, unsigned &y) { auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false); x = r[0]; y = r[1]; }
template <int CTRL> __device__ __forceinline__ unsigned dpp(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
// lane bit 3 <-> slot bit 0
__device__ __forceinline__ void swap8(unsigned &x, unsigned &y, bool hi)
{
    const unsigned send = hi ? x : y;
    const unsigned got = dpp<0x128>(send);                  // row_ror:8 = lane ^ 8 inside a row of 16
    x = hi ? got : x;
    y = hi ? y : got;
}
__global__ void hi_kernel(unsigned *out)
{
    const int lane = threadIdx.x;
    unsigned e[8];
    for (int s = 0; s < 8; s++) e[s] = 64 * s + lane;                          // after stages 0, 1: slot s of lane j holds point 64 s + j
    for (int k = 0; k < 4; k++) swap32(e[k], e[k + 4]);
    for (int k = 0; k < 8; k++) if (!(k & 2)) swap16(e[k], e[k + 2]);
    for (int k = 0; k < 8; k += 2) swap8(e[k], e[k + 1], (lane & 8) != 0);
    for (int k = 0; k < 8; k++) out[lane * 8 + k] = e[k];                      // wanted: 64 g + jp + 8 k, g = lane >> 3, jp = lane & 7
}
int main()
{
    unsigned *d, h[512];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(hi_kernel, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int k = 0; k < 8; k++) bad += h[l * 8 + k] != (unsigned)(64 * (l >> 3) + (l & 7) + 8 * k);
    printf("lane bits 5:3 <-> slot: %s (%d wrong)\n", bad ? "WRONG" : "ok", bad);
    if (bad) for (int l = 0; l < 64; l += 9) { printf("lane %2d:", l); for (int k = 0; k < 8; k++) printf(" %3u", h[l * 8 + k]); printf("\n"); }
    return 0;
}
