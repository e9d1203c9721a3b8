// tools/ubench_valu.hip -- issue rate of the integer VALU instructions the QOA / FLAC kernels lean on, gfx950.
// Each kernel runs 8 independent dependency chains of one instruction per lane, 4096 x 8 instructions per wavefront, 8
// wavefronts per SIMD; the result is cycles per wave64 instruction per SIMD (4 = full rate).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_valu.bin tools/ubench_valu.hip && tools/ubench_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define KERNEL(name, text)                                                                                          \
    __global__ __launch_bounds__(256) void k_##name(int *out, int a, int b)                                         \
    {                                                                                                               \
        int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < 4096; i++) {                                                                            \
            asm volatile(text "\n" : "+v"(x0) : "v"(a), "v"(b));                                                    \
            asm volatile(text "\n" : "+v"(x1) : "v"(a), "v"(b));                                                    \
            asm volatile(text "\n" : "+v"(x2) : "v"(a), "v"(b));                                                    \
            asm volatile(text "\n" : "+v"(x3) : "v"(a), "v"(b));                                                    \
            asm volatile(text "\n" : "+v"(x4) : "v"(a), "v"(b));                                                    \
            asm volatile(text "\n" : "+v"(x5) : "v"(a), "v"(b));                                                    \
            asm volatile(text "\n" : "+v"(x6) : "v"(a), "v"(b));                                                    \
            asm volatile(text "\n" : "+v"(x7) : "v"(a), "v"(b));                                                    \
        }                                                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                \
    }

KERNEL(add_u32, "v_add_u32 %0, %0, %1")
KERNEL(mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(mul_i32_i24, "v_mul_i32_i24 %0, %0, %1")
KERNEL(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL(mad_i32_i16, "v_mad_i32_i16 %0, %0, %1, %2")
KERNEL(dot2_i32_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
KERNEL(dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %0")
KERNEL(pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2")
KERNEL(fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL(xad_u32, "v_xad_u32 %0, %0, %1, %2")
KERNEL(med3_i32, "v_med3_i32 %0, %0, %1, %2")
KERNEL(ashr, "v_ashrrev_i32 %0, 1, %0")
KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(bfe_u32, "v_bfe_u32 %0, %0, 3, 7")
KERNEL(perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(mul_hi_i32_i24, "v_mul_hi_i32_i24 %0, %0, %1")

template <typename K> static void run(const char *name, K kernel, int *d_out, double clock_hz)
{
    const int blocks = 256 * 8;                       // 8 workgroups of 4 wavefronts per CU: 8 wavefronts per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 3, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = 8.0 * 4096 * 8;     // wavefronts per SIMD x iterations x chains
    printf("%-16s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD\n", name, ms, ms * 1e-3 * clock_hz / instr_per_simd);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double hz = p.clockRate * 1e3;
    printf("%s  %d CUs  %.0f MHz\n", p.name, p.multiProcessorCount, hz / 1e6);
    int *d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * sizeof(int));
#define RUN(name) run(#name, k_##name, d_out, hz)
    RUN(add_u32); RUN(mad_i32_i24); RUN(mul_i32_i24); RUN(mad_u32_u24); RUN(mul_lo_u32); RUN(mad_i32_i16); RUN(dot2_i32_i16);
    RUN(dot4_i32_i8); RUN(pk_mad_i16); RUN(fma_f32); RUN(add3_u32); RUN(xad_u32); RUN(med3_i32); RUN(ashr); RUN(lshl_add);
    RUN(lshl_or); RUN(and_or); RUN(bfe_u32); RUN(perm); RUN(cvt_f32_i32); RUN(mul_hi_i32_i24);
    return 0;
}
