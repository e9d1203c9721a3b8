// ubench_imul2.hip -- issue cost of integer multiply instructions on gfx950, by inline assembly (the compiler cannot
// substitute or hoist anything): 8 independent chains per lane, N wavefronts per SIMD.  Prints SIMD cycles per
// wavefront-instruction (wall time x clock / instructions issued per SIMD) for 1, 2 and 4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x(0) x(1) x(2) x(3) x(4) x(5) x(6) x(7)

template <int OP>
__global__ void k(const int *in, int *out, int iters)
{
    int a0 = in[threadIdx.x & 63], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int b = in[(threadIdx.x * 3) & 63] | 1;
    long long w0 = a0, w1 = a1, w2 = a2, w3 = a3;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (OP == 1) asm volatile("v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n"
                                  "v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8\n"
                                  "v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n"
                                  "v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8\n"
                                  "v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n"
                                  "v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8\n"
                                  "v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n"
                                  "v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8\n"
                                  "v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n"
                                  "v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8\n"
                                  "v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n"
                                  "v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8\n"
                                  "v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n"
                                  "v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8\n"
                                  "v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n"
                                  "v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (OP == 2) asm volatile("v_mad_i32_i24 %0, %0, %8, %1\n v_mad_i32_i24 %1, %1, %8, %2\n v_mad_i32_i24 %2, %2, %8, %3\n v_mad_i32_i24 %3, %3, %8, %4\n"
                                  "v_mad_i32_i24 %4, %4, %8, %5\n v_mad_i32_i24 %5, %5, %8, %6\n v_mad_i32_i24 %6, %6, %8, %7\n v_mad_i32_i24 %7, %7, %8, %0\n"
                                  "v_mad_i32_i24 %0, %0, %8, %1\n v_mad_i32_i24 %1, %1, %8, %2\n v_mad_i32_i24 %2, %2, %8, %3\n v_mad_i32_i24 %3, %3, %8, %4\n"
                                  "v_mad_i32_i24 %4, %4, %8, %5\n v_mad_i32_i24 %5, %5, %8, %6\n v_mad_i32_i24 %6, %6, %8, %7\n v_mad_i32_i24 %7, %7, %8, %0\n"
                                  "v_mad_i32_i24 %0, %0, %8, %1\n v_mad_i32_i24 %1, %1, %8, %2\n v_mad_i32_i24 %2, %2, %8, %3\n v_mad_i32_i24 %3, %3, %8, %4\n"
                                  "v_mad_i32_i24 %4, %4, %8, %5\n v_mad_i32_i24 %5, %5, %8, %6\n v_mad_i32_i24 %6, %6, %8, %7\n v_mad_i32_i24 %7, %7, %8, %0\n"
                                  "v_mad_i32_i24 %0, %0, %8, %1\n v_mad_i32_i24 %1, %1, %8, %2\n v_mad_i32_i24 %2, %2, %8, %3\n v_mad_i32_i24 %3, %3, %8, %4\n"
                                  "v_mad_i32_i24 %4, %4, %8, %5\n v_mad_i32_i24 %5, %5, %8, %6\n v_mad_i32_i24 %6, %6, %8, %7\n v_mad_i32_i24 %7, %7, %8, %0\n"
                                  "v_mad_i32_i24 %0, %0, %8, %1\n v_mad_i32_i24 %1, %1, %8, %2\n v_mad_i32_i24 %2, %2, %8, %3\n v_mad_i32_i24 %3, %3, %8, %4\n"
                                  "v_mad_i32_i24 %4, %4, %8, %5\n v_mad_i32_i24 %5, %5, %8, %6\n v_mad_i32_i24 %6, %6, %8, %7\n v_mad_i32_i24 %7, %7, %8, %0\n"
                                  "v_mad_i32_i24 %0, %0, %8, %1\n v_mad_i32_i24 %1, %1, %8, %2\n v_mad_i32_i24 %2, %2, %8, %3\n v_mad_i32_i24 %3, %3, %8, %4\n"
                                  "v_mad_i32_i24 %4, %4, %8, %5\n v_mad_i32_i24 %5, %5, %8, %6\n v_mad_i32_i24 %6, %6, %8, %7\n v_mad_i32_i24 %7, %7, %8, %0\n"
                                  "v_mad_i32_i24 %0, %0, %8, %1\n v_mad_i32_i24 %1, %1, %8, %2\n v_mad_i32_i24 %2, %2, %8, %3\n v_mad_i32_i24 %3, %3, %8, %4\n"
                                  "v_mad_i32_i24 %4, %4, %8, %5\n v_mad_i32_i24 %5, %5, %8, %6\n v_mad_i32_i24 %6, %6, %8, %7\n v_mad_i32_i24 %7, %7, %8, %0\n"
                                  "v_mad_i32_i24 %0, %0, %8, %1\n v_mad_i32_i24 %1, %1, %8, %2\n v_mad_i32_i24 %2, %2, %8, %3\n v_mad_i32_i24 %3, %3, %8, %4\n"
                                  "v_mad_i32_i24 %4, %4, %8, %5\n v_mad_i32_i24 %5, %5, %8, %6\n v_mad_i32_i24 %6, %6, %8, %7\n v_mad_i32_i24 %7, %7, %8, %0\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                                  : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(a0), "v"(b) : "vcc");
        if (OP == 4) asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                  "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                                  "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                  "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                                  "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                  "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                                  "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                  "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                                  "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                  "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                                  "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                  "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                                  "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                  "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                                  "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                  "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (OP == 5) asm volatile("v_dot2_i32_i16 %0, %0, %8, %1\n v_dot2_i32_i16 %1, %1, %8, %2\n v_dot2_i32_i16 %2, %2, %8, %3\n v_dot2_i32_i16 %3, %3, %8, %4\n"
                                  "v_dot2_i32_i16 %4, %4, %8, %5\n v_dot2_i32_i16 %5, %5, %8, %6\n v_dot2_i32_i16 %6, %6, %8, %7\n v_dot2_i32_i16 %7, %7, %8, %0\n"
                                  "v_dot2_i32_i16 %0, %0, %8, %1\n v_dot2_i32_i16 %1, %1, %8, %2\n v_dot2_i32_i16 %2, %2, %8, %3\n v_dot2_i32_i16 %3, %3, %8, %4\n"
                                  "v_dot2_i32_i16 %4, %4, %8, %5\n v_dot2_i32_i16 %5, %5, %8, %6\n v_dot2_i32_i16 %6, %6, %8, %7\n v_dot2_i32_i16 %7, %7, %8, %0\n"
                                  "v_dot2_i32_i16 %0, %0, %8, %1\n v_dot2_i32_i16 %1, %1, %8, %2\n v_dot2_i32_i16 %2, %2, %8, %3\n v_dot2_i32_i16 %3, %3, %8, %4\n"
                                  "v_dot2_i32_i16 %4, %4, %8, %5\n v_dot2_i32_i16 %5, %5, %8, %6\n v_dot2_i32_i16 %6, %6, %8, %7\n v_dot2_i32_i16 %7, %7, %8, %0\n"
                                  "v_dot2_i32_i16 %0, %0, %8, %1\n v_dot2_i32_i16 %1, %1, %8, %2\n v_dot2_i32_i16 %2, %2, %8, %3\n v_dot2_i32_i16 %3, %3, %8, %4\n"
                                  "v_dot2_i32_i16 %4, %4, %8, %5\n v_dot2_i32_i16 %5, %5, %8, %6\n v_dot2_i32_i16 %6, %6, %8, %7\n v_dot2_i32_i16 %7, %7, %8, %0\n"
                                  "v_dot2_i32_i16 %0, %0, %8, %1\n v_dot2_i32_i16 %1, %1, %8, %2\n v_dot2_i32_i16 %2, %2, %8, %3\n v_dot2_i32_i16 %3, %3, %8, %4\n"
                                  "v_dot2_i32_i16 %4, %4, %8, %5\n v_dot2_i32_i16 %5, %5, %8, %6\n v_dot2_i32_i16 %6, %6, %8, %7\n v_dot2_i32_i16 %7, %7, %8, %0\n"
                                  "v_dot2_i32_i16 %0, %0, %8, %1\n v_dot2_i32_i16 %1, %1, %8, %2\n v_dot2_i32_i16 %2, %2, %8, %3\n v_dot2_i32_i16 %3, %3, %8, %4\n"
                                  "v_dot2_i32_i16 %4, %4, %8, %5\n v_dot2_i32_i16 %5, %5, %8, %6\n v_dot2_i32_i16 %6, %6, %8, %7\n v_dot2_i32_i16 %7, %7, %8, %0\n"
                                  "v_dot2_i32_i16 %0, %0, %8, %1\n v_dot2_i32_i16 %1, %1, %8, %2\n v_dot2_i32_i16 %2, %2, %8, %3\n v_dot2_i32_i16 %3, %3, %8, %4\n"
                                  "v_dot2_i32_i16 %4, %4, %8, %5\n v_dot2_i32_i16 %5, %5, %8, %6\n v_dot2_i32_i16 %6, %6, %8, %7\n v_dot2_i32_i16 %7, %7, %8, %0\n"
                                  "v_dot2_i32_i16 %0, %0, %8, %1\n v_dot2_i32_i16 %1, %1, %8, %2\n v_dot2_i32_i16 %2, %2, %8, %3\n v_dot2_i32_i16 %3, %3, %8, %4\n"
                                  "v_dot2_i32_i16 %4, %4, %8, %5\n v_dot2_i32_i16 %5, %5, %8, %6\n v_dot2_i32_i16 %6, %6, %8, %7\n v_dot2_i32_i16 %7, %7, %8, %0\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (OP == 6) asm volatile("v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8\n"
                                  "v_mul_hi_i32 %4, %4, %8\n v_mul_hi_i32 %5, %5, %8\n v_mul_hi_i32 %6, %6, %8\n v_mul_hi_i32 %7, %7, %8\n"
                                  "v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8\n"
                                  "v_mul_hi_i32 %4, %4, %8\n v_mul_hi_i32 %5, %5, %8\n v_mul_hi_i32 %6, %6, %8\n v_mul_hi_i32 %7, %7, %8\n"
                                  "v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8\n"
                                  "v_mul_hi_i32 %4, %4, %8\n v_mul_hi_i32 %5, %5, %8\n v_mul_hi_i32 %6, %6, %8\n v_mul_hi_i32 %7, %7, %8\n"
                                  "v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8\n"
                                  "v_mul_hi_i32 %4, %4, %8\n v_mul_hi_i32 %5, %5, %8\n v_mul_hi_i32 %6, %6, %8\n v_mul_hi_i32 %7, %7, %8\n"
                                  "v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8\n"
                                  "v_mul_hi_i32 %4, %4, %8\n v_mul_hi_i32 %5, %5, %8\n v_mul_hi_i32 %6, %6, %8\n v_mul_hi_i32 %7, %7, %8\n"
                                  "v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8\n"
                                  "v_mul_hi_i32 %4, %4, %8\n v_mul_hi_i32 %5, %5, %8\n v_mul_hi_i32 %6, %6, %8\n v_mul_hi_i32 %7, %7, %8\n"
                                  "v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8\n"
                                  "v_mul_hi_i32 %4, %4, %8\n v_mul_hi_i32 %5, %5, %8\n v_mul_hi_i32 %6, %6, %8\n v_mul_hi_i32 %7, %7, %8\n"
                                  "v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8\n"
                                  "v_mul_hi_i32 %4, %4, %8\n v_mul_hi_i32 %5, %5, %8\n v_mul_hi_i32 %6, %6, %8\n v_mul_hi_i32 %7, %7, %8\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (OP == 7) asm volatile("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                                  "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                                  "v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                                  "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                                  "v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                                  "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                                  "v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                                  "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                                  "v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                                  "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                                  "v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                                  "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                                  "v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                                  "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                                  "v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                                  "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (OP == 8) asm volatile("v_mad_i32_i16 %0, %0, %8, %1\n v_mad_i32_i16 %1, %1, %8, %2\n v_mad_i32_i16 %2, %2, %8, %3\n v_mad_i32_i16 %3, %3, %8, %4\n"
                                  "v_mad_i32_i16 %4, %4, %8, %5\n v_mad_i32_i16 %5, %5, %8, %6\n v_mad_i32_i16 %6, %6, %8, %7\n v_mad_i32_i16 %7, %7, %8, %0\n"
                                  "v_mad_i32_i16 %0, %0, %8, %1\n v_mad_i32_i16 %1, %1, %8, %2\n v_mad_i32_i16 %2, %2, %8, %3\n v_mad_i32_i16 %3, %3, %8, %4\n"
                                  "v_mad_i32_i16 %4, %4, %8, %5\n v_mad_i32_i16 %5, %5, %8, %6\n v_mad_i32_i16 %6, %6, %8, %7\n v_mad_i32_i16 %7, %7, %8, %0\n"
                                  "v_mad_i32_i16 %0, %0, %8, %1\n v_mad_i32_i16 %1, %1, %8, %2\n v_mad_i32_i16 %2, %2, %8, %3\n v_mad_i32_i16 %3, %3, %8, %4\n"
                                  "v_mad_i32_i16 %4, %4, %8, %5\n v_mad_i32_i16 %5, %5, %8, %6\n v_mad_i32_i16 %6, %6, %8, %7\n v_mad_i32_i16 %7, %7, %8, %0\n"
                                  "v_mad_i32_i16 %0, %0, %8, %1\n v_mad_i32_i16 %1, %1, %8, %2\n v_mad_i32_i16 %2, %2, %8, %3\n v_mad_i32_i16 %3, %3, %8, %4\n"
                                  "v_mad_i32_i16 %4, %4, %8, %5\n v_mad_i32_i16 %5, %5, %8, %6\n v_mad_i32_i16 %6, %6, %8, %7\n v_mad_i32_i16 %7, %7, %8, %0\n"
                                  "v_mad_i32_i16 %0, %0, %8, %1\n v_mad_i32_i16 %1, %1, %8, %2\n v_mad_i32_i16 %2, %2, %8, %3\n v_mad_i32_i16 %3, %3, %8, %4\n"
                                  "v_mad_i32_i16 %4, %4, %8, %5\n v_mad_i32_i16 %5, %5, %8, %6\n v_mad_i32_i16 %6, %6, %8, %7\n v_mad_i32_i16 %7, %7, %8, %0\n"
                                  "v_mad_i32_i16 %0, %0, %8, %1\n v_mad_i32_i16 %1, %1, %8, %2\n v_mad_i32_i16 %2, %2, %8, %3\n v_mad_i32_i16 %3, %3, %8, %4\n"
                                  "v_mad_i32_i16 %4, %4, %8, %5\n v_mad_i32_i16 %5, %5, %8, %6\n v_mad_i32_i16 %6, %6, %8, %7\n v_mad_i32_i16 %7, %7, %8, %0\n"
                                  "v_mad_i32_i16 %0, %0, %8, %1\n v_mad_i32_i16 %1, %1, %8, %2\n v_mad_i32_i16 %2, %2, %8, %3\n v_mad_i32_i16 %3, %3, %8, %4\n"
                                  "v_mad_i32_i16 %4, %4, %8, %5\n v_mad_i32_i16 %5, %5, %8, %6\n v_mad_i32_i16 %6, %6, %8, %7\n v_mad_i32_i16 %7, %7, %8, %0\n"
                                  "v_mad_i32_i16 %0, %0, %8, %1\n v_mad_i32_i16 %1, %1, %8, %2\n v_mad_i32_i16 %2, %2, %8, %3\n v_mad_i32_i16 %3, %3, %8, %4\n"
                                  "v_mad_i32_i16 %4, %4, %8, %5\n v_mad_i32_i16 %5, %5, %8, %6\n v_mad_i32_i16 %6, %6, %8, %7\n v_mad_i32_i16 %7, %7, %8, %0\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %8\n v_add_u32 %1, %1, %0\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %3, %3, %2\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_add_u32 %5, %5, %4\n v_mul_lo_u32 %6, %6, %8\n v_add_u32 %7, %7, %6\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_add_u32 %1, %1, %0\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %3, %3, %2\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_add_u32 %5, %5, %4\n v_mul_lo_u32 %6, %6, %8\n v_add_u32 %7, %7, %6\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_add_u32 %1, %1, %0\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %3, %3, %2\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_add_u32 %5, %5, %4\n v_mul_lo_u32 %6, %6, %8\n v_add_u32 %7, %7, %6\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_add_u32 %1, %1, %0\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %3, %3, %2\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_add_u32 %5, %5, %4\n v_mul_lo_u32 %6, %6, %8\n v_add_u32 %7, %7, %6\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_add_u32 %1, %1, %0\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %3, %3, %2\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_add_u32 %5, %5, %4\n v_mul_lo_u32 %6, %6, %8\n v_add_u32 %7, %7, %6\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_add_u32 %1, %1, %0\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %3, %3, %2\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_add_u32 %5, %5, %4\n v_mul_lo_u32 %6, %6, %8\n v_add_u32 %7, %7, %6\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_add_u32 %1, %1, %0\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %3, %3, %2\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_add_u32 %5, %5, %4\n v_mul_lo_u32 %6, %6, %8\n v_add_u32 %7, %7, %6\n"
                                  "v_mul_lo_u32 %0, %0, %8\n v_add_u32 %1, %1, %0\n v_mul_lo_u32 %2, %2, %8\n v_add_u32 %3, %3, %2\n"
                                  "v_mul_lo_u32 %4, %4, %8\n v_add_u32 %5, %5, %4\n v_mul_lo_u32 %6, %6, %8\n v_add_u32 %7, %7, %6\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (int)(w0 + w1 + w2 + w3);
}

template <int OP>
void run(const char *name, const int *in, int *out)
{
    const int iters = 4000;
    printf("%-28s", name);
    for (int threads : { 256, 512, 1024 }) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 64 * (threads / 256);
        printf("  %d wave/SIMD: %6.2f cyc", threads / 256, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
    printf("   (cycles at 2.4 GHz per wavefront-instruction per SIMD)\n");
}

int main()
{
    int *in, *out;
    hipMalloc(&in, 64 * 4); hipMalloc(&out, 1 << 22);
    std::vector<int> h(64);
    for (int i = 0; i < 64; i++) h[i] = i * 37 + 5;
    hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice);
    run<4>("v_add_u32", in, out);
    run<7>("v_fma_f32", in, out);
    run<0>("v_mul_lo_u32", in, out);
    run<9>("v_mul_lo_u32 + v_add_u32 /2", in, out);
    run<6>("v_mul_hi_i32", in, out);
    run<1>("v_mul_i32_i24", in, out);
    run<2>("v_mad_i32_i24", in, out);
    run<8>("v_mad_i32_i16", in, out);
    run<3>("v_mad_u64_u32", in, out);
    run<5>("v_dot2_i32_i16", in, out);
    return 0;
}
