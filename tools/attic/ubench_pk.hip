// Micro-benchmark: issue rate of packed fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) against the
// scalar forms, 8 independent chains per lane, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ void k(float *out, float a0)
{
    const float a = a0 + threadIdx.x * 1e-9f;
    float s[8];
    f2 p[8];
    for (int i = 0; i < 8; i++) { s[i] = 1.0f + i; p[i] = f2{ 1.0f + i, 2.0f + i }; }
    const f2 a2 = f2{ a, a };
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) s[j] = s[j] * a;                                    // v_mul_f32
            if (OP == 1) s[j] = s[j] + a;                                    // v_add_f32
            if (OP == 2) s[j] = __builtin_fmaf(s[j], a, a);                  // v_fma_f32
            if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(a2));
            if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(a2));
            if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j]) : "v"(a2));
            if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "+v"(p[j]) : "v"(a2));
        }
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += s[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 1024 * 4);
    const char *names[] = {"v_mul_f32", "v_add_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f32+mods"};
    for (int waves = 1; waves <= 4; waves *= 2)
    for (int op = 0; op < 7; op++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const dim3 g(256 * 8), b(256 * waves);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, d, 1.0000001f);
            if (op == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, d, 1.0000001f);
            if (op == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, d, 1.0000001f);
            if (op == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, d, 1.0000001f);
            if (op == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, d, 1.0000001f);
            if (op == 5) hipLaunchKernelGGL(k<5>, g, b, 0, 0, d, 1.0000001f);
            if (op == 6) hipLaunchKernelGGL(k<6>, g, b, 0, 0, d, 1.0000001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // 8 blocks per CU, each of `waves` waves per SIMD, N*8 instructions per wave
        const double instr_per_simd = 8.0 * waves * N * 8;
        printf("%-20s waves/SIMD=%d  %.3f ms  => %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n",
               names[op], waves, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
    return 0;
}
