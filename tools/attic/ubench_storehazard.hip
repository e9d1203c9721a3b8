// tools/ubench_storehazard.hip -- does a 16-byte buffer store with a REGISTER scalar offset read its data registers before
// the next vector instruction of the wavefront overwrites them?  (LLVM's hazard recogniser spaces such a write from the
// store only when the scalar offset is NOT a register: GCNHazardRecognizer::createsVALUHazard.)
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_storehazard.hip -o tools/ubench_storehazard.bin && tools/ubench_storehazard.bin
// Every lane stores {x, x+1, x+2, x+3} and the instruction after the store writes a marker into the first / all data
// registers; the host counts markers that reached memory.  Variants: scalar offset in a register or the literal 0, with 0,
// 1 or 2 wait states between the store and the overwrite.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MARK 0xdead0000u

template <int VARIANT>
__global__ __launch_bounds__(64) void k(uint32_t *out, int rounds, uint32_t rows_per_wave)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t wave = blockIdx.x;
    // the FLAC pattern: 4 rows x 256 bytes per instruction, rows 32 KB apart, the position inside the row in the scalar offset
    uint32_t *base = out + wave * rows_per_wave * 8192ull;
    const uint64_t a = (uint64_t)(uintptr_t)base;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)(((uint64_t)hi << 32) | lo), 0, 0xffffffff, 0x00020000);
    for (int it = 0; it < rounds; it++) {
        for (uint32_t i = 0; i < rows_per_wave / 4; i++) {
            const uint32_t row = 4 * i + lane / 16;
            const uint32_t voff = row * 32768u + (lane % 16) * 16u;
            const int soff = __builtin_amdgcn_readfirstlane(it * 256);
            const uint32_t x = (uint32_t)(wave * 131 + row * 17 + it) & 0xffffu;
            if (VARIANT == 0)
                asm volatile("v_mov_b32 v10, %0\n v_add_u32 v11, 1, %0\n v_add_u32 v12, 2, %0\n v_add_u32 v13, 3, %0\n s_nop 4\n"
                             "buffer_store_dwordx4 v[10:13], %1, %2, %3 offen\n"
                             "v_mov_b32 v10, %4\n v_mov_b32 v11, %4\n v_mov_b32 v12, %4\n v_mov_b32 v13, %4\n"
                             : : "v"(x), "v"(voff), "s"(r), "s"(soff), "v"(MARK) : "v10", "v11", "v12", "v13", "memory");
            else if (VARIANT == 1)
                asm volatile("v_mov_b32 v10, %0\n v_add_u32 v11, 1, %0\n v_add_u32 v12, 2, %0\n v_add_u32 v13, 3, %0\n s_nop 4\n"
                             "buffer_store_dwordx4 v[10:13], %1, %2, %3 offen\n s_nop 0\n"
                             "v_mov_b32 v10, %4\n v_mov_b32 v11, %4\n v_mov_b32 v12, %4\n v_mov_b32 v13, %4\n"
                             : : "v"(x), "v"(voff), "s"(r), "s"(soff), "v"(MARK) : "v10", "v11", "v12", "v13", "memory");
            else if (VARIANT == 2) {
                const uint32_t v2 = voff + (uint32_t)soff;
                asm volatile("v_mov_b32 v10, %0\n v_add_u32 v11, 1, %0\n v_add_u32 v12, 2, %0\n v_add_u32 v13, 3, %0\n s_nop 4\n"
                             "buffer_store_dwordx4 v[10:13], %1, %2, 0 offen\n"
                             "v_mov_b32 v10, %3\n v_mov_b32 v11, %3\n v_mov_b32 v12, %3\n v_mov_b32 v13, %3\n"
                             : : "v"(x), "v"(v2), "s"(r), "v"(MARK) : "v10", "v11", "v12", "v13", "memory");
            } else if (VARIANT == 3) {
                const uint32_t v2 = voff + (uint32_t)soff;
                asm volatile("v_mov_b32 v10, %0\n v_add_u32 v11, 1, %0\n v_add_u32 v12, 2, %0\n v_add_u32 v13, 3, %0\n s_nop 4\n"
                             "buffer_store_dwordx4 v[10:13], %1, %2, 0 offen\n s_nop 0\n"
                             "v_mov_b32 v10, %3\n v_mov_b32 v11, %3\n v_mov_b32 v12, %3\n v_mov_b32 v13, %3\n"
                             : : "v"(x), "v"(v2), "s"(r), "v"(MARK) : "v10", "v11", "v12", "v13", "memory");
            } else {                                           // an LDS read landing in the data registers instead of a vector write
                __shared__ uint32_t l[256];
                l[lane] = MARK; l[lane + 64] = MARK; l[lane + 128] = MARK; l[lane + 192] = MARK;
                const uint32_t la = (uint32_t)(uintptr_t)(l) + lane * 16u;
                asm volatile("v_mov_b32 v10, %0\n v_add_u32 v11, 1, %0\n v_add_u32 v12, 2, %0\n v_add_u32 v13, 3, %0\n s_nop 4\n s_waitcnt lgkmcnt(0)\n"
                             "buffer_store_dwordx4 v[10:13], %1, %2, %3 offen\n"
                             "ds_read_b128 v[10:13], %4\n s_waitcnt lgkmcnt(0)\n"
                             : : "v"(x), "v"(voff), "s"(r), "s"(soff), "v"(la) : "v10", "v11", "v12", "v13", "memory");
            }
        }
    }
}

__global__ void count(const uint32_t *d, size_t words, unsigned long long *res)
{
    unsigned long long marks = 0, written = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
        marks += d[i] == MARK;
        written += d[i] != 0;
    }
    atomicAdd(res, marks);
    atomicAdd(res + 1, written);
}

template <int V> static void run(const char *what, uint32_t *d, size_t words, int waves, int rounds)
{
    unsigned long long *res, h[2] = { 0, 0 };
    (void)hipMalloc(&res, 16);
    (void)hipMemset(res, 0, 16);
    (void)hipMemset(d, 0, words * 4);
    hipLaunchKernelGGL(k<V>, dim3(waves), dim3(64), 0, 0, d, rounds, 32u);
    hipLaunchKernelGGL(count, dim3(4096), dim3(256), 0, 0, d, words, res);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, res, 16, hipMemcpyDeviceToHost);
    printf("%-70s markers in memory: %llu of %llu stored words\n", what, h[0], h[1]);
    (void)hipFree(res);
}

int main()
{
    const int waves = 8192, rounds = 32;                       // 32 rows x 32 KB per wave: 8 GB
    const size_t words = (size_t)waves * 32 * 8192;
    uint32_t *d;
    if (hipMalloc(&d, words * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<0>("register scalar offset, overwrite in the next instruction", d, words, waves, rounds);
    run<1>("register scalar offset, s_nop 0 between", d, words, waves, rounds);
    run<2>("literal 0 scalar offset, overwrite in the next instruction", d, words, waves, rounds);
    run<3>("literal 0 scalar offset, s_nop 0 between", d, words, waves, rounds);
    run<4>("register scalar offset, ds_read_b128 into the data registers next", d, words, waves, rounds);
    return 0;
}
