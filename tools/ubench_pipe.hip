// Micro-benchmark (round 6): the copy pipeline of afg_batch_decode's device stages in isolation -- chunks of `up` MB go host -> device on
// one stream, a token kernel runs behind each, `down` MB come back on a second stream behind an event -- for a few page-locked
// allocation flavours, chunk sizes and copy arrangements.  Prints per-direction and total GB/s.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_pipe.hip -o tools/ubench_pipe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void token(float *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Case { const char *name; unsigned flags; size_t up_mb, down_mb; int chunks; int mode; };   // mode 0: both, 1: up only, 2: down only, 3: down split in 4 pieces, 4: down on two streams alternating

extern "C" int ubench_pipe_run()
{
    const size_t MB = 1 << 20;
    std::vector<Case> cases = {
        { "portable   up 32 down 64 x8", hipHostMallocPortable, 32, 64, 8, 0 },
        { "portable   up only  32 x8", hipHostMallocPortable, 32, 64, 8, 1 },
        { "portable   down only 64 x8", hipHostMallocPortable, 32, 64, 8, 2 },
        { "default    up 32 down 64 x8", hipHostMallocDefault, 32, 64, 8, 0 },
        { "default    down only 64 x8", hipHostMallocDefault, 32, 64, 8, 2 },
        { "noncoherent up 32 down 64 x8", hipHostMallocNonCoherent, 32, 64, 8, 0 },
        { "noncoherent down only 64 x8", hipHostMallocNonCoherent, 32, 64, 8, 2 },
        { "numauser   down only 64 x8", hipHostMallocNumaUser, 32, 64, 8, 2 },
        { "portable   up 256 down 512 x8", hipHostMallocPortable, 256, 512, 8, 0 },
        { "portable   down only 512 x8", hipHostMallocPortable, 256, 512, 8, 2 },
        { "portable   up 32 down 64 x8, down in 4 pieces", hipHostMallocPortable, 32, 64, 8, 3 },
        { "portable   up 32 down 64 x8, down on two streams", hipHostMallocPortable, 32, 64, 8, 4 },
        { "portable   up 8 down 16 x32", hipHostMallocPortable, 8, 16, 32, 0 },
        { "up 32 down 64 x8, streams created per run (6th..)", hipHostMallocPortable, 32, 64, 8, 5 },
        { "up 32 down 64 x8, per run, down = high priority", hipHostMallocPortable, 32, 64, 8, 6 },
        { "up 32 down 64 x8, per run, down = low priority", hipHostMallocPortable, 32, 64, 8, 7 },
        { "up 32 down 64 x8, fork/join side kernel", hipHostMallocPortable, 32, 64, 8, 8 },
        { "up 32 down 64 x8, fork/join, down = high priority", hipHostMallocPortable, 32, 64, 8, 9 },
        { "up 32 down 64 x8, 3 copies up", hipHostMallocPortable, 32, 64, 8, 10 },
        { "portable   up 128 down 256 x2", hipHostMallocPortable, 128, 256, 2, 0 },
    };
    hipStream_t up, down, down2;
    CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&down2, hipStreamNonBlocking));
    for (const Case &c : cases) {
        const size_t ub = c.up_mb * MB, db = c.down_mb * MB;
        char *h_in, *h_out, *d_in, *d_out;
        CK(hipHostMalloc((void **)&h_in, ub * c.chunks, c.flags));
        CK(hipHostMalloc((void **)&h_out, db * c.chunks, c.flags));
        CK(hipMalloc((void **)&d_in, ub * c.chunks));
        CK(hipMalloc((void **)&d_out, db * c.chunks));
        memset(h_in, 1, ub * c.chunks);
        memset(h_out, 2, db * c.chunks);
        std::vector<hipEvent_t> ev(c.chunks);
        for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        double best = 1e30;
        static hipStream_t side = nullptr;
        static hipEvent_t fork_ev, join_ev;
        if (!side) { CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming)); }
        for (int rep = 0; rep < 6; rep++) {
            CK(hipDeviceSynchronize());
            hipStream_t up0 = up, down0 = down;
            hipStream_t up = up0, down = down0;
            if (c.mode >= 5 && c.mode <= 9) {
                int lo = 0, hi = 0;
                CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
                if (c.mode == 6 || c.mode == 9) CK(hipStreamCreateWithPriority(&down, hipStreamNonBlocking, hi));
                else if (c.mode == 7) CK(hipStreamCreateWithPriority(&down, hipStreamNonBlocking, lo));
                else CK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
            }
            const double t0 = now_ms();
            for (int k = 0; k < c.chunks; k++) {
                if (c.mode == 10) {
                    CK(hipMemcpyAsync(d_in + k * ub, h_in + k * ub, 65536, hipMemcpyHostToDevice, up));
                    CK(hipMemcpyAsync(d_in + k * ub + 65536, h_in + k * ub + 65536, 65536, hipMemcpyHostToDevice, up));
                    CK(hipMemcpyAsync(d_in + k * ub + 131072, h_in + k * ub + 131072, ub - 131072, hipMemcpyHostToDevice, up));
                } else
                if (c.mode != 2) CK(hipMemcpyAsync(d_in + k * ub, h_in + k * ub, ub, hipMemcpyHostToDevice, up));
                if (c.mode == 8 || c.mode == 9) {
                    CK(hipEventRecord(fork_ev, up));
                    CK(hipStreamWaitEvent(side, fork_ev, 0));
                    hipLaunchKernelGGL(token, dim3(64), dim3(256), 0, side, (float *)(d_out + k * db) + 32768, (size_t)16384);
                    CK(hipEventRecord(join_ev, side));
                }
                hipLaunchKernelGGL(token, dim3(64), dim3(256), 0, up, (float *)(d_out + k * db), (size_t)16384);
                if (c.mode == 8 || c.mode == 9) CK(hipStreamWaitEvent(up, join_ev, 0));
                CK(hipEventRecord(ev[k], up));
                hipStream_t ds = (c.mode == 4 && (k & 1)) ? down2 : down;
                CK(hipStreamWaitEvent(ds, ev[k], 0));
                if (c.mode == 1) continue;
                if (c.mode == 3) {
                    for (int q = 0; q < 4; q++) CK(hipMemcpyAsync(h_out + k * db + q * (db / 4), d_out + k * db + q * (db / 4), db / 4, hipMemcpyDeviceToHost, ds));
                } else CK(hipMemcpyAsync(h_out + k * db, d_out + k * db, db, hipMemcpyDeviceToHost, ds));
            }
            CK(hipStreamSynchronize(up));
            CK(hipStreamSynchronize(down));
            CK(hipStreamSynchronize(down2));
            const double t = now_ms() - t0;
            if (t < best) best = t;
            if (c.mode >= 5 && c.mode <= 9) { CK(hipStreamDestroy(up)); CK(hipStreamDestroy(down)); }
        }
        const double upb = c.mode == 2 ? 0 : (double)ub * c.chunks, dnb = c.mode == 1 ? 0 : (double)db * c.chunks;
        printf("%-52s %7.2f ms   up %5.1f GB/s  down %5.1f GB/s  total %5.1f GB/s\n", c.name, best, upb / best / 1e6, dnb / best / 1e6, (upb + dnb) / best / 1e6);
        fflush(stdout);
        for (auto &e : ev) CK(hipEventDestroy(e));
        CK(hipHostFree(h_in)); CK(hipHostFree(h_out)); CK(hipFree(d_in)); CK(hipFree(d_out));
    }
    return 0;
}

#ifndef UBENCH_PIPE_LIB
int main() { return ubench_pipe_run(); }
#endif
