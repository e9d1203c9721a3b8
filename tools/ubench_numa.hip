// Micro-benchmark (round 6): does the NUMA node a page-locked buffer lives on decide the PCIe rate?  For every node of the box:
// bind the allocation (set_mempolicy MPOL_BIND + hipHostMallocNumaUser), then time 8 x 64 MB device -> host and host -> device.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench_numa.hip -o tools/ubench_numa.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <sys/syscall.h>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    int nodes = 0;
    if (DIR *d = opendir("/sys/devices/system/node")) {
        while (dirent *e = readdir(d)) if (!strncmp(e->d_name, "node", 4) && e->d_name[4] >= '0' && e->d_name[4] <= '9') nodes++;
        closedir(d);
    }
    printf("NUMA nodes: %d\n", nodes);
    const size_t MB = 1 << 20, piece = 64 * MB, n = 8;
    char *dev;
    CK(hipMalloc((void **)&dev, piece * n));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int node = -1; node < (nodes ? nodes : 1); node++) {
        unsigned long mask[16] = {};
        if (node >= 0) {
            mask[node / 64] = 1ul << (node % 64);
            if (syscall(SYS_set_mempolicy, 2 /* MPOL_BIND */, mask, 1024) != 0) { perror("set_mempolicy"); continue; }
        }
        char *h;
        CK(hipHostMalloc((void **)&h, piece * n, node >= 0 ? hipHostMallocNumaUser : hipHostMallocPortable));
        memset(h, 1, piece * n);
        if (node >= 0) syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0);
        double best_d = 1e30, best_u = 1e30;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipDeviceSynchronize());
            double t0 = now_ms();
            for (size_t k = 0; k < n; k++) CK(hipMemcpyAsync(h + k * piece, dev + k * piece, piece, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            double t = now_ms() - t0;
            if (t < best_d) best_d = t;
            t0 = now_ms();
            for (size_t k = 0; k < n; k++) CK(hipMemcpyAsync(dev + k * piece, h + k * piece, piece, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            t = now_ms() - t0;
            if (t < best_u) best_u = t;
        }
        int cpu = sched_getcpu();
        printf("node %2d%s: device -> host %5.1f GB/s   host -> device %5.1f GB/s   (thread on cpu %d)\n", node, node < 0 ? " (default policy, hipHostMallocPortable)" : "",
               piece * n / best_d / 1e6, piece * n / best_u / 1e6, cpu);
        fflush(stdout);
        CK(hipHostFree(h));
    }
    return 0;
}
