// afg_api.hip -- status strings, device checks and small utilities of the C ABI.
#include "afg_common.h"

#include <atomic>
#include <cstdlib>
#include <string>

namespace afg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return AFG_ERR_NO_DEVICE;
    }
    int dev = 0;
    AFG_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    AFG_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; kernels are built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return AFG_ERR_NO_DEVICE;
    }
    return AFG_OK;
}

int device_slot(int *dev, const char *who)
{
    *dev = 0;
    AFG_HIP_CHECK(hipGetDevice(dev));
    if (*dev < 0 || *dev >= AFG_MAX_DEVICES) {
        set_error("%s: device index %d does not fit this library's per-device tables (AFG_MAX_DEVICES = %d)", who, *dev,
                  AFG_MAX_DEVICES);
        return AFG_ERR_INVALID;
    }
    return AFG_OK;
}

static std::atomic<int> g_numeric_mode{ -1 };                 // -1: not set by the caller, the environment decides

int numeric_mode()
{
    const int m = g_numeric_mode.load(std::memory_order_relaxed);
    if (m >= 0) return m;
    const char *e = getenv("AFG_NUMERIC");
    if (e && !strcmp(e, "exact")) return AFG_NUMERIC_EXACT;
    return AFG_NUMERIC_TOLERANCE;
}

static std::atomic<long> g_dev_option[kDevCount] = { { -1 }, { -1 }, { -1 }, { -1 }, { -1 }, { -1 }, { -1 }, { -1 }, { -1 }, { -1 }, { -1 }, { -1 } };
static const char *const k_dev_option_name[kDevCount] = { "celt_path", "celt_de_seq", "celt_de_duo", "celt_seg_recs", "celt_whole_frames",
                                                          "vorbis_single", "mp3_chunks", "mp3_float_upload", "vorbis_host_floor",
                                                          "flac_host_res32", "vorbis_seg_packets", "batch_groups" };

long dev_option(DevOption which) { return g_dev_option[which].load(std::memory_order_relaxed); }

int DeviceArray::upload(const void *host, size_t nbytes)
{
    release();
    if (nbytes == 0) return AFG_OK;
    hipError_t e = hipMalloc(&ptr, nbytes);
    if (e != hipSuccess) {
        ptr = nullptr;
        set_error("hipMalloc(%zu) failed: %s", nbytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? AFG_ERR_OOM : AFG_ERR_HIP;
    }
    bytes = nbytes;
    AFG_HIP_CHECK(hipMemcpy(ptr, host, nbytes, hipMemcpyHostToDevice));
    return AFG_OK;
}

void DeviceArray::release()
{
    if (ptr && owned) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
    owned = true;
}

}  // namespace afg

extern "C" {

int afg_abi_version(void) { return AFG_ABI_VERSION; }

const char *afg_status_string(int status)
{
    switch (status) {
    case AFG_OK: return "ok";
    case AFG_ERR_INVALID: return "invalid argument";
    case AFG_ERR_NO_DEVICE: return "no usable gfx950 device";
    case AFG_ERR_HIP: return "HIP runtime error";
    case AFG_ERR_OOM: return "out of memory";
    case AFG_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
    }
}

const char *afg_last_error(void) { return afg::g_err; }

int afg_get_numeric_mode(void) { return afg::numeric_mode(); }

int afg_dev_option(const char *name, int value)
{
    for (int i = 0; name && i < afg::kDevCount; i++)
        if (!strcmp(name, afg::k_dev_option_name[i])) {
            afg::g_dev_option[i].store(value < 0 ? -1 : value, std::memory_order_relaxed);
            return AFG_OK;
        }
    afg::set_error("afg_dev_option: no option named '%s'", name ? name : "(null)");
    return AFG_ERR_INVALID;
}

int afg_set_numeric_mode(int mode)
{
    if (mode != AFG_NUMERIC_EXACT && mode != AFG_NUMERIC_TOLERANCE && mode != AFG_NUMERIC_FROM_ENV) {
        afg::set_error("afg_set_numeric_mode: unknown mode %d", mode);
        return AFG_ERR_INVALID;
    }
    const int prev = afg::numeric_mode();
    afg::g_numeric_mode.store(mode, std::memory_order_relaxed);
    return prev;
}

int afg_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        afg::set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return e == hipErrorNoDevice ? 0 : AFG_ERR_HIP;
    }
    return n;
}

int afg_device_name(int device, char *buf, size_t buflen)
{
    if (!buf || buflen == 0) return AFG_ERR_INVALID;
    hipDeviceProp_t prop;
    AFG_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return AFG_OK;
}

int afg_device_malloc(void **d_ptr, size_t bytes)
{
    if (!d_ptr) return AFG_ERR_INVALID;
    *d_ptr = nullptr;
    if (int rc = afg::require_device()) return rc;
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        afg::set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? AFG_ERR_OOM : AFG_ERR_HIP;
    }
    return AFG_OK;
}

int afg_device_free(void *d_ptr)
{
    if (d_ptr) AFG_HIP_CHECK(hipFree(d_ptr));
    return AFG_OK;
}

int afg_memcpy_h2d(void *d_dst, const void *src, size_t bytes, void *hip_stream)
{
    AFG_HIP_CHECK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)hip_stream));
    return AFG_OK;
}

int afg_memcpy_d2h(void *dst, const void *d_src, size_t bytes, void *hip_stream)
{
    AFG_HIP_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    return AFG_OK;
}

namespace {
typedef float probe_f4 __attribute__((ext_vector_type(4)));
// one 16-byte element per thread, no grid-stride loop: the access pattern that reaches the highest copy
// rate measured on MI355X (tools/ubench_bw.hip: ~6.3 TB/s against ~4.6 TB/s for hipMemcpy device-to-device)
__global__ __launch_bounds__(256) void copy_probe_kernel(const probe_f4 *__restrict__ in, probe_f4 *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void lds_fill_kernel(uint32_t word, uint32_t *sink)
{
    extern __shared__ uint32_t lds_words[];
    for (uint32_t i = threadIdx.x; i < 160u * 1024u / 4u; i += 256) lds_words[i] = word;
    __syncthreads();
    if (sink && lds_words[(threadIdx.x * 97u) % (160u * 1024u / 4u)] != word) *sink = 1;     // keeps the stores alive
}
}  // namespace

int afg_lds_fill_probe_hip(uint32_t word, void *hip_stream)
{
    if (int rc = afg::require_device()) return rc;
    static uint32_t *sink = nullptr;                          // per process; never read back
    if (!sink) AFG_HIP_CHECK(hipMalloc(&sink, 256));
    AFG_HIP_CHECK(hipFuncSetAttribute((const void *)lds_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // one workgroup owns a whole CU's LDS while it runs: 16 x the CU count visits every CU with near certainty
    hipLaunchKernelGGL(lds_fill_kernel, dim3(256 * 16), dim3(256), 160 * 1024, (hipStream_t)hip_stream, word, sink);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}

int afg_copy_probe_hip(void *d_dst, const void *d_src, size_t bytes, void *hip_stream)
{
    if (!d_dst || !d_src || (bytes & 15) || (((uintptr_t)d_dst | (uintptr_t)d_src) & 15)) {
        afg::set_error("afg_copy_probe_hip: pointers and size must be 16-byte aligned");
        return AFG_ERR_INVALID;
    }
    if (int rc = afg::require_device()) return rc;
    const size_t n = bytes / 16, blocks = (n + 255) / 256;
    if (blocks == 0) return AFG_OK;
    if (blocks > 0x7fffffffull) {
        afg::set_error("afg_copy_probe_hip: at most 2^31 blocks of 4 KiB per call");
        return AFG_ERR_INVALID;
    }
    hipLaunchKernelGGL(copy_probe_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)hip_stream,
                       (const probe_f4 *)d_src, (probe_f4 *)d_dst, n);
    AFG_HIP_CHECK(hipGetLastError());
    return AFG_OK;
}

int afg_stream_synchronize(void *hip_stream)
{
    AFG_HIP_CHECK(hipStreamSynchronize((hipStream_t)hip_stream));
    return AFG_OK;
}

}  // extern "C"
