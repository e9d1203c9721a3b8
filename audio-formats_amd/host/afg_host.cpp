// afg_host.cpp -- host front-ends and the AudioStream-shaped surface of the C ABI.
//
// What stays on the host (bitstream / entropy parsing) for the formats whose
// front-ends exist so far:
//   FLAC  container + frame/subframe headers + Rice residuals   (reference drflac.d:680-1695,
//         :1887-2153; the prediction half of drflac.d:1235 is NOT done here: residuals and
//         subframe parameters become afg_flac_subframe / afg_flac_frame records)
//   QOA   file/frame headers only (reference qoa.d:413-486): the device reads the raw bytes
// and the outer surface mirroring AudioStream (stream.d:150-170 openFromMemory, :295-412
// getters, :429-637 readSamplesFloat): parse the file into transform-stage records, restore the
// samples on the device, serve interleaved floats.
#include "../csrc/afg_common.h"
#include "afg_flac_front.h"
#include "afg_mp3_front.h"
#include "afg_opus_front.h"
#include "afg_vorbis_front.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace afg_front;

namespace {

// error strings of the reference (internals.d:16-23; stream.d:1379)
const char *const kErrorUnknownFormat = "Cannot decode stream: unrecognized encoding.";
const char *const kErrorDecodingError = "Decoder encountered an error";
const char *const kErrorDecoderInitializationFailed = "Decoder initialization failed";
const char *const kErrorNotInitialized = "Stream not initialized";
// this library's own: the reference decodes such files, the device path does not (out of scope)
const char *const kErrorOpusMode = "Cannot decode stream: Opus SILK / hybrid packets are not supported (CELT-only).";

// ---------------------------------------------------------------------------------------------
// decoded files: one result plane for a whole batch
// ---------------------------------------------------------------------------------------------
struct Decoded {
    int status = AFG_OK;
    const char *message = nullptr;
    int format = AFG_FORMAT_UNKNOWN;
    int channels = 0;
    float samplerate = 0;
    int64_t frames = 0;                 // frames actually decoded
    int64_t declared_frames = AFG_UNKNOWN_LENGTH;
    size_t pcm_off = 0;                 // float offset of this file's interleaved PCM in the result plane
    bool in_mp3_plane = false;          // ... or in the batch's MP3 plane (staging layout, afg_batch_decode)
    bool in_opus_plane = false;         // ... or in the batch's Opus plane (decoded by the pipelined stage of afg_batch_decode)
};

struct Parsed {
    int format = AFG_FORMAT_UNKNOWN;
    FlacInfo fi;
    FlacRecords flac;
    QoaInfo qi;
    std::vector<afg_qoa_frame> qoa;
    afg_mp3::File mp3;
    afg_vorbis::File ogg;
    afg_opus::File opus;
    bool opus_mode = false;               // an Ogg Opus file with SILK / hybrid packets: reported, not decoded
    const float *mp3_coef() const { return mp3.ext_coef ? mp3.ext_coef : mp3.coef.data(); }
    const uint32_t *mp3_flags() const { return mp3.ext_flags ? mp3.ext_flags : mp3.flags.data(); }
};

// Vorbis: inverse coupling and floor curves on the device (default) or in the host parser (AFG_VORBIS_HOST_FLOOR=1)
// debugging aids, read ONCE when the library is loaded (set them before that): poisoned allocations (the test-suite runs
// with it: an output byte the library forgets to write shows up as NaN) and a trace of the staging pool
static const bool g_poison_alloc = std::getenv("AFG_POISON_ALLOC") != nullptr;
static const bool g_trace = std::getenv("AFG_TRACE") != nullptr;
static bool vorbis_floor_on_device() { return afg::dev_option(afg::kDevVorbisHostFloor) <= 0; }

// FLAC: residual rows that fit 16 bits are packed as int16 (default) or left as int32 (AFG_FLAC_HOST_RES32=1)
static bool flac_rows_int16() { return afg::dev_option(afg::kDevFlacHostRes32) <= 0; }

// Device buffers of the batch path are kept between calls (round 6).  A batch call used to hipMalloc its planes and hipFree them
// on the way out; the driver wipes freed video memory with the copy engines, and that wipe ran into the NEXT call's transfers:
// of back-to-back calls over the 2048-file FLAC batch every one but the first after a pause moved its 0.8 GB in 21.8 ms, the
// first in 15.4 (AFG_TRACE; the pipeline alone, tools/ubench_pipe.hip: 10.9 ms).  The pool keeps what it has seen, per device,
// bounded in count and bytes; afg_host_pool_trim() frees it.
class DevicePool {
public:
    int take(size_t bytes, void **out, size_t *cap_out)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        const size_t want = bytes ? bytes : 1;
        {
            std::lock_guard<std::mutex> lk(mu_);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].dev == dev && free_[i].cap >= want && free_[i].cap <= 2 * want + ((size_t)1 << 20) &&
                    (best == free_.size() || free_[i].cap < free_[best].cap)) best = i;
            if (best != free_.size()) {
                *out = free_[best].p; *cap_out = free_[best].cap;
                held_ -= free_[best].cap;
                free_.erase(free_.begin() + (long)best);
                return AFG_OK;
            }
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipErrorOutOfMemory && trim()) e = hipMalloc(&p, want);          // what the pool holds may be what is missing
        if (e != hipSuccess) { afg::set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return AFG_ERR_OOM; }
        *out = p; *cap_out = want;
        return AFG_OK;
    }
    void give_back(void *p, size_t cap)
    {
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (known && free_.size() < 128 && held_ + cap <= ((size_t)64 << 30)) {
                free_.push_back({ dev, p, cap });
                held_ += cap;
                return;
            }
        }
        (void)hipFree(p);
    }
    size_t trim()
    {
        std::vector<Buf> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            drop.swap(free_);
            held_ = 0;
        }
        size_t bytes = 0;
        for (auto &b : drop) { (void)hipFree(b.p); bytes += b.cap; }
        return bytes;
    }
private:
    struct Buf { int dev; void *p; size_t cap; };
    std::mutex mu_;
    std::vector<Buf> free_;
    size_t held_ = 0;
};
DevicePool g_devpool;

struct DeviceBuf {
    void *p = nullptr;
    size_t cap = 0;
    ~DeviceBuf() { if (p) g_devpool.give_back(p, cap); }
    int alloc(size_t bytes)
    {
        if (int rc = g_devpool.take(bytes, &p, &cap)) { p = nullptr; return rc; }
        // AFG_POISON_ALLOC=1 (tests): device buffers start out holding NaN patterns, so that a stage that reads what nobody wrote
        // shows (hipMemset runs on the null stream and returns early; the stages copy on non-blocking streams, which do not wait
        // for it: without the synchronisation the fill can land on top of an upload)
        if (bytes && g_poison_alloc) {
            (void)hipMemset(p, 0xff, bytes);
            (void)hipStreamSynchronize(nullptr);
        }
        return AFG_OK;
    }
};

// page-locked host memory: H2D / D2H run at PCIe rate without a staging copy
struct PinnedBuf {
    void *p = nullptr;
    ~PinnedBuf() { release(); }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; }
    int alloc(size_t bytes)
    {
        hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
        if (e != hipSuccess) { p = nullptr; afg::set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return AFG_ERR_OOM; }
        return AFG_OK;
    }
};

// Staging buffers (gathered inputs, the raw MP3 PCM plane) are reused across calls: pinning memory costs about as
// much as moving it.  The pool keeps the few largest buffers it has seen, bounded in count and bytes.
class StagingPool {
public:
    struct Lease {
        StagingPool *pool = nullptr;
        void *p = nullptr;
        size_t cap = 0;
        Lease() = default;
        Lease(const Lease &) = delete;
        Lease &operator=(const Lease &) = delete;
        ~Lease() { if (pool && p) pool->give_back(p, cap); }
    };
    int take(size_t bytes, Lease &out)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].second >= bytes && (best == free_.size() || free_[i].second < free_[best].second)) best = i;
            if (best != free_.size()) {
                out.pool = this; out.p = free_[best].first; out.cap = free_[best].second;
                held_ -= out.cap;
                free_.erase(free_.begin() + (long)best);
                if (g_poison_alloc) std::memset(out.p, 0xff, out.cap);     // (tests: see DeviceBuf::alloc)
                return AFG_OK;
            }
        }
        void *p = nullptr;
        const size_t cap = bytes ? bytes : 1;
        if (g_trace) fprintf(stderr, "[afg] staging pool miss: pinning %.1f MB\n", cap / 1e6);
        hipError_t e = hipHostMalloc(&p, cap, hipHostMallocPortable);
        if (e != hipSuccess) { afg::set_error("hipHostMalloc(%zu) failed: %s", cap, hipGetErrorString(e)); return AFG_ERR_OOM; }
        out.pool = this; out.p = p; out.cap = cap;
        if (g_poison_alloc) std::memset(out.p, 0xff, out.cap);
        return AFG_OK;
    }
    // Frees every buffer that is not on lease (afg_host_pool_trim): a long-lived process gives the pinned memory back.
    size_t trim()
    {
        std::vector<std::pair<void *, size_t>> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            drop.swap(free_);
            held_ = 0;
        }
        size_t bytes = 0;
        for (auto &b : drop) { (void)hipHostFree(b.first); bytes += b.second; }
        return bytes;
    }
private:
    void give_back(void *p, size_t cap)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (free_.size() < 128 && held_ + cap <= ((size_t)24 << 30)) {      // (a grouped batch leases a set of buffers per group)
            free_.emplace_back(p, cap);
            held_ += cap;
        } else {
            (void)hipHostFree(p);
        }
    }
    std::mutex mu_;
    std::vector<std::pair<void *, size_t>> free_;
    size_t held_ = 0;
};
StagingPool g_staging;

// The upload / download stream pair of a device stage, kept between calls.  Creating and destroying two streams cost 2.8 ms of
// every batch call (hipStreamCreateWithFlags 0.8 ms, hipStreamDestroy 0.6 ms each), and a freshly created pair moves its first
// several hundred megabytes at half the rate of a pair that has been used before (round 6, AFG_TRACE on the 2048-file FLAC
// batch: 67 MB downloads 2.8 ms each on new streams, 1.4 ms on streams a probe had just run 1 GB through; `tools/ubench_pipe.hip`
// is the pipeline alone).  A stage leases a pair (stages of one call may run on two host threads) and gives it back drained.
class StreamPool {
public:
    // `mid` (optional): a third stream for the kernels of a stage whose uploads should never wait behind them
    hipError_t take(hipStream_t *up, hipStream_t *down, hipStream_t *mid = nullptr)
    {
        *up = *down = nullptr;
        if (mid) *mid = nullptr;
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev)) return e;
        Pair got{ -1, nullptr, nullptr, nullptr };
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].dev == dev) {
                    got = free_[i];
                    free_.erase(free_.begin() + (long)i);
                    break;
                }
        }
        hipError_t e = hipSuccess;
        if (!got.up) e = hipStreamCreateWithFlags(&got.up, hipStreamNonBlocking);
        if (e == hipSuccess && !got.down) e = hipStreamCreateWithFlags(&got.down, hipStreamNonBlocking);
        if (e == hipSuccess && mid && !got.mid) e = hipStreamCreateWithFlags(&got.mid, hipStreamNonBlocking);
        if (e != hipSuccess) {
            for (hipStream_t st : { got.up, got.down, got.mid }) if (st) (void)hipStreamDestroy(st);
            return e;
        }
        *up = got.up; *down = got.down;
        if (mid) *mid = got.mid;
        else if (got.mid) {                              // not wanted this time: keep it for a later lease
            std::lock_guard<std::mutex> lk(mu_);
            spare_mid_.push_back({ dev, got.mid });
        }
        return e;
    }
    void give(hipStream_t up, hipStream_t down, hipStream_t mid = nullptr)
    {
        int dev = 0;
        const bool known = up && down && hipGetDevice(&dev) == hipSuccess;
        std::unique_lock<std::mutex> lk(mu_);
        if (known && !mid)
            for (size_t i = 0; i < spare_mid_.size(); i++)
                if (spare_mid_[i].first == dev) { mid = spare_mid_[i].second; spare_mid_.erase(spare_mid_.begin() + (long)i); break; }
        if (known && free_.size() < 16) { free_.push_back({ dev, up, down, mid }); return; }
        lk.unlock();
        for (hipStream_t st : { up, down, mid }) if (st) (void)hipStreamDestroy(st);
    }
private:
    struct Pair { int dev; hipStream_t up, down, mid; };
    std::vector<std::pair<int, hipStream_t>> spare_mid_;
    std::mutex mu_;
    std::vector<Pair> free_;
};
StreamPool g_streams;

// AFG_TRACE=1: wall-clock of the host stages on stderr (development aid)
struct StageTimer {
    bool on = g_trace;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[afg] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// Helper threads are kept between calls: the batch path runs one parallel_for per chunk of files, and creating a
// few hundred threads each time cost more than the parsing they did.
class HelperPool {
public:
    ~HelperPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    // Runs work() on the caller and on up to `helpers` pooled threads; returns when all of them are done.
    // One job at a time: a second caller (another host thread in the library) just runs its work alone.
    void run(unsigned helpers, const std::function<void()> &work)
    {
        if (helpers == 0) { work(); return; }                // (without touching the pool: the caller wants no helpers)
        std::unique_lock<std::mutex> job_lock(job_mu_, std::try_to_lock);
        if (!job_lock.owns_lock()) { work(); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            while (workers_.size() < helpers && workers_.size() < 1024) workers_.emplace_back([this] { loop(); });
            job_ = &work; want_ = helpers; started_ = finished_ = 0; epoch_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        want_ = started_;                                    // no helper may still pick the job up
        done_cv_.wait(lk, [&] { return finished_ == started_; });
        job_ = nullptr;
    }
private:
    void loop()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && started_ < want_); });
            if (stop_) return;
            seen = epoch_;
            started_++;
            const std::function<void()> *job = job_;
            lk.unlock();
            (*job)();
            lk.lock();
            finished_++;
            done_cv_.notify_all();
        }
    }
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    const std::function<void()> *job_ = nullptr;
    unsigned want_ = 0, started_ = 0, finished_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};
HelperPool g_helpers;
// A multi-device batch runs one host thread per device, each with its own helpers (g_helpers serves one job at a time)
HelperPool g_device_helpers[16];
// ... and the second host thread of a grouped batch (afg_batch_decode_ex) its own: a pool serves one job at a time, and a
// group's parse pass must not find it taken by the other group's small gather jobs (it would run on one thread)
HelperPool g_group_helpers[16];
thread_local HelperPool *tl_helpers = nullptr;
// chunks a device stage cuts its files into (8; a group of a grouped batch -- below -- is itself a piece of a pipeline and takes 2)
thread_local unsigned tl_stage_chunks = 8;

template <typename F>
void parallel_for(size_t n, unsigned threads, F fn)
{
    if (n == 0) return;
    threads = (unsigned)std::min<size_t>(std::max(1u, threads), n);
    std::atomic<size_t> next{ 0 };
    const std::function<void()> work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) return;
            fn(i);
        }
    };
    (tl_helpers ? *tl_helpers : g_helpers).run(threads - 1, work);
}

struct BatchOut {
    std::vector<Decoded> files;
    StagingPool::Lease plane;           // all PCM of the batch: FLAC files, then QOA files, then MP3 files; page-locked,
    size_t plane_floats = 0;            // returned to the pool by afg_batch_free / afg_close
    StagingPool::Lease mp3_plane;       // batch path: the MP3 PCM in staging layout, served in place
    StagingPool::Lease opus_plane;      // batch path: the Opus PCM, files back to back
    std::unique_ptr<BatchOut> early;    // batch path: the FLAC / QOA files, decoded on a second host thread meanwhile
};

// Device stage for a set of parsed files: every FLAC record of the batch in one launch, every QOA frame
// in another; inputs are gathered (by `threads` host threads) into one page-locked buffer per kind and
// the results come back as one plane.
// Where the batch path parsed its MP3 files: one page-locked buffer, file i at block base[i] (gaps between files).
// The device planes and the MP3 result plane mirror that layout, so a chunk of files moves in ONE copy each way and
// a file's PCM is served where it lands (a per-file copy costs ~20 us of submission: 2 x 2048 of them were the whole
// end-to-end time of a 2048-file batch).
struct Mp3Stage {
    const float *coef = nullptr;        // float upload: dequantised spectra, blocks * 576 ...
    const int16_t *q = nullptr;         // ... or quantised upload: Huffman values, blocks * 576, and one record slot
    const afg_mp3_qgranule *recs = nullptr;   // per block (the slot of a granule's first block is used, nch = 0 elsewhere)
    const uint32_t *flags = nullptr;
    size_t blocks = 0;
    const size_t *base = nullptr;
    float *plane = nullptr;             // host PCM plane, blocks * 576 floats (page-locked)
};

// Where the batch path parsed its Ogg Vorbis files: file i's spectra at float base[i] of one page-locked buffer
struct OggStage {
    const float *spec = nullptr;
    size_t floats = 0;
    const size_t *base = nullptr;
};

// Where the batch path parsed its FLAC files: file i's residual plane at word base[i] of one page-locked buffer
struct FlacStage {
    const int32_t *res = nullptr;
    size_t words = 0;
    const size_t *base = nullptr;
};

// H2D -> kernel on stream `up`, D2H on stream `down` behind an event: chunk k+1 uploads and transforms while chunk
// k's PCM goes back (PCIe is full duplex) -- and while the host threads parse chunk k+2.
struct Mp3Pipe {
    const Mp3Stage *st = nullptr;
    DeviceBuf d_in, d_pcm;
    uint32_t *d_flags = nullptr;
    hipStream_t up = nullptr, down = nullptr;
    std::vector<afg_mp3_plan *> plans;
    std::vector<hipEvent_t> events;
    DeviceBuf d_tables;                 // plan tables: at most one 16-byte segment and stream record per block
    StagingPool::Lease h_tables;
    afg::PlanArena arena;
    int rc = AFG_OK;
    hipError_t e = hipSuccess;

    DeviceBuf d_qin;                    // quantised upload: int16 plane, record slots, stereo descriptors
    int16_t *d_q = nullptr;
    afg_mp3_qgranule *d_recs = nullptr;
    afg_mp3_sdesc *d_sdesc = nullptr;
    StagingPool::Lease h_sdesc;
    size_t sdesc_cap = 0, sdesc_used = 0;
    uint64_t h2d_bytes = 0;

    int open(const Mp3Stage &stage)
    {
        st = &stage;
        const size_t coef_bytes = stage.blocks * 576 * sizeof(float), flag_bytes = (stage.blocks * 4 + 15) & ~(size_t)15;
        if (int r = d_in.alloc(coef_bytes + flag_bytes)) return r;
        if (int r = d_pcm.alloc(coef_bytes)) return r;
        if (stage.q) {
            const size_t q_bytes = (stage.blocks * 576 * sizeof(int16_t) + 15) & ~(size_t)15;
            const size_t rec_bytes = stage.blocks * sizeof(afg_mp3_qgranule);
            sdesc_cap = 4096;                            // intensity-stereo granules of the whole batch (grown on demand: rare)
            if (int r = d_qin.alloc(q_bytes + rec_bytes + sdesc_cap * sizeof(afg_mp3_sdesc))) return r;
            d_q = (int16_t *)d_qin.p;
            d_recs = (afg_mp3_qgranule *)((uint8_t *)d_qin.p + q_bytes);
            d_sdesc = (afg_mp3_sdesc *)((uint8_t *)d_recs + rec_bytes);
            if (int r = g_staging.take(sdesc_cap * sizeof(afg_mp3_sdesc), h_sdesc)) return r;
        }
        d_flags = (uint32_t *)((uint8_t *)d_in.p + coef_bytes);
        e = g_streams.take(&up, &down);
        if (e != hipSuccess) { afg::set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); return AFG_ERR_HIP; }
        const size_t tab_bytes = stage.blocks * 32 + 4096;
        if (int r = d_tables.alloc(tab_bytes)) return r;
        if (int r = g_staging.take(tab_bytes, h_tables)) return r;
        arena.host = (uint8_t *)h_tables.p; arena.dev = (uint8_t *)d_tables.p; arena.cap = tab_bytes; arena.stream = up;
        return AFG_OK;
    }
    // files [f0, f1) have been parsed into the stage: plan, upload, transform, download
    void submit(const std::vector<Parsed> &parsed, size_t f0, size_t f1)
    {
        if (rc || e != hipSuccess) return;
        std::vector<uint32_t> granules;
        std::vector<uint8_t> channels;
        std::vector<uint64_t> bases;
        size_t b0 = 0, b1 = 0;
        for (size_t i = f0; i < f1; i++) {
            const Parsed &p = parsed[i];
            if (p.format != AFG_FORMAT_MP3 || !p.mp3.blocks()) continue;
            if (granules.empty()) b0 = st->base[i];
            uint64_t at = st->base[i];
            for (uint32_t g : p.mp3.run_granules) {
                granules.push_back(g);
                channels.push_back((uint8_t)p.mp3.channels);
                bases.push_back(at);
                at += (uint64_t)g * (uint64_t)p.mp3.channels;
            }
            b1 = st->base[i] + p.mp3.blocks();
        }
        if (granules.empty()) return;
        afg_mp3_plan *plan = nullptr;
        rc = afg::mp3_plan_create_at(&plan, (uint32_t)granules.size(), granules.data(), channels.data(), bases.data(), 0, &arena);
        if (rc) return;
        plans.push_back(plan);
        hipEvent_t done = nullptr;
        e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (e != hipSuccess) return;
        events.push_back(done);
        const size_t nb = b1 - b0;
        if (st->q) {
            // quantised upload: 2 bytes per line + a record per granule, requantised on the device into the plane the
            // transform reads (afg_mp3_requant_hip); the stereo descriptors of intensity frames are gathered per chunk
            afg_mp3_qgranule *hrecs = const_cast<afg_mp3_qgranule *>(st->recs);
            const size_t sd0 = sdesc_used;
            for (size_t i = f0; i < f1; i++) {
                const Parsed &p = parsed[i];
                if (p.format != AFG_FORMAT_MP3 || p.mp3.sdesc.empty()) continue;
                if (sdesc_used + p.mp3.sdesc.size() > sdesc_cap) { afg::set_error("MP3 stage: more than %zu intensity-stereo granules in one batch", sdesc_cap); rc = AFG_ERR_UNSUPPORTED; return; }
                std::memcpy((afg_mp3_sdesc *)h_sdesc.p + sdesc_used, p.mp3.sdesc.data(), p.mp3.sdesc.size() * sizeof(afg_mp3_sdesc));
                for (size_t k = st->base[i]; k < st->base[i] + p.mp3.blocks(); k++)
                    if (hrecs[k].nch && hrecs[k].sdesc != AFG_MP3_NO_SDESC) hrecs[k].sdesc += (uint32_t)sdesc_used;
                sdesc_used += p.mp3.sdesc.size();
            }
            e = hipMemcpyAsync(d_q + b0 * 576, st->q + b0 * 576, nb * 576 * sizeof(int16_t), hipMemcpyHostToDevice, up);
            if (e == hipSuccess) e = hipMemcpyAsync(d_recs + b0, st->recs + b0, nb * sizeof(afg_mp3_qgranule), hipMemcpyHostToDevice, up);
            if (e == hipSuccess && sdesc_used > sd0)
                e = hipMemcpyAsync(d_sdesc + sd0, (afg_mp3_sdesc *)h_sdesc.p + sd0, (sdesc_used - sd0) * sizeof(afg_mp3_sdesc), hipMemcpyHostToDevice, up);
            if (e == hipSuccess) e = hipMemcpyAsync(d_flags + b0, st->flags + b0, nb * sizeof(uint32_t), hipMemcpyHostToDevice, up);
            if (e != hipSuccess) return;
            h2d_bytes += nb * (576 * sizeof(int16_t) + sizeof(afg_mp3_qgranule) + sizeof(uint32_t));
            rc = afg_mp3_requant_hip(nb, d_recs + b0, d_q, d_sdesc, (float *)d_in.p, up);
            if (rc) return;
        } else {
            e = hipMemcpyAsync((float *)d_in.p + b0 * 576, st->coef + b0 * 576, nb * 576 * sizeof(float), hipMemcpyHostToDevice, up);
            if (e == hipSuccess) e = hipMemcpyAsync(d_flags + b0, st->flags + b0, nb * sizeof(uint32_t), hipMemcpyHostToDevice, up);
            if (e != hipSuccess) return;
            h2d_bytes += nb * (576 * sizeof(float) + sizeof(uint32_t));
        }
        rc = afg_mp3_transform_hip(plan, (const float *)d_in.p, d_flags, (float *)d_pcm.p, nullptr, up);
        if (rc) return;
        e = hipEventRecord(done, up);
        if (e == hipSuccess) e = hipStreamWaitEvent(down, done, 0);
        if (e == hipSuccess)
            e = hipMemcpyAsync(st->plane + b0 * 576, (const float *)d_pcm.p + b0 * 576, nb * 576 * sizeof(float), hipMemcpyDeviceToHost, down);
    }
    int close()
    {
        if (up) { hipError_t e2 = hipStreamSynchronize(up); if (e == hipSuccess) e = e2; }
        if (down) { hipError_t e2 = hipStreamSynchronize(down); if (e == hipSuccess) e = e2; }
        for (afg_mp3_plan *p : plans) afg_mp3_plan_destroy(p);
        for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
        plans.clear(); events.clear();
        g_streams.give(up, down);
        up = down = nullptr;
        if (rc) return rc;
        if (e != hipSuccess) { afg::set_error("MP3 stage failed: %s", hipGetErrorString(e)); return AFG_ERR_HIP; }
        return AFG_OK;
    }
    ~Mp3Pipe() { (void)close(); }
};

// Decoder state an MP3 stream carries from one chunk of frames to the next (chunked AudioStream reads): the overlap and
// polyphase history of the run that was open when the previous chunk ended, as the transform kernel left it.
struct Mp3Carry {
    DeviceBuf state;                    // AFG_MP3_STATE_FLOATS floats
    bool valid = false;                 // `state` holds the end of the previous chunk
    bool continues = false;             // this chunk's first run goes on from it
};

// What an Opus stream carries from one chunk of packets to the next on the device: the transform stage's per-channel
// memory (overlap, post-filter history, de-emphasis), as afg_celt_transform_hip reads and rewrites it.
struct OpusCarry {
    DeviceBuf states;                   // channels * AFG_CELT_STATE_FLOATS floats
    bool valid = false;
};

int decode_parsed(std::vector<Parsed> &parsed, const uint8_t *const *data, const size_t *len, unsigned threads, BatchOut &out,
                  const Mp3Stage *stage = nullptr, const OggStage *ogg_stage = nullptr, const FlacStage *flac_stage = nullptr,
                  const uint8_t *own = nullptr, Mp3Carry *carry = nullptr, OpusCarry *opus_carry = nullptr,
                  const size_t *opus_done_at = nullptr)
{
    // opus_done_at (batch path): the Opus files are already decoded, file i's PCM at float opus_done_at[i] of the batch's
    // Opus plane; only their metadata is filled in here
    // `own` (optional, one byte per file): the files this call is responsible for.  The batch path decodes its FLAC /
    // QOA files on a second host thread while the first still parses MP3 / Ogg files: a call never looks at (not even
    // the format of) a file it does not own.
    auto fmt_of = [&](const Parsed &q) -> int { return (!own || own[&q - parsed.data()]) ? q.format : -1; };
    const size_t nf = parsed.size();
    out.files.assign(nf, Decoded());
    StageTimer tm;
    // ---- layout ----
    std::vector<size_t> res_base(nf, 0), fr_base(nf, 0), sf_base(nf, 0), qbyte_base(nf, 0), qfr_base(nf, 0);
    size_t res_total = 0, fr_total = 0, sf_total = 0, flac_out = 0, qbytes = 0, qframes = 0, qoa_out = 0;
    const bool flac_staged = flac_stage && flac_stage->words;
    for (size_t i = 0; i < nf; i++) {
        Parsed &p = parsed[i];
        if (fmt_of(p) != AFG_FORMAT_FLAC) continue;
        if (!flac_staged) res_total = (res_total + 3) & ~(size_t)3;     // 16-byte aligned planes (int16 rows: afg_flac_frame.res16)
        res_base[i] = flac_staged ? flac_stage->base[i] : res_total; fr_base[i] = fr_total; sf_base[i] = sf_total;
        out.files[i].pcm_off = flac_out;
        res_total += p.flac.res_size(); fr_total += p.flac.frames.size(); sf_total += p.flac.subframes.size();
        flac_out += p.flac.out_samples;
    }
    if (flac_staged) res_total = flac_stage->words;          // the device plane mirrors the staging layout (gaps and all)
    for (size_t i = 0; i < nf; i++) {
        Parsed &p = parsed[i];
        if (fmt_of(p) != AFG_FORMAT_QOA) continue;
        qbyte_base[i] = qbytes; qfr_base[i] = qframes;
        out.files[i].pcm_off = flac_out + qoa_out;
        qbytes += (len[i] + 15) & ~(size_t)15;
        qframes += p.qoa.size();
        qoa_out += p.qoa.back().out_off + (size_t)p.qoa.back().samples * p.qoa.back().channels;
    }
    std::vector<size_t> mp3_blk_base(nf, 0);
    size_t mp3_blocks = 0, mp3_out = 0;
    for (size_t i = 0; i < nf; i++) {
        Parsed &p = parsed[i];
        if (fmt_of(p) != AFG_FORMAT_MP3) continue;
        mp3_blk_base[i] = mp3_blocks;
        out.files[i].pcm_off = flac_out + qoa_out + mp3_out;
        mp3_blocks += p.mp3.blocks();
        mp3_out += (size_t)p.mp3.pcm_samples;
    }
    // Staged batch (afg_batch_decode): the MP3 files are already decoded, in staging layout, in stage->plane
    const bool staged = stage && stage->blocks && mp3_blocks;
    if (staged) {
        mp3_out = 0;
        for (size_t i = 0; i < nf; i++) {
            Parsed &p = parsed[i];
            if (fmt_of(p) != AFG_FORMAT_MP3) continue;
            const uint64_t first = p.mp3.copies.empty() ? 0 : p.mp3.copies[0].src;
            out.files[i].pcm_off = stage->base[i] * 576 + (size_t)first;
            out.files[i].in_mp3_plane = true;
        }
    }
    // Vorbis: chunks of files, one plan each; the Vorbis part of the result plane is the plans' output planes back to
    // back, so a chunk's PCM comes back in one copy and a file is served where it lands (first piece onwards)
    struct OggChunk {
        size_t f0 = 0, f1 = 0, spec0 = 0, spec_n = 0, out0 = 0, out_n = 0;
        afg_vorbis_plan *plan = nullptr;
        hipEvent_t done = nullptr;
        std::vector<size_t> spec_at;                         // per file of the chunk: float offset of its spectra in the chunk
    };
    struct OggChunks {
        std::vector<OggChunk> v;
        ~OggChunks()
        {
            for (OggChunk &c : v) {
                if (c.plan) afg_vorbis_plan_destroy(c.plan);
                if (c.done) (void)hipEventDestroy(c.done);
            }
        }
    } ogg;
    struct OggPiece { size_t file; uint64_t from, count; };  // pieces of files that are not served as one run
    std::vector<OggPiece> ogg_pieces;
    std::vector<size_t> ogg_broken;
    size_t ogg_out = 0, ogg_packets = 0, ogg_spec = 0;
    const bool ogg_staged = ogg_stage && ogg_stage->floats;
    {
        size_t total = 0;
        for (size_t i = 0; i < nf; i++)
            if (fmt_of(parsed[i]) == AFG_FORMAT_OGG) { total += parsed[i].ogg.n_spec; ogg_packets += parsed[i].ogg.pflags.size(); }
        const size_t target = std::max<size_t>((total + tl_stage_chunks - 1) / tl_stage_chunks, (size_t)4 << 20);
        for (size_t f0 = 0; f0 < nf && ogg_packets;) {
            size_t f1 = f0, acc = 0;
            while (f1 < nf && acc < target) { if (fmt_of(parsed[f1]) == AFG_FORMAT_OGG) acc += parsed[f1].ogg.n_spec; f1++; }
            OggChunk c;
            c.f0 = f0; c.f1 = f1; c.spec0 = ogg_spec; c.out0 = ogg_out;
            std::vector<uint32_t> npk;
            std::vector<uint8_t> chans, pflags;
            std::vector<uint16_t> b0, b1;
            std::vector<uint64_t> sbase;                     // staged: where each stream's spectra sit in the staging buffer
            c.spec_at.assign(f1 - f0, 0);
            size_t at = 0, span0 = 0, span1 = 0;
            for (size_t i = f0; i < f1; i++) {
                const Parsed &p = parsed[i];
                if (fmt_of(p) != AFG_FORMAT_OGG) continue;
                c.spec_at[i - f0] = at;
                at += p.ogg.n_spec;
                if (p.ogg.pflags.empty()) continue;
                if (ogg_staged) {
                    if (sbase.empty()) span0 = ogg_stage->base[i];
                    span1 = ogg_stage->base[i] + p.ogg.n_spec;
                    sbase.push_back(ogg_stage->base[i]);
                }
                npk.push_back((uint32_t)p.ogg.pflags.size());
                chans.push_back((uint8_t)p.ogg.channels);
                b0.push_back((uint16_t)p.ogg.blocksize0);
                b1.push_back((uint16_t)p.ogg.blocksize1);
                pflags.insert(pflags.end(), p.ogg.pflags.begin(), p.ogg.pflags.end());
            }
            f0 = f1;
            if (npk.empty()) continue;
            if (int rc = afg::vorbis_plan_create_at(&c.plan, (uint32_t)npk.size(), npk.data(), chans.data(), b0.data(), b1.data(),
                                                    pflags.data(), ogg_staged ? sbase.data() : nullptr, 0))
                return rc;
            ogg.v.push_back(std::move(c));
            OggChunk &k = ogg.v.back();
            k.spec_n = (size_t)afg_vorbis_plan_spec_floats(k.plan);
            k.out_n = (size_t)afg_vorbis_plan_out_floats(k.plan);
            if (ogg_staged) {                                // the plan addresses the staging layout: the chunk is a span of it
                if (k.spec_n != span1) {
                    afg::set_error("Vorbis stage: spectrum layout mismatch (%zu vs %zu floats)", k.spec_n, span1);
                    return AFG_ERR_INVALID;
                }
                k.spec0 = span0;
                k.spec_n = span1 - span0;
            } else if (k.spec_n != at) {
                afg::set_error("Vorbis stage: spectrum layout mismatch (%zu vs %zu floats)", k.spec_n, at);
                return AFG_ERR_INVALID;
            }
            std::vector<uint64_t> out_off(pflags.size());
            if (int rc = afg_vorbis_plan_offsets(k.plan, nullptr, out_off.data())) return rc;
            // delivery = the pull API's share of every packet's output: normally one run from the first packet on
            size_t pk = 0;
            for (size_t i = k.f0; i < k.f1; i++) {
                const Parsed &p = parsed[i];
                if (fmt_of(p) != AFG_FORMAT_OGG) continue;
                const size_t n = p.ogg.pflags.size(), C = (size_t)p.ogg.channels;
                const size_t first_piece = ogg_pieces.size();
                for (size_t q = 0; q < n;) {
                    if (p.ogg.take_count[q] <= 0) { q++; continue; }
                    uint64_t from = out_off[pk + q] + (uint64_t)p.ogg.take_from[q] * C, cnt = (uint64_t)p.ogg.take_count[q] * C;
                    size_t j = q + 1;
                    while (j < n && p.ogg.take_count[j] > 0 && out_off[pk + j] + (uint64_t)p.ogg.take_from[j] * C == from + cnt)
                        cnt += (uint64_t)p.ogg.take_count[j++] * C;
                    ogg_pieces.push_back(OggPiece{ i, ogg_out + from, cnt });
                    q = j;
                }
                out.files[i].pcm_off = flac_out + qoa_out + mp3_out + (ogg_pieces.size() > first_piece ? (size_t)ogg_pieces[first_piece].from : ogg_out);
                if (ogg_pieces.size() - first_piece > 1) ogg_broken.push_back(i);
                else if (ogg_pieces.size() > first_piece) ogg_pieces.pop_back();          // one run: nothing to move
                pk += n;
            }
            if (!ogg_staged) ogg_spec += k.spec_n;
            ogg_out += k.out_n;
        }
    }
    if (staged) {
        // delivery in place: a file whose copy plan is one piece (every undamaged file) is served where it landed;
        // the pieces of a damaged file are closed up towards its first piece (ascending, so memmove order is safe)
        std::vector<size_t> broken;                          // files whose pieces are not already back to back
        for (size_t i = 0; i < nf; i++) {
            const Parsed &p = parsed[i];
            if (fmt_of(p) != AFG_FORMAT_MP3) continue;
            for (size_t k = 1; k < p.mp3.copies.size(); k++)
                if (p.mp3.copies[k].src != p.mp3.copies[k - 1].src + p.mp3.copies[k - 1].count) { broken.push_back(i); break; }
        }
        parallel_for(broken.size(), threads, [&](size_t bi) {
            const size_t i = broken[bi];
            const Parsed &p = parsed[i];
            float *file_plane = stage->plane + stage->base[i] * 576;
            float *dst = file_plane + p.mp3.copies[0].src;
            for (const afg_mp3::Copy &c : p.mp3.copies) {
                if (dst != file_plane + c.src) std::memmove(dst, file_plane + c.src, (size_t)c.count * sizeof(float));
                dst += c.count;
            }
        });
        tm.lap("mp3 delivery (in place)");
    }
    // Opus: one channel sequence per output channel of every file; the PCM plane holds the files back to back, interleaved
    std::vector<size_t> opus_rec_base(nf, 0), opus_coef_base(nf, 0), opus_pcm_base(nf, 0);
    size_t opus_out = 0, opus_recs = 0, opus_coefs = 0, opus_seqs = 0;
    for (size_t i = 0; i < nf; i++) {
        const Parsed &p = parsed[i];
        if (fmt_of(p) != AFG_FORMAT_OPUS) continue;
        if (opus_done_at) {
            out.files[i].pcm_off = opus_done_at[i];
            out.files[i].in_opus_plane = true;
            continue;
        }
        opus_rec_base[i] = opus_recs; opus_coef_base[i] = opus_coefs; opus_pcm_base[i] = opus_out;
        out.files[i].pcm_off = flac_out + qoa_out + mp3_out + ogg_out + opus_out;
        opus_recs += p.opus.frames.size() * (size_t)p.opus.channels;
        opus_coefs += p.opus.coeffs.size();
        opus_out += (size_t)p.opus.pcm_frames * (size_t)p.opus.channels;
        // The transform stage walks sequences 2p and 2p + 1 together when they are the two channels of a stream (one
        // wavefront, half each: afg.h).  A stereo file behind an odd number of mono files would sit across two such
        // slots -- walked one channel at a time, and, in AFG_NUMERIC_TOLERANCE, to samples that depend on what else is
        // in the batch -- so an empty sequence goes in front of it.
        if (p.opus.channels == 2 && (opus_seqs & 1)) opus_seqs++;
        opus_seqs += (size_t)p.opus.channels;
    }
    out.plane_floats = flac_out + qoa_out + mp3_out + ogg_out + opus_out;
    if (out.plane_floats == 0) goto metadata;
    {
        if (int rc = g_staging.take(out.plane_floats * sizeof(float), out.plane)) return rc;
        tm.lap("layout + plane alloc");
        DeviceBuf d_out;
        if (int rc = d_out.alloc(out.plane_floats * sizeof(float))) return rc;
        hipStream_t stream = nullptr;
        // ---- FLAC ----
        if (flac_out) {
            const size_t rec_bytes = fr_total * sizeof(afg_flac_frame) + sf_total * sizeof(afg_flac_subframe);
            const size_t rec_pad = (rec_bytes + 15) & ~(size_t)15;
            StagingPool::Lease h_in;
            DeviceBuf d_in;
            if (int rc = g_staging.take(rec_pad + (flac_staged ? 0 : res_total * 4), h_in)) return rc;
            if (int rc = d_in.alloc(rec_pad + res_total * 4)) return rc;
            afg_flac_frame *hf = (afg_flac_frame *)h_in.p;
            afg_flac_subframe *hs = (afg_flac_subframe *)(hf + fr_total);
            int32_t *hr = (int32_t *)((uint8_t *)h_in.p + rec_pad);           // (not staged: the residuals are gathered here)
            const int32_t *hres = flac_staged ? flac_stage->res : hr;
            // Chunks of files: gather (host threads) -> upload + kernel on `up` -> download on `down` behind an event,
            // so the gather of chunk k+1, the upload of chunk k and the download of chunk k-1 overlap.
            const afg_flac_frame *df = (const afg_flac_frame *)d_in.p;
            const afg_flac_subframe *ds = (const afg_flac_subframe *)(df + fr_total);
            const int32_t *dr = (const int32_t *)((const uint8_t *)d_in.p + rec_pad);
            // three streams: uploads only on `up` (the next chunk's never waits behind this chunk's kernel), kernels on `mid`
            // behind the upload's event, downloads on `down` behind the kernel's
            hipStream_t up = nullptr, down = nullptr, mid = nullptr;
            std::vector<hipEvent_t> events;
            hipError_t e = g_streams.take(&up, &down, &mid);
            int rc = AFG_OK;
            const size_t target = std::max<size_t>((res_total + tl_stage_chunks - 1) / tl_stage_chunks, (size_t)4 << 20);
            // AFG_TRACE: host wall-clock of every chunk's gather and submission, device time of its upload, kernel and download
            struct ChunkTrace { double t_begin, t_gathered, t_queued; hipEvent_t e_up0, e_up1, e_k1, e_d0, e_d1; };
            std::vector<ChunkTrace> ctrace;
            const auto t_stage = std::chrono::steady_clock::now();
            auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_stage).count(); };
            hipEvent_t e_stage = nullptr;
            if (g_trace && e == hipSuccess) { (void)hipEventCreate(&e_stage); (void)hipEventRecord(e_stage, up); }
            auto mark = [&](hipStream_t st) { hipEvent_t ev = nullptr; (void)hipEventCreate(&ev); (void)hipEventRecord(ev, st); return ev; };
            for (size_t f0 = 0; f0 < nf && !rc && e == hipSuccess;) {
                size_t f1 = f0, acc = 0;
                while (f1 < nf && acc < target) { if (fmt_of(parsed[f1]) == AFG_FORMAT_FLAC) acc += parsed[f1].flac.res_size(); f1++; }
                size_t first = nf, last = nf;                // first / last FLAC file of the chunk
                for (size_t i = f0; i < f1; i++)
                    if (fmt_of(parsed[i]) == AFG_FORMAT_FLAC) { if (first == nf) first = i; last = i; }
                if (first == nf) { f0 = f1; continue; }
                ChunkTrace ct{};
                ct.t_begin = since();
                parallel_for(f1 - f0, threads, [&](size_t k) {
                    const size_t i = f0 + k;
                    Parsed &p = parsed[i];
                    if (fmt_of(p) != AFG_FORMAT_FLAC) return;
                    for (size_t q = 0; q < p.flac.frames.size(); q++) {
                        afg_flac_frame f = p.flac.frames[q];
                        f.in_off += (f.res16 ? 2 : 1) * (uint64_t)res_base[i]; f.out_off += out.files[i].pcm_off; f.sf_index += (uint32_t)sf_base[i];
                        hf[fr_base[i] + q] = f;
                    }
                    std::memcpy(hs + sf_base[i], p.flac.subframes.data(), p.flac.subframes.size() * sizeof(afg_flac_subframe));
                    if (!flac_staged) {
                        std::memcpy(hr + res_base[i], p.flac.res_data(), p.flac.res_size() * 4);
                        std::vector<int32_t>().swap(p.flac.res);       // the residual plane is the big one: drop it early
                    }
                });
                const size_t fr0 = fr_base[first], fr1 = fr_base[last] + parsed[last].flac.frames.size();
                const size_t sf0 = sf_base[first], sf1 = sf_base[last] + parsed[last].flac.subframes.size();
                const size_t r0 = res_base[first];
                size_t r1 = r0;
                for (size_t q = fr0; q < fr1; q++)                           // (a packed frame keeps the words it was parsed into)
                    r1 = std::max<size_t>(r1, (size_t)(hf[q].res16 ? hf[q].in_off / 2 : hf[q].in_off) + (size_t)hf[q].channels * hf[q].block_size);
                const size_t o0 = out.files[first].pcm_off, o1 = out.files[last].pcm_off + parsed[last].flac.out_samples;
                ct.t_gathered = since();
                if (g_trace) ct.e_up0 = mark(up);
                e = hipMemcpyAsync((void *)(df + fr0), hf + fr0, (fr1 - fr0) * sizeof(afg_flac_frame), hipMemcpyHostToDevice, up);
                if (e == hipSuccess) e = hipMemcpyAsync((void *)(ds + sf0), hs + sf0, (sf1 - sf0) * sizeof(afg_flac_subframe), hipMemcpyHostToDevice, up);
                if (e == hipSuccess) e = hipMemcpyAsync((void *)(dr + r0), hres + r0, (r1 - r0) * 4, hipMemcpyHostToDevice, up);
                if (e != hipSuccess) break;
                if (g_trace) ct.e_up1 = mark(up);
                hipEvent_t landed = nullptr, done = nullptr;
                e = hipEventCreateWithFlags(&landed, hipEventDisableTiming);
                if (e != hipSuccess) break;
                events.push_back(landed);
                e = hipEventRecord(landed, up);
                if (e == hipSuccess) e = hipStreamWaitEvent(mid, landed, 0);
                if (e != hipSuccess) break;
                // (the records are still here in host memory: only the populated instantiations are launched)
                rc = afg_flac_transform_variants_hip(fr1 - fr0, df + fr0, ds, dr, nullptr, (float *)d_out.p,
                                                     afg_flac_variants(fr1 - fr0, hf + fr0, hs), mid);
                if (rc) break;
                if (g_trace) ct.e_k1 = mark(mid);
                e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
                if (e != hipSuccess) break;
                events.push_back(done);
                e = hipEventRecord(done, mid);
                if (e == hipSuccess) e = hipStreamWaitEvent(down, done, 0);
                if (g_trace) ct.e_d0 = mark(down);
                if (e == hipSuccess)
                    e = hipMemcpyAsync((float *)out.plane.p + o0, (const float *)d_out.p + o0, (o1 - o0) * sizeof(float), hipMemcpyDeviceToHost, down);
                if (g_trace) { ct.e_d1 = mark(down); ct.t_queued = since(); ctrace.push_back(ct); }
                f0 = f1;
            }
            const double t_loop = since();
            if (up) { hipError_t e2 = hipStreamSynchronize(up); if (e == hipSuccess) e = e2; }
            if (mid) { hipError_t e2 = hipStreamSynchronize(mid); if (e == hipSuccess) e = e2; }
            const double t_up = since();
            if (down) { hipError_t e2 = hipStreamSynchronize(down); if (e == hipSuccess) e = e2; }
            if (g_trace) {
                std::fprintf(stderr, "[afg] flac stage: loop done %.2f ms, up drained %.2f, down drained %.2f\n", t_loop, t_up, since());
                for (size_t k = 0; k < ctrace.size(); k++) {
                    const ChunkTrace &c = ctrace[k];
                    float u0 = 0, u1 = 0, k1 = 0, d0 = 0, d1 = 0;
                    (void)hipEventElapsedTime(&u0, e_stage, c.e_up0); (void)hipEventElapsedTime(&u1, e_stage, c.e_up1);
                    (void)hipEventElapsedTime(&k1, e_stage, c.e_k1); (void)hipEventElapsedTime(&d0, e_stage, c.e_d0);
                    (void)hipEventElapsedTime(&d1, e_stage, c.e_d1);
                    std::fprintf(stderr, "[afg]   chunk %zu: host begin %.2f gathered %.2f queued %.2f | device up %.2f-%.2f kernel -%.2f down %.2f-%.2f\n",
                                 k, c.t_begin, c.t_gathered, c.t_queued, u0, u1, k1, d0, d1);
                    for (hipEvent_t ev : { c.e_up0, c.e_up1, c.e_k1, c.e_d0, c.e_d1 }) (void)hipEventDestroy(ev);
                }
                if (e_stage) (void)hipEventDestroy(e_stage);
            }
            for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
            g_streams.give(up, down, mid);
            if (rc) return rc;
            if (e != hipSuccess) { afg::set_error("FLAC stage failed: %s", hipGetErrorString(e)); return AFG_ERR_HIP; }
            tm.lap("flac gather | h2d | kernel | d2h (chunks overlapped)");
        }
        // ---- QOA ----
        if (qoa_out) {
            const size_t rec_pad = (qframes * sizeof(afg_qoa_frame) + 15) & ~(size_t)15;
            StagingPool::Lease h_in;
            DeviceBuf d_in;
            if (int rc = g_staging.take(rec_pad + qbytes, h_in)) return rc;
            if (int rc = d_in.alloc(rec_pad + qbytes)) return rc;
            afg_qoa_frame *hq = (afg_qoa_frame *)h_in.p;
            uint8_t *hb = (uint8_t *)h_in.p + rec_pad;
            parallel_for(nf, threads, [&](size_t i) {
                Parsed &p = parsed[i];
                if (fmt_of(p) != AFG_FORMAT_QOA) return;
                for (size_t k = 0; k < p.qoa.size(); k++) {
                    afg_qoa_frame f = p.qoa[k];
                    f.byte_off += qbyte_base[i]; f.out_off += out.files[i].pcm_off;
                    hq[qfr_base[i] + k] = f;
                }
                std::memcpy(hb + qbyte_base[i], data[i], len[i]);
            });
            AFG_HIP_CHECK(hipMemcpyAsync(d_in.p, h_in.p, rec_pad + qbytes, hipMemcpyHostToDevice, stream));
            if (int rc = afg_qoa_transform_hip(qframes, (const afg_qoa_frame *)d_in.p, (const uint8_t *)d_in.p + rec_pad, nullptr,
                                               (float *)d_out.p, stream))
                return rc;
            AFG_HIP_CHECK(hipStreamSynchronize(stream));
        }
        if (qoa_out) {                                       // (the FLAC part came back chunk by chunk)
            AFG_HIP_CHECK(hipMemcpyAsync((float *)out.plane.p + flac_out, (const float *)d_out.p + flac_out, qoa_out * sizeof(float),
                                         hipMemcpyDeviceToHost, stream));
            AFG_HIP_CHECK(hipStreamSynchronize(stream));
        }
        tm.lap("qoa stage");
        // ---- MP3: spectra of every decoded granule -> PCM plane -> the samples mp3dec_ex_read would deliver ----
        if (mp3_blocks && !staged) {
            const size_t coef_bytes = mp3_blocks * 576 * sizeof(float), flag_bytes = (mp3_blocks * 4 + 15) & ~(size_t)15;
            DeviceBuf d_in, d_pcm;
            if (int rc = d_in.alloc(coef_bytes + flag_bytes)) return rc;
            if (int rc = d_pcm.alloc(coef_bytes)) return rc;
            // The files are cut into a few chunks of similar size, each with its own plan: the upload and kernel of
            // chunk k+1 (stream `up`) run while chunk k's PCM goes back (stream `down`) -- PCIe is full duplex.
            struct Chunk { size_t f0, f1, blk0, blocks; afg_mp3_plan *plan; hipEvent_t done; size_t runs = 0; };
            std::vector<Chunk> chunks;
            {
                size_t want = 8;
                if (afg::dev_option(afg::kDevMp3Chunks) > 0) want = (size_t)afg::dev_option(afg::kDevMp3Chunks);
                const size_t target = std::max<size_t>((mp3_blocks + want - 1) / want, 8192);
                Chunk c{ 0, 0, 0, 0, nullptr, nullptr, 0 };
                for (size_t i = 0; i < nf; i++) {
                    if (fmt_of(parsed[i]) != AFG_FORMAT_MP3) continue;
                    if (c.blocks == 0) { c.f0 = i; c.blk0 = mp3_blk_base[i]; }
                    c.blocks += parsed[i].mp3.blocks();
                    c.f1 = i + 1;
                    if (c.blocks >= target) { chunks.push_back(c); c = Chunk{ 0, 0, 0, 0, nullptr, nullptr, 0 }; }
                }
                if (c.blocks) chunks.push_back(c);
            }
            StagingPool::Lease hfl_lease;                    // page-locked: the flag words travel asynchronously too
            if (int rc = g_staging.take(mp3_blocks * sizeof(uint32_t), hfl_lease)) return rc;
            uint32_t *hfl = (uint32_t *)hfl_lease.p;
            hipStream_t up = nullptr, down = nullptr;
            hipError_t e = g_streams.take(&up, &down);
            int rc = AFG_OK;
            for (Chunk &c : chunks) {                        // plans first: their tables are uploaded synchronously
                if (rc || e != hipSuccess) break;
                std::vector<uint32_t> granules;
                std::vector<uint8_t> channels;
                for (size_t i = c.f0; i < c.f1; i++) {
                    const Parsed &p = parsed[i];
                    if (fmt_of(p) != AFG_FORMAT_MP3) continue;
                    for (uint32_t g : p.mp3.run_granules) {
                        granules.push_back(g);
                        channels.push_back((uint8_t)p.mp3.channels);
                    }
                    if (p.mp3.blocks()) std::memcpy(hfl + mp3_blk_base[i], p.mp3_flags(), p.mp3.blocks() * sizeof(uint32_t));
                }
                rc = afg_mp3_plan_create(&c.plan, (uint32_t)granules.size(), granules.data(), channels.data(), 0);
                if (!rc) e = hipEventCreateWithFlags(&c.done, hipEventDisableTiming);
                c.runs = granules.size();
            }
            tm.lap("mp3 plans");
            uint32_t *d_flags = (uint32_t *)((uint8_t *)d_in.p + coef_bytes);
            for (Chunk &c : chunks) {
                if (rc || e != hipSuccess) break;
                for (size_t i = c.f0; i < c.f1 && e == hipSuccess; i++) {
                    const Parsed &p = parsed[i];
                    if (fmt_of(p) != AFG_FORMAT_MP3 || !p.mp3.blocks()) continue;
                    // the batch path parsed this file straight into page-locked staging: one asynchronous copy per file
                    // into the packed device plane (a file parsed on its own comes from ordinary memory)
                    e = hipMemcpyAsync((float *)d_in.p + mp3_blk_base[i] * 576, p.mp3_coef(), p.mp3.blocks() * 576 * sizeof(float),
                                       hipMemcpyHostToDevice, up);
                }
                if (e == hipSuccess)
                    e = hipMemcpyAsync(d_flags + c.blk0, hfl + c.blk0, c.blocks * sizeof(uint32_t), hipMemcpyHostToDevice, up);
                if (e != hipSuccess) break;
                // chunked stream: one state blob per run of the chunk, zero (a fresh decoder) except the first when the chunk
                // goes on from the previous one; the last run's blob is what the next chunk goes on from
                DeviceBuf d_states;
                float *states = nullptr;
                if (carry && chunks.size() == 1 && c.runs) {
                    const size_t sb = AFG_MP3_STATE_FLOATS * sizeof(float);
                    if ((rc = d_states.alloc(c.runs * sb)) != AFG_OK) break;
                    states = (float *)d_states.p;
                    e = hipMemsetAsync(states, 0, c.runs * sb, up);
                    if (e == hipSuccess && carry->continues && carry->valid)
                        e = hipMemcpyAsync(states, carry->state.p, sb, hipMemcpyDeviceToDevice, up);
                    if (e != hipSuccess) break;
                }
                rc = afg_mp3_transform_hip(c.plan, (const float *)d_in.p + c.blk0 * 576, d_flags + c.blk0,
                                           (float *)d_pcm.p + c.blk0 * 576, states, up);
                if (rc) break;
                if (states) {
                    const size_t sb = AFG_MP3_STATE_FLOATS * sizeof(float);
                    if (!carry->state.p && (rc = carry->state.alloc(sb)) != AFG_OK) break;
                    e = hipMemcpyAsync(carry->state.p, states + (c.runs - 1) * AFG_MP3_STATE_FLOATS, sb, hipMemcpyDeviceToDevice, up);
                    if (e == hipSuccess) e = hipStreamSynchronize(up);          // d_states goes out of scope below
                    if (e != hipSuccess) break;
                    carry->valid = true;
                }
                e = hipEventRecord(c.done, up);
                if (e == hipSuccess) e = hipStreamWaitEvent(down, c.done, 0);
                // delivery: the copy plan of each file, merged into maximal contiguous pieces (one per undamaged file),
                // straight from the device PCM plane into the page-locked result plane
                for (size_t i = c.f0; i < c.f1 && e == hipSuccess; i++) {
                    const Parsed &p = parsed[i];
                    if (fmt_of(p) != AFG_FORMAT_MP3) continue;
                    const float *src = (const float *)d_pcm.p + mp3_blk_base[i] * 576;
                    float *dst = (float *)out.plane.p + out.files[i].pcm_off;
                    const std::vector<afg_mp3::Copy> &cp = p.mp3.copies;
                    for (size_t k = 0; k < cp.size() && e == hipSuccess;) {
                        uint64_t from = cp[k].src, cnt = cp[k].count;
                        size_t j = k + 1;
                        while (j < cp.size() && cp[j].src == from + cnt) cnt += cp[j++].count;
                        e = hipMemcpyAsync(dst, src + from, (size_t)cnt * sizeof(float), hipMemcpyDeviceToHost, down);
                        dst += cnt;
                        k = j;
                    }
                }
            }
            if (up) { hipError_t e2 = hipStreamSynchronize(up); if (e == hipSuccess) e = e2; }
            if (down) { hipError_t e2 = hipStreamSynchronize(down); if (e == hipSuccess) e = e2; }
            for (Chunk &c : chunks) {
                if (c.plan) afg_mp3_plan_destroy(c.plan);
                if (c.done) (void)hipEventDestroy(c.done);
            }
            g_streams.give(up, down);
            tm.lap("mp3 h2d | kernel | d2h (chunks overlapped)");
            if (rc) return rc;
            if (e != hipSuccess) { afg::set_error("MP3 stage failed: %s", hipGetErrorString(e)); return AFG_ERR_HIP; }
            tm.lap("mp3 delivery copies");
        }
        // ---- Vorbis: per chunk gather (host threads) -> upload + kernel on `up` -> download on `down` ----
        if (!ogg.v.empty()) {
            StagingPool::Lease h_spec;
            DeviceBuf d_spec, d_pcm;
            if (!ogg_staged)
                if (int rc = g_staging.take(ogg_spec * sizeof(float), h_spec)) return rc;
            if (int rc = d_spec.alloc((ogg_staged ? ogg_stage->floats : ogg_spec) * sizeof(float))) return rc;
            if (int rc = d_pcm.alloc(ogg_out * sizeof(float))) return rc;
            float *ogg_plane = (float *)out.plane.p + flac_out + qoa_out + mp3_out;
            hipStream_t up = nullptr, down = nullptr;
            hipError_t e = g_streams.take(&up, &down);
            int rc = AFG_OK;
            // Files parsed with the floor left to the device: their packets' coupling / floor records go up
            // with the chunk (one page-locked block: packets, curves, points, steps of chunk after chunk) and
            // afg_vorbis_floor_hip turns the residue vectors into spectra in place, in front of the transform.
            std::vector<size_t> f_pk(nf + 1, 0), f_cv(nf + 1, 0), f_pt(nf + 1, 0), f_st(nf + 1, 0);
            for (size_t i = 0; i < nf; i++) {
                const bool on = fmt_of(parsed[i]) == AFG_FORMAT_OGG && parsed[i].ogg.device_floor;
                f_pk[i + 1] = f_pk[i] + (on ? parsed[i].ogg.fl_packets.size() : 0);
                f_cv[i + 1] = f_cv[i] + (on ? parsed[i].ogg.fl_curves.size() : 0);
                f_pt[i + 1] = f_pt[i] + (on ? parsed[i].ogg.fl_points.size() / 2 : 0);
                f_st[i + 1] = f_st[i] + (on ? parsed[i].ogg.fl_steps.size() / 2 : 0);
            }
            const size_t fl_pk_bytes = f_pk[nf] * sizeof(afg_vorbis_floor_packet), fl_cv_bytes = f_cv[nf] * sizeof(afg_vorbis_floor_curve),
                         fl_pt_bytes = f_pt[nf] * 8, fl_st_bytes = f_st[nf] * 2;
            const size_t fl_bytes = fl_pk_bytes + fl_cv_bytes + fl_pt_bytes + fl_st_bytes;
            StagingPool::Lease h_fl;
            DeviceBuf d_fl;
            if (f_pk[nf]) {
                if (int rc2 = g_staging.take(fl_bytes, h_fl)) return rc2;
                if (int rc2 = d_fl.alloc(fl_bytes)) return rc2;
            }
            auto fl_at = [&](void *base0, int which, size_t index) -> uint8_t * {   // 0 packets, 1 curves, 2 points, 3 steps
                uint8_t *b = (uint8_t *)base0;
                if (which == 0) return b + index * sizeof(afg_vorbis_floor_packet);
                if (which == 1) return b + fl_pk_bytes + index * sizeof(afg_vorbis_floor_curve);
                if (which == 2) return b + fl_pk_bytes + fl_cv_bytes + index * 8;
                return b + fl_pk_bytes + fl_cv_bytes + fl_pt_bytes + index * 2;
            };
            for (size_t ci = 0; ci < ogg.v.size(); ci++) {
                OggChunk &c = ogg.v[ci];
                if (rc || e != hipSuccess) break;
                const float *hs = ogg_staged ? ogg_stage->spec + c.spec0 : (const float *)h_spec.p + c.spec0;
                const size_t npk_c = f_pk[c.f1] - f_pk[c.f0];
                if (npk_c) {
                    parallel_for(c.f1 - c.f0, threads, [&](size_t k) {
                        const size_t i = c.f0 + k;
                        const Parsed &p = parsed[i];
                        if (f_pk[i + 1] == f_pk[i]) return;
                        // chunk-local indices (the kernel gets the chunk's slices), absolute spectrum offsets
                        const size_t spec_base = ogg_staged ? ogg_stage->base[i] : c.spec0 + c.spec_at[k];
                        afg_vorbis_floor_packet *pk = (afg_vorbis_floor_packet *)fl_at(h_fl.p, 0, f_pk[i]);
                        for (size_t q = 0; q < p.ogg.fl_packets.size(); q++) {
                            afg_vorbis_floor_packet r = p.ogg.fl_packets[q];
                            r.spec_off += spec_base;
                            r.curve_index += (uint32_t)(f_cv[i] - f_cv[c.f0]);
                            r.step_off += (uint32_t)(f_st[i] - f_st[c.f0]);
                            pk[q] = r;
                        }
                        afg_vorbis_floor_curve *cv = (afg_vorbis_floor_curve *)fl_at(h_fl.p, 1, f_cv[i]);
                        for (size_t q = 0; q < p.ogg.fl_curves.size(); q++) {
                            afg_vorbis_floor_curve r = p.ogg.fl_curves[q];
                            r.point_off += (uint32_t)(f_pt[i] - f_pt[c.f0]);
                            cv[q] = r;
                        }
                        if (!p.ogg.fl_points.empty()) std::memcpy(fl_at(h_fl.p, 2, f_pt[i]), p.ogg.fl_points.data(), p.ogg.fl_points.size() * sizeof(int32_t));
                        if (!p.ogg.fl_steps.empty()) std::memcpy(fl_at(h_fl.p, 3, f_st[i]), p.ogg.fl_steps.data(), p.ogg.fl_steps.size());
                    });
                }
                if (!ogg_staged) {
                    float *hw = (float *)h_spec.p + c.spec0;
                    parallel_for(c.f1 - c.f0, threads, [&](size_t k) {
                        Parsed &p = parsed[c.f0 + k];
                        if (fmt_of(p) != AFG_FORMAT_OGG || !p.ogg.n_spec) return;
                        std::memcpy(hw + c.spec_at[k], p.ogg.spectra(), p.ogg.n_spec * sizeof(float));
                        std::vector<float>().swap(p.ogg.spec);         // the big one: released here, by many threads
                    });
                }
                e = hipMemcpyAsync((float *)d_spec.p + c.spec0, hs, c.spec_n * sizeof(float), hipMemcpyHostToDevice, up);
                if (e != hipSuccess) break;
                if (npk_c) {
                    const struct { int which; size_t i0, i1, unit; } part[4] = {
                        { 0, f_pk[c.f0], f_pk[c.f1], sizeof(afg_vorbis_floor_packet) }, { 1, f_cv[c.f0], f_cv[c.f1], sizeof(afg_vorbis_floor_curve) },
                        { 2, f_pt[c.f0], f_pt[c.f1], 8 }, { 3, f_st[c.f0], f_st[c.f1], 2 } };
                    for (const auto &pt : part) {
                        if (pt.i1 == pt.i0 || e != hipSuccess) continue;
                        e = hipMemcpyAsync(fl_at(d_fl.p, pt.which, pt.i0), fl_at(h_fl.p, pt.which, pt.i0), (pt.i1 - pt.i0) * pt.unit, hipMemcpyHostToDevice, up);
                    }
                    if (e != hipSuccess) break;
                    rc = afg_vorbis_floor_hip(npk_c, (const afg_vorbis_floor_packet *)fl_at(d_fl.p, 0, f_pk[c.f0]),
                                              (const afg_vorbis_floor_curve *)fl_at(d_fl.p, 1, f_cv[c.f0]), (const int32_t *)fl_at(d_fl.p, 2, f_pt[c.f0]),
                                              (const uint8_t *)fl_at(d_fl.p, 3, f_st[c.f0]), (float *)d_spec.p, up);
                    if (rc) break;
                }
                // a staged plan addresses the staging layout from float 0; a gathered one is packed from its chunk's start
                rc = afg_vorbis_transform_hip(c.plan, (const float *)d_spec.p + (ogg_staged ? 0 : c.spec0), (float *)d_pcm.p + c.out0, up);
                if (rc) break;
                e = hipEventCreateWithFlags(&c.done, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(c.done, up);
                if (e == hipSuccess) e = hipStreamWaitEvent(down, c.done, 0);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(ogg_plane + c.out0, (const float *)d_pcm.p + c.out0, c.out_n * sizeof(float), hipMemcpyDeviceToHost, down);
            }
            if (up) { hipError_t e2 = hipStreamSynchronize(up); if (e == hipSuccess) e = e2; }
            if (down) { hipError_t e2 = hipStreamSynchronize(down); if (e == hipSuccess) e = e2; }
            g_streams.give(up, down);
            if (rc) return rc;
            if (e != hipSuccess) { afg::set_error("Vorbis stage failed: %s", hipGetErrorString(e)); return AFG_ERR_HIP; }
            // files delivered as several runs (seek-style trims, damaged streams): close the runs up, in place
            for (size_t bi = 0, at = 0; bi < ogg_broken.size(); bi++) {
                const size_t i = ogg_broken[bi];
                while (at < ogg_pieces.size() && ogg_pieces[at].file != i) at++;
                float *dst = (float *)out.plane.p + out.files[i].pcm_off;
                for (; at < ogg_pieces.size() && ogg_pieces[at].file == i; at++) {
                    const float *src = ogg_plane + ogg_pieces[at].from;
                    if (dst != src) std::memmove(dst, src, (size_t)ogg_pieces[at].count * sizeof(float));
                    dst += ogg_pieces[at].count;
                }
            }
            tm.lap("vorbis gather | h2d | kernel | d2h (chunks overlapped)");
        }
        // ---- Opus (CELT): records + coefficients -> transform -> gain / int16 round trip -> result plane ----
        if (opus_out) {
            if (opus_seqs > 0xffffffffull) { afg::set_error("Opus stage: too many channel sequences"); return AFG_ERR_INVALID; }
            const size_t base_bytes = ((opus_seqs + 1) * sizeof(uint64_t) + 15) & ~(size_t)15;
            const size_t rec_bytes = (opus_recs * sizeof(afg_celt_frame) + 15) & ~(size_t)15;
            StagingPool::Lease h_in;
            DeviceBuf d_in, d_pcm;
            if (int rc = g_staging.take(base_bytes + rec_bytes + opus_coefs * sizeof(float), h_in)) return rc;
            if (int rc = d_in.alloc(base_bytes + rec_bytes + opus_coefs * sizeof(float))) return rc;
            if (int rc = d_pcm.alloc(opus_out * sizeof(float))) return rc;
            uint64_t *hb = (uint64_t *)h_in.p;
            afg_celt_frame *hr = (afg_celt_frame *)((uint8_t *)h_in.p + base_bytes);
            float *hc = (float *)((uint8_t *)h_in.p + base_bytes + rec_bytes);
            {
                size_t seq = 0;
                for (size_t i = 0; i < nf; i++) {
                    const Parsed &p = parsed[i];
                    if (fmt_of(p) != AFG_FORMAT_OPUS) continue;
                    if (p.opus.channels == 2 && (seq & 1)) hb[seq++] = opus_rec_base[i];          // the empty sequence (above)
                    for (int c = 0; c < p.opus.channels; c++) hb[seq++] = opus_rec_base[i] + (size_t)c * p.opus.frames.size();
                }
                hb[seq] = opus_recs;
            }
            parallel_for(nf, threads, [&](size_t i) {
                Parsed &p = parsed[i];
                if (fmt_of(p) != AFG_FORMAT_OPUS) return;
                const size_t n = p.opus.frames.size();
                for (int c = 0; c < p.opus.channels; c++)
                    for (size_t k = 0; k < n; k++) {
                        afg_celt_frame f = p.opus.frames[k];
                        f.coef_off += opus_coef_base[i] + (uint64_t)c * f.frame_size;
                        f.out_off += opus_pcm_base[i] + (uint64_t)c;
                        hr[opus_rec_base[i] + (size_t)c * n + k] = f;
                    }
                if (!p.opus.coeffs.empty()) std::memcpy(hc + opus_coef_base[i], p.opus.coeffs.data(), p.opus.coeffs.size() * sizeof(float));
                std::vector<float>().swap(p.opus.coeffs);
            });
            AFG_HIP_CHECK(hipMemcpyAsync(d_in.p, h_in.p, base_bytes + rec_bytes + opus_coefs * sizeof(float), hipMemcpyHostToDevice, stream));
            // chunked stream (one file): the channel states live on the device between chunks, zero for a fresh decoder
            float *states = nullptr;
            if (opus_carry) {
                const size_t sb = opus_seqs * AFG_CELT_STATE_FLOATS * sizeof(float);
                if (!opus_carry->states.p) {
                    if (int rc = opus_carry->states.alloc(sb)) return rc;
                    opus_carry->valid = false;
                }
                if (!opus_carry->valid) AFG_HIP_CHECK(hipMemsetAsync(opus_carry->states.p, 0, sb, stream));
                opus_carry->valid = true;
                states = (float *)opus_carry->states.p;
            }
            if (int rc = afg_celt_transform_hip((uint32_t)opus_seqs, (const uint64_t *)d_in.p, (const afg_celt_frame *)((const uint8_t *)d_in.p + base_bytes),
                                                (const float *)((const uint8_t *)d_in.p + base_bytes + rec_bytes), (float *)d_pcm.p, states, stream))
                return rc;
            // output gain (when the file asks for one) and the reference's int16 round trip, in place
            bool any_gain = false;
            for (size_t i = 0; i < nf; i++) any_gain = any_gain || (fmt_of(parsed[i]) == AFG_FORMAT_OPUS && parsed[i].opus.gain_i != 0);
            if (!any_gain) {
                if (int rc = afg_opus_output_hip(opus_out, (const float *)d_pcm.p, nullptr, (float *)d_pcm.p, stream)) return rc;
            } else {
                for (size_t i = 0; i < nf; i++) {
                    const Parsed &p = parsed[i];
                    if (fmt_of(p) != AFG_FORMAT_OPUS) continue;
                    float *at = (float *)d_pcm.p + opus_pcm_base[i];
                    const uint64_t n = p.opus.pcm_frames * (uint64_t)p.opus.channels;
                    const int rc = p.opus.gain_i ? afg_opus_output_gain_hip(n, at, p.opus.gain, nullptr, at, stream)
                                                 : afg_opus_output_hip(n, at, nullptr, at, stream);
                    if (rc) return rc;
                }
            }
            AFG_HIP_CHECK(hipMemcpyAsync((float *)out.plane.p + flac_out + qoa_out + mp3_out + ogg_out, d_pcm.p, opus_out * sizeof(float),
                                         hipMemcpyDeviceToHost, stream));
            AFG_HIP_CHECK(hipStreamSynchronize(stream));
            tm.lap("opus gather | h2d | kernel | d2h");
        }
    }
metadata:
    for (size_t i = 0; i < nf; i++) {
        Parsed &p = parsed[i];
        Decoded &dcd = out.files[i];
        dcd.format = fmt_of(p);
        if (fmt_of(p) == AFG_FORMAT_FLAC) {
            dcd.channels = (int)p.fi.channels;
            dcd.samplerate = (float)p.fi.sample_rate;
            dcd.frames = (int64_t)(p.flac.out_samples / p.fi.channels);
            dcd.declared_frames = (int64_t)p.fi.total_samples;      // totalSampleCount / channels, stream.d:1631
        } else if (fmt_of(p) == AFG_FORMAT_MP3) {
            dcd.channels = p.mp3.channels;
            dcd.samplerate = (float)p.mp3.hz;
            dcd.frames = (int64_t)(p.mp3.pcm_samples / (uint64_t)p.mp3.channels);
            dcd.declared_frames = (int64_t)(p.mp3.declared_samples / (uint64_t)p.mp3.channels);   // stream.d:1737
        } else if (fmt_of(p) == AFG_FORMAT_OGG) {
            dcd.channels = p.ogg.channels;
            dcd.samplerate = (float)p.ogg.sample_rate;
            dcd.frames = (int64_t)p.ogg.pcm_frames;
            dcd.declared_frames = (int64_t)p.ogg.total_samples;    // stb_vorbis_stream_length_in_samples, stream.d:1696
        } else if (fmt_of(p) == AFG_FORMAT_OPUS) {
            dcd.channels = p.opus.channels;
            dcd.samplerate = 48000.0f;                              // OpusFileCtx.rate (dopus.d:7973)
            dcd.declared_frames = p.opus.declared_frames;           // smpduration(), stream.d:1609
            // the reference never reads past the declared length (stream.d:439-442); the pre-skip samples are not dropped
            dcd.frames = std::min<int64_t>((int64_t)p.opus.pcm_frames, std::max<int64_t>(p.opus.declared_frames, 0));
            if (p.opus.error) {                                     // a packet failed: the read that reaches it reports an error
                dcd.status = AFG_ERR_INVALID;
                dcd.message = kErrorDecoderInitializationFailed;    // (the string stream.d:454 sets)
            }
        } else if (fmt_of(p) == AFG_FORMAT_QOA) {
            dcd.channels = (int)p.qi.channels;
            dcd.samplerate = (float)p.qi.samplerate;
            dcd.frames = (int64_t)(p.qoa.back().out_off / p.qi.channels) + p.qoa.back().samples;
            dcd.declared_frames = (int64_t)p.qi.samples;
        } else {
            dcd.status = AFG_ERR_UNSUPPORTED;
            dcd.message = p.opus_mode ? kErrorOpusMode : kErrorUnknownFormat;
        }
    }
    return AFG_OK;
}

}  // namespace

// The AudioStream surface decodes as the caller pulls (stream.d:429-637 keeps O(1) state per stream): open parses
// the container only, a read decodes the next chunk of frames (about 1.5 s of audio) on the device when the FIFO of
// delivered samples runs dry.  Memory per handle: the file bytes (the reference copies them too, stream.d:2031-2041),
// one chunk of PCM, and for MP3 the 6 KB decoder state that lives on the device between chunks.
struct afg_stream {
    const char *error = kErrorNotInitialized;      // stream.d:1379
    int format = AFG_FORMAT_UNKNOWN, channels = 0;
    float samplerate = 0;
    int64_t declared_frames = AFG_UNKNOWN_LENGTH;
    std::vector<uint8_t> bytes;
    // per-format readers
    FlacInfo fi;
    size_t flac_pos = 0;
    QoaInfo qi;
    std::vector<afg_qoa_frame> qoa;
    size_t qoa_next = 0;
    std::unique_ptr<afg_mp3::Reader> mp3;
    std::unique_ptr<afg_vorbis::Reader> ogg;
    std::unique_ptr<afg_opus::Reader> opus;
    Mp3Carry carry;
    OpusCarry opus_carry;
    int opus_gain_i = 0;
    float opus_gain = 1.0f;
    int64_t opus_decoded = 0;           // frames decoded so far (the declared length cuts the delivery, stream.d:439-442)
    bool opus_failed = false;           // a packet could not be framed: the next refill reports it (stream.d:452-456)
    // delivered samples not yet read
    std::vector<float> fifo;
    size_t fifo_at = 0;                 // floats of `fifo` already handed out
    int64_t position = 0;               // frames handed out so far (tellPosition)
    bool ended = false;                 // nothing further can be decoded

    static constexpr int kMp3Frames = 64, kOggPackets = 64, kFlacFrames = 16, kQoaFrames = 16, kOpusPackets = 64;

    // (re)start the readers at the head of the stream
    bool rewind()
    {
        fifo.clear();
        fifo_at = 0;
        position = 0;
        ended = false;
        flac_pos = 0;
        qoa_next = 0;
        carry.valid = carry.continues = false;
        opus_carry.valid = false;
        opus_decoded = 0;
        opus_failed = false;
        if (format == AFG_FORMAT_OPUS) {
            afg_opus::File meta;
            opus.reset(new afg_opus::Reader);
            return opus->open(bytes.data(), bytes.size(), meta) == afg_opus::kOpened;
        }
        if (format == AFG_FORMAT_MP3) {
            afg_mp3::File meta;
            mp3.reset(new afg_mp3::Reader);
            return mp3->open(bytes.data(), bytes.size(), meta);
        }
        if (format == AFG_FORMAT_OGG) {
            afg_vorbis::File meta;
            ogg.reset(new afg_vorbis::Reader);
            return ogg->open(bytes.data(), bytes.size(), meta, vorbis_floor_on_device());
        }
        return true;
    }

    // decode the next chunk into the FIFO; false: end of stream (or error state set)
    bool refill()
    {
        if (ended) return false;
        if (fifo_at == fifo.size()) { fifo.clear(); fifo_at = 0; }
        std::vector<Parsed> parsed(1);
        Parsed &p = parsed[0];
        const uint8_t *dp[1] = { bytes.data() };
        size_t lp[1] = { bytes.size() };
        Mp3Carry *cr = nullptr;
        OpusCarry *ocr = nullptr;
        if (format == AFG_FORMAT_OPUS) {
            if (opus_failed) { error = kErrorDecoderInitializationFailed; ended = true; return false; }      // stream.d:454
            if (opus_decoded >= declared_frames || !opus->more(p.opus, kOpusPackets)) { ended = true; return false; }
            if (p.opus.error) opus_failed = true;
            p.opus.error = false;                               // this chunk's good frames are delivered first
            if (p.opus.frames.empty()) { error = kErrorDecoderInitializationFailed; ended = true; return false; }
            p.opus.channels = channels;
            p.opus.gain_i = opus_gain_i;
            p.opus.gain = opus_gain;
            p.opus.declared_frames = declared_frames - opus_decoded;
            opus_decoded += (int64_t)p.opus.pcm_frames;
            ocr = &opus_carry;
            p.format = AFG_FORMAT_OPUS;
        } else if (format == AFG_FORMAT_FLAC) {
            bool done = false;
            p.fi = fi;
            p.flac.pack16 = flac_rows_int16();
            const int got = flac_parse_frames(bytes.data(), bytes.size(), fi, p.flac, &flac_pos, kFlacFrames, &done);
            if (done) ended = true;
            if (!got) { ended = true; return false; }
            p.format = AFG_FORMAT_FLAC;
        } else if (format == AFG_FORMAT_QOA) {
            if (qoa_next >= qoa.size()) { ended = true; return false; }
            const size_t k1 = std::min(qoa.size(), qoa_next + (size_t)kQoaFrames);
            const uint64_t byte0 = qoa[qoa_next].byte_off, out0 = qoa[qoa_next].out_off;
            for (size_t k = qoa_next; k < k1; k++) {
                afg_qoa_frame f = qoa[k];
                f.byte_off -= byte0;
                f.out_off -= out0;
                p.qoa.push_back(f);
            }
            const uint64_t byte1 = k1 < qoa.size() ? qoa[k1].byte_off : (uint64_t)bytes.size();
            dp[0] = bytes.data() + byte0;
            lp[0] = (size_t)(byte1 - byte0);
            p.qi = qi;
            p.format = AFG_FORMAT_QOA;
            qoa_next = k1;
            if (qoa_next >= qoa.size()) ended = true;
        } else if (format == AFG_FORMAT_OGG) {
            if (!ogg->more(p.ogg, kOggPackets)) { ended = true; return false; }
            p.format = AFG_FORMAT_OGG;
        } else if (format == AFG_FORMAT_MP3) {
            bool continues = false;
            if (!mp3->more(p.mp3, kMp3Frames, &continues)) { ended = true; return false; }
            carry.continues = continues;
            cr = &carry;
            p.format = AFG_FORMAT_MP3;
        } else {
            ended = true;
            return false;
        }
        BatchOut out;
        if (decode_parsed(parsed, dp, lp, 1, out, nullptr, nullptr, nullptr, nullptr, cr, ocr) != AFG_OK || out.files[0].status != AFG_OK) {
            error = kErrorDecodingError;
            ended = true;
            return false;
        }
        const Decoded &d = out.files[0];
        const size_t want = (size_t)std::max<int64_t>(d.frames, 0) * (size_t)channels;
        if (want) {
            if (d.pcm_off + want > out.plane_floats) { error = kErrorDecodingError; ended = true; return false; }
            const float *src = (const float *)out.plane.p + d.pcm_off;
            fifo.insert(fifo.end(), src, src + want);
        }
        return true;
    }
};

extern "C" {

afg_stream *afg_open_from_memory(const uint8_t *data, size_t length)
{
    afg_stream *s = new (std::nothrow) afg_stream;
    if (!s) return nullptr;
    if (!data || length == 0) { s->error = kErrorUnknownFormat; return s; }
    try {
        s->bytes.assign(data, data + length);
        const uint8_t *d = s->bytes.data();
        // startDecoding's probe order for the formats handled here (stream.d:1586-1838): FLAC, QOA, OGG, then MP3
        afg_mp3::File m3;
        afg_vorbis::File og;
        afg_opus::File op;
        // the reference tries Opus before anything else (stream.d:1596-1614)
        afg_opus::Status ost = afg_opus::kNotOpus;
        if (length >= 4 && std::memcmp(d, "OggS", 4) == 0) {
            s->opus.reset(new afg_opus::Reader);
            ost = s->opus->open(d, length, op);
        }
        if (ost == afg_opus::kUnsupported) {
            s->error = kErrorOpusMode;
            return s;
        }
        if (ost == afg_opus::kOpened) {
            s->format = AFG_FORMAT_OPUS;
            s->channels = op.channels;
            s->samplerate = 48000.0f;                                     // OpusFileCtx.rate, stream.d:1607
            s->declared_frames = op.declared_frames;                      // smpduration(), stream.d:1609
            s->opus_gain_i = op.gain_i;
            s->opus_gain = op.gain;
        } else if (flac_open_info(d, length, s->fi)) {
            s->format = AFG_FORMAT_FLAC;
            s->channels = (int)s->fi.channels;
            s->samplerate = (float)s->fi.sample_rate;
            s->declared_frames = (int64_t)s->fi.total_samples;            // totalSampleCount / channels, stream.d:1631
        } else if (qoa_parse(d, length, s->qi, s->qoa)) {
            s->format = AFG_FORMAT_QOA;
            s->channels = (int)s->qi.channels;
            s->samplerate = (float)s->qi.samplerate;
            s->declared_frames = (int64_t)s->qi.samples;
        } else if ((s->ogg.reset(new afg_vorbis::Reader), s->ogg->open(d, length, og, vorbis_floor_on_device()))) {
            s->format = AFG_FORMAT_OGG;
            s->channels = og.channels;
            s->samplerate = (float)og.sample_rate;
            s->declared_frames = (int64_t)og.total_samples;               // stb_vorbis_stream_length_in_samples, stream.d:1696
        } else if (afg_mp3::looks_like_mp3(d, length) && (s->mp3.reset(new afg_mp3::Reader), s->mp3->open(d, length, m3))) {
            s->format = AFG_FORMAT_MP3;
            s->channels = m3.channels;
            s->samplerate = (float)m3.hz;
            s->declared_frames = (int64_t)(m3.declared_samples / (uint64_t)std::max(1, m3.channels));   // stream.d:1737
        } else {
            s->error = kErrorUnknownFormat;
            return s;
        }
        if (afg::require_device() != AFG_OK) { s->error = kErrorDecoderInitializationFailed; return s; }
        s->error = nullptr;
    } catch (...) {
        s->error = kErrorDecoderInitializationFailed;      // out of memory
    }
    return s;
}

int afg_is_error(const afg_stream *s) { return !s || s->error != nullptr; }
const char *afg_error_message(const afg_stream *s) { return s ? s->error : kErrorNotInitialized; }
int afg_get_format(const afg_stream *s) { return (s && !s->error) ? s->format : AFG_FORMAT_UNKNOWN; }
int afg_get_num_channels(const afg_stream *s) { return (s && !s->error) ? s->channels : 0; }
float afg_get_samplerate(const afg_stream *s) { return (s && !s->error) ? s->samplerate : 0.0f; }

int64_t afg_get_length_in_frames(const afg_stream *s)
{
    if (!s || s->error) return AFG_UNKNOWN_LENGTH;
    return s->declared_frames;         // stream.d:404-407: whatever the container declares (FLAC: may be 0)
}

int afg_read_samples_float(afg_stream *s, float *out, int frames)
{
    if (!s || s->error || frames <= 0) return 0;
    // stream.d:498: a FLAC stream stops once the position equals the declared length (a STREAMINFO that
    // declares 0 samples therefore reads nothing); the check is made on entry only, like the reference's.
    if (s->format == AFG_FORMAT_FLAC && s->position == s->declared_frames) return 0;
    try {
        const size_t C = (size_t)std::max(1, s->channels);
        int done = 0;
        while (done < frames) {
            if (s->fifo_at == s->fifo.size() && !s->refill()) break;
            const size_t avail = (s->fifo.size() - s->fifo_at) / C;
            const size_t n = std::min<size_t>(avail, (size_t)(frames - done));
            if (out && n) std::memcpy(out + (size_t)done * C, s->fifo.data() + s->fifo_at, n * C * sizeof(float));
            s->fifo_at += n * C;
            done += (int)n;
        }
        if (s->error && s->format == AFG_FORMAT_OPUS) return 0;   // stream.d:452-456: the failing read returns 0
        s->position += done;
        return done;
    } catch (...) {
        s->error = kErrorDecodingError;
        return 0;
    }
}

int afg_can_seek(const afg_stream *s) { return s && !s->error; }

int afg_seek_position(afg_stream *s, int frame)
{
    if (!s || s->error) return 0;
    // the reference bounds a seek by the declared length (stream.d:1104, :1113, :1137); what can actually be reached is
    // bounded by what decodes.  Backwards: the readers start over; forwards: chunks are decoded and dropped (a chunk
    // is ~1.5 s of audio and takes about a millisecond on the device).
    const int64_t limit = std::max<int64_t>(s->declared_frames, 0);
    if (frame < 0 || frame > limit) return 0;
    try {
        if (frame < s->position) {
            // still inside the FIFO?  then just step back
            const size_t C = (size_t)std::max(1, s->channels);
            const int64_t fifo_first = s->position - (int64_t)(s->fifo_at / C);
            if (frame >= fifo_first) {
                s->fifo_at -= (size_t)(s->position - frame) * C;
                s->position = frame;
                return 1;
            }
            if (!s->rewind()) { s->error = kErrorDecodingError; return 0; }
        }
        const size_t C = (size_t)std::max(1, s->channels);
        while (s->position < frame) {
            if (s->fifo_at == s->fifo.size() && !s->refill()) break;
            const size_t avail = (s->fifo.size() - s->fifo_at) / C;
            const size_t n = std::min<size_t>(avail, (size_t)(frame - s->position));
            s->fifo_at += n * C;
            s->position += (int64_t)n;
        }
        return s->error ? 0 : 1;
    } catch (...) {
        s->error = kErrorDecodingError;
        return 0;
    }
}

int afg_tell_position(const afg_stream *s) { return (s && !s->error) ? (int)s->position : -1; }

void afg_close(afg_stream *s) { delete s; }

namespace {
struct FlacParsedOwner {
    FlacRecords rec;
};
}  // namespace

int afg_flac_parse(const uint8_t *data, size_t length, afg_flac_parsed *out)
{
    try {
        if (!out) return AFG_ERR_INVALID;
        std::memset(out, 0, sizeof(*out));
        if (!data) return AFG_ERR_INVALID;
        std::unique_ptr<FlacParsedOwner> own_guard(new (std::nothrow) FlacParsedOwner);               // (freed if the parser throws)
        auto *own = own_guard.get();
        if (!own) return AFG_ERR_OOM;
        FlacInfo fi;
        if (!flac_parse(data, length, fi, own->rec)) {
            own_guard.reset();
            afg::set_error("afg_flac_parse: not a native FLAC stream");
            return AFG_ERR_UNSUPPORTED;
        }
        out->sample_rate = fi.sample_rate;
        out->channels = fi.channels;
        out->bps = fi.bps;
        out->max_block = fi.max_block;
        out->total_samples = fi.total_samples;
        out->n_frames = own->rec.frames.size();
        out->n_subframes = own->rec.subframes.size();
        out->n_res = own->rec.res.size();
        out->out_samples = own->rec.out_samples;
        out->frames = own->rec.frames.data();
        out->subframes = own->rec.subframes.data();
        out->res = own->rec.res.data();
        out->owner = own_guard.release();
        return AFG_OK;
    } catch (...) {
        afg::set_error("out of host memory");
        return AFG_ERR_OOM;
    }
}

void afg_flac_parsed_free(afg_flac_parsed *p)
{
    if (!p) return;
    delete (FlacParsedOwner *)p->owner;
    std::memset(p, 0, sizeof(*p));
}

int afg_mp3_parse(const uint8_t *data, size_t length, afg_mp3_parsed *out)
{
    try {
        if (!out) return AFG_ERR_INVALID;
        std::memset(out, 0, sizeof(*out));
        if (!data) return AFG_ERR_INVALID;
        std::unique_ptr<afg_mp3::File> own_guard(new (std::nothrow) afg_mp3::File);               // (freed if the parser throws)
        auto *own = own_guard.get();
        if (!own) return AFG_ERR_OOM;
        if (!afg_mp3::parse_file(data, length, *own)) {
            own_guard.reset();
            afg::set_error("afg_mp3_parse: no MPEG Layer III stream found");
            return AFG_ERR_UNSUPPORTED;
        }
        static_assert(sizeof(afg_mp3::Copy) == sizeof(afg_mp3_copy), "copy plan layout");
        out->channels = own->channels;
        out->hz = own->hz;
        out->tagged = own->tagged ? 1 : 0;
        out->start_delay = own->start_delay;
        out->detected_samples = own->detected_samples;
        out->declared_samples = own->declared_samples;
        out->pcm_samples = own->pcm_samples;
        out->n_runs = own->run_granules.size();
        out->n_blocks = own->blocks();
        out->n_copies = own->copies.size();
        out->run_granules = own->run_granules.data();
        out->coef = own->coef.data();
        out->flags = own->flags.data();
        out->copies = (afg_mp3_copy *)own->copies.data();
        out->owner = own_guard.release();
        return AFG_OK;
    } catch (...) {
        afg::set_error("out of host memory");
        return AFG_ERR_OOM;
    }
}

void afg_mp3_parsed_free(afg_mp3_parsed *p)
{
    if (!p) return;
    delete (afg_mp3::File *)p->owner;
    std::memset(p, 0, sizeof(*p));
}

int afg_mp3_parse_q(const uint8_t *data, size_t length, afg_mp3_parsed_q *out)
{
    try {
        if (!out) return AFG_ERR_INVALID;
        std::memset(out, 0, sizeof(*out));
        if (!data) return AFG_ERR_INVALID;
        std::unique_ptr<afg_mp3::File> own_guard(new (std::nothrow) afg_mp3::File);               // (freed if the parser throws)
        auto *own = own_guard.get();
        if (!own) return AFG_ERR_OOM;
        own->quantised = true;
        if (!afg_mp3::parse_file(data, length, *own)) {
            own_guard.reset();
            afg::set_error("afg_mp3_parse_q: no MPEG Layer III stream found");
            return AFG_ERR_UNSUPPORTED;
        }
        if (own->q_unsupported) {
            own_guard.reset();
            afg::set_error("afg_mp3_parse_q: the stream holds granules the device requantiser does not cover (MPEG-2.5 8 kHz mixed blocks, or a mono frame with the intensity bit set)");
            return AFG_ERR_UNSUPPORTED;
        }
        afg_mp3_parsed &b = out->base;
        b.channels = own->channels;
        b.hz = own->hz;
        b.tagged = own->tagged ? 1 : 0;
        b.start_delay = own->start_delay;
        b.detected_samples = own->detected_samples;
        b.declared_samples = own->declared_samples;
        b.pcm_samples = own->pcm_samples;
        b.n_runs = own->run_granules.size();
        b.n_blocks = own->blocks();
        b.n_copies = own->copies.size();
        b.run_granules = own->run_granules.data();
        b.coef = nullptr;
        b.flags = own->flags.data();
        b.copies = (afg_mp3_copy *)own->copies.data();
        b.owner = own;                                  // (released from the guard at the end)
        out->n_granules = own->qgr.size();
        out->n_sdesc = own->sdesc.size();
        out->q = own->q.data();
        out->granules = own->qgr.data();
        out->sdesc = own->sdesc.data();
        own_guard.release();
        return AFG_OK;
    } catch (...) {
        afg::set_error("out of host memory");
        return AFG_ERR_OOM;
    }
}

void afg_mp3_qtables(uint8_t band_of_line[24][576], uint16_t dst_of_src[24][576], float pow43[145])
{
    const afg_mp3::QTables &t = afg_mp3::qtables();
    if (band_of_line) std::memcpy(band_of_line, t.band_of_line, sizeof(t.band_of_line));
    if (dst_of_src) std::memcpy(dst_of_src, t.dst_of_src, sizeof(t.dst_of_src));
    if (pow43) std::memcpy(pow43, t.pow43, sizeof(t.pow43));
}

void afg_mp3_parsed_q_free(afg_mp3_parsed_q *p)
{
    if (!p) return;
    delete (afg_mp3::File *)p->base.owner;
    std::memset(p, 0, sizeof(*p));
}

static int vorbis_parse_any(const uint8_t *data, size_t length, afg_vorbis_parsed *out, bool device_floor)
{
    try {
        if (!out) return AFG_ERR_INVALID;
        std::memset(out, 0, sizeof(*out));
        if (!data) return AFG_ERR_INVALID;
        std::unique_ptr<afg_vorbis::File> own_guard(new (std::nothrow) afg_vorbis::File);               // (freed if the parser throws)
        auto *own = own_guard.get();
        if (!own) return AFG_ERR_OOM;
        if (!afg_vorbis::parse_file(data, length, *own, device_floor)) {
            own_guard.reset();
            afg::set_error("afg_vorbis_parse: not an Ogg Vorbis I stream");
            return AFG_ERR_UNSUPPORTED;
        }
        out->channels = own->channels;
        out->blocksize0 = own->blocksize0;
        out->blocksize1 = own->blocksize1;
        out->sample_rate = own->sample_rate;
        out->total_samples = own->total_samples;
        out->n_packets = own->pflags.size();
        out->spec_floats = own->spec.size();
        out->pcm_frames = own->pcm_frames;
        out->pflags = own->pflags.data();
        out->spec = own->spec.data();
        out->take_from = own->take_from.data();
        out->take_count = own->take_count.data();
        out->owner = own_guard.release();
        return AFG_OK;
    } catch (...) {
        afg::set_error("out of host memory");
        return AFG_ERR_OOM;
    }
}

int afg_vorbis_parse(const uint8_t *data, size_t length, afg_vorbis_parsed *out) { return vorbis_parse_any(data, length, out, false); }

void afg_vorbis_parsed_free(afg_vorbis_parsed *p)
{
    if (!p) return;
    delete (afg_vorbis::File *)p->owner;
    std::memset(p, 0, sizeof(*p));
}

int afg_vorbis_parse_r(const uint8_t *data, size_t length, afg_vorbis_parsed_r *out)
{
    if (!out) return AFG_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    if (int rc = vorbis_parse_any(data, length, &out->base, true)) return rc;
    afg_vorbis::File *own = (afg_vorbis::File *)out->base.owner;
    out->n_curves = own->fl_curves.size();
    out->n_points = own->fl_points.size() / 2;
    out->n_steps = own->fl_steps.size() / 2;
    out->packets = own->fl_packets.data();
    out->curves = own->fl_curves.data();
    out->points = own->fl_points.data();
    out->steps = own->fl_steps.data();
    return AFG_OK;
}

void afg_vorbis_parsed_r_free(afg_vorbis_parsed_r *p)
{
    if (!p) return;
    afg_vorbis_parsed_free(&p->base);
    std::memset(p, 0, sizeof(*p));
}

int afg_opus_parse(const uint8_t *data, size_t length, afg_opus_parsed *out)
{
    try {
        if (!out) return AFG_ERR_INVALID;
        std::memset(out, 0, sizeof(*out));
        if (!data) return AFG_ERR_INVALID;
        std::unique_ptr<afg_opus::File> own_guard(new (std::nothrow) afg_opus::File);               // (freed if the parser throws)
        auto *own = own_guard.get();
        if (!own) return AFG_ERR_OOM;
        const afg_opus::Status st = afg_opus::parse_file(data, length, *own);
        if (st != afg_opus::kOpened) {
            if (st == afg_opus::kUnsupported) {
                out->channels = own->channels;
                out->preskip = own->preskip;
                afg::set_error("afg_opus_parse: the stream holds SILK or hybrid packets (only CELT-only Opus is decoded)");
            } else {
                afg::set_error("afg_opus_parse: not an Ogg Opus stream");
            }
            own_guard.reset();
            return AFG_ERR_UNSUPPORTED;
        }
        out->channels = own->channels;
        out->preskip = own->preskip;
        out->gain_i = own->gain_i;
        out->error = own->error ? 1 : 0;
        out->gain = own->gain;
        out->declared_frames = own->declared_frames;
        out->pcm_frames = own->pcm_frames;
        out->n_frames = own->frames.size();
        out->n_coeffs = own->coeffs.size();
        out->frames = own->frames.data();
        out->coeffs = own->coeffs.data();
        out->owner = own_guard.release();
        return AFG_OK;
    } catch (...) {
        afg::set_error("out of host memory");
        return AFG_ERR_OOM;
    }
}

void afg_opus_parsed_free(afg_opus_parsed *p)
{
    if (!p) return;
    delete (afg_opus::File *)p->owner;
    std::memset(p, 0, sizeof(*p));
}

int afg_qoa_parse(const uint8_t *data, size_t length, uint32_t *channels, uint32_t *samplerate, uint32_t *samples,
                  afg_qoa_frame *frames, size_t frame_cap, size_t *n_frames)
{
    if (!data) return AFG_ERR_INVALID;
    QoaInfo qi;
    std::vector<afg_qoa_frame> fr;
    if (!qoa_parse(data, length, qi, fr)) {
        afg::set_error("afg_qoa_parse: not a QOA file");
        return AFG_ERR_UNSUPPORTED;
    }
    if (channels) *channels = qi.channels;
    if (samplerate) *samplerate = qi.samplerate;
    if (samples) *samples = qi.samples;
    if (n_frames) *n_frames = fr.size();
    if (frames) std::memcpy(frames, fr.data(), std::min(frame_cap, fr.size()) * sizeof(afg_qoa_frame));
    return AFG_OK;
}

}  // extern "C"

namespace {

// The CPU time the process may actually use: a container's cgroup quota in CPUs (0: none / unknown).
double cpu_quota()
{
    static const double quota = [] {
        double q = 0;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                    // cgroup v2: "<quota|max> <period>"
            char w[32] = { 0 };
            double period = 0;
            if (std::fscanf(f, "%31s %lf", w, &period) == 2 && period > 0 && std::strcmp(w, "max") != 0) q = std::atof(w) / period;
            std::fclose(f);
        } else if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { // cgroup v1
            double us = -1, period = 100000;
            if (std::fscanf(g, "%lf", &us) != 1) us = -1;
            std::fclose(g);
            if (FILE *h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(h, "%lf", &period) != 1) period = 100000;
                std::fclose(h);
            }
            if (us > 0 && period > 0) q = us / period;
        }
        return q;
    }();
    return quota;
}

// Host threads of a batch when the caller leaves the choice to the library: one per physical core of an SMT-2 host (with one
// per logical CPU the parse stages ran up to 10x longer on a shared 256-CPU box: the stragglers wait for a CPU) -- and under a
// cgroup quota ONE AND A HALF times the quota's CPUs.  The GPU boxes of this project show 256 logical CPUs behind a 16-CPU
// quota.  What counts for back-to-back calls -- what a service does and what bench.py times -- is the CPU seconds a call
// costs, and threads the quota cannot run cost more of them (descheduled in the middle of a file, cold caches): round 6, 2048-file
// batches, CPU seconds per call with 16 / 24 / 32 / 64 threads: FLAC 0.27 / 0.28 / 0.31 / 0.37, Vorbis 1.09 / 1.16 / 1.24 / 1.39,
// and MP3 5.8e9 samples/s with 64 threads, 7.0e9 with 24 (tools/gpu_r06_e2e_threads.sh, gpu_r06_groups.sh).  Rounds 3-5 used four
// times the quota, sized on single calls that finish inside the burst the quota allows.
unsigned default_threads()
{
    static const unsigned n = [] {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        unsigned t = hw >= 16 ? hw / 2 : hw;
        const double quota = cpu_quota();
        if (quota > 0) t = std::min(t, std::max(1u, (unsigned)(1.5 * quota + 0.5)));
        return std::max(1u, t);
    }();
    return n;
}

// ... and for a stage that keeps every CPU busy for half a second by itself (the Opus range / PVQ decode of a 2048-file batch):
// twice the quota (1.7e9 samples/s with 128 threads, 2.3e9 with 32).
unsigned sustained_threads()
{
    static const unsigned n = [] {
        unsigned t = default_threads();
        const double quota = cpu_quota();
        if (quota > 0) t = std::min(t, std::max(1u, (unsigned)(2 * quota + 0.5)));
        return std::max(1u, t);
    }();
    return n;
}

// What a batch result owns: one BatchOut per device the batch ran on.
struct BatchOwner {
    std::vector<std::unique_ptr<BatchOut>> parts;
};

// The whole batch path for the files handed in, on the calling thread's current device; fills items[0..n_files).
// `parse_done` (optional) is called once, when the host passes that keep every helper thread busy are over and what remains is
// the device stages of decode_parsed: a grouped batch lets its next group start parsing then.
int batch_decode_device(const uint8_t *const *data, const size_t *length, int n_files, int n_threads, afg_batch_item *items,
                        std::unique_ptr<BatchOut> &keep, const std::function<void()> *parse_done = nullptr)
{
    {
        if (int rc = afg::require_device()) return rc;
        int cur_dev = 0;
        AFG_HIP_CHECK(hipGetDevice(&cur_dev));
        StageTimer tm;
        struct ExitLap { StageTimer *t; const char *what; ~ExitLap() { t->lap(what); } } exit_lap{ &tm, "records released (call ends)" };
        std::vector<Parsed> parsed((size_t)n_files);
        tm.lap("call set-up");
        // default: one thread per physical core of an SMT-2 host (half the logical CPUs).  With one thread per logical
        // CPU the parse stages ran up to 10x longer on a shared 256-CPU box: the stragglers wait for a CPU.
        const unsigned nt = n_threads > 0 ? (unsigned)n_threads : default_threads();
        const unsigned nt_long = n_threads > 0 ? (unsigned)n_threads : sustained_threads();
        // pass 1: containers with a signature are parsed at once; MP3 candidates only get an upper bound of their
        // record count, so that pass 2 can parse them straight into one page-locked staging buffer
        std::vector<size_t> bound((size_t)n_files, 0), base((size_t)n_files, 0), ogg_bound((size_t)n_files, 0), ogg_base((size_t)n_files, 0),
            flac_bound((size_t)n_files, 0), flac_base((size_t)n_files, 0);
        std::vector<uint8_t> opus_open((size_t)n_files, 0);
        parallel_for((size_t)n_files, nt, [&](size_t i) {
            if (!data[i] || !length[i]) return;
            Parsed &p = parsed[i];
            try {
            if (length[i] >= 4 && std::memcmp(data[i], "OggS", 4) == 0) {                             // Opus is tried first (stream.d:1596)
                afg_opus::Reader rd;                                                                  // headers + sizes; decoded in pass 1c
                const afg_opus::Status st = rd.open(data[i], length[i], p.opus);
                if (st == afg_opus::kOpened) { opus_open[i] = 1; return; }
                p.opus = afg_opus::File();
                if (st == afg_opus::kUnsupported) { p.opus_mode = true; return; }
            }
            if ((flac_bound[i] = flac_res_bound(data[i], length[i])) != 0) return;                    // parsed in pass 1b
            p.flac.pack16 = flac_rows_int16();
            if (flac_parse(data[i], length[i], p.fi, p.flac)) { p.format = AFG_FORMAT_FLAC; return; }
            p.flac = FlacRecords();
            if (qoa_parse(data[i], length[i], p.qi, p.qoa)) { p.format = AFG_FORMAT_QOA; return; }
            if ((ogg_bound[i] = afg_vorbis::max_spec_floats(data[i], length[i])) != 0) return;       // parsed in pass 1b
            if (afg_mp3::looks_like_mp3(data[i], length[i])) bound[i] = afg_mp3::max_blocks(data[i], length[i]);
            } catch (...) { p = Parsed(); }
        });
        tm.lap("pass 1: flac / qoa parse, ogg + mp3 bounds");
        size_t total_bound = 0, ogg_total = 0, flac_total = 0;
        for (size_t i = 0; i < (size_t)n_files; i++) {
            base[i] = total_bound; total_bound += bound[i];
            ogg_base[i] = ogg_total; ogg_total += ogg_bound[i];
            flac_base[i] = flac_total; flac_total += (flac_bound[i] + 3) & ~(size_t)3;        // 16-byte aligned planes
        }
        // pass 1b: FLAC files of known length straight into one page-locked residual buffer
        StagingPool::Lease flac_lease;
        FlacStage flac_stage;
        if (flac_total) {
            if (int rc = g_staging.take(flac_total * sizeof(int32_t), flac_lease)) return rc;
            int32_t *res0 = (int32_t *)flac_lease.p;
            std::atomic<bool> lost{ false };
            parallel_for((size_t)n_files, nt, [&](size_t i) {
                if (!flac_bound[i]) return;
                Parsed &p = parsed[i];
                bool ok = false;
                try {
                    p.flac.pack16 = flac_rows_int16();
                    ok = flac_parse_into(data[i], length[i], p.fi, p.flac, res0 + flac_base[i], flac_bound[i]);
                    if (ok && p.flac.overflow) {                 // more audio than STREAMINFO declares: the file's own buffer
                        p.flac = FlacRecords();
                        p.flac.pack16 = flac_rows_int16();
                        ok = flac_parse(data[i], length[i], p.fi, p.flac);
                        lost = true;
                    }
                } catch (...) { ok = false; }
                if (ok) p.format = AFG_FORMAT_FLAC;
                else p.flac = FlacRecords();
            });
            // the staged layout is used only when every FLAC file of the batch is in it (a file of undeclared length
            // was parsed into its own buffer in pass 1): otherwise everything is gathered, wherever it sits
            bool all_staged = !lost;
            for (size_t i = 0; i < (size_t)n_files && all_staged; i++)
                if (parsed[i].format == AFG_FORMAT_FLAC && !parsed[i].flac.ext_res && parsed[i].flac.res_size()) all_staged = false;
            if (all_staged) { flac_stage.res = res0; flac_stage.words = flac_total; flac_stage.base = flac_base.data(); }
            tm.lap("pass 1b: flac parse into staging");
        }
        BatchOut *owner = new (std::nothrow) BatchOut;
        if (!owner) return AFG_ERR_OOM;
        std::unique_ptr<BatchOut> guard(owner);
        // ---- pass 1c: Ogg Opus.  The open scan gave exact sizes, so every file is decoded (range decoder + CELT frame decoder,
        // by all helper threads) straight into one page-locked buffer, a chunk of files at a time; the chunk's upload, transform,
        // output conversion and download are queued on two streams and run while the helpers decode the next chunk.
        std::vector<size_t> opus_pcm_at((size_t)n_files, 0);
        bool opus_staged = false;
        {
            size_t n_opus = 0, recs_total = 0, coefs_total = 0, seqs_total = 0;
            std::vector<size_t> rec_at((size_t)n_files, 0), coef_at((size_t)n_files, 0), seq_at((size_t)n_files, 0);
            for (size_t i = 0; i < (size_t)n_files; i++) {
                if (!opus_open[i]) continue;
                const afg_opus::File &m = parsed[i].opus;
                rec_at[i] = recs_total; coef_at[i] = coefs_total;
                opus_pcm_at[i] = coefs_total;                     // one PCM float per coefficient
                recs_total += m.bound_frames * (size_t)m.channels;
                coefs_total += m.bound_coeffs;
                n_opus++;
            }
            // Channel sequences.  The transform stage walks sequences 2p and 2p + 1 of a launch together when they are the two
            // channels of a stream (one wavefront, half each: afg.h), so a stereo file starts on an even index of its launch:
            // an empty sequence goes in front of it after an odd number of mono files, and in front of a chunk (a launch, below)
            // that would start on an odd one.  Without it such a file is walked one channel at a time -- slower, and in
            // AFG_NUMERIC_TOLERANCE to samples that depend on what else is in the batch (tools/soak_damaged.py found one).
            const size_t target = std::max<size_t>((coefs_total + 7) / 8, (size_t)4 << 20);      // coefficients per chunk
            std::vector<uint8_t> seq_pad((size_t)n_files, 0);
            for (size_t f0 = 0; f0 < (size_t)n_files;) {
                size_t f1 = f0, acc = 0;
                while (f1 < (size_t)n_files && acc < target) { if (opus_open[f1]) acc += parsed[f1].opus.bound_coeffs; f1++; }
                bool first = true;
                for (size_t i = f0; i < f1; i++) {
                    if (!opus_open[i]) continue;
                    const size_t C = (size_t)parsed[i].opus.channels;
                    if ((seqs_total & 1) && (first || C == 2)) { seq_pad[i] = 1; seqs_total++; }
                    seq_at[i] = seqs_total;
                    seqs_total += C;
                    first = false;
                }
                f0 = f1;
            }
            if (n_opus && seqs_total <= 0xffffffffull) {
                const size_t base_bytes = ((seqs_total + 1) * sizeof(uint64_t) + 15) & ~(size_t)15;
                const size_t rec_bytes = (recs_total * sizeof(afg_celt_frame) + 15) & ~(size_t)15;
                StagingPool::Lease h_in;
                DeviceBuf d_in, d_pcm;
                if (int rc = g_staging.take(base_bytes + rec_bytes + coefs_total * sizeof(float), h_in)) return rc;
                if (int rc = g_staging.take(std::max<size_t>(coefs_total, 1) * sizeof(float), owner->opus_plane)) return rc;
                if (int rc = d_in.alloc(base_bytes + rec_bytes + coefs_total * sizeof(float))) return rc;
                if (int rc = d_pcm.alloc(std::max<size_t>(coefs_total, 1) * sizeof(float))) return rc;
                uint64_t *hb = (uint64_t *)h_in.p;
                afg_celt_frame *hr = (afg_celt_frame *)((uint8_t *)h_in.p + base_bytes);
                float *hc = (float *)((uint8_t *)h_in.p + base_bytes + rec_bytes);
                const uint64_t *db = (const uint64_t *)d_in.p;
                const afg_celt_frame *dr = (const afg_celt_frame *)((const uint8_t *)d_in.p + base_bytes);
                const float *dc = (const float *)((const uint8_t *)d_in.p + base_bytes + rec_bytes);
                for (size_t i = 0; i < (size_t)n_files; i++) {
                    if (!opus_open[i]) continue;
                    const afg_opus::File &m = parsed[i].opus;
                    if (seq_pad[i]) hb[seq_at[i] - 1] = rec_at[i];                                  // the empty sequence
                    for (int c = 0; c < m.channels; c++) hb[seq_at[i] + (size_t)c] = rec_at[i] + (size_t)c * m.bound_frames;
                }
                hb[seqs_total] = recs_total;
                hipStream_t up = nullptr, down = nullptr;
                std::vector<hipEvent_t> events;
                hipError_t e = g_streams.take(&up, &down);
                if (e == hipSuccess) e = hipMemcpyAsync(d_in.p, hb, base_bytes, hipMemcpyHostToDevice, up);
                int rc = AFG_OK;
                for (size_t f0 = 0; f0 < (size_t)n_files && !rc && e == hipSuccess;) {
                    size_t f1 = f0, acc = 0;
                    while (f1 < (size_t)n_files && acc < target) { if (opus_open[f1]) acc += parsed[f1].opus.bound_coeffs; f1++; }
                    size_t first = (size_t)n_files, last = (size_t)n_files;
                    for (size_t i = f0; i < f1; i++)
                        if (opus_open[i]) { if (first == (size_t)n_files) first = i; last = i; }
                    if (first == (size_t)n_files) { f0 = f1; continue; }
                    parallel_for(f1 - f0, nt_long, [&](size_t k) {
                        const size_t i = f0 + k;
                        if (!opus_open[i]) return;
                        Parsed &p = parsed[i];
                        const size_t nfr = p.opus.bound_frames, nco = p.opus.bound_coeffs;
                        const int C = p.opus.channels;
                        afg_celt_frame *recs = hr + rec_at[i];
                        bool ok = false;
                        try {
                            afg_opus::File f;
                            ok = afg_opus::parse_file_into(data[i], length[i], f, recs, nfr, hc + coef_at[i], nco) == afg_opus::kOpened &&
                                 !f.overflow && f.n_frames == nfr && f.n_coeffs == nco;
                            f.ext_frames = nullptr;                // (the staging outlives this record)
                            f.ext_coeffs = nullptr;
                            if (ok) p.opus = f;
                        } catch (...) { ok = false; }
                        if (!ok) {                                 // cannot happen (the sizes are exact): an empty, failed file
                            p.opus.error = true;
                            p.opus.pcm_frames = 0;
                            std::memset((void *)recs, 0, nfr * (size_t)C * sizeof(afg_celt_frame));
                            std::memset(hc + coef_at[i], 0, nco * sizeof(float));
                            for (size_t q = 0; q < nfr * (size_t)C; q++) { recs[q].frame_size = 120; recs[q].blocks = 1; recs[q].out_stride = 1; recs[q].imdct_scale = 1.0f; recs[q].out_off = opus_pcm_at[i]; recs[q].coef_off = coef_at[i]; }
                            p.format = AFG_FORMAT_OPUS;
                            return;
                        }
                        // channel 0's records are in place (file-relative offsets): make them plane-absolute, derive the others
                        for (int c = C - 1; c >= 0; c--)
                            for (size_t q = 0; q < nfr; q++) {
                                afg_celt_frame r = recs[q];
                                r.coef_off += coef_at[i] + (uint64_t)c * r.frame_size;
                                r.out_off += opus_pcm_at[i] + (uint64_t)c;
                                recs[(size_t)c * nfr + q] = r;
                            }
                        p.format = AFG_FORMAT_OPUS;
                    });
                    const size_t r0 = rec_at[first], r1 = rec_at[last] + parsed[last].opus.bound_frames * (size_t)parsed[last].opus.channels;
                    const size_t c0 = coef_at[first], c1 = coef_at[last] + parsed[last].opus.bound_coeffs;
                    const size_t s0 = seq_at[first], s1 = seq_at[last] + (size_t)parsed[last].opus.channels;
                    e = hipMemcpyAsync((void *)(dr + r0), hr + r0, (r1 - r0) * sizeof(afg_celt_frame), hipMemcpyHostToDevice, up);
                    if (e == hipSuccess && c1 > c0) e = hipMemcpyAsync((void *)(dc + c0), hc + c0, (c1 - c0) * sizeof(float), hipMemcpyHostToDevice, up);
                    if (e != hipSuccess) break;
                    if (c1 > c0) {
                        rc = afg_celt_transform_hip((uint32_t)(s1 - s0), db + s0, dr, dc, (float *)d_pcm.p, nullptr, up);
                        if (rc) break;
                        bool any_gain = false;
                        for (size_t i = first; i <= last; i++) any_gain = any_gain || (opus_open[i] && parsed[i].opus.gain_i != 0);
                        if (!any_gain) {
                            rc = afg_opus_output_hip(c1 - c0, (const float *)d_pcm.p + c0, nullptr, (float *)d_pcm.p + c0, up);
                        } else {
                            for (size_t i = first; i <= last && !rc; i++) {
                                if (!opus_open[i] || !parsed[i].opus.bound_coeffs) continue;
                                float *at = (float *)d_pcm.p + opus_pcm_at[i];
                                const afg_opus::File &m = parsed[i].opus;
                                rc = m.gain_i ? afg_opus_output_gain_hip(m.bound_coeffs, at, m.gain, nullptr, at, up)
                                              : afg_opus_output_hip(m.bound_coeffs, at, nullptr, at, up);
                            }
                        }
                        if (rc) break;
                        hipEvent_t done = nullptr;
                        e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
                        if (e != hipSuccess) break;
                        events.push_back(done);
                        e = hipEventRecord(done, up);
                        if (e == hipSuccess) e = hipStreamWaitEvent(down, done, 0);
                        if (e == hipSuccess)
                            e = hipMemcpyAsync((float *)owner->opus_plane.p + c0, (const float *)d_pcm.p + c0, (c1 - c0) * sizeof(float), hipMemcpyDeviceToHost, down);
                    }
                    f0 = f1;
                }
                if (up) { hipError_t e2 = hipStreamSynchronize(up); if (e == hipSuccess) e = e2; }
                if (down) { hipError_t e2 = hipStreamSynchronize(down); if (e == hipSuccess) e = e2; }
                for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
                g_streams.give(up, down);
                if (rc) return rc;
                if (e != hipSuccess) { afg::set_error("Opus stage failed: %s", hipGetErrorString(e)); return AFG_ERR_HIP; }
                opus_staged = true;
                tm.lap("pass 1c: opus decode into staging | h2d | kernels | d2h (chunks overlapped)");
            } else if (n_opus) {
                afg::set_error("Opus stage: too many channel sequences");
                return AFG_ERR_INVALID;
            }
        }
        // The FLAC and QOA files are complete now: their device stage (mostly PCIe time) runs on a second host thread
        // while this one parses the MP3 and Ogg files.  Each call of decode_parsed only touches the files it owns.
        std::vector<uint8_t> own_early((size_t)n_files, 0), own_late((size_t)n_files, 1);
        size_t n_early = 0;
        for (size_t i = 0; i < (size_t)n_files; i++)
            if (parsed[i].format == AFG_FORMAT_FLAC || parsed[i].format == AFG_FORMAT_QOA) { own_early[i] = 1; own_late[i] = 0; n_early++; }
        struct EarlyJob {
            std::thread th;
            int rc = AFG_OK;
            std::string error;
            ~EarlyJob() { if (th.joinable()) th.join(); }
        } early_job;
        const bool split = n_early && (total_bound || ogg_total);
        if (split) {
            owner->early.reset(new BatchOut);
            BatchOut *eo = owner->early.get();
            const FlacStage *fs = flac_stage.words ? &flac_stage : nullptr;
            early_job.th = std::thread([&, eo, fs, cur_dev] {
                try {
                    // HIP's current device is per host thread and starts at 0: this thread works for the caller's device
                    if (hipSetDevice(cur_dev) != hipSuccess) { afg::set_error("hipSetDevice(%d) failed", cur_dev); early_job.rc = AFG_ERR_HIP; return; }
                    early_job.rc = decode_parsed(parsed, data, length, 1 /* no helpers: they are parsing */, *eo, nullptr, nullptr, fs,
                                                 own_early.data());
                } catch (...) {
                    afg::set_error("out of host memory");
                    early_job.rc = AFG_ERR_OOM;
                }
                if (early_job.rc) early_job.error = afg_last_error();
            });
        }
        StagingPool::Lease mp3_stage;
        Mp3Stage stage;
        Mp3Pipe pipe;
        bool fallback = false;
        if (total_bound) {
            // Quantised upload by default: int16 Huffman values + a record per granule, requantised on the device.
            // afg_dev_option("mp3_float_upload", 1) keeps the float spectra of round 1 (A/B of the bytes that cross the bus).
            const bool qmode = afg::dev_option(afg::kDevMp3FloatUpload) <= 0;
            const size_t per_block = qmode ? 576 * sizeof(int16_t) + sizeof(afg_mp3_qgranule) + sizeof(uint32_t)
                                           : 576 * sizeof(float) + sizeof(uint32_t);
            if (int rc = g_staging.take(total_bound * per_block + 64, mp3_stage)) return rc;
            if (int rc = g_staging.take(total_bound * 576 * sizeof(float), owner->mp3_plane)) return rc;
            float *coef0 = nullptr;
            int16_t *q0 = nullptr;
            afg_mp3_qgranule *recs0 = nullptr;
            uint32_t *flags0 = nullptr;
            if (qmode) {
                recs0 = (afg_mp3_qgranule *)mp3_stage.p;                           // 8-byte aligned records first
                flags0 = (uint32_t *)(recs0 + total_bound);
                q0 = (int16_t *)(flags0 + total_bound);
                stage.q = q0; stage.recs = recs0;
            } else {
                coef0 = (float *)mp3_stage.p;
                flags0 = (uint32_t *)(coef0 + total_bound * 576);
                stage.coef = coef0;
            }
            stage.flags = flags0; stage.blocks = total_bound; stage.base = base.data();
            stage.plane = (float *)owner->mp3_plane.p;
            if (int rc = pipe.open(stage)) return rc;
            tm.lap("mp3 pipeline set-up (device planes, streams, table arena)");
            // pass 2, chunk by chunk: all host threads parse a chunk of files, its device work is queued, and they go on
            // with the next chunk while the copies and the kernel of this one run
            size_t want = 8;
            if (afg::dev_option(afg::kDevMp3Chunks) > 0) want = (size_t)afg::dev_option(afg::kDevMp3Chunks);
            const size_t target = std::max<size_t>((total_bound + want - 1) / want, 8192);
            for (size_t f0 = 0; f0 < (size_t)n_files;) {
                size_t f1 = f0, acc = 0;
                while (f1 < (size_t)n_files && acc < target) acc += bound[f1++];
                std::atomic<bool> lost{ false };
                parallel_for(f1 - f0, nt, [&](size_t k) {
                    const size_t i = f0 + k;
                    if (!bound[i]) return;
                    Parsed &p = parsed[i];
                    bool ok = false;
                    try {
                        if (qmode) {
                            std::memset(recs0 + base[i], 0, bound[i] * sizeof(afg_mp3_qgranule));     // nch = 0: slot unused
                            p.mp3.quantised = true;
                            p.mp3.ext_q = q0 + base[i] * 576;
                            p.mp3.ext_qgr = recs0 + base[i];
                            ok = afg_mp3::parse_file_into(data[i], length[i], p.mp3, nullptr, flags0 + base[i], bound[i]);
                            if (ok && !p.mp3.overflow && !p.mp3.q_unsupported) {
                                const uint64_t shift = (uint64_t)base[i] * 576;                       // file-relative -> plane offsets
                                for (size_t k = 0; k < p.mp3.blocks(); k++)
                                    if (recs0[base[i] + k].nch) { recs0[base[i] + k].q_off += shift; recs0[base[i] + k].coef_off += shift; }
                            }
                        } else {
                            ok = afg_mp3::parse_file_into(data[i], length[i], p.mp3, coef0 + base[i] * 576, flags0 + base[i], bound[i]);
                        }
                        if (ok && (p.mp3.overflow || p.mp3.q_unsupported)) {
                            // overflow cannot happen; a stream the device requantiser does not cover (MPEG-2.5 8 kHz mixed
                            // blocks) can: either way the file is parsed into its own float buffers and the batch takes the
                            // gathered path
                            p.mp3 = afg_mp3::File();
                            ok = afg_mp3::parse_file(data[i], length[i], p.mp3);
                            lost = true;
                        }
                    } catch (...) {
                        // the file may have left records with file-relative offsets in the staged plane: the chunk must not
                        // be submitted as it stands (afg_mp3_requant_hip walks every slot with nch != 0)
                        ok = false;
                        if (qmode) std::memset(recs0 + base[i], 0, bound[i] * sizeof(afg_mp3_qgranule));
                        lost = true;
                    }
                    if (ok) p.format = AFG_FORMAT_MP3;
                    else p.mp3 = afg_mp3::File();
                });
                if (lost) fallback = true;
                if (!fallback) pipe.submit(parsed, f0, f1);
                f0 = f1;
            }
            if (fallback && qmode) {
                // the gathered path works from float records: the files that were parsed into the quantised staging are
                // parsed again into their own buffers (rare: one file of the batch is outside the requantiser's coverage)
                parallel_for((size_t)n_files, nt, [&](size_t i) {
                    if (!bound[i] || parsed[i].format != AFG_FORMAT_MP3 || !parsed[i].mp3.quantised) return;
                    Parsed &p = parsed[i];
                    bool ok = false;
                    try {
                        p.mp3 = afg_mp3::File();
                        ok = afg_mp3::parse_file(data[i], length[i], p.mp3);
                    } catch (...) { ok = false; }
                    if (!ok) { p.mp3 = afg_mp3::File(); p.format = AFG_FORMAT_UNKNOWN; }
                });
            }
            tm.lap("mp3 parse (all threads) | h2d | kernel | d2h");
        }
        // pass 1b: Ogg Vorbis files straight into one page-locked staging buffer (no per-file megabyte vectors to
        // fault in, gather and unmap) -- while the MP3 chunks queued above are still moving
        StagingPool::Lease ogg_lease;
        OggStage ogg_stage;
        if (ogg_total) {
            if (int rc = g_staging.take(ogg_total * sizeof(float), ogg_lease)) return rc;
            float *spec0 = (float *)ogg_lease.p;
            std::atomic<bool> lost{ false };
            parallel_for((size_t)n_files, nt, [&](size_t i) {
                if (!ogg_bound[i]) return;
                Parsed &p = parsed[i];
                bool ok = false;
                try {
                    ok = afg_vorbis::parse_file_into(data[i], length[i], p.ogg, spec0 + ogg_base[i], ogg_bound[i], vorbis_floor_on_device());
                    if (ok && p.ogg.overflow) {                  // cannot happen; be safe: the file's own buffer
                        ok = afg_vorbis::parse_file(data[i], length[i], p.ogg, vorbis_floor_on_device());
                        lost = true;
                    }
                } catch (...) { ok = false; }
                if (ok) p.format = AFG_FORMAT_OGG;
                else p.ogg = afg_vorbis::File();
            });
            if (!lost) { ogg_stage.spec = spec0; ogg_stage.floats = ogg_total; ogg_stage.base = ogg_base.data(); }
            tm.lap("pass 1b: ogg parse into staging");
        }
        if (total_bound) {
            const int prc = pipe.close();
            tm.lap("mp3 pipeline drain");
            if (prc) return prc;
            if (fallback) stage.blocks = 0;                   // decode_parsed does those files from their own buffers
        }
        if (parse_done) (*parse_done)();
        int rc = decode_parsed(parsed, data, length, nt, *owner, stage.blocks ? &stage : nullptr, ogg_stage.floats ? &ogg_stage : nullptr,
                               split ? nullptr : (flac_stage.words ? &flac_stage : nullptr), split ? own_late.data() : nullptr, nullptr, nullptr,
                               opus_staged ? opus_pcm_at.data() : nullptr);
        tm.lap("decode_parsed total");
        if (split) {
            early_job.th.join();
            tm.lap("flac / qoa thread joined");
            if (!rc && early_job.rc) { afg::set_error("%s", early_job.error.c_str()); rc = early_job.rc; }
        }
        if (rc) return rc;
        for (int i = 0; i < n_files; i++) {
            const BatchOut *src = (split && own_early[(size_t)i]) ? owner->early.get() : owner;
            const Decoded &d = src->files[(size_t)i];
            items[i].status = d.status;
            items[i].message = d.message;
            items[i].format = d.format;
            items[i].channels = d.channels;
            items[i].samplerate = d.samplerate;
            items[i].frames = d.frames;
            const float *plane = d.in_mp3_plane ? (const float *)owner->mp3_plane.p
                                 : d.in_opus_plane ? (const float *)owner->opus_plane.p : (const float *)src->plane.p;
            items[i].pcm = (d.status == AFG_OK && d.frames > 0) ? (float *)plane + d.pcm_off : nullptr;
        }
        keep = std::move(guard);
        tm.lap("items filled");
        return AFG_OK;
    }
}

}  // namespace

extern "C" {

int afg_set_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        afg::set_error("no HIP device available (%s); this library has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return AFG_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        afg::set_error("afg_set_device(%d): %d device(s) visible", device, n);
        return AFG_ERR_INVALID;
    }
    AFG_HIP_CHECK(hipSetDevice(device));
    return afg::require_device();
}

uint64_t afg_host_pool_trim(void) { return (uint64_t)g_staging.trim() + (uint64_t)g_devpool.trim(); }

int afg_get_device(void)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        afg::set_error("hipGetDevice failed: %s", hipGetErrorString(e));
        return e == hipErrorNoDevice ? AFG_ERR_NO_DEVICE : AFG_ERR_HIP;
    }
    return dev;
}

int afg_batch_decode_ex(const uint8_t *const *data, const size_t *length, int n_files, const afg_batch_opts *opts, afg_batch_result *out)
{
    try {
        if (!out || n_files < 0 || (n_files && (!data || !length))) return AFG_ERR_INVALID;
        out->n_files = 0; out->items = nullptr; out->owner = nullptr;
        if (opts && opts->struct_size < sizeof(afg_batch_opts)) { afg::set_error("afg_batch_opts.struct_size too small"); return AFG_ERR_INVALID; }
        if (n_files == 0) return AFG_OK;
        // ---- which devices ----
        std::vector<int> devs;
        const int want = opts ? opts->n_devices : 0;
        if (want != 0) {
            const int visible = afg_device_count();
            if (visible <= 0) { afg::set_error("no HIP device available; this library has no CPU fallback"); return AFG_ERR_NO_DEVICE; }
            if (want < 0) for (int d = 0; d < visible; d++) devs.push_back(d);
            else for (int k = 0; k < want; k++) {
                const int d = opts->devices ? opts->devices[k] : k;
                if (d < 0 || d >= visible) { afg::set_error("afg_batch_decode_ex: device %d of %d visible", d, visible); return AFG_ERR_INVALID; }
                devs.push_back(d);
            }
            if (devs.size() > 16) { afg::set_error("afg_batch_decode_ex: at most 16 devices"); return AFG_ERR_INVALID; }
        }
        const int n_threads = opts ? opts->n_threads : 0;
        auto owner = std::unique_ptr<BatchOwner>(new BatchOwner);
        afg_batch_item *items = (afg_batch_item *)std::calloc((size_t)n_files, sizeof(afg_batch_item));
        if (!items) return AFG_ERR_OOM;
        struct ItemsGuard { afg_batch_item *p; ~ItemsGuard() { std::free(p); } } items_guard{ items };
        if (devs.size() <= 1) {
            // one device: the caller's current one, or the one named
            int restore = -1;
            if (devs.size() == 1) {
                int cur = 0;
                AFG_HIP_CHECK(hipGetDevice(&cur));
                if (cur != devs[0]) { restore = cur; AFG_HIP_CHECK(hipSetDevice(devs[0])); }
            }
            // A large batch runs as a pipeline of GROUPS of files (round 6): while one group's device stages run -- the host mostly
            // waiting -- the next group is parsed.  Two host threads take the groups in turn; a token serialises their parse passes
            // (one set of helper threads at a time), handed on when a group reaches its device stages.  Results do not depend on
            // the grouping (files are independent).  afg_dev_option("batch_groups", n) forces n (1: off).
            size_t total_bytes = 0;
            for (int i = 0; i < n_files; i++) total_bytes += length[i];
            // By default only batches of (almost only) native FLAC and Ogg Vorbis files, from 512 files and 16 MB up, in 4 groups:
            // their calls are a parse pass followed by device stages; the MP3 and Opus paths overlap their parsing with their own
            // transfers already and lose 5-10 % to the split (2048-file batches, 24 helper threads, samples/s ungrouped -> 4 groups:
            // Vorbis 5.3e9 -> 6.2e9, FLAC 5.0e9 -> 5.7e9, MP3 6.5e9 -> 6.0e9, Opus 2.5e9 -> 2.3e9, mixed 4.9e9 -> 4.5e9).
            long groups = afg::dev_option(afg::kDevBatchGroups);
            if (groups <= 0) {
                groups = 1;
                if (n_files >= 512 && total_bytes >= ((size_t)16 << 20)) {
                    int staged = 0;
                    for (int i = 0; i < n_files; i++) {
                        const uint8_t *b = data[i];
                        const size_t n = length[i];
                        if (n >= 4 && !std::memcmp(b, "fLaC", 4)) staged++;
                        else if (n >= 28 && !std::memcmp(b, "OggS", 4) && n >= (size_t)27 + b[26] + 7 &&
                                 !std::memcmp(b + 27 + b[26], "\x01vorbis", 7)) staged++;      // (the first page's payload: the identification header)
                    }
                    if ((size_t)staged * 10 >= (size_t)n_files * 9) groups = 4;
                }
            }
            if (groups > n_files) groups = n_files;
            int rc = AFG_OK;
            if (groups <= 1) {
                owner->parts.emplace_back();
                rc = batch_decode_device(data, length, n_files, n_threads, items, owner->parts.back());
            } else {
                const size_t G = (size_t)groups;
                owner->parts.resize(G);
                std::vector<int> first(G + 1, n_files);          // contiguous ranges of about equal compressed size
                {
                    size_t acc = 0, g = 0;
                    first[0] = 0;
                    for (int i = 0; i < n_files && g + 1 < G; i++) {
                        acc += length[i];
                        if (acc * G >= total_bytes * (g + 1)) first[++g] = i + 1;
                    }
                }
                int dev = 0;
                AFG_HIP_CHECK(hipGetDevice(&dev));
                HelperPool *const helpers = tl_helpers;
                std::mutex token;
                std::atomic<bool> failed{ false };
                struct Job { int rc = AFG_OK; std::string error; } jobs[2];
                auto work = [&](int w) {
                    try {
                        if (w && hipSetDevice(dev) != hipSuccess) { jobs[w].rc = AFG_ERR_HIP; jobs[w].error = "hipSetDevice failed"; failed = true; return; }
                        tl_helpers = w ? &g_group_helpers[dev & 15] : helpers;
                        tl_stage_chunks = 2;
                        for (size_t g = (size_t)w; g < G && !failed; g += 2) {
                            const int f0 = first[g], f1 = first[g + 1];
                            if (f1 <= f0) continue;
                            std::unique_lock<std::mutex> lk(token);
                            bool released = false;
                            const std::function<void()> done = [&] { if (!released) { released = true; lk.unlock(); } };
                            const int r = batch_decode_device(data + f0, length + f0, f1 - f0, n_threads, items + f0, owner->parts[g], &done);
                            done();
                            if (r) { jobs[w].rc = r; jobs[w].error = afg_last_error(); failed = true; }
                        }
                    } catch (...) {
                        jobs[w].rc = AFG_ERR_OOM; jobs[w].error = "out of host memory"; failed = true;
                    }
                    tl_stage_chunks = 8;
                    tl_helpers = helpers;
                };
                {
                    struct Joiner { std::thread th; ~Joiner() { if (th.joinable()) th.join(); } } j;
                    j.th = std::thread(work, 1);
                    work(0);
                }
                for (const Job &jb : jobs)
                    if (jb.rc && !rc) { afg::set_error("%s", jb.error.c_str()); rc = jb.rc; }
            }
            if (restore >= 0) (void)hipSetDevice(restore);
            if (rc) return rc;
        } else {
            // Files are independent (stream.d:1363-1434 is all per-instance): the batch shards by file, longest file first
            // onto the least loaded device (compressed bytes stand for decode work), no exchange between devices.
            const size_t nd = devs.size();
            std::vector<size_t> order((size_t)n_files);
            for (size_t i = 0; i < order.size(); i++) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return length[a] > length[b]; });
            std::vector<std::vector<size_t>> mine(nd);
            std::vector<uint64_t> load(nd, 0);
            for (size_t f : order) {
                size_t best = 0;
                for (size_t k = 1; k < nd; k++) if (load[k] < load[best]) best = k;
                mine[best].push_back(f);
                load[best] += length[f] + 1;
            }
            for (auto &v : mine) std::sort(v.begin(), v.end());
            const unsigned total_threads = n_threads > 0 ? (unsigned)n_threads : default_threads();
            const int per_dev_threads = (int)std::max<unsigned>(1u, total_threads / (unsigned)nd);
            owner->parts.resize(nd);
            struct Part {
                std::vector<const uint8_t *> data;
                std::vector<size_t> len;
                std::vector<afg_batch_item> items;
                int rc = AFG_OK;
                std::string error;
            };
            std::vector<Part> parts(nd);
            int caller_dev = 0;
            AFG_HIP_CHECK(hipGetDevice(&caller_dev));
            auto run_part = [&](size_t k) {
                Part &p = parts[k];
                try {
                    for (size_t f : mine[k]) { p.data.push_back(data[f]); p.len.push_back(length[f]); }
                    p.items.assign(mine[k].size(), afg_batch_item{});
                    if (mine[k].empty()) return;
                    if (hipSetDevice(devs[k]) != hipSuccess) { p.rc = AFG_ERR_HIP; p.error = "hipSetDevice failed"; return; }
                    tl_helpers = &g_device_helpers[k];
                    p.rc = batch_decode_device(p.data.data(), p.len.data(), (int)mine[k].size(), per_dev_threads, p.items.data(), owner->parts[k]);
                    tl_helpers = nullptr;
                    if (p.rc) p.error = afg_last_error();
                } catch (...) {
                    tl_helpers = nullptr;
                    p.rc = AFG_ERR_OOM; p.error = "out of host memory";
                }
            };
            {
                // joins on every way out: a thread that cannot be created after others have started must not take the
                // process down (std::terminate on a joinable std::thread) instead of returning AFG_ERR_OOM
                struct Joiner {
                    std::vector<std::thread> th;
                    ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); }
                } j;
                j.th.reserve(nd);
                for (size_t k = 1; k < nd; k++) j.th.emplace_back(run_part, k);
                run_part(0);
            }
            (void)hipSetDevice(caller_dev);
            for (size_t k = 0; k < nd; k++)
                if (parts[k].rc) { afg::set_error("device %d: %s", devs[k], parts[k].error.c_str()); return parts[k].rc; }
            for (size_t k = 0; k < nd; k++)
                for (size_t j = 0; j < mine[k].size(); j++) items[mine[k][j]] = parts[k].items[j];
        }
        items_guard.p = nullptr;
        out->n_files = n_files;
        out->items = items;
        out->owner = owner.release();
        return AFG_OK;
    } catch (...) {
        afg::set_error("out of host memory");
        return AFG_ERR_OOM;
    }
}

int afg_batch_decode(const uint8_t *const *data, const size_t *length, int n_files, int n_threads, afg_batch_result *out)
{
    afg_batch_opts o;
    std::memset(&o, 0, sizeof(o));
    o.struct_size = (uint32_t)sizeof(o);
    o.n_threads = n_threads;
    return afg_batch_decode_ex(data, length, n_files, &o, out);
}

void afg_batch_free(afg_batch_result *r)
{
    if (!r) return;
    StageTimer tm;
    std::free(r->items);
    delete (BatchOwner *)r->owner;
    r->items = nullptr; r->owner = nullptr; r->n_files = 0;
    tm.lap("afg_batch_free");
}

}  // extern "C"
