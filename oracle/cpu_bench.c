/*
 * cpu_bench.c -- native thread pool that times the CPU oracle on the host cores.
 *
 * TEST INFRASTRUCTURE ONLY (see afg_oracle.h): used by bench.py's `cpu_baseline` leg, never by the product.
 * "the scalar restatement is timed in the same process run on the box's host cores: single-thread and
 * all-cores, one file per task".  A task is one file through the transform-stage oracle; worker threads draw
 * (repeat, task) pairs from one atomic counter and write into a private output buffer, so nothing is allocated
 * or shared while the clock runs.  The caller gets the wall time and the CPU time the workers actually burned
 * (CLOCK_THREAD_CPUTIME_ID): cpu / wall is the parallelism achieved, whatever the cgroup quota allowed.
 *
 * Round 4: file tasks (codec 10..13) take a whole file from bytes to delivered PCM -- the oracle front-end (Huffman / code
 * book / range decoding: mp3_frontend.c, vorbis_frontend.c, opus_frontend.c) and then the transform-stage oracle -- the CPU
 * side of SURVEY 8d (c), bench.py's `cpu_baseline_e2e`.  Round 5: FLAC too (flac_frontend.c: the reference's scalar 32-bit
 * cache reader with the prediction fused into the Rice loop, as drflac decodes); until then its task borrowed the
 * product's parser.
 */
#define _GNU_SOURCE
#include "afg_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct afgo_bench_task {
    int32_t  codec;                  /* 0 MP3, 1 Vorbis, 2 FLAC, 3 CELT (transform stage); 10 MP3, 11 Ogg Vorbis, 12 Ogg Opus, 13 FLAC:
                                        a file in memory, a = bytes, out_floats = byte count */
    uint32_t n;                      /* MP3: granules; Vorbis: packets; FLAC: frames; CELT: channel sequences */
    uint32_t channels;               /* MP3 / Vorbis */
    uint16_t bs0, bs1;               /* Vorbis block sizes */
    const void *a;                   /* MP3 coef    | Vorbis spec     | FLAC frames    | CELT rec_base */
    const void *b;                   /* MP3 flags   | Vorbis pflags   | FLAC subframes | CELT recs     */
    const void *c;                   /*             | Vorbis spec_off | FLAC res       | CELT coeffs   */
    const void *d;                   /*             | Vorbis out_off  |                |               */
    uint64_t out_floats;             /* size of the task's output (floats or int32) */
} afgo_bench_task;

typedef struct pool {
    const afgo_bench_task *tasks;
    long n_tasks, total;
    atomic_long next;
    uint64_t max_out;
    atomic_ullong samples;
    int failed;
} pool;

typedef struct worker {
    pool *p;
    pthread_t th;
    double cpu_s;
} worker;

static double now_s(clockid_t c)
{
    struct timespec ts;
    clock_gettime(c, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* bytes -> delivered PCM of one file; returns the number of samples (channels included), 0 on failure */
static uint64_t run_file(const afgo_bench_task *t)
{
    const uint8_t *data = (const uint8_t *)t->a;
    const size_t size = (size_t)t->out_floats;
    uint64_t samples = 0;
    switch (t->codec) {
    case 10: {
        afgo_mp3_file f;
        memset(&f, 0, sizeof(f));
        if (afgo_mp3_decode_file(data, size, &f) == 0) {          /* front-end + transform + delivery (mp3_frontend.c) */
            samples = f.pcm_samples;
            afgo_mp3_file_free(&f);
        }
        break;
    }
    case 11: {
        afgo_vorbis_file f;
        memset(&f, 0, sizeof(f));
        if (afgo_vorbis_decode_file(data, size, &f) != 0) break;
        const uint32_t n = f.n_packets;
        uint64_t *so = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (n ? n : 1)), *oo = so ? so + n : NULL, st = 0;
        if (so && n) {
            const uint64_t total = afgo_vorbis_layout(n, f.channels, f.blocksize0, f.blocksize1, f.pflags, 0, 0, so, oo, &st);
            float *pcm = (float *)malloc(sizeof(float) * (size_t)(total ? total : 1));
            if (pcm) {
                const uint8_t nch = (uint8_t)f.channels;
                const uint16_t b0 = (uint16_t)f.blocksize0, b1 = (uint16_t)f.blocksize1;
                if (afgo_vorbis_transform(1, &n, &nch, &b0, &b1, f.pflags, so, oo, f.spec, pcm) == 0) samples = f.pcm_frames * (uint64_t)f.channels;
                free(pcm);
            }
        }
        free(so);
        afgo_vorbis_file_free(&f);
        break;
    }
    case 12: {
        afgo_opus_file f;
        memset(&f, 0, sizeof(f));
        if (afgo_opus_decode_file(data, size, &f) != 0) break;
        const uint64_t nf = f.n_frames, ch = (uint64_t)f.channels, total = f.pcm_frames * ch;
        afgo_celt_frame *recs = (afgo_celt_frame *)malloc(sizeof(afgo_celt_frame) * (size_t)((nf && ch) ? nf * ch : 1));
        float *pcm = (float *)malloc(sizeof(float) * (size_t)(total ? total : 1));
        if (recs && pcm && nf) {
            uint64_t base[3] = { 0, nf, 2 * nf };
            for (uint64_t c = 0; c < ch; c++)
                for (uint64_t i = 0; i < nf; i++) {                /* channel c of frame i: afg_oracle.h, afgo_opus_file */
                    recs[c * nf + i] = f.frames[i];
                    recs[c * nf + i].coef_off += c * f.frames[i].frame_size;
                    recs[c * nf + i].out_off += c;
                }
            afgo_celt_transform((uint32_t)ch, base, recs, f.coeffs, pcm, NULL);
            if (f.gain_i)
                for (uint64_t i = 0; i < total; i++) pcm[i] *= f.gain;
            afgo_opus_output(total, pcm, NULL, pcm);               /* int16 round trip / 32767 (dopus.d:8098-8105, stream.d:480) */
            samples = total;
        }
        free(recs);
        free(pcm);
        afgo_opus_file_free(&f);
        break;
    }
    case 13: {                                          /* FLAC: the oracle's own front-end (flac_frontend.c) + stream.d:505-511 */
        afgo_flac_file f;
        if (afgo_flac_decode_file(data, size, &f) != 0) break;
        float *pcf = (float *)malloc(4 * (size_t)(f.n_samples ? f.n_samples : 1));
        if (pcf) {
            const double factor = 1.0 / 2147483647.0;
            for (uint64_t i = 0; i < f.n_samples; i++) pcf[i] = (float)(f.pcm[i] * factor);
            samples = f.n_samples;
        }
        free(pcf);
        afgo_flac_file_free(&f);
        break;
    }
    default: break;
    }
    return samples;
}

static void run_task(const afgo_bench_task *t, void *out)
{
    switch (t->codec) {
    case 0: {
        const uint32_t ngr = t->n;
        const uint8_t nch = (uint8_t)t->channels;
        afgo_mp3_transform(1, &ngr, &nch, (const float *)t->a, (const uint32_t *)t->b, (float *)out, NULL);
        break;
    }
    case 1: {
        const uint32_t npkt = t->n;
        const uint8_t nch = (uint8_t)t->channels;
        (void)afgo_vorbis_transform(1, &npkt, &nch, &t->bs0, &t->bs1, (const uint8_t *)t->b, (const uint64_t *)t->c,
                                    (const uint64_t *)t->d, (const float *)t->a, (float *)out);
        break;
    }
    case 2:
        afgo_flac_transform(t->n, (const afgo_flac_frame *)t->a, (const afgo_flac_subframe *)t->b, (const int32_t *)t->c,
                            (int32_t *)out, NULL);
        break;
    case 3:
        afgo_celt_transform(t->n, (const uint64_t *)t->a, (const afgo_celt_frame *)t->b, (const float *)t->c, (float *)out, NULL);
        break;
    default: break;
    }
}

static void *work(void *arg)
{
    worker *w = (worker *)arg;
    pool *p = w->p;
    void *out = malloc((size_t)(p->max_out ? p->max_out : 1) * 4);
    if (!out) { p->failed = 1; return NULL; }
    memset(out, 0, (size_t)(p->max_out ? p->max_out : 1) * 4);      /* fault the pages in before the clock matters */
    const double c0 = now_s(CLOCK_THREAD_CPUTIME_ID);
    unsigned long long done = 0;
    for (;;) {
        const long i = atomic_fetch_add(&p->next, 1);
        if (i >= p->total) break;
        const afgo_bench_task *t = &p->tasks[i % p->n_tasks];
        if (t->codec >= 10) {
            done += run_file(t);
        } else {
            run_task(t, out);
            done += t->out_floats;
        }
    }
    w->cpu_s = now_s(CLOCK_THREAD_CPUTIME_ID) - c0;
    atomic_fetch_add(&p->samples, done);
    free(out);
    return NULL;
}

/* Runs `repeats` passes over the task list on n_threads threads.  Returns the wall seconds (< 0 on failure);
 * *cpu_seconds = CPU time summed over the workers, *samples = output values produced. */
double afgo_bench_run(const afgo_bench_task *tasks, int n_tasks, int repeats, int n_threads, double *cpu_seconds,
                      uint64_t *samples)
{
    if (!tasks || n_tasks <= 0 || repeats <= 0 || n_threads <= 0) return -1.0;
    pool p;
    memset(&p, 0, sizeof(p));
    p.tasks = tasks;
    p.n_tasks = n_tasks;
    p.total = (long)n_tasks * repeats;
    atomic_init(&p.next, 0);
    atomic_init(&p.samples, 0);
    for (int i = 0; i < n_tasks; i++)
        if (tasks[i].codec < 10 && tasks[i].out_floats > p.max_out) p.max_out = tasks[i].out_floats;
    worker *w = (worker *)calloc((size_t)n_threads, sizeof(worker));
    if (!w) return -1.0;
    const double t0 = now_s(CLOCK_MONOTONIC);
    int started = 0;
    for (int i = 0; i < n_threads; i++) {
        w[i].p = &p;
        if (pthread_create(&w[i].th, NULL, work, &w[i]) != 0) break;
        started++;
    }
    double cpu = 0;
    for (int i = 0; i < started; i++) {
        pthread_join(w[i].th, NULL);
        cpu += w[i].cpu_s;
    }
    const double wall = now_s(CLOCK_MONOTONIC) - t0;
    free(w);
    if (started == 0 || p.failed) return -1.0;
    if (cpu_seconds) *cpu_seconds = cpu;
    if (samples) *samples = (uint64_t)atomic_load(&p.samples);
    return wall;
}
