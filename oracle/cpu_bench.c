/*
 * cpu_bench.c -- native thread pool that times the CPU oracle on the host cores.
 *
 * TEST INFRASTRUCTURE ONLY (see afg_oracle.h): used by bench.py's `cpu_baseline` leg, never by the product.
 * "the scalar restatement is timed in the same process run on the box's host cores: single-thread and
 * all-cores, one file per task".  A task is one file through the transform-stage oracle; worker threads draw
 * (repeat, task) pairs from one atomic counter and write into a private output buffer, so nothing is allocated
 * or shared while the clock runs.  The caller gets the wall time and the CPU time the workers actually burned
 * (CLOCK_THREAD_CPUTIME_ID): cpu / wall is the parallelism achieved, whatever the cgroup quota allowed.
 */
#define _GNU_SOURCE
#include "afg_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct afgo_bench_task {
    int32_t  codec;                  /* 0 MP3, 1 Vorbis, 2 FLAC, 3 CELT */
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
        run_task(t, out);
        done += t->out_floats;
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
        if (tasks[i].out_floats > p.max_out) p.max_out = tasks[i].out_floats;
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
