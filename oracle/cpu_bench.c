/* TEST / BENCH INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg).
 *
 * pthread harness that times the CPU oracle the way the reference's command-line tool uses its cores
 * (mrgingham-from-image.cc:50, :374-379): T worker threads, each takes the next frame and runs the whole
 * per-frame schedule on it alone -- detect at `start_level`, refine down to level 0 (mrgingham.cc:50, :81-99) --
 * with the reference's per-call allocations kept (every level call allocates and frees its level image and its
 * zeroed response, find_chessboard_corners.cc:449, :506).  No Python, no GIL, no thread-pool hand-over in the timed
 * region; the run lasts at least `min_seconds`.
 *
 * oracle_bench_fn times any `void f(int16_t*, const uint8_t*, int w, int h, int stride)` (the upstream ChESS.c
 * built into oracle/_ref) the same way, one frame per thread at a time. */
#define _GNU_SOURCE
#include <malloc.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mrgingham_oracle.h"

typedef void (*chess_fn)(int16_t*, const uint8_t*, int, int, int);

typedef struct {
    const uint8_t* frames;
    int nframes, H, W, start_level;
    double min_seconds;
    chess_fn fn; /* NULL: the chain */
    atomic_long next;
    atomic_long points;
    atomic_int stop;
    struct timespec t0;
} bench_t;

static double since(const struct timespec* t0)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)(t.tv_sec - t0->tv_sec) + 1e-9 * (double)(t.tv_nsec - t0->tv_nsec);
}

static void* worker(void* arg)
{
    bench_t* b = (bench_t*)arg;
    const size_t npx = (size_t)b->H * (size_t)b->W;
    const int cap = 4096;
    int32_t* xy = (int32_t*)malloc(sizeof(int32_t) * 2 * cap);
    double* pts = (double*)malloc(sizeof(double) * 2 * cap);
    signed char* lv = (signed char*)malloc(cap);
    int16_t* resp = b->fn ? (int16_t*)malloc(sizeof(int16_t) * npx) : NULL;
    long done = 0;
    while (!atomic_load(&b->stop)) {
        const long i = atomic_fetch_add(&b->next, 1);
        const uint8_t* img = b->frames + (size_t)(i % b->nframes) * npx;
        if (b->fn) {
            b->fn(resp, img, b->W, b->H, b->W);
        } else {
            int n = oracle_find_corners(xy, cap, img, b->H, b->W, b->W, b->start_level);
            if (n > cap) n = cap;
            for (int k = 0; k < n; k++) {
                pts[2 * k] = xy[2 * k] / 1000.0;
                pts[2 * k + 1] = xy[2 * k + 1] / 1000.0;
                lv[k] = (signed char)b->start_level;
            }
            for (int L = b->start_level - 1; L >= 0 && n > 0; L--)
                if (oracle_refine_corners(pts, lv, n, img, b->H, b->W, b->W, L) <= 0) break;
            atomic_fetch_add(&b->points, n);
        }
        done++;
        if (since(&b->t0) >= b->min_seconds) atomic_store(&b->stop, 1);
    }
    free(xy); free(pts); free(lv); free(resp);
    return (void*)done;
}

/* Returns the number of frame passes completed (every pass that was begun is finished and counted) and the wall
 * time from the start of the first pass to the end of the last in *elapsed; *points = candidates seen (so that the
 * work cannot be optimised away and the caller can check it). */
static long run(bench_t* b, int nthreads, double* elapsed, long* points)
{
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    atomic_store(&b->next, 0);
    atomic_store(&b->points, 0);
    atomic_store(&b->stop, 0);
    clock_gettime(CLOCK_MONOTONIC, &b->t0);
    for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, worker, b);
    long total = 0;
    for (int i = 0; i < nthreads; i++) {
        void* r;
        pthread_join(th[i], &r);
        total += (long)r;
    }
    *elapsed = since(&b->t0);
    if (points) *points = atomic_load(&b->points);
    free(th);
    return total;
}

long oracle_bench_chain(const uint8_t* frames, int nframes, int H, int W, int start_level, int nthreads,
                        double min_seconds, double* elapsed, long* points)
{
    bench_t b;
    memset(&b, 0, sizeof b);
    b.frames = frames; b.nframes = nframes; b.H = H; b.W = W; b.start_level = start_level;
    b.min_seconds = min_seconds; b.fn = NULL;
    return run(&b, nthreads, elapsed, points);
}

long oracle_bench_fn(void* fn, const uint8_t* frames, int nframes, int H, int W, int nthreads, double min_seconds,
                     double* elapsed)
{
    bench_t b;
    memset(&b, 0, sizeof b);
    b.frames = frames; b.nframes = nframes; b.H = H; b.W = W;
    b.min_seconds = min_seconds; b.fn = (chess_fn)fn;
    return run(&b, nthreads, elapsed, NULL);
}

/* Allocator policy of the process for the runs that follow.  on = 1: blocks of any size come from the heap and
 * freed memory is kept (M_MMAP_THRESHOLD / M_TRIM_THRESHOLD at their maxima) -- the per-call malloc / calloc / free
 * of the schedule stay, but they stop being mmap / munmap / page-fault storms, which is what a 25 MB calloc per
 * level call is under glibc's defaults once 128 threads do it at once.  on = 0: glibc's defaults. */
void oracle_bench_heap_reuse(int on)
{
    if (on) {
        mallopt(M_MMAP_THRESHOLD, 1 << 30);
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_ARENA_MAX, 512);
    } else {
        mallopt(M_MMAP_THRESHOLD, 128 * 1024);
        mallopt(M_TRIM_THRESHOLD, 128 * 1024);
    }
}
