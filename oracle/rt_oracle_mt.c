/*
 * rt_oracle_mt.c -- the CPU oracle on several host cores: the reference's
 * static partitioning (PARTIONING_STRATEGY 1, src/RayTracer.cpp:904-923 with
 * CORE_NUM > 1: every core renders one contiguous share of the image, nothing
 * is exchanged) with POSIX threads in place of Tilera tiles.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h): this is what bench.py's
 * cpu_baseline leg times next to the GPU.  The arithmetic is orc_render()'s,
 * untouched; the scene is read-only and shared, the work counters are
 * per-thread.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <time.h>

typedef struct job {
    const orc_scene *scene;
    const orc_camera *camera;
    int W, H, max_depth;
    const int *chunk_x0;
    int first_chunk, last_chunk, chunk_cols;      /* this thread's contiguous block of chunks */
    int cpu;                                      /* logical CPU to run on, or -1 */
    float *out;
    int rc;
} job;

static void *worker(void *arg) {
    job *j = (job *)arg;
    int k;
    if (j->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(j->cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);   /* best effort */
    }
    for (k = j->first_chunk; k < j->last_chunk; k++) {
        const int x0 = j->chunk_x0[k];
        const int x1 = x0 + j->chunk_cols < j->W ? x0 + j->chunk_cols : j->W;
        float *dst = j->out + (size_t)k * (size_t)j->chunk_cols * (size_t)j->H * 3;
        if (orc_render(j->scene, j->camera, j->W, j->H, x0, x1, j->max_depth, dst)) j->rc = 1;
    }
    return NULL;
}

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int orc_render_static_partition(const orc_scene *s, const orc_camera *c, int W, int H, int max_depth,
                                const int *chunk_x0, int n_chunks, int chunk_cols,
                                int n_threads, const int *cpus, float *out, double *seconds) {
    pthread_t *th;
    job *jobs;
    int t, started = 0, rc = 0;
    double t0;
    if (!s || !c || !out || !chunk_x0 || n_chunks <= 0 || chunk_cols <= 0 || n_threads <= 0) return 1;
    if (n_threads > n_chunks) n_threads = n_chunks;
    th = (pthread_t *)calloc((size_t)n_threads, sizeof(*th));
    jobs = (job *)calloc((size_t)n_threads, sizeof(*jobs));
    if (!th || !jobs) { free(th); free(jobs); return 1; }
    t0 = now_s();                                 /* starting a thread costs microseconds, a share seconds */
    for (t = 0; t < n_threads; t++) {
        job *j = &jobs[t];
        j->scene = s; j->camera = c; j->W = W; j->H = H; j->max_depth = max_depth;
        j->chunk_x0 = chunk_x0; j->chunk_cols = chunk_cols;
        j->first_chunk = (int)((long long)n_chunks * t / n_threads);          /* contiguous shares, like the tiles' strips */
        j->last_chunk = (int)((long long)n_chunks * (t + 1) / n_threads);
        j->cpu = cpus ? cpus[t] : -1;
        j->out = out; j->rc = 0;
        if (pthread_create(&th[t], NULL, worker, j)) { rc = 1; break; }       /* its share stays unrendered: an error */
        started++;
    }
    for (t = 0; t < started; t++) { pthread_join(th[t], NULL); rc |= jobs[t].rc; }
    if (seconds) *seconds = now_s() - t0;
    free(th); free(jobs);
    return rc;
}
