/*
 * oracle_bench.c -- TEST INFRASTRUCTURE ONLY.
 * Multi-threaded driver that runs the oracle's whole per-query path (k-mer search
 * -> family selection -> DAG -> mesh DP -> backtrack -> NAST) over a sample of
 * queries, one query per thread at a time (the reference's TBB nodes likewise run
 * one tray per worker, src/sina.cpp:497-519).  Used for the cpu_baseline leg of
 * bench.py and nowhere else.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include "sina_oracle.h"

typedef struct {
    const so_index *idx;
    const so_cseq *const *refs;
    const so_cseq *const *queries;
    uint32_t nq;
    const so_ff_opts *ff;
    const so_align_opts *al;
    atomic_uint next;
    atomic_ullong cells;
    atomic_uint aligned;
    int warm;                 /* 1: untimed first-touch pass, one query per thread */
    pthread_barrier_t *bar;
    struct timespec t_start;
} bench_job;

static void *bench_worker(void *arg) {
    bench_job *j = (bench_job *)arg;
    uint32_t n = so_index_size(j->idx);
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    float *sc = (float *)malloc(sizeof(float) * (n ? n : 1));
    const so_cseq **fam = (const so_cseq **)malloc(sizeof(so_cseq *) * (n ? n : 1));
    int warm = j->warm;
    for (;;) {
        uint32_t q;
        if (warm) {
            /* steady state is what the metric means (100k queries): touch this thread's mesh
             * scratch once before the clock starts */
            q = 0;
        } else {
            q = atomic_fetch_add(&j->next, 1);
            if (q >= j->nq) break;
        }
        so_log lg;
        so_log_init(&lg);
        uint32_t nf = so_famfinder(j->idx, j->refs, j->queries[q], j->ff, ids, sc, n, &lg);
        if (nf) {
            for (uint32_t i = 0; i < nf; i++) fam[i] = j->refs[ids[i]];
            so_cseq *out = so_cseq_new("out");
            so_align_result res;
            so_align(fam, nf, j->queries[q], j->al, out, &res, &lg);
            if (!warm) {
                atomic_fetch_add(&j->cells, (unsigned long long)res.cells);
                if (res.status == 0 || res.status == 1) atomic_fetch_add(&j->aligned, 1);
            }
            so_cseq_free(out);
        }
        so_log_free(&lg);
        if (warm) {
            warm = 0;
            if (pthread_barrier_wait(j->bar) == PTHREAD_BARRIER_SERIAL_THREAD) clock_gettime(CLOCK_MONOTONIC, &j->t_start);
            pthread_barrier_wait(j->bar);
        }
    }
    free(ids);
    free(sc);
    free(fam);
    return NULL;
}

/* Memory policy of the calling thread and the threads it starts afterwards: on != 0 interleaves new
 * pages over all online NUMA nodes (what `numactl --interleave=all` does; the mesh scratch of many
 * threads otherwise lands on the node of whoever touched it first), 0 restores the default.
 * Returns 0 on success, -1 when the kernel refuses (single-node hosts: nothing to do). */
int so_bench_mempolicy_interleave(int on) {
#ifdef SYS_set_mempolicy
    unsigned long mask[16] = {0};
    if (!on) return (int)syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, NULL, 0);
    FILE *f = fopen("/sys/devices/system/node/online", "r");
    if (!f) return -1;
    char buf[256] = {0};
    if (!fgets(buf, sizeof buf, f)) buf[0] = 0;
    fclose(f);
    int any = 0;
    for (char *p = buf; *p;) { /* "0-3,8" */
        char *e;
        long a = strtol(p, &e, 10), b;
        if (e == p) break;
        b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long x = a; x <= b && x < 1024; x++) mask[x / (8 * sizeof(long))] |= 1UL << (x % (8 * sizeof(long))), any++;
        p = (*e == ',') ? e + 1 : e;
        if (*e != ',') break;
    }
    if (any < 2) return -1;
    return (int)syscall(SYS_set_mempolicy, 3 /* MPOL_INTERLEAVE */, mask, 1024 + 1);
#else
    (void)on;
    return -1;
#endif
}

/* returns wall seconds; *cells = mesh cells filled, *aligned = queries with a result */
double so_bench_run(const so_index *idx, const so_cseq *const *refs, const so_cseq *const *queries, uint32_t nq,
                    const so_ff_opts *ff, const so_align_opts *al, uint32_t threads, uint64_t *cells,
                    uint32_t *aligned) {
    bench_job j;
    memset(&j, 0, sizeof(j));
    j.idx = idx;
    j.refs = refs;
    j.queries = queries;
    j.nq = nq;
    j.ff = ff;
    j.al = al;
    atomic_init(&j.next, 0);
    atomic_init(&j.cells, 0);
    atomic_init(&j.aligned, 0);
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, threads);
    j.bar = &bar;
    j.warm = nq > 0;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &j.t_start);
    for (uint32_t i = 1; i < threads; i++) pthread_create(&th[i], NULL, bench_worker, &j);
    bench_worker(&j);
    for (uint32_t i = 1; i < threads; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &b);
    a = j.t_start;
    pthread_barrier_destroy(&bar);
    free(th);
    if (cells) *cells = (uint64_t)atomic_load(&j.cells);
    if (aligned) *aligned = atomic_load(&j.aligned);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

/* parallel index build helper: so_index_build is single-threaded like IndexBuilder on one
 * range; building 100k references for the baseline takes a while, so let callers know. */
