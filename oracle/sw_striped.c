/*
 * sw_striped.c — batch driver over the scalar restatement (and, later, the striped SIMD
 * restatement used as the timed CPU baseline).  TEST INFRASTRUCTURE ONLY (see fade_oracle.h).
 */
#include "fade_oracle.h"
#include <pthread.h>
#include <stdlib.h>

typedef struct {
    const fo_params *p; int n, tid, nthreads; const uint8_t *q; const int64_t *q_off;
    const uint8_t *r; const int64_t *r_off; int32_t *res; uint32_t *ops; int max_ops, variant, rc;
} sw_batch_arg;

static void *sw_batch_worker(void *v) {
    sw_batch_arg *a = (sw_batch_arg *)v;
    const int64_t lo = (int64_t)a->n * a->tid / a->nthreads, hi = (int64_t)a->n * (a->tid + 1) / a->nthreads;
    for (int64_t k = lo; k < hi; k++) {
        fo_sw_result res;
        int rc = fo_sw_trace(a->p, (const char *)a->q + a->q_off[k], (int)(a->q_off[k + 1] - a->q_off[k]),
                             (const char *)a->r + a->r_off[k], (int)(a->r_off[k + 1] - a->r_off[k]), &res,
                             a->ops + (size_t)k * a->max_ops, a->max_ops);
        if (rc) a->rc = rc;
        int32_t *o = a->res + 6 * k;
        o[0] = res.score; o[1] = res.end_query; o[2] = res.end_ref;
        o[3] = res.beg_query; o[4] = res.beg_ref; o[5] = res.n_ops;
    }
    return NULL;
}

int fo_sw_batch(const fo_params *p, int n, int threads, const uint8_t *q, const int64_t *q_off,
                const uint8_t *r, const int64_t *r_off, int32_t *res, uint32_t *ops, int max_ops,
                int variant) {
    if (variant != 0) return -2;
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    sw_batch_arg *args = (sw_batch_arg *)malloc(sizeof(sw_batch_arg) * (size_t)threads);
    int rc = 0;
    for (int t = 0; t < threads; t++) {
        args[t] = (sw_batch_arg){p, n, t, threads, q, q_off, r, r_off, res, ops, max_ops, variant, 0};
        if (t > 0) pthread_create(&th[t], NULL, sw_batch_worker, &args[t]);
    }
    sw_batch_worker(&args[0]);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    for (int t = 0; t < threads; t++) if (args[t].rc) rc = args[t].rc;
    free(th); free(args);
    return rc;
}
