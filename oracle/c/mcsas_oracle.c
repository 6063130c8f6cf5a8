/* mcsas_oracle.c — TEST INFRASTRUCTURE ONLY (second CPU checker and the compiled CPU baseline).
 *
 * Plain-C restatement of the Monte-Carlo hot path of BAMresearch/McSAS for the Sphere model, the
 * model of BASELINE config 2.  It follows the same reference lines as oracle/mcsas_oracle.py:
 *   McSAS.analyse   mcsas/mcsas.py:191-285   repetition loop, up to maxRetries+1 attempts each
 *   McSAS.mcFit     mcsas/mcsas.py:287-439   one chain: proposal, test = ft - old + new, fit, accept
 *   Sphere          models/sphere.py:32-63   F = 3 (sin x - x cos x)/x^3, V = 4 pi/3 r^3
 *   calcIntensity   bases/model/sasmodel.py:37-79   it = F^2 V^(2c)
 *   fit / chi^2     mcsas/backgroundscalingfit.py:46-139, as the closed-form minimiser of the same
 *                   weighted sum of squares (what MINPACK converges to), chi^2 from the residuals (:72-77)
 *   proposals       bases/model/scatteringmodel.py:117-127, bases/algorithm/numbergenerator.py:28-31
 * Random numbers: a replayed uniform stream (what numpy.random.uniform returned to the reference) or the
 * build's Philox4x32-10 stream keyed by (seed, chain) — identical to the numpy oracle and to the device.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (mcsas_amd/) never does.  Parity: pinned through tests/test_oracle_golden.py (reference trajectories).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t nq, n_contrib, n_reps, max_retries;
    const double *q, *intensity, *sigma;
    double gen_lo, gen_hi, clip_lo, clip_hi, sld, comp_exp, conv_crit, start_value;
    int64_t max_iter;
    int32_t find_bg, pos_bg, start_from_min, rep_offset;
    uint64_t seed;
    const double *replay;          /* [n_reps][replay_len] or NULL */
    int64_t replay_len;
    /* outputs, caller allocated */
    double *contribs;              /* [n_contrib][n_reps]  (one active parameter) */
    double *fit;                   /* [nq][n_reps] */
    double *chisq, *scaling, *background;      /* [n_reps] */
    int64_t *num_iter, *num_moves, *draws, *total_steps;
    int32_t *attempts, *converged, *overflow;
    int32_t *accepted;             /* optional [n_reps][accepted_cap]: iteration index of every accepted move (last attempt) */
    int64_t accepted_cap;
} mcsas_c_problem;

/* ---- Philox4x32-10 (Salmon et al., SC'11), draw idx of chain `chain` (oracle/mcsas_oracle.py philox_uniform) */
static double philox_uniform(uint64_t seed, uint32_t chain, uint64_t idx) {
    const uint64_t blk = idx >> 1;
    uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = chain, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t a = (idx & 1) ? c2 : c0, b = (idx & 1) ? c3 : c1;
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

typedef struct { const mcsas_c_problem *p; int rep; uint64_t pos; int overflow; } stream_t;
static double draw(stream_t *s) {
    const mcsas_c_problem *p = s->p;
    double u;
    if (p->replay) {
        if ((int64_t)s->pos < p->replay_len) u = p->replay[(size_t)s->rep * p->replay_len + s->pos];
        else { s->overflow = 1; u = 0.5; }
    } else u = philox_uniform(p->seed, (uint32_t)(p->rep_offset + s->rep), s->pos);
    s->pos++;
    return u;
}

/* SASModel.calcIntensity for one sphere (sasmodel.py:46-79, sphere.py:32-63); radius clipped into its
 * valueRange the way Parameter.setValue does (bases/algorithm/parameter.py:405-414) */
static void sphere_row(const mcsas_c_problem *p, double radius, double *it) {
    const double r = fmin(fmax(radius, p->clip_lo), p->clip_hi);
    const double vol = (M_PI * 4. / 3.) * r * r * r;
    const double w = pow(vol, 2. * p->comp_exp);
    for (int k = 0; k < p->nq; ++k) {
        const double x = p->q[k] * r;
        const double f = 3. * (sin(x) - x * cos(x)) / (x * x * x);
        it[k] = f * f * w;
    }
}

/* closed-form argmin_{A,b} sum w (I - A C - b)^2 and the reduced chi^2 of its residuals */
static double fit_chisq(const mcsas_c_problem *p, const double *w, double Sw, double SwI, const double *C,
                        double *A_out, double *b_out) {
    double SwC = 0., SwCC = 0., SwIC = 0.;
    for (int k = 0; k < p->nq; ++k) { const double wc = w[k] * C[k]; SwC += wc; SwCC += wc * C[k]; SwIC += wc * p->intensity[k]; }
    double A, b;
    if (p->find_bg) {
        const double det = Sw * SwCC - SwC * SwC;
        A = (Sw * SwIC - SwI * SwC) / det;
        b = (SwI - A * SwC) / Sw;
        if (p->pos_bg && b < 0.) { A = SwIC / SwCC; b = 0.; }
    } else { A = SwIC / SwCC; b = 0.; }
    double chi = 0.;
    for (int k = 0; k < p->nq; ++k) { const double r = p->intensity[k] - (A * C[k] + b); chi += w[k] * r * r; }
    *A_out = A; *b_out = b;
    return chi / (double)p->nq;
}

/* McSAS.mcFit (mcsas.py:287-439) with the per-contribution intensity rows kept (bit-identical to re-evaluating
 * `old`, mcsas.py:362).  Returns the iterations done. */
static int64_t mc_fit(const mcsas_c_problem *p, stream_t *s, const double *w, double Sw, double SwI, double *rset,
                      double *rows, double *ft, double *test, double *newrow, double *chi_out, double *A_out,
                      double *b_out, int64_t *moves_out, int32_t *accepted, int64_t acc_cap) {
    const int N = p->n_contrib, Q = p->nq;
    for (int n = 0; n < N; ++n)                         /* generateParameters(N): N draws (scatteringmodel.py:117-127) */
        rset[n] = p->start_from_min ? p->start_value : draw(s) * (p->gen_hi - p->gen_lo) + p->gen_lo;
    memset(ft, 0, sizeof(double) * Q);
    for (int n = 0; n < N; ++n) {                       /* model.calc: rows summed in contribution order */
        sphere_row(p, rset[n], rows + (size_t)n * Q);
        for (int k = 0; k < Q; ++k) ft[k] += rows[(size_t)n * Q + k];
    }
    double A, b, chi = fit_chisq(p, w, Sw, SwI, ft, &A, &b);
    int64_t it = 0, moves = 0;
    int ri = 0;
    while (N > 1 && chi > p->conv_crit && it < p->max_iter) {       /* mcsas.py:354-357 */
        const double rt = draw(s) * (p->gen_hi - p->gen_lo) + p->gen_lo;
        sphere_row(p, rt, newrow);
        const double *old = rows + (size_t)ri * Q;
        for (int k = 0; k < Q; ++k) test[k] = ft[k] - old[k] + newrow[k];   /* :367 */
        double At, bt;
        const double chit = fit_chisq(p, w, Sw, SwI, test, &At, &bt);
        if (chit < chi) {                                                  /* :379-390 */
            rset[ri] = rt; chi = chit; A = At; b = bt;
            memcpy(ft, test, sizeof(double) * Q);
            memcpy(rows + (size_t)ri * Q, newrow, sizeof(double) * Q);
            if (accepted && moves < acc_cap) accepted[moves] = (int32_t)it;
            ++moves;
        }
        ri = (ri + 1 == N) ? 0 : ri + 1;                                   /* :403-404 */
        ++it;
    }
    chi = fit_chisq(p, w, Sw, SwI, ft, &A, &b);                            /* :424-425 */
    *chi_out = chi; *A_out = A; *b_out = b; *moves_out = moves;
    return it;
}

typedef struct { const mcsas_c_problem *p; int first, stride; } job_t;

static void *worker(void *arg) {
    const job_t *j = (const job_t *)arg;
    const mcsas_c_problem *p = j->p;
    const int N = p->n_contrib, Q = p->nq, R = p->n_reps;
    double *w = malloc(sizeof(double) * Q), *rset = malloc(sizeof(double) * N), *rows = malloc(sizeof(double) * (size_t)N * Q);
    double *ft = malloc(sizeof(double) * Q), *test = malloc(sizeof(double) * Q), *newrow = malloc(sizeof(double) * Q);
    double Sw = 0., SwI = 0.;
    for (int k = 0; k < Q; ++k) {
        const double e = p->sigma[k] == 0. ? 1. : p->sigma[k];      /* backgroundscalingfit.py:117 */
        w[k] = 1. / (e * e); Sw += w[k]; SwI += w[k] * p->intensity[k];
    }
    for (int rep = j->first; rep < R; rep += j->stride) {
        stream_t s = {p, rep, 0, 0};
        double chi = 0., A = 1., b = 0.;
        int64_t it = 0, moves = 0, total = 0;
        int attempts = 0, conv = 0;
        for (int a = 0; a <= p->max_retries; ++a) {                  /* mcsas.py:220-246 */
            ++attempts;
            it = mc_fit(p, &s, w, Sw, SwI, rset, rows, ft, test, newrow, &chi, &A, &b, &moves,
                        p->accepted ? p->accepted + (size_t)rep * p->accepted_cap : NULL, p->accepted_cap);
            total += it;
            conv = !(chi > p->conv_crit);
            if (conv) break;
        }
        for (int n = 0; n < N; ++n) p->contribs[(size_t)n * R + rep] = rset[n];
        for (int k = 0; k < Q; ++k) p->fit[(size_t)k * R + rep] = ft[k] * A + b;       /* :430 */
        p->chisq[rep] = chi; p->scaling[rep] = A; p->background[rep] = b;
        p->num_iter[rep] = it; p->num_moves[rep] = moves; p->draws[rep] = (int64_t)s.pos; p->total_steps[rep] = total;
        p->attempts[rep] = attempts; p->converged[rep] = conv; p->overflow[rep] = s.overflow;
    }
    free(w); free(rset); free(rows); free(ft); free(test); free(newrow);
    return NULL;
}

/* McSAS.analyse: repetitions spread over `n_threads` host threads (the reference runs them one after the other) */
int mcsas_c_analyse(const mcsas_c_problem *p, int n_threads) {
    if (!p || p->nq < 1 || p->n_contrib < 1 || p->n_reps < 1 || n_threads < 1) return -1;
    if (n_threads > p->n_reps) n_threads = p->n_reps;
    pthread_t *th = malloc(sizeof(pthread_t) * n_threads);
    job_t *jobs = malloc(sizeof(job_t) * n_threads);
    for (int t = 0; t < n_threads; ++t) {
        jobs[t].p = p; jobs[t].first = t; jobs[t].stride = n_threads;
        if (pthread_create(&th[t], NULL, worker, &jobs[t])) return -2;
    }
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return 0;
}

double mcsas_c_philox_uniform(uint64_t seed, uint32_t chain, uint64_t idx) { return philox_uniform(seed, chain, idx); }
