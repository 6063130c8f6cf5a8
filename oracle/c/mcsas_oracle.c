/* mcsas_oracle.c — TEST INFRASTRUCTURE ONLY (second CPU checker and the compiled CPU baseline).
 *
 * Plain-C restatement of the Monte-Carlo hot path of BAMresearch/McSAS for the models of BASELINE configs 2-4:
 * Sphere, CylindersIsotropic, EllipsoidalCoreShell.  It follows the same reference lines as oracle/mcsas_oracle.py:
 *   McSAS.analyse          mcsas/mcsas.py:191-285   repetition loop, up to maxRetries+1 attempts each
 *   McSAS.mcFit            mcsas/mcsas.py:287-439   one chain: proposal, test = ft - old + new, fit, accept
 *   Sphere                 models/sphere.py:32-63   F = 3 (sin x - x cos x)/x^3, V = 4 pi/3 r^3
 *   CylindersIsotropic     models/cylindersisotropic.py:50-101   trapezoid over x in [0,1] of (J1(qR sqrt(1-x^2)) sin(qLx/2) /
 *                          (qR sqrt(1-x^2) qLx))^2 with the two analytic end columns (:79-82); V = pi R^2 2 halfLength
 *   EllipsoidalCoreShell   models/ellipsoidalcoreshell.py:59-97   mean over mu of the core + shell amplitude squared
 *   calcIntensity          bases/model/sasmodel.py:37-79   it = F^2 V^(2c); parameters clipped into their valueRange the way
 *                          Parameter.setValue does (bases/algorithm/parameter.py:405-414)
 *   fit / chi^2            mcsas/backgroundscalingfit.py:46-139, as the closed-form minimiser of the same weighted sum of
 *                          squares (what MINPACK converges to), chi^2 from the residuals (:72-77)
 *   proposals              bases/model/scatteringmodel.py:117-127 (column by column, `count` draws per active parameter),
 *                          bases/algorithm/numbergenerator.py:28-31 (uniform), :168-191 (RandomExponential 1/2/3 decades)
 * Third-party arithmetic: sin / cos / pow / sqrt from libm; the Bessel function J1 (scipy.special.j1 in the reference, which is
 * Cephes' j1.c) restated from the published Cephes algorithm below (rational approximations RP/RQ on [0, 5], the asymptotic
 * modulus / phase rationals PP/PQ, QP/QQ beyond), pinned against scipy on the CPU (tests/test_oracle_golden.py).
 * Random numbers: a replayed uniform stream (what numpy.random.uniform returned to the reference) or the build's
 * Philox4x32-10 stream keyed by (seed, chain) — identical to the numpy oracle and to the device.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product (mcsas_amd/) never
 * does.  Parity: pinned through tests/test_oracle_golden.py (the reference's form-factor vectors G1/G2 and its trajectories
 * G4 / G9 / G14 / G17 for all three models).
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { MODEL_SPHERE = 0, MODEL_CYL_ISO = 1, MODEL_ELL_CS = 2 };
#define MAX_ACTIVE 4
#define MAX_PARAMS 8

typedef struct {
    int32_t model, nq, n_contrib, n_reps, max_retries, n_active;
    int32_t active[MAX_ACTIVE];        /* index of every active parameter in the model's full parameter vector (ascending) */
    int32_t gen[MAX_ACTIVE];           /* 0: RandomUniform; K = 1, 2, 3: RandomExponential over K decades */
    const double *q, *intensity, *sigma;
    double gen_lo[MAX_ACTIVE], gen_hi[MAX_ACTIVE];      /* activeRange ∩ valueRange */
    double start_value[MAX_ACTIVE];    /* startFromMinimum (mcsas.py:310-315) */
    double values[MAX_PARAMS];         /* full parameter vector; inactive entries are used as they are */
    double clip_lo[MAX_PARAMS], clip_hi[MAX_PARAMS];    /* valueRange of every parameter */
    double comp_exp, conv_crit;
    int64_t max_iter;
    int32_t find_bg, pos_bg, start_from_min, rep_offset;
    uint64_t seed;
    const double *replay;          /* [n_reps][replay_len] or NULL */
    int64_t replay_len;
    /* outputs, caller allocated */
    double *contribs;              /* [n_contrib][n_active][n_reps] */
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

/* NumberGenerator.get + generateValues (numbergenerator.py:28-31,168-191; bases/algorithm/parameter.py:66-84) */
static double generate(const mcsas_c_problem *p, int col, double u) {
    if (p->gen[col] != 0) {
        const double upper = (double)p->gen[col];
        u = (pow(10., 0. + (upper - 0.) * u) - 1.) / pow(10., upper - 0.);
    }
    return u * (p->gen_hi[col] - p->gen_lo[col]) + p->gen_lo[col];
}

/* ---- Bessel J1: the Cephes algorithm (j1.c, S. Moshier; what scipy.special.j1 runs).  polevl evaluates
 * c[0] x^n + ... + c[n]; p1evl the same with an implicit leading coefficient 1. */
static double polevl(double x, const double *c, int n) { double a = c[0]; for (int i = 1; i <= n; ++i) a = a * x + c[i]; return a; }
static double p1evl(double x, const double *c, int n) { double a = x + c[0]; for (int i = 1; i < n; ++i) a = a * x + c[i]; return a; }
static const double J1_RP[4] = {-8.99971225705559398224E8, 4.52228297998194034323E11, -7.27494245221818276015E13, 3.68295732863852883286E15};
static const double J1_RQ[8] = {6.20836478118054335476E2, 2.56987256757748830383E5, 8.35146791431949253037E7, 2.21511595479792499675E10,
                                4.74914122079991414898E12, 7.84369607876235854894E14, 8.95222336184627338078E16, 5.32278620332680085395E18};
static const double J1_PP[7] = {7.62125616208173112003E-4, 7.31397056940917570436E-2, 1.12719608129684925192E0, 5.11207951146807644818E0,
                                8.42404590141772420927E0, 5.21451598682361504063E0, 1.00000000000000000254E0};
static const double J1_PQ[7] = {5.71323128072548699714E-4, 6.88455908754495404082E-2, 1.10514232634061696926E0, 5.07386386128601488557E0,
                                8.39985554327604159757E0, 5.20982848682361821619E0, 9.99999999999999997461E-1};
static const double J1_QP[8] = {5.10862594750176621635E-2, 4.98213872951233449420E0, 7.58238284132545283818E1, 3.66779609360150777800E2,
                                7.10856304998926107277E2, 5.97489612400613639965E2, 2.11688757100572135698E2, 2.52070205858023719784E1};
static const double J1_QQ[7] = {7.42373277035675149943E1, 1.05644886038262816351E3, 4.98641058337653607651E3, 9.56231892404756170795E3,
                                7.99704160447350683650E3, 2.82619278517639096600E3, 3.36093607810698293419E2};
static double cephes_j1(double x) {
    if (x < 0.) return -cephes_j1(-x);
    if (x <= 5.0) {
        const double z = x * x;
        const double w = polevl(z, J1_RP, 3) / p1evl(z, J1_RQ, 8);
        return w * x * (z - 1.46819706421238932572E1) * (z - 4.92184563216946036703E1);
    }
    const double w = 5.0 / x, z = w * w;
    const double pp = polevl(z, J1_PP, 6) / polevl(z, J1_PQ, 6);
    const double qq = polevl(z, J1_QP, 7) / p1evl(z, J1_QQ, 7);
    const double xn = x - 2.35619449019234492885;                 /* 3 pi / 4 */
    return (pp * cos(xn) - w * qq * sin(xn)) * 7.9788456080286535587989E-1 / sqrt(x);
}
double mcsas_c_j1(double x) { return cephes_j1(x); }

/* ---- SASModel.calcIntensity for one contribution (sasmodel.py:46-79): it[k] = F(q_k)^2 volume()^(2c).  `par`: the full
 * parameter vector with the active columns already set and clipped.  `tab`: scratch of at least 4 * intDiv doubles. */
static void row_sphere(const mcsas_c_problem *p, const double *par, double *it) {             /* sphere.py:32-63 */
    const double r = par[0];
    const double vol = (M_PI * 4. / 3.) * r * r * r;
    const double w = pow(vol, 2. * p->comp_exp);
    for (int k = 0; k < p->nq; ++k) {
        const double x = p->q[k] * r;
        const double f = 3. * (sin(x) - x * cos(x)) / (x * x * x);
        it[k] = f * f * w;
    }
}
static void row_cylinders_isotropic(const mcsas_c_problem *p, const double *par, double *it, double *tab) {   /* cylindersisotropic.py:50-101 */
    const double r = par[0], length = par[2], aspect = par[3];
    const int use_aspect = par[1] != 0., K = (int)par[4];
    const double hl = use_aspect ? r * aspect : 0.5 * length;                                 /* :65-68 */
    const double vol = M_PI * (r * r) * (hl * 2.);                                              /* :92-98 */
    const double w = pow(vol, 2. * p->comp_exp);
    const double step = 1. / (double)(K - 1);                                                 /* numpy.linspace(0, 1, K, retstep) */
    double *rs = tab, *lx = tab + K;
    for (int j = 0; j < K; ++j) {
        double x = (j == K - 1) ? 1. : (double)j * step;
        if (j == 0 || j == K - 1) x = 0.5;                                                    /* :60-61 (overwritten below) */
        rs[j] = r * sqrt(1. - x * x);
        lx[j] = 2. * hl * x;
    }
    for (int k = 0; k < p->nq; ++k) {
        const double q = p->q[k];
        double prev = 0., sum = 0.;
        for (int j = 0; j < K; ++j) {
            double f;
            if (j == 0) f = 0.5 * (cephes_j1(q * r) / (q * r));                               /* :79-80 */
            else if (j == K - 1) f = sin(q * hl) / (q * hl);                                  /* :82 */
            else {
                const double a = q * rs[j], b = q * lx[j];
                f = (cephes_j1(a) * sin(b / 2.)) / (a * b);                                   /* :73-75 */
            }
            const double y = f * f;
            if (j) sum += step * (y + prev) / 2.0;                                            /* numpy.trapz(y, dx = step) */
            prev = y;
        }
        const double ff = sqrt(16 * sum);                                                     /* :90 */
        it[k] = ff * ff * w;
    }
}
static void row_ellipsoidal_core_shell(const mcsas_c_problem *p, const double *par, double *it, double *tab) {   /* ellipsoidalcoreshell.py:59-97 */
    const double a = par[0], b = par[1], t = par[2], eta_c = par[3], eta_s = par[4], eta_sol = par[5];
    const int K = (int)par[6];
    const double vol = 4. / 3 * M_PI * (a + t) * ((b + t) * (b + t));                         /* :92-94 */
    const double w = pow(vol, 2. * p->comp_exp);
    const double vc = 4. / 3. * M_PI * a * (b * b);
    const double v_ratio = vc / vol;                                                          /* :75-77 */
    double *sc = tab, *st = tab + K;
    for (int j = 0; j < K; ++j) {
        const double mu = (j == K - 1) ? 1. : (double)j * (1. / (double)(K - 1));      /* numpy.linspace(0, 1, K) */
        sc[j] = sqrt(a * a * (mu * mu) + b * b * (1 - mu * mu));                              /* :63-67 */
        st[j] = sqrt((a + t) * (a + t) * (mu * mu) + (b + t) * (b + t) * (1 - mu * mu));      /* :69-73 */
    }
    for (int k = 0; k < p->nq; ++k) {
        const double q = p->q[k];
        double sum = 0.;
        for (int j = 0; j < K; ++j) {
            const double xc = q * sc[j], xt = q * st[j];
            double s1, c1, s2, c2;
            sincos(xc, &s1, &c1); sincos(xt, &s2, &c2);
            const double jc = (s1 - xc * c1) / (xc * xc), jt = (s2 - xt * c2) / (xt * xt);    /* :60-61 */
            const double f = (eta_c - eta_s) * v_ratio * (3 * jc / xc) + (eta_s - eta_sol) * 1. * (3 * jt / xt);   /* :81-86 */
            sum += f * f;
        }
        const double ff = sqrt(sum / (double)K);                                              /* :88 numpy.mean */
        it[k] = ff * ff * w;
    }
}
static void calc_row(const mcsas_c_problem *p, const double *row, double *it, double *tab) {
    double par[MAX_PARAMS];
    memcpy(par, p->values, sizeof(par));
    for (int c = 0; c < p->n_active; ++c) {
        const int i = p->active[c];
        par[i] = fmin(fmax(row[c], p->clip_lo[i]), p->clip_hi[i]);
    }
    if (p->model == MODEL_SPHERE) row_sphere(p, par, it);
    else if (p->model == MODEL_CYL_ISO) row_cylinders_isotropic(p, par, it, tab);
    else row_ellipsoidal_core_shell(p, par, it, tab);
}
/* ScatteringModel.calc for one parameter set (scatteringmodel.py:79-105): the summed intensity; test hook */
int mcsas_c_model_calc(const mcsas_c_problem *p, const double *pset, int n, double *cum) {
    double *it = malloc(sizeof(double) * p->nq), *tab = malloc(sizeof(double) * 4 * 10000);
    memset(cum, 0, sizeof(double) * p->nq);
    for (int i = 0; i < n; ++i) {
        calc_row(p, pset + (size_t)i * p->n_active, it, tab);
        for (int k = 0; k < p->nq; ++k) cum[k] += it[k];
    }
    free(it); free(tab);
    return 0;
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
                      double *rows, double *ft, double *test, double *newrow, double *tab, double *chi_out, double *A_out,
                      double *b_out, int64_t *moves_out, int32_t *accepted, int64_t acc_cap) {
    const int N = p->n_contrib, Q = p->nq, P = p->n_active;
    for (int c = 0; c < P; ++c)                         /* generateParameters(N): column by column, N draws each (scatteringmodel.py:117-127) */
        for (int n = 0; n < N; ++n)
            rset[(size_t)n * P + c] = p->start_from_min ? p->start_value[c] : generate(p, c, draw(s));
    memset(ft, 0, sizeof(double) * Q);
    for (int n = 0; n < N; ++n) {                       /* model.calc: rows summed in contribution order */
        calc_row(p, rset + (size_t)n * P, rows + (size_t)n * Q, tab);
        for (int k = 0; k < Q; ++k) ft[k] += rows[(size_t)n * Q + k];
    }
    double A, b, chi = fit_chisq(p, w, Sw, SwI, ft, &A, &b);
    int64_t it = 0, moves = 0;
    int ri = 0;
    double rt[MAX_ACTIVE];
    while (N > 1 && chi > p->conv_crit && it < p->max_iter) {       /* mcsas.py:354-357 */
        for (int c = 0; c < P; ++c) rt[c] = generate(p, c, draw(s));       /* :358 */
        calc_row(p, rt, newrow, tab);                                      /* :360 */
        const double *old = rows + (size_t)ri * Q;
        for (int k = 0; k < Q; ++k) test[k] = ft[k] - old[k] + newrow[k];   /* :367 */
        double At, bt;
        const double chit = fit_chisq(p, w, Sw, SwI, test, &At, &bt);
        if (chit < chi) {                                                  /* :379-390 */
            for (int c = 0; c < P; ++c) rset[(size_t)ri * P + c] = rt[c];
            chi = chit; A = At; b = bt;
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
    const int N = p->n_contrib, Q = p->nq, R = p->n_reps, P = p->n_active;
    double *w = malloc(sizeof(double) * Q), *rset = malloc(sizeof(double) * N * P), *rows = malloc(sizeof(double) * (size_t)N * Q);
    double *ft = malloc(sizeof(double) * Q), *test = malloc(sizeof(double) * Q), *newrow = malloc(sizeof(double) * Q);
    double *tab = malloc(sizeof(double) * 4 * 10000);                /* intDiv <= 1e4 (valueRange of both models) */
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
            it = mc_fit(p, &s, w, Sw, SwI, rset, rows, ft, test, newrow, tab, &chi, &A, &b, &moves,
                        p->accepted ? p->accepted + (size_t)rep * p->accepted_cap : NULL, p->accepted_cap);
            total += it;
            conv = !(chi > p->conv_crit);
            if (conv) break;
        }
        for (int n = 0; n < N; ++n)
            for (int c = 0; c < P; ++c) p->contribs[((size_t)n * P + c) * R + rep] = rset[(size_t)n * P + c];
        for (int k = 0; k < Q; ++k) p->fit[(size_t)k * R + rep] = ft[k] * A + b;       /* :430 */
        p->chisq[rep] = chi; p->scaling[rep] = A; p->background[rep] = b;
        p->num_iter[rep] = it; p->num_moves[rep] = moves; p->draws[rep] = (int64_t)s.pos; p->total_steps[rep] = total;
        p->attempts[rep] = attempts; p->converged[rep] = conv; p->overflow[rep] = s.overflow;
    }
    free(w); free(rset); free(rows); free(ft); free(test); free(newrow); free(tab);
    return NULL;
}

/* McSAS.analyse: repetitions spread over `n_threads` host threads (the reference runs them one after the other) */
int mcsas_c_analyse(const mcsas_c_problem *p, int n_threads) {
    if (!p || p->nq < 1 || p->n_contrib < 1 || p->n_reps < 1 || n_threads < 1) return -1;
    if (p->model < MODEL_SPHERE || p->model > MODEL_ELL_CS || p->n_active < 1 || p->n_active > MAX_ACTIVE) return -3;
    if (p->model != MODEL_SPHERE && (p->values[p->model == MODEL_CYL_ISO ? 4 : 6] < 2. || p->values[p->model == MODEL_CYL_ISO ? 4 : 6] > 1e4)) return -4;
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
