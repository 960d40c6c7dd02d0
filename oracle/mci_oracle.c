/*
 * mci_oracle.c -- CPU ORACLE (test infrastructure, NOT the product; see mci_oracle.h).
 *
 * Plain-C restatement of MCIntegration.jl's VEGAS / VegasMC path.  All indices that
 * mirror Julia arrays are kept 1-based (arrays are allocated with one spare leading
 * element) so that every line can be checked against the cited reference line.
 *
 * "ref:" comments give /root/reference/<file>:<line>.
 */
#include "mci_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif


#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * RNG: Philox4x32-10 (Salmon et al., SC'11; Random123 v1.14 constants).  Replaces
 * Random.MersenneTwister (ref: src/configuration.jl:190) -- see "PINNING STATUS" in the header.
 * ---------------------------------------------------------------------------------------- */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

/* Philox4x32-R (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): R = 10 is the Random123 default, R = 7 the
 * fewest rounds its authors found crush-resistant -- the opt-in cheaper stream (mci_set_rng_rounds) */
void mcio_philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* rounds of every stream below: process-wide like the library's per-problem setting is problem-wide (the oracle runs one
 * configuration at a time); mirror of mci_set_rng_rounds */
static int g_rng_rounds = 10;
void mcio_set_rng_rounds(int rounds) { g_rng_rounds = rounds == 7 ? 7 : 10; }
int mcio_get_rng_rounds(void) { return g_rng_rounds; }
void mcio_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { mcio_philox4x32_r(ctr, key, out, 10); }

/* Stream contract shared with the HIP path (DESIGN.md "RNG streams"):
 *   key = (seed lo, seed hi); ctr = (index lo, index hi, k>>1, stream)
 *   draw k uses words (2*(k&1), 2*(k&1)+1); 52 mantissa bits, [1,2)-1 -> [0,1).
 * rand(config.rng) in the reference is likewise a [0,1) Float64 (ref: sampler.jl:296,361). */
double mcio_uniform(uint64_t seed, uint32_t stream, uint64_t index, uint32_t k) {
    uint32_t ctr[4] = {(uint32_t)index, (uint32_t)(index >> 32), k >> 1, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    mcio_philox4x32_r(ctr, key, o, g_rng_rounds);
    uint32_t a = o[2 * (k & 1)], b = o[2 * (k & 1) + 1];
    uint64_t bits = ((((uint64_t)b << 32) | a) >> 12) | 0x3FF0000000000000ull; /* [1,2) */
    double d;
    memcpy(&d, &bits, sizeof d);
    return d - 1.0; /* 52 random mantissa bits, like Julia's MersenneTwister rand(Float64) */
}

/* The opt-in 32-bit stream of solver = :vegas (mci_set_rng_bits): ONE Philox word per draw -- block k >> 2, word k & 3 -- whose 32 bits
 * become the top 32 mantissa bits of a double in [1, 2); y = that - 1 lies on a 2^-32 lattice. */
double mcio_uniform32(uint64_t seed, uint32_t stream, uint64_t index, uint32_t k) {
    uint32_t ctr[4] = {(uint32_t)index, (uint32_t)(index >> 32), k >> 2, stream};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    mcio_philox4x32_r(ctr, key, o, g_rng_rounds);
    uint64_t bits = ((uint64_t)o[k & 3] << 20) | 0x3FF0000000000000ull; /* [1,2) */
    double d;
    memcpy(&d, &bits, sizeof d);
    return d - 1.0;
}

enum { STREAM_VEGAS = 0, STREAM_POOLINIT = 1, STREAM_MC_INIT = 2, STREAM_MC_STEP = 3, STREAM_MCMC_INIT = 4, STREAM_MCMC_STEP = 5, STREAM_MCMC_GROUP = 6, STREAM_MC_GROUP = 7 };
static inline uint32_t stream_id(uint32_t iteration, int purpose) { return iteration * 8u + (uint32_t)purpose; }
/* chain solvers: a chain is identified by (block, chain within the block); the block index rides in the top 12 bits of the
   stream word so that a block's streams do not depend on how many chains other blocks run (iteration < 131072, block < 4096) */
static inline uint32_t stream_id_block(uint32_t iteration, int purpose, long block) { return stream_id(iteration, purpose) + ((uint32_t)block << 20); }

/* ------------------------------------------------------------------------------------------
 * src/distribution/common.jl
 * ---------------------------------------------------------------------------------------- */

/* ref: common.jl:8-36.  acc is 0-based C storage of a Julia vector of length n; returns the
 * 1-based idx with acc[idx] <= p < acc[idx+1]; -1 where the reference raises error(). */
long mcio_locate(const double *acc, long n, double p) {
    if (acc[0] > p || acc[n - 1] <= p) return -1; /* :10-13 */
    long jl = 1, ju = n + 1;                      /* :16-17 */
    while (ju - jl > 1) {                         /* :18 */
        long jm = (jl + ju) / 2;                  /* :19 */
        if (p < acc[jm - 1]) ju = jm;             /* :20-21 */
        else jl = jm;                             /* :22-23 */
    }
    return jl;                                    /* :34 */
}

/* ref: common.jl:43-54 */
void mcio_smooth(const double *dist, long n, double factor, double *out) {
    if (n <= 1) { /* :44-46 */
        for (long i = 0; i < n; ++i) out[i] = dist[i];
        return;
    }
    out[0] = (dist[0] * (factor + 1) + dist[1]) / (factor + 2);             /* :48 */
    out[n - 1] = (dist[n - 1] * (factor + 1) + dist[n - 2]) / (factor + 2); /* :49 */
    for (long i = 1; i < n - 1; ++i)                                        /* :50-52 */
        out[i] = (dist[i - 1] + dist[i] * factor + dist[i + 1]) / (factor + 2);
}

/* Julia's sum() over a Vector{Float64} shorter than pairwise_blocksize = 1024 (Base.mapreduce_impl, base/reduce.jl) is an
 * `@simd` loop: LLVM vectorises the reduction, so its association is the host CPU's (vector lanes x interleave), not left to
 * right -- the reference has no single order.  The oracle (and the device) fix ONE association of that family: 16 interleaved
 * partial sums (element i -> partial i mod 16, each left to right), folded p[l] += p[l + h] for h = 8, 4, 2, 1 -- the lanes x
 * interleave of an AVX2 build.  It is not any particular Julia binary's order to the last bit: Base._mapreduce sums vectors of fewer
 * than 16 elements left to right, and mapreduce_impl adds the first two elements before its @simd loop starts at the third (with a
 * scalar tail behind the vector body); differences are in the last bit of a sum of ~1000 positive terms, below every tolerance here.
 * Used where the reference sums a histogram-length vector: rescale (common.jl:72) and f_ninc (variable.jl:226).
 * From 1025 elements on mapreduce_impl first splits the range at its midpoint, imid = ifirst + (ilast - ifirst) >> 1, sums the two
 * halves the same way and adds the two results (base/reduce.jl mapreduce_impl, pairwise_blocksize = 1024): mcio_sum_julia below, and
 * sum_julia on the device (mci_train.h). */
double mcio_sum16(const double *v, long n) {
    double p[16] = {0};
    for (long i = 0; i < n; ++i) p[i & 15] += v[i];
    for (int h = 8; h >= 1; h >>= 1)
        for (int l = 0; l < h; ++l) p[l] += p[l + h];
    return p[0];
}

/* b ^ alpha the way Julia's ^(::Float64, ::Float64) evaluates it for the exponents the reference's constructors hand out
 * (alpha = 2 by default, variable.jl:137; 3 in the bubble example): an integer-valued exponent takes Base.Math.pow_body's
 * compensated power-by-squaring, whose result for n = 2 is the correctly rounded x*x and for n = 3 literally x*x*x; libm's pow is
 * within an ulp of both, but not always the same bits.  Anything else: pow(). */
static double mcio_pow_julia(double b, double alpha) {
    if (alpha == 2.0) return b * b;
    if (alpha == 1.0) return b;
    if (alpha == 3.0) return b * b * b;
    return pow(b, alpha);
}

/* Julia's sum() of a Vector{Float64} of any length: the @simd block below 1025 elements, pairwise halves above */
double mcio_sum_julia(const double *v, long n) {
    if (n <= 1024) return mcio_sum16(v, n);
    const long h = ((n - 1) >> 1) + 1; /* [ifirst, imid] holds (ilast - ifirst) >> 1 + 1 elements */
    return mcio_sum_julia(v, h) + mcio_sum_julia(v + h, n - h);
}

/* ref: common.jl:67-82.  Output is NOT renormalised (:81-82).  Returns 1/2 where the
 * reference's @assert (:71 / :79) fires.  sum() is mcio_sum_julia (above). */
int mcio_rescale(double *dist, long n, double alpha) {
    if (n == 1) return 0; /* :68-70 */
    for (long i = 0; i < n; ++i)
        if (!(dist[i] > 0)) return 1; /* :71 */
    const double s = mcio_sum_julia(dist, n);
    for (long i = 0; i < n; ++i) dist[i] /= s; /* :72 */
    for (long i = 0; i < n; ++i)               /* :74-78 */
        if (dist[i] > 0 && dist[i] <= 0.99999999) dist[i] = mcio_pow_julia(-(1 - dist[i]) / log(dist[i]), alpha);
    for (long i = 0; i < n; ++i)
        if (!isfinite(dist[i])) return 2; /* :79 */
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * src/distribution/variable.jl : train!
 * ---------------------------------------------------------------------------------------- */

/* ref: variable.jl:206-239 (Continuous).  grid[npts] and hist[npts-1] are 0-based storage. */
int mcio_train_continuous(double *grid, long npts, double *hist, double alpha) {
    long N = npts - 1;
    for (long i = 0; i < N; ++i)
        if (!isfinite(hist[i])) return 3; /* :212 */
    for (long i = 0; i < N; ++i)
        if (!(hist[i] > 0)) return 4; /* :213 */
    double *d = (double *)malloc(sizeof(double) * (size_t)N);
    double *newgrid = (double *)malloc(sizeof(double) * (size_t)npts);
    mcio_smooth(hist, N, 6.0, d);     /* :214 */
    int rc = mcio_rescale(d, N, alpha); /* :215 */
    if (rc) { free(d); free(newgrid); return rc; }
    newgrid[0] = grid[0];               /* :217 */
    newgrid[npts - 1] = grid[npts - 1]; /* :218 */
    long j = 0;                         /* :221 */
    double acc_f = 0.0;                 /* :222 */
    double f_ninc = mcio_sum_julia(d, N) / (double)N; /* :226 */
    for (long i = 2; i <= npts - 1; ++i) { /* :227 (1-based i) */
        while (acc_f < f_ninc) {           /* :228 */
            j += 1;                        /* :229 */
            acc_f += d[j - 1];             /* :230 */
        }
        acc_f -= f_ninc;                   /* :232 */
        /* :233  T.grid[j+1] - (acc_f/avg_f[j])*(T.grid[j+1]-T.grid[j]) with 1-based j */
        newgrid[i - 1] = grid[j] - (acc_f / d[j - 1]) * (grid[j] - grid[j - 1]);
    }
    newgrid[npts - 1] = grid[npts - 1]; /* :235 */
    memcpy(grid, newgrid, sizeof(double) * (size_t)npts); /* :236 */
    for (long i = 0; i < N; ++i) hist[i] = 1.0e-10;       /* :238 -> :565 */
    free(d);
    free(newgrid);
    return 0;
}

/* ref: variable.jl:369-382 (Discrete).  No smoothing. */
int mcio_train_discrete(double *hist, long K, double alpha, double *distribution, double *accumulation) {
    double *d = (double *)malloc(sizeof(double) * (size_t)K);
    memcpy(d, hist, sizeof(double) * (size_t)K); /* :373 */
    int rc = mcio_rescale(d, K, alpha);          /* :374 */
    if (rc) { free(d); return rc; }
    double s = 0.0;
    for (long i = 0; i < K; ++i) s += d[i];
    for (long i = 0; i < K; ++i) d[i] /= s;      /* :375 */
    accumulation[0] = 0.0;                        /* :377 */
    double run = 0.0;
    for (long i = 0; i < K; ++i) {                /* :376 sum(distribution[1:i]) */
        run += d[i];
        accumulation[i + 1] = run;
    }
    memcpy(distribution, d, sizeof(double) * (size_t)K); /* :378 */
    for (long i = 0; i < K; ++i) hist[i] = 1.0e-10;      /* :381 */
    free(d);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Configuration  (src/configuration.jl:105-194) and variable constructors
 * ---------------------------------------------------------------------------------------- */

#define MAXORDER 16 /* ref: distribution.jl:59 */

void mcio_maxdof(const int *dof, int nd, int npool, int *out) { /* ref: configuration.jl:229-236 */
    for (int v = 0; v < npool; ++v) {
        int m = 0;
        for (int i = 0; i < nd; ++i)
            if (dof[i * npool + v] > m) m = dof[i * npool + v];
        out[v] = m;
    }
}

static void leaf_alloc_pool(mcio_leaf *L, int P) {
    L->P = P;
    L->data = (double *)calloc(((size_t)P + 1) * (size_t)(L->width > 0 ? L->width : 1), sizeof(double));
    L->gidx = (long *)calloc((size_t)P + 1, sizeof(long));
    L->prob = (double *)calloc((size_t)P + 1, sizeof(double));
}

/* ref: variable.jl:137-153 (Continuous ctor) / :299-325 (Discrete ctor) */
static void leaf_init(mcio_leaf *L, int kind, int pool, double lower, double upper, int npts, double alpha,
                      int adapt, int P) {
    memset(L, 0, sizeof(*L));
    L->kind = kind;
    L->pool = pool;
    L->lower = lower;
    L->upper = upper;
    L->alpha = alpha;
    L->adapt = adapt;
    L->width = kind == MCIO_FERMIK ? npts : 1;
    leaf_alloc_pool(L, P);
    if (kind == MCIO_FERMIK) { /* variable.jl:11-19: k = kF/sqrt(dim) in every component, prob = 1, histogram = [0.0] */
        L->npts = npts;
        L->nbin = 1;
        L->adapt = 0;
        L->hist = (double *)calloc(1, sizeof(double));
        for (int i = 1; i <= P; ++i) {
            for (int j = 0; j < npts; ++j) L->data[i * npts + j] = lower / sqrt((double)npts);
            L->prob[i] = 1.0;
        }
    } else if (kind == MCIO_CONTINUOUS) {
        L->npts = npts;
        L->nbin = npts - 1; /* :147 */
        L->grid = (double *)malloc(sizeof(double) * (size_t)npts);
        /* :137 grid = collect(LinRange(lower, upper, ninc)); Julia's LinRange lerps
           (1-t)*a + t*b with t = i/(n-1) */
        for (int i = 0; i < npts; ++i) {
            double t = (double)i / (double)(npts - 1);
            L->grid[i] = (1.0 - t) * lower + t * upper;
        }
        L->grid[0] = lower;
        L->grid[npts - 1] = upper;
        L->hist = (double *)malloc(sizeof(double) * (size_t)L->nbin);
        for (int i = 0; i < L->nbin; ++i) L->hist[i] = MCIO_TINY; /* :149 */
        /* :141-145 deterministic initial pool contents */
        for (int i = 1; i <= P; ++i) {
            double a = lower + (upper - lower) / P, b = upper - (upper - lower) / P;
            double t = (P > 1) ? (double)(i - 1) / (double)(P - 1) : 0.0;
            L->data[i] = (1.0 - t) * a + t * b;
            long g = mcio_locate(L->grid, npts, L->data[i]);
            L->gidx[i] = g < 1 ? 1 : g;
            L->prob[i] = 1.0;
        }
    } else {
        int K = (int)(upper - lower) + 1;
        L->npts = K;
        L->nbin = K; /* :305 */
        L->hist = (double *)malloc(sizeof(double) * (size_t)K);
        L->distribution = (double *)malloc(sizeof(double) * (size_t)K);
        L->accumulation = (double *)malloc(sizeof(double) * (size_t)(K + 1));
        for (int i = 0; i < K; ++i) L->distribution[i] = MCIO_TINY; /* :305-307 */
        double s = 0.0;
        for (int i = 0; i < K; ++i) s += L->distribution[i];
        for (int i = 0; i < K; ++i) L->distribution[i] /= s; /* :312 */
        L->accumulation[0] = 0.0;                             /* :313-314 */
        double run = 0.0;
        for (int i = 0; i < K; ++i) {
            run += L->distribution[i];
            L->accumulation[i + 1] = run;
        }
        for (int i = 1; i <= P; ++i) { /* :302, :316-317 */
            L->data[i] = lower + (double)((i - 1) % K);
            L->prob[i] = 1.0 / P;
        }
        for (int i = 0; i < K; ++i) L->hist[i] = 1.0e-10; /* :323 */
    }
}

mcio_config *mcio_config_create(int nleaf, const int *kind, const int *pool, const double *lower,
                                const double *upper, const int *npts, const double *alpha,
                                const int *adapt, const int *pool_offset, int npool, int Ni,
                                const int *dof, const int *obs_nbin, const int *obs_bin_draw) {
    mcio_config *c = (mcio_config *)calloc(1, sizeof(mcio_config));
    c->nleaf = nleaf;
    c->npool = npool;
    c->Ni = Ni;
    int Nd = Ni + 1;
    c->dof = (int *)calloc((size_t)Nd * npool, sizeof(int));
    memcpy(c->dof, dof, sizeof(int) * (size_t)Ni * npool); /* last row: zeros(Int, length(var)) ref: configuration.jl:153 */
    c->maxdof = (int *)calloc((size_t)npool, sizeof(int));
    mcio_maxdof(c->dof, Nd, npool, c->maxdof); /* :155 */
    c->pool_leaf0 = (int *)calloc((size_t)npool, sizeof(int));
    c->pool_nleaf = (int *)calloc((size_t)npool, sizeof(int));
    c->pool_offset = (int *)calloc((size_t)npool, sizeof(int));
    for (int v = 0; v < npool; ++v) c->pool_leaf0[v] = -1;
    for (int l = 0; l < nleaf; ++l) {
        int v = pool[l];
        if (c->pool_leaf0[v] < 0) c->pool_leaf0[v] = l;
        c->pool_nleaf[v] += 1;
    }
    c->leaf = (mcio_leaf *)calloc((size_t)nleaf, sizeof(mcio_leaf));
    c->pool_prob = (double **)calloc((size_t)npool, sizeof(double *));
    c->pool_prob_cache = (double *)calloc((size_t)npool, sizeof(double));
    for (int v = 0; v < npool; ++v) {
        int off = pool_offset ? pool_offset[v] : 0;
        c->pool_offset[v] = off;
        /* pool size: MaxOrder+1 (ref: variable.jl:139), grown to maxdof+2+offset (ref: configuration.jl:156-160) */
        int P = MAXORDER + 1;
        if (c->maxdof[v] + off >= P - 2) P = c->maxdof[v] + 2 + off;
        for (int l = c->pool_leaf0[v]; l < c->pool_leaf0[v] + c->pool_nleaf[v]; ++l)
            leaf_init(&c->leaf[l], kind[l], v, lower[l], upper[l], npts[l], alpha[l], adapt[l], P);
        if (c->pool_nleaf[v] == 1) {
            c->pool_prob[v] = c->leaf[c->pool_leaf0[v]].prob;
        } else { /* CompositeVar.prob = ones(size)  ref: variable.jl:425 */
            c->pool_prob[v] = (double *)calloc((size_t)P + 1, sizeof(double));
            for (int i = 0; i <= P; ++i) c->pool_prob[v][i] = 1.0;
        }
        c->pool_prob_cache[v] = 1.0;
    }
    c->pool_width = (int *)calloc((size_t)npool, sizeof(int));
    for (int v = 0; v < npool; ++v) {
        c->pool_width[v] = 0;
        for (int l = c->pool_leaf0[v]; l < c->pool_leaf0[v] + c->pool_nleaf[v]; ++l) c->pool_width[v] += c->leaf[l].width;
    }
    c->ndraw = 0;
    for (int v = 0; v < npool; ++v) c->ndraw += c->maxdof[v] * c->pool_width[v];
    c->draw_leaf = (int *)calloc((size_t)(c->ndraw > 0 ? c->ndraw : 1), sizeof(int));
    c->draw_slot = (int *)calloc((size_t)(c->ndraw > 0 ? c->ndraw : 1), sizeof(int));
    c->draw_comp = (int *)calloc((size_t)(c->ndraw > 0 ? c->ndraw : 1), sizeof(int));
    int k = 0;
    for (int v = 0; v < npool; ++v)
        for (int idx = 1; idx <= c->maxdof[v]; ++idx)
            for (int l = 0; l < c->pool_nleaf[v]; ++l)
                for (int j = 0; j < c->leaf[c->pool_leaf0[v] + l].width; ++j) {
                    c->draw_leaf[k] = c->pool_leaf0[v] + l;
                    c->draw_slot[k] = idx;
                    c->draw_comp[k] = j;
                    ++k;
                }
    c->obs_off = (int *)calloc((size_t)Ni, sizeof(int));
    c->obs_nbin = (int *)calloc((size_t)Ni, sizeof(int));
    c->obs_bin_draw = (int *)calloc((size_t)Ni, sizeof(int));
    c->nobs = 0;
    for (int i = 0; i < Ni; ++i) {
        c->obs_off[i] = c->nobs;
        c->obs_nbin[i] = obs_nbin ? obs_nbin[i] : 1;
        c->obs_bin_draw[i] = obs_bin_draw ? obs_bin_draw[i] : -1;
        c->nobs += c->obs_nbin[i];
    }
    c->observable = (double *)calloc((size_t)c->nobs, sizeof(double));
    c->reweight = (double *)calloc((size_t)Nd, sizeof(double));
    for (int i = 0; i < Nd; ++i) c->reweight[i] = 1.0 / Nd; /* ref: configuration.jl:110,172-173 */
    c->visited = (double *)calloc((size_t)Nd, sizeof(double));
    for (int i = 0; i < Nd; ++i) c->visited[i] = 1.0e-8;    /* :182 */
    c->rng_bits = 52;
    c->pam = Nd > npool ? Nd : npool;     /* :185 max(Nd, Nv) */
    c->npa = 3 * Nd * c->pam;
    c->propose = (double *)calloc((size_t)c->npa, sizeof(double));
    c->accept = (double *)calloc((size_t)c->npa, sizeof(double));
    for (int v = 0; v < c->npa; ++v) c->propose[v] = 1.0e-8; /* :186 */
    c->hold_hist = (unsigned long long *)calloc(64, sizeof(unsigned long long));
    /* default neighbor graph  ref: configuration.jl:201-210 (1-based there, 0-based here; index Nd-1 = normalisation) */
    c->nneighbor = (int *)calloc((size_t)Nd, sizeof(int));
    c->neighbor = (int **)calloc((size_t)Nd, sizeof(int *));
    for (int d = 0; d < Nd; ++d) {
        c->neighbor[d] = (int *)calloc(2, sizeof(int));
        c->neighbor[d][0] = d - 1; /* :205 [d-1, d+1] */
        c->neighbor[d][1] = d + 1;
        c->nneighbor[d] = 2;
    }
    if (Nd == 2) { c->neighbor[0][0] = 1; c->nneighbor[0] = 1; }           /* :206 */
    else { c->neighbor[0][0] = Nd - 1; c->neighbor[0][1] = 1; }             /* :206 [Nd, 2] */
    c->neighbor[Nd - 1][0] = 0; c->nneighbor[Nd - 1] = 1;                   /* :207 norm -> first */
    if (Nd >= 3) { c->neighbor[Nd - 2][0] = Nd - 3; c->nneighbor[Nd - 2] = 1; } /* :208 */
    c->thermal_ratio = 0.1;
    c->ncomp = 1;
    c->measure_fn = NULL;
    c->normalization = 1.0e-10;                              /* :179 */
    c->neval = 0;
    c->prob_mode = MCIO_PROB_CREATE;
    c->carry = (mcio_carry *)calloc(1, sizeof(mcio_carry));
    c->carry->mode = -1;
    c->carry_owner = 1;
    return c;
}

void mcio_set_chain_carry(mcio_config *c, int mode) {
    c->carry->mode = mode == 0 ? 0 : mode > 0 ? 1 : -1;
    if (mode == 0) c->carry->valid = 0;
}

static void leaf_free(mcio_leaf *L) {
    free(L->grid); free(L->hist); free(L->accumulation); free(L->distribution);
    free(L->data); free(L->gidx); free(L->prob);
}

void mcio_config_destroy(mcio_config *c) {
    if (!c) return;
    for (int v = 0; v < c->npool; ++v)
        if (c->pool_nleaf[v] != 1) free(c->pool_prob[v]);
    for (int l = 0; l < c->nleaf; ++l) leaf_free(&c->leaf[l]);
    free(c->leaf); free(c->pool_leaf0); free(c->pool_nleaf); free(c->pool_offset);
    free(c->pool_prob); free(c->pool_prob_cache); free(c->dof); free(c->maxdof);
    free(c->draw_leaf); free(c->draw_slot); free(c->draw_comp); free(c->pool_width); free(c->obs_off); free(c->obs_nbin); free(c->obs_bin_draw);
    free(c->observable); free(c->reweight); free(c->visited); free(c->propose); free(c->accept);
    for (int d = 0; d < c->Ni + 1; ++d) free(c->neighbor[d]);
    free(c->neighbor); free(c->nneighbor); free(c->reweight_goal); free(c->hold_hist);
    if (c->carry_owner && c->carry) {
        for (int b = 0; b < 2; ++b) { free(c->carry->x[b]); free(c->carry->curr[b]); free(c->carry->P[b]); }
        free(c->carry->rw_used);
        free(c->carry->src);
        free(c->carry);
    }
    free(c);
}

static void *dup_mem(const void *p, size_t n) {
    if (!p) return NULL;
    void *q = malloc(n ? n : 1);
    memcpy(q, p, n);
    return q;
}

/* deepcopy(config)  ref: main.jl:130-131 */
mcio_config *mcio_config_clone(const mcio_config *s) {
    mcio_config *c = (mcio_config *)calloc(1, sizeof(mcio_config));
    *c = *s;
    int Nd = s->Ni + 1;
    c->leaf = (mcio_leaf *)calloc((size_t)s->nleaf, sizeof(mcio_leaf));
    for (int l = 0; l < s->nleaf; ++l) {
        const mcio_leaf *a = &s->leaf[l];
        mcio_leaf *b = &c->leaf[l];
        *b = *a;
        b->grid = (double *)dup_mem(a->grid, sizeof(double) * (size_t)a->npts);
        b->hist = (double *)dup_mem(a->hist, sizeof(double) * (size_t)a->nbin);
        b->accumulation = a->accumulation ? (double *)dup_mem(a->accumulation, sizeof(double) * (size_t)(a->nbin + 1)) : NULL;
        b->distribution = a->distribution ? (double *)dup_mem(a->distribution, sizeof(double) * (size_t)a->nbin) : NULL;
        b->data = (double *)dup_mem(a->data, sizeof(double) * (size_t)(a->P + 1) * (size_t)(a->width > 0 ? a->width : 1));
        b->gidx = (long *)dup_mem(a->gidx, sizeof(long) * (size_t)(a->P + 1));
        b->prob = (double *)dup_mem(a->prob, sizeof(double) * (size_t)(a->P + 1));
    }
    c->pool_leaf0 = (int *)dup_mem(s->pool_leaf0, sizeof(int) * (size_t)s->npool);
    c->pool_nleaf = (int *)dup_mem(s->pool_nleaf, sizeof(int) * (size_t)s->npool);
    c->pool_offset = (int *)dup_mem(s->pool_offset, sizeof(int) * (size_t)s->npool);
    c->pool_prob_cache = (double *)dup_mem(s->pool_prob_cache, sizeof(double) * (size_t)s->npool);
    c->pool_prob = (double **)calloc((size_t)s->npool, sizeof(double *));
    for (int v = 0; v < s->npool; ++v) {
        if (s->pool_nleaf[v] == 1) c->pool_prob[v] = c->leaf[c->pool_leaf0[v]].prob;
        else c->pool_prob[v] = (double *)dup_mem(s->pool_prob[v], sizeof(double) * (size_t)(s->leaf[s->pool_leaf0[v]].P + 1));
    }
    c->dof = (int *)dup_mem(s->dof, sizeof(int) * (size_t)Nd * s->npool);
    c->maxdof = (int *)dup_mem(s->maxdof, sizeof(int) * (size_t)s->npool);
    c->draw_leaf = (int *)dup_mem(s->draw_leaf, sizeof(int) * (size_t)(s->ndraw > 0 ? s->ndraw : 1));
    c->draw_slot = (int *)dup_mem(s->draw_slot, sizeof(int) * (size_t)(s->ndraw > 0 ? s->ndraw : 1));
    c->draw_comp = (int *)dup_mem(s->draw_comp, sizeof(int) * (size_t)(s->ndraw > 0 ? s->ndraw : 1));
    c->pool_width = (int *)dup_mem(s->pool_width, sizeof(int) * (size_t)s->npool);
    c->obs_off = (int *)dup_mem(s->obs_off, sizeof(int) * (size_t)s->Ni);
    c->obs_nbin = (int *)dup_mem(s->obs_nbin, sizeof(int) * (size_t)s->Ni);
    c->obs_bin_draw = (int *)dup_mem(s->obs_bin_draw, sizeof(int) * (size_t)s->Ni);
    c->observable = (double *)dup_mem(s->observable, sizeof(double) * (size_t)s->nobs);
    c->reweight = (double *)dup_mem(s->reweight, sizeof(double) * (size_t)Nd);
    c->visited = (double *)dup_mem(s->visited, sizeof(double) * (size_t)Nd);
    c->propose = (double *)dup_mem(s->propose, sizeof(double) * (size_t)s->npa);
    c->accept = (double *)dup_mem(s->accept, sizeof(double) * (size_t)s->npa);
    c->nneighbor = (int *)dup_mem(s->nneighbor, sizeof(int) * (size_t)Nd);
    c->neighbor = (int **)calloc((size_t)Nd, sizeof(int *));
    for (int d = 0; d < Nd; ++d) c->neighbor[d] = (int *)dup_mem(s->neighbor[d], sizeof(int) * (size_t)s->nneighbor[d]);
    c->reweight_goal = s->reweight_goal ? (double *)dup_mem(s->reweight_goal, sizeof(double) * (size_t)Nd) : NULL;
    c->hold_hist = (unsigned long long *)dup_mem(s->hold_hist, 64 * sizeof(unsigned long long));
    c->carry = (mcio_carry *)calloc(1, sizeof(mcio_carry)); /* a copy keeps no chains (run_blocks shares the parent's state with its clones) */
    c->carry->mode = s->carry ? s->carry->mode : -1;
    c->carry_owner = 1;
    c->carry_load = c->carry_store = 0;
    return c;
}

int mcio_set_grid(mcio_config *c, int leaf, const double *grid, int npts) {
    mcio_leaf *L = &c->leaf[leaf];
    if (L->kind != MCIO_CONTINUOUS) return 1;
    free(L->grid);
    free(L->hist);
    L->npts = npts;
    L->nbin = npts - 1;
    L->grid = (double *)dup_mem(grid, sizeof(double) * (size_t)npts);
    L->hist = (double *)malloc(sizeof(double) * (size_t)L->nbin);
    for (int i = 0; i < L->nbin; ++i) L->hist[i] = MCIO_TINY;
    for (int i = 1; i <= L->P; ++i) {
        long g = mcio_locate(L->grid, npts, L->data[i]);
        L->gidx[i] = g < 1 ? 1 : g;
    }
    return 0;
}

/* Discrete(...; distribution=...)  ref: variable.jl:306-315 */
int mcio_set_distribution(mcio_config *c, int leaf, const double *dist) {
    mcio_leaf *L = &c->leaf[leaf];
    if (L->kind != MCIO_DISCRETE) return 1;
    int K = L->nbin;
    double s = 0.0;
    for (int i = 0; i < K; ++i) {
        if (!(dist[i] >= 0.0)) return 2; /* :309 */
        s += dist[i];
    }
    double run = 0.0;
    L->accumulation[0] = 0.0;
    for (int i = 0; i < K; ++i) {
        L->distribution[i] = dist[i] / s; /* :312 */
        run += L->distribution[i];
        L->accumulation[i + 1] = run;     /* :313-314 */
    }
    return 0;
}

/* ref: configuration.jl:238-250 and variable.jl:565 */
void mcio_clear_statistics(mcio_config *c) {
    for (int i = 0; i < c->nobs; ++i) c->observable[i] = 0.0;
    c->neval = 0;
    c->normalization = 1.0e-10;
    for (int i = 0; i < c->Ni + 1; ++i) c->visited[i] = 1.0e-8;
    for (int v = 0; v < c->npa; ++v) {
        c->propose[v] = 1.0e-8;
        c->accept[v] = 1.0e-10;
    }
    for (int l = 0; l < c->nleaf; ++l)
        for (int i = 0; i < c->leaf[l].nbin; ++i) c->leaf[l].hist[i] = 1.0e-10;
    for (int b = 0; b < 64; ++b) c->hold_hist[b] = 0;
}

/* ref: configuration.jl:252-262 and variable.jl:567 */
void mcio_add_config(mcio_config *c, const mcio_config *ic) {
    for (int i = 0; i < c->Ni + 1; ++i) c->visited[i] += ic->visited[i];
    for (int v = 0; v < c->npa; ++v) {
        c->accept[v] += ic->accept[v];
        c->propose[v] += ic->propose[v];
    }
    c->neval += ic->neval;
    c->normalization += ic->normalization;
    for (int i = 0; i < c->nobs; ++i) c->observable[i] += ic->observable[i];
    for (int l = 0; l < c->nleaf; ++l)
        for (int i = 0; i < c->leaf[l].nbin; ++i) c->leaf[l].hist[i] += ic->leaf[l].hist[i];
    for (int b = 0; b < 64; ++b) c->hold_hist[b] += ic->hold_hist[b];
}

/* Dist.train!(v) for every variable  ref: main.jl:194-195, variable.jl:479-483 */
void mcio_train(mcio_config *c) {
    if (c->carry) c->carry->ntrain += 1;
    for (int l = 0; l < c->nleaf; ++l) {
        mcio_leaf *L = &c->leaf[l];
        if (!L->adapt) continue; /* variable.jl:208, :370 */
        int rc;
        if (L->kind == MCIO_CONTINUOUS) rc = mcio_train_continuous(L->grid, L->npts, L->hist, L->alpha);
        else rc = mcio_train_discrete(L->hist, L->nbin, L->alpha, L->distribution, L->accumulation);
        if (rc) fprintf(stderr, "[mci_oracle] train! assertion %d on leaf %d\n", rc, l);
    }
}

/* ------------------------------------------------------------------------------------------
 * src/distribution/sampler.jl
 * ---------------------------------------------------------------------------------------- */

static inline long locate_clamped(const mcio_leaf *L, double u) {
    long g = mcio_locate(L->accumulation, L->nbin + 1, u);
    /* the reference raises error() when accumulation[end] <= u < 1 through rounding
       (common.jl:10-12); the oracle clamps to the last bin instead (probability ~1e-16). */
    if (g < 1) g = L->nbin;
    return g;
}

/* create!  ref: sampler.jl:293-305 (Continuous), :13-22 (Discrete).  idx is 1-based incl. offset. */
double mcio_create(mcio_config *c, int leaf, int idx, double u) {
    mcio_leaf *T = &c->leaf[leaf];
    if (T->kind == MCIO_CONTINUOUS) {
        long N = T->npts - 1;                 /* :295 */
        double y = u;                          /* :296 */
        long iy = (long)floor(y * N) + 1;     /* :297 */
        double dy = y * N - (iy - 1);          /* :298 */
        double x = T->grid[iy - 1] + dy * (T->grid[iy] - T->grid[iy - 1]); /* :299 */
        T->data[idx] = x;                      /* :300 */
        T->gidx[idx] = iy;                     /* :301 */
        T->prob[idx] = 1.0 / (N * (T->grid[iy] - T->grid[iy - 1])); /* :303 */
        return 1.0 / T->prob[idx];             /* :304 */
    } else {
        long gidx = locate_clamped(T, u);                  /* :17 */
        T->data[idx] = T->lower + (double)(gidx - 1);      /* :18 */
        T->gidx[idx] = gidx;
        T->prob[idx] = T->distribution[gidx - 1];          /* :20 */
        return 1.0 / T->distribution[gidx - 1];            /* :21 */
    }
}

/* shift!  ref: sampler.jl:336-386 (Continuous), :57-71 (Discrete) */
double mcio_shift(mcio_config *c, int leaf, int idx, double u) {
    mcio_leaf *T = &c->leaf[leaf];
    int end = T->P;
    if (T->kind == MCIO_CONTINUOUS) {
        T->data[end] = T->data[idx]; /* :338 */
        T->gidx[end] = T->gidx[idx]; /* :339 */
        T->prob[end] = T->prob[idx]; /* :340 */
        long cur = T->gidx[idx];     /* :341 */
        long N = T->npts - 1;        /* :343 */
        double y = u;                 /* :361 */
        long iy = (long)floor(y * N) + 1; /* :378 */
        double dy = y * N - (iy - 1);      /* :379 */
        double x = T->grid[iy - 1] + dy * (T->grid[iy] - T->grid[iy - 1]); /* :380 */
        T->data[idx] = x;             /* :381 */
        T->gidx[idx] = iy;            /* :382 */
        double ratio = (T->grid[cur] - T->grid[cur - 1]) / (T->grid[iy] - T->grid[iy - 1]); /* :383 */
        T->prob[idx] *= ratio;        /* :384 */
        return 1.0 / ratio;           /* :385 */
    } else {
        T->data[end] = T->data[idx];  /* :62 */
        T->prob[end] = T->prob[idx];  /* :63 */
        T->gidx[end] = T->gidx[idx];
        long cur = (long)(T->data[idx] - T->lower) + 1; /* :64 */
        long gidx = locate_clamped(T, u);                /* :65 */
        T->data[idx] = T->lower + (double)(gidx - 1);    /* :66 */
        T->gidx[idx] = gidx;
        double ratio = T->distribution[gidx - 1] / T->distribution[cur - 1]; /* :68 */
        T->prob[idx] *= ratio;                           /* :69 */
        return 1.0 / ratio;                              /* :70 */
    }
}

/* shiftRollback!  ref: sampler.jl:388-393, :73-77 */
void mcio_shift_rollback(mcio_config *c, int leaf, int idx) {
    mcio_leaf *T = &c->leaf[leaf];
    int end = T->P;
    T->data[idx] = T->data[end];
    T->gidx[idx] = T->gidx[end];
    T->prob[idx] = T->prob[end];
}

/* ---- FermiK{D}  ref: sampler.jl:109-281.  K.prob is written but never read on the :mcmc path; it is kept for the record. */
static double fermik_create(mcio_leaf *K, int idx, const double *u) { /* :109-148, u = D uniforms */
    const int D = K->width;
    const double kF = K->lower, dk = K->upper;
    double *k = &K->data[idx * D];
    const double Kamp = kF + (u[0] - 0.5) * 2.0 * dk;  /* :121 */
    if (Kamp <= 0.0) return 0.0;                          /* :122 */
    const double phi = 2.0 * M_PI * u[1];                 /* :124 */
    double prop;
    if (D == 3) {
        const double theta = M_PI * u[2];                 /* :126 */
        k[0] = Kamp * cos(phi) * sin(theta);              /* :129-131 */
        k[1] = Kamp * sin(phi) * sin(theta);
        k[2] = Kamp * cos(theta);
        prop = 2 * dk * 2 * M_PI * M_PI * (sin(theta) * Kamp * Kamp); /* :132 */
    } else {
        k[0] = Kamp * cos(phi);                           /* :139-140 */
        k[1] = Kamp * sin(phi);
        prop = 2 * dk * 2 * M_PI * Kamp;                  /* :141 */
    }
    K->prob[idx] = 1.0 / prop;
    return prop;
}

static double fermik_remove(mcio_leaf *K, int idx) { /* :158-188 */
    const int D = K->width;
    const double kF = K->lower, dk = K->upper;
    const double *k = &K->data[idx * D];
    double k2 = 0.0;
    for (int j = 0; j < D; ++j) k2 += k[j] * k[j];
    const double Kamp = sqrt(k2);                                   /* :171 */
    if (!(kF - dk < Kamp && Kamp < kF + dk)) return 0.0;            /* :172-174 */
    double prop;
    if (D == 3) {
        const double sint = sqrt(k[0] * k[0] + k[1] * k[1]) / Kamp; /* :177 */
        if (sint < 1.0e-15) return 0.0;                             /* :178 */
        prop = 1.0 / (2 * dk * 2 * M_PI * M_PI * sint * Kamp * Kamp); /* :179 */
    } else {
        prop = 1.0 / (2 * dk * 2 * M_PI * Kamp);                    /* :183 */
    }
    K->prob[idx] = 1.0 / prop;
    return prop;
}

/* shift!  :198-246 ; upick selects the move, u = up to D more uniforms */
static double fermik_shift(mcio_leaf *K, int idx, double upick, const double *u) {
    const int D = K->width, end = K->P;
    double *k = &K->data[idx * D];
    for (int j = 0; j < D; ++j) K->data[end * D + j] = k[j]; /* :201 save current K */
    K->prob[end] = K->prob[idx];
    if (upick < 1.0 / 3) {                                   /* :206-212 scale */
        const double lambda = 1.5;
        const double ratio = 1.0 / lambda + u[0] * (lambda - 1.0 / lambda);
        for (int j = 0; j < D; ++j) k[j] *= ratio;
        return D == 2 ? 1.0 : ratio;
    } else if (upick < 2.0 / 3) {                            /* :213-229 rotate */
        const double phi = u[0] * 2.0 * M_PI;
        if (D == 3) {
            const double theta = acos(1.0 - 2.0 * u[1]);
            const double Kamp = sqrt(k[0] * k[0] + k[1] * k[1] + k[2] * k[2]);
            k[0] = Kamp * cos(phi) * sin(theta);
            k[1] = Kamp * sin(phi) * sin(theta);
            k[2] = Kamp * cos(theta);
        } else {
            const double Kamp = sqrt(k[0] * k[0] + k[1] * k[1]);
            k[0] = Kamp * cos(phi);
            k[1] = Kamp * sin(phi);
        }
        return 1.0;
    }
    for (int j = 0; j < D; ++j) k[j] += (u[j] - 0.5) * K->upper; /* :231-243 shift by (rand - 0.5) dk per component */
    return 1.0;
}

static void fermik_shift_rollback(mcio_leaf *K, int idx) { /* :248-252 */
    const int D = K->width, end = K->P;
    for (int j = 0; j < D; ++j) K->data[idx * D + j] = K->data[end * D + j];
    K->prob[idx] = K->prob[end];
}

static void fermik_swap(mcio_leaf *K, int i1, int i2) { /* :254-281 */
    const int D = K->width;
    double t = K->prob[i1]; K->prob[i1] = K->prob[i2]; K->prob[i2] = t;
    for (int j = 0; j < D; ++j) {
        t = K->data[i1 * D + j]; K->data[i1 * D + j] = K->data[i2 * D + j]; K->data[i2 * D + j] = t;
    }
}

/* FermiK-aware shift of a pool slot: u = the pool's width uniforms, upick = the extra uniform of FermiK's move selection */
static double pool_shift_any(mcio_config *c, int vi, int idx, double upick, const double *u) {
    mcio_leaf *L0 = &c->leaf[c->pool_leaf0[vi]];
    if (L0->kind == MCIO_FERMIK) return fermik_shift(L0, idx, upick, u);
    return mcio_pool_shift(c, vi, idx, u);
}

/* pool-level shift!: plain variable, or CompositeVar  ref: sampler.jl:431-440 */
double mcio_pool_shift(mcio_config *c, int vi, int idx, const double *u) {
    int l0 = c->pool_leaf0[vi], nl = c->pool_nleaf[vi];
    if (nl == 1) return mcio_shift(c, l0, idx, u[0]);
    double prop = 1.0;                              /* :432 */
    c->pool_prob_cache[vi] = c->pool_prob[vi][idx]; /* :433 */
    c->pool_prob[vi][idx] = 1.0;                    /* :434 */
    for (int l = 0; l < nl; ++l) {                  /* :435-438 */
        prop *= mcio_shift(c, l0 + l, idx, u[l]);
        c->pool_prob[vi][idx] *= c->leaf[l0 + l].prob[idx];
    }
    return prop;
}

/* pool-level create!  ref: sampler.jl:410-418 */
double mcio_pool_create(mcio_config *c, int vi, int idx, const double *u) {
    int l0 = c->pool_leaf0[vi], nl = c->pool_nleaf[vi];
    if (c->leaf[l0].kind == MCIO_FERMIK) return fermik_create(&c->leaf[l0], idx, u);
    if (nl == 1) return mcio_create(c, l0, idx, u[0]);
    double prop = 1.0;             /* :411 */
    c->pool_prob[vi][idx] = 1.0;   /* :412 */
    for (int l = 0; l < nl; ++l) { /* :413-416 */
        prop *= mcio_create(c, l0 + l, idx, u[l]);
        c->pool_prob[vi][idx] *= c->leaf[l0 + l].prob[idx];
    }
    return prop;
}

/* ref: sampler.jl:441-446 */
void mcio_pool_shift_rollback(mcio_config *c, int vi, int idx) {
    int l0 = c->pool_leaf0[vi], nl = c->pool_nleaf[vi];
    if (c->leaf[l0].kind == MCIO_FERMIK) {
        fermik_shift_rollback(&c->leaf[l0], idx);
        return;
    }
    for (int l = 0; l < nl; ++l) mcio_shift_rollback(c, l0 + l, idx);
    if (nl != 1) c->pool_prob[vi][idx] = c->pool_prob_cache[vi];
}

/* remove!  ref: sampler.jl:318-323 (Continuous), :36-40 (Discrete): the probability of the slot that goes away */
double mcio_remove(mcio_config *c, int leaf, int idx) {
    mcio_leaf *T = &c->leaf[leaf];
    if (T->kind == MCIO_CONTINUOUS) {
        long iy = T->gidx[idx];                                                 /* :321 */
        return 1.0 / ((T->grid[iy] - T->grid[iy - 1]) * (double)(T->npts - 1)); /* :322 */
    }
    long gidx = (long)(T->data[idx] - T->lower) + 1; /* :38 */
    return T->distribution[gidx - 1];                /* :39 */
}

/* ref: sampler.jl:422-428 */
double mcio_pool_remove(mcio_config *c, int vi, int idx) {
    if (c->leaf[c->pool_leaf0[vi]].kind == MCIO_FERMIK) return fermik_remove(&c->leaf[c->pool_leaf0[vi]], idx);
    double prop = 1.0;
    for (int l = c->pool_leaf0[vi]; l < c->pool_leaf0[vi] + c->pool_nleaf[vi]; ++l) prop *= mcio_remove(c, l, idx);
    return prop;
}

/* swap! == swapRollback!  ref: sampler.jl:395-408 (Continuous), :86-97 (Discrete), :448-462 (CompositeVar) */
double mcio_pool_swap(mcio_config *c, int vi, int idx1, int idx2) {
    int l0 = c->pool_leaf0[vi], nl = c->pool_nleaf[vi];
    if (c->leaf[l0].kind == MCIO_FERMIK) {
        fermik_swap(&c->leaf[l0], idx1, idx2);
        return 1.0;
    }
    if (nl != 1) { /* :450 */
        double t = c->pool_prob[vi][idx1];
        c->pool_prob[vi][idx1] = c->pool_prob[vi][idx2];
        c->pool_prob[vi][idx2] = t;
    }
    for (int l = l0; l < l0 + nl; ++l) {
        mcio_leaf *T = &c->leaf[l];
        double d = T->data[idx1]; T->data[idx1] = T->data[idx2]; T->data[idx2] = d;
        long g = T->gidx[idx1]; T->gidx[idx1] = T->gidx[idx2]; T->gidx[idx2] = g;
        double p = T->prob[idx1]; T->prob[idx1] = T->prob[idx2]; T->prob[idx2] = p;
    }
    return 1.0;
}

/* ref: variable.jl:587-599 */
double mcio_total_probability(const mcio_config *c) {
    double prob = 1.0;
    for (int vi = 0; vi < c->npool; ++vi)
        for (int pos = 1; pos <= c->maxdof[vi]; ++pos) prob *= c->pool_prob[vi][pos + c->pool_offset[vi]];
    return prob;
}

/* ref: variable.jl:606-619 ; i is 0-based here, i == Ni is the normalisation integrand */
double mcio_probability(const mcio_config *c, int i) {
    double prob = 1.0;
    for (int vi = 0; vi < c->npool; ++vi)
        for (int pos = 1; pos <= c->dof[i * c->npool + vi]; ++pos) prob *= c->pool_prob[vi][pos + c->pool_offset[vi]];
    return prob;
}

/* ref: variable.jl:628-641 */
double mcio_padding_probability(const mcio_config *c, int i) {
    double prob = 1.0;
    for (int vi = 0; vi < c->npool; ++vi)
        for (int pos = c->dof[i * c->npool + vi] + 1; pos <= c->maxdof[vi]; ++pos)
            prob *= c->pool_prob[vi][pos + c->pool_offset[vi]];
    return prob;
}

/* accumulate!  ref: variable.jl:196-200 (Continuous), :362-367 (Discrete), :474-478 (CompositeVar) */
static inline void pool_accumulate(mcio_config *c, int vi, int idx, double weight) {
    int l0 = c->pool_leaf0[vi], nl = c->pool_nleaf[vi];
    for (int l = l0; l < l0 + nl; ++l) {
        mcio_leaf *T = &c->leaf[l];
        if (!T->adapt) continue;
        long g = (T->kind == MCIO_CONTINUOUS) ? T->gidx[idx] : (long)(T->data[idx] - T->lower) + 1;
        T->hist[g - 1] += weight;
    }
}

/* gather the flat draw vector the integrand sees (x[i] == var.data[i], ref: distribution.jl:27) */
static inline void gather_x(const mcio_config *c, double *x) {
    for (int k = 0; k < c->ndraw; ++k) {
        const mcio_leaf *T = &c->leaf[c->draw_leaf[k]];
        x[k] = T->data[(c->draw_slot[k] + c->pool_offset[T->pool]) * T->width + c->draw_comp[k]];
    }
}

/* abs(weights[i]) (vegas/montecarlo.jl:173 etc.): |w| or the complex modulus (Julia abs(::Complex) = hypot) */
static inline double absw(const mcio_config *c, const double *w, int i) {
    return c->ncomp == 1 ? fabs(w[i]) : hypot(w[2 * i], w[2 * i + 1]);
}

/* default measure (ref: vegas/montecarlo.jl:151-153), "bin by a Discrete draw" (ref: example/bubble.jl:81-84),
   or the user's measure (vegas/montecarlo.jl:156-161).  relw has Ni*ncomp entries. */
static inline void measure(mcio_config *c, const double *x, const double *relw, const double *ud) {
    if (c->measure_fn) {
        c->measure_fn(x, relw, ud, -1, c->observable);
        return;
    }
    for (int i = 0; i < c->Ni; ++i) {
        if (c->obs_bin_draw[i] >= 0) {
            const mcio_leaf *T = &c->leaf[c->draw_leaf[c->obs_bin_draw[i]]];
            int bin = (int)(x[c->obs_bin_draw[i]] - T->lower);
            if (bin < 0 || bin >= c->obs_nbin[i]) continue;
            c->observable[c->obs_off[i] + bin] += relw[i];
        } else {
            for (int q = 0; q < c->ncomp; ++q) c->observable[c->obs_off[i] + q] += relw[i * c->ncomp + q];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * src/vegas/montecarlo.jl:72-191
 * ---------------------------------------------------------------------------------------- */
/* propose / accept [update][integrand][target], configuration.jl:185-186 (0-based, row-major) */
#define PA_IDX(c, ut, curr, target) ((((ut) * ((c)->Ni + 1)) + (curr)) * (c)->pam + (target))
#define MCIO_MAXDRAW 256
#define MCIO_MAXNI 64

int mcio_vegas_block(mcio_config *c, mcio_integrand_fn f, const double *ud, uint64_t seed,
                     uint32_t iteration, long block_index, long neval, long measurefreq) {
    const int Ni = c->Ni, npool = c->npool;
    if (c->ndraw > MCIO_MAXDRAW || Ni > MCIO_MAXNI || measurefreq <= 0) return -1; /* :77 */
    for (int l = 0; l < c->nleaf; ++l)
        if (c->leaf[l].kind == MCIO_FERMIK) return -4; /* "vegas doesn't work with FermiK variable yet" test/bubble_FermiK.jl:2 */
    const int nc = c->ncomp;
    double relw[2 * MCIO_MAXNI], weights[2 * MCIO_MAXNI], pad[MCIO_MAXNI]; /* :79-81 */
    int diff[MCIO_MAXNI];
    double x[MCIO_MAXDRAW], u[MCIO_MAXDRAW];
    for (int i = 0; i < Ni; ++i) {
        weights[nc * i] = weights[nc * i + nc - 1] = 0.0;
        pad[i] = 1.0;
        diff[i] = 1; /* :82 dof[i] == maxdof */
        for (int v = 0; v < npool; ++v)
            if (c->dof[i * npool + v] != c->maxdof[v]) diff[i] = 0;
    }
    /* :108-110 Dist.initialize! -> create! on every live slot (variable.jl:576-580).  Its
       uniforms come from their own stream; in PROB_CREATE mode the values are overwritten
       before use and only the RNG accounting of the reference is lost. */
    {
        uint32_t st = stream_id(iteration, STREAM_POOLINIT);
        uint32_t kk = 0;
        for (int v = 0; v < npool; ++v) {
            int P = c->leaf[c->pool_leaf0[v]].P;
            for (int idx = 1 + c->pool_offset[v]; idx <= P - 2; ++idx) {
                for (int l = 0; l < c->pool_nleaf[v]; ++l) u[l] = mcio_uniform(seed, st, (uint64_t)block_index, kk++);
                mcio_pool_create(c, v, idx, u);
            }
        }
    }
    const uint32_t st = stream_id(iteration, STREAM_VEGAS);
    for (long ne = 1; ne <= neval; ++ne) { /* :117 */
        c->neval += 1;                      /* :118 */
        const uint64_t gs = (uint64_t)block_index * (uint64_t)neval + (uint64_t)(ne - 1);
        double jac = 1.0;                   /* :121 */
        int k = 0;
        for (int vi = 0; vi < npool; ++vi) { /* :122 */
            const int off = c->pool_offset[vi], nl = c->pool_nleaf[vi];
            for (int idx = 1; idx <= c->maxdof[vi]; ++idx) { /* :124 */
                for (int l = 0; l < nl; ++l)
                    u[l] = c->rng_bits == 32 ? mcio_uniform32(seed, st, gs, (uint32_t)(k + l)) : mcio_uniform(seed, st, gs, (uint32_t)(k + l));
                if (c->prob_mode == MCIO_PROB_SHIFT) mcio_pool_shift(c, vi, idx + off, u); /* :125 */
                else mcio_pool_create(c, vi, idx + off, u);                                /* :128-129 */
                jac /= c->pool_prob[vi][idx + off];                                        /* :126 */
                k += nl;
            }
        }
        for (int i = 0; i < Ni; ++i) /* :133-137 */
            if (!diff[i]) pad[i] = mcio_padding_probability(c, i);
        gather_x(c, x);
        f(x, weights, ud); /* :140-144 */
        if (ne % measurefreq == 0) { /* :148 */
            for (int i = 0; i < Ni; ++i)
                for (int q = 0; q < nc; ++q) relw[nc * i + q] = weights[nc * i + q] * pad[i] * jac; /* :152 / :157 */
            measure(c, x, relw, ud);
            c->normalization += 1.0; /* :164 */
        }
        for (int vi = 0; vi < npool; ++vi) { /* :170 */
            const int off = c->pool_offset[vi];
            for (int i = 0; i < Ni; ++i) {   /* :172 */
                double w2 = absw(c, weights, i); /* :173 */
                double j2 = jac;              /* :174 */
                for (int pos = 1; pos <= c->dof[i * npool + vi]; ++pos) /* :179 */
                    pool_accumulate(c, vi, pos + off, (w2 * j2) * (w2 * j2)); /* :180 */
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * src/vegas_mc/montecarlo.jl:112-241 + src/vegas_mc/updates.jl:45-106
 * The block's neval steps are run as `nchain` independent chains of neval/nchain steps
 * (nchain = 1 is the reference).  Chain ch of the block (its block index rides in the stream word, stream_id_block) draws
 *   init  : stream MC_INIT, index g,            k = flat draw
 *   step s: stream MC_STEP, index (g<<32 | s),  k = 0 pool pick, 1 slot pick, 2 accept, 3+l leaf l
 * ---------------------------------------------------------------------------------------- */
/* A carried chain's slot (pool vi, 1-based idx incl. offset) from its stored x entries: the values, and on the CURRENT map the
 * bin that holds each of them with prob = 1/(N dx) (sampler.jl:303) | distribution[bin] (:20) -- mirror of mci_device.h relocate_draw */
static void carried_slot(mcio_config *c, int vi, int idx, const double *xs) {
    const int l0 = c->pool_leaf0[vi], nl = c->pool_nleaf[vi];
    mcio_leaf *L0 = &c->leaf[l0];
    if (L0->kind == MCIO_FERMIK) {
        for (int j = 0; j < L0->width; ++j) L0->data[idx * L0->width + j] = xs[j];
        return;
    }
    double pp = 1.0;
    for (int l = 0; l < nl; ++l) {
        mcio_leaf *T = &c->leaf[l0 + l];
        const double x = xs[l];
        T->data[idx] = x;
        if (T->kind == MCIO_CONTINUOUS) {
            const long N = T->npts - 1;
            long lo = 0, hi = N - 1; /* the largest increment whose lower edge is <= x */
            while (lo < hi) {
                const long mid = (lo + hi + 1) >> 1;
                if (T->grid[mid] <= x) lo = mid;
                else hi = mid - 1;
            }
            T->gidx[idx] = lo + 1;
            T->prob[idx] = 1.0 / ((T->grid[lo + 1] - T->grid[lo]) * (double)N);
        } else {
            long g = (long)(x - T->lower);
            if (g < 0) g = 0;
            if (g >= T->nbin) g = T->nbin - 1;
            T->gidx[idx] = g + 1;
            T->prob[idx] = T->distribution[g];
        }
        pp *= T->prob[idx];
    }
    if (nl != 1) c->pool_prob[vi][idx] = pp;
}
/* the stored chain that new chain `ch` of this block continues: the stored chains resampled to the moved target (run_blocks:
   :mcmc mcio_resample_chains, :vegasmc mcio_resample_weighted) */
static long carried_slot_of(const mcio_config *c, long ch, long nchain, int mcmc) {
    const mcio_carry *cy = c->carry;
    (void)mcmc;
    return c->carry_lb * cy->load_nchain + cy->src[c->carry_lb * nchain + ch];
}
static void load_carried(mcio_config *c, long slot) {
    const mcio_carry *cy = c->carry;
    double xs[64];
    for (int vi = 0, k = 0; vi < c->npool; ++vi)
        for (int idx = 1; idx <= c->maxdof[vi]; ++idx) {
            const int w = c->pool_width[vi];
            for (int l = 0; l < w; ++l) xs[l] = cy->x[cy->rd][(long)(k + l) * cy->cap[cy->rd] + slot];
            carried_slot(c, vi, idx + c->pool_offset[vi], xs);
            k += w;
        }
}
static void store_carried(mcio_config *c, long ch, long nchain, int curr) {
    mcio_carry *cy = c->carry;
    const long slot = c->carry_lb * nchain + ch;
    double x[MCIO_MAXDRAW];
    gather_x(c, x);
    for (int k = 0; k < c->ndraw; ++k) cy->x[cy->wr][(long)k * cy->cap[cy->wr] + slot] = x[k];
    cy->curr[cy->wr][slot] = curr;
}
/* config.probability (vegas_mc/montecarlo.jl:155-166) of the configuration the pools hold: the target density of a :vegasmc chain */
static double vegasmc_target(mcio_config *c, mcio_integrand_fn f, const double *ud, double *weights_out /* [ncomp * Ni] or NULL */, double *pad /* [Ni+1] */) {
    const int N = c->Ni, norm = c->Ni, nc = c->ncomp;
    double x[MCIO_MAXDRAW], _weights[2 * MCIO_MAXNI];
    gather_x(c, x);
    f(x, _weights, ud); /* :155-159 */
    for (int i = 0; i <= N; ++i) pad[i] = mcio_padding_probability(c, i); /* :161 */
    double probability = c->reweight[norm] * pad[norm];                   /* :162 */
    for (int i = 0; i < N; ++i) {                                         /* :163-166 */
        if (weights_out)
            for (int q = 0; q < nc; ++q) weights_out[nc * i + q] = _weights[nc * i + q];
        probability += absw(c, _weights, i) * c->reweight[i] * pad[i];
    }
    return probability;
}

int mcio_vegasmc_block(mcio_config *c, mcio_integrand_fn f, const double *ud, uint64_t seed,
                       uint32_t iteration, long block_index, long neval, long measurefreq,
                       long nchain) {
    const int N = c->Ni, npool = c->npool, norm = c->Ni;
    if (c->ndraw > MCIO_MAXDRAW || N + 1 > MCIO_MAXNI || measurefreq <= 0 || nchain < 1) return -1;
    for (int l = 0; l < c->nleaf; ++l)
        if (c->leaf[l].kind == MCIO_FERMIK) return -4; /* test/bubble_FermiK.jl:126 "vegasmc can not handle this" */
    const int nc = c->ncomp;
    double weights[2 * MCIO_MAXNI], _weights[2 * MCIO_MAXNI], relw[2 * MCIO_MAXNI];
    double pad[MCIO_MAXNI], _pad[MCIO_MAXNI]; /* :147-148 */
    double x[MCIO_MAXDRAW], u[MCIO_MAXDRAW];
    const long steps = neval / nchain;
    /* first measured step: `ne >= neval/100` (:213) for the reference's single chain; with nchain > 1 each short
       chain additionally skips min(steps/2, 64*nslots) steps (the many-chain decomposition is this engine's own) */
    int nslots = 0;
    for (int vi = 0; vi < npool; ++vi) nslots += c->maxdof[vi];
    double burnin = (double)steps / 100.0;
    if (nchain > 1 && !c->carry_load) { /* (carried chains keep the reference's own term only) */
        double fl = 64.0 * (double)nslots;
        if (fl > (double)steps / 2.0) fl = (double)steps / 2.0;
        if (fl > burnin) burnin = fl;
    }
    const uint32_t st_init = stream_id_block(iteration, STREAM_MC_INIT, block_index), st_step = stream_id_block(iteration, STREAM_MC_STEP, block_index);
    for (long ch = 0; ch < nchain; ++ch) {
        const uint64_t g = (uint64_t)ch;
        /* :151-153 initialize! (only the slots that are ever read: 1..maxdof) */
        int k = 0;
        if (c->carry_load) load_carried(c, carried_slot_of(c, ch, nchain, 0)); /* continues the previous iteration's chain (mci_set_chain_carry) */
        else
        for (int vi = 0; vi < npool; ++vi)
            for (int idx = 1; idx <= c->maxdof[vi]; ++idx) {
                int nl = c->pool_nleaf[vi];
                for (int l = 0; l < nl; ++l) u[l] = mcio_uniform(seed, st_init, g, (uint32_t)(k + l));
                mcio_pool_create(c, vi, idx + c->pool_offset[vi], u);
                k += nl;
            }
        double probability = vegasmc_target(c, f, ud, weights, pad); /* :155-166 */
        for (long ne = 1; ne <= steps; ++ne) { /* :184 */
            const uint64_t sidx = (g << 32) | (uint64_t)(ne - 1);
            /* ---- changeVariable  ref: updates.jl:45-106 ---- */
            do {
                /* :50; with many chains per block, chains (ch & ~63) .. (ch | 63) of a block share the pool-pick
                   sequence (stream MC_GROUP), which does not depend on the chain states */
                double upool = mcio_uniform(seed, st_step, sidx, 0);
                if (npool > 1 && nchain > 1) {
                    const uint64_t gidx = ((uint64_t)(ch & ~63L) << 32) | (uint64_t)(ne - 1);
                    upool = mcio_uniform(seed, stream_id_block(iteration, STREAM_MC_GROUP, block_index), gidx, 0);
                }
                int vi = (int)floor(upool * npool);
                if (vi >= npool) vi = npool - 1;
                const mcio_leaf *v0 = &c->leaf[c->pool_leaf0[vi]];
                if (c->pool_nleaf[vi] == 1 && v0->kind == MCIO_DISCRETE && v0->nbin == 1) break; /* :52-54 */
                if (c->maxdof[vi] <= 0) break;                                                    /* :55-57 */
                int slot = (int)floor(mcio_uniform(seed, st_step, sidx, 1) * c->maxdof[vi]) + 1;
                if (slot > c->maxdof[vi]) slot = c->maxdof[vi];
                int idx = c->pool_offset[vi] + slot; /* :58 */
                for (int l = 0; l < c->pool_nleaf[vi]; ++l) u[l] = mcio_uniform(seed, st_step, sidx, (uint32_t)(3 + l));
                double prop = mcio_pool_shift(c, vi, idx, u); /* :60 */
                if (prop <= 4.9406564584124654e-324) break;   /* :63-65 */
                gather_x(c, x);
                f(x, _weights, ud);                            /* :67-75 */
                c->neval += 1;                                 /* :77 */
                for (int i = 0; i <= N; ++i) _pad[i] = mcio_padding_probability(c, i); /* :79-81 */
                double newp = c->reweight[norm] * _pad[norm];                          /* :84 */
                for (int i = 0; i < N; ++i) newp += absw(c, _weights, i) * c->reweight[i] * _pad[i]; /* :85-87 */
                double R = prop * newp / probability;          /* :88 */
                c->propose[PA_IDX(c, 1, 0, vi)] += 1.0;        /* :90 propose[2, 1, vi] */
                if (mcio_uniform(seed, st_step, sidx, 2) < R) { /* :91 */
                    c->accept[PA_IDX(c, 1, 0, vi)] += 1.0;     /* :92 */
                    for (int i = 0; i < N * nc; ++i) weights[i] = _weights[i]; /* :93-95 */
                    for (int i = 0; i <= N; ++i) pad[i] = _pad[i];        /* :96-98 */
                    probability = newp;                        /* :100 */
                } else {
                    mcio_pool_shift_rollback(c, vi, idx);      /* :102 */
                }
            } while (0);
            /* ---- histogram  ref: montecarlo.jl:198-211 ---- */
            for (int i = 0; i < N; ++i) {
                double f2 = absw(c, weights, i) * absw(c, weights, i) / mcio_probability(c, i); /* :203 */
                double wf2 = f2 * pad[i] / probability;                                   /* :204 */
                for (int vi = 0; vi < npool; ++vi)                                        /* :205 */
                    for (int pos = 1; pos <= c->dof[i * npool + vi]; ++pos)               /* :207 */
                        pool_accumulate(c, vi, pos + c->pool_offset[vi], wf2);            /* :208 */
            }
            /* ---- measurement  ref: montecarlo.jl:213-232 ---- */
            if (ne % measurefreq == 0 && (double)ne >= burnin) { /* :213 */
                for (int i = 0; i < N; ++i) {
                    c->visited[i] += absw(c, weights, i) * fabs(pad[i] * c->reweight[i]) / probability; /* :216 */
                    for (int q = 0; q < nc; ++q) relw[nc * i + q] = weights[nc * i + q] * pad[i] / probability; /* :218/:220 */
                }
                gather_x(c, x); /* measure() reads the current (accepted) variables */
                measure(c, x, relw, ud);
                c->normalization += 1.0 * pad[norm] / probability;                   /* :229 */
                c->visited[norm] += c->reweight[norm] * pad[norm] / probability;    /* :230 */
            }
        }
        if (c->carry_store) {
            store_carried(c, ch, nchain, 0);
            c->carry->P[c->carry->wr][c->carry_lb * nchain + ch] = probability; /* the target at the configuration the chain stopped at */
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * src/main.jl
 * ---------------------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------------
 * src/mcmc/montecarlo.jl:72-184 + src/mcmc/updates.jl:1-147
 * The Markov chain walks over (integrand index curr, live variables).  A block's neval measured steps are
 * run as `nchain` independent chains of neval/nchain measured steps, each preceded by its burn-in
 * (nchain = 1 is the reference).  Chain ch of the block (its block index rides in the stream word, stream_id_block) draws
 *   init try t: stream MCMC_INIT, index g*16384 + t,  k = flat draw
 *   step s    : stream MCMC_STEP, index (g<<32 | s),  k = 0 update pick, 1 neighbor/pool pick, 2 slot pick,
 *               3 second slot pick (swap), 4 accept, 5 + flat draw index of a created/shifted (pool, slot, leaf)
 * ---------------------------------------------------------------------------------------- */
long mcio_mcmc_burnin(long steps, long nchain, int nslots, int Nd, int npool, double thermal_ratio) {
    long nburn = (long)floor((double)steps * thermal_ratio); /* :133 */
    if (nchain > 1) { /* many short chains: each must forget its start (this engine's own decomposition) */
        long fl = 64L * nslots + 16L * (npool + 1) * Nd;
        if (fl > nburn) nburn = fl;
    }
    return nburn;
}

int mcio_set_neighbor(mcio_config *c, const int *offsets, const int *list) {
    const int Nd = c->Ni + 1;
    for (int d = 0; d < Nd; ++d) {
        int n = offsets[d + 1] - offsets[d];
        if (n < 1) return 1;
        for (int j = 0; j < n; ++j)
            if (list[offsets[d] + j] < 0 || list[offsets[d] + j] >= Nd) return 2;
    }
    for (int d = 0; d < Nd; ++d) {
        int n = offsets[d + 1] - offsets[d];
        free(c->neighbor[d]);
        c->neighbor[d] = (int *)dup_mem(list + offsets[d], sizeof(int) * (size_t)n);
        c->nneighbor[d] = n;
    }
    return 0;
}

void mcio_set_thermal_ratio(mcio_config *c, double r) { c->thermal_ratio = r; }
void mcio_set_ncomp(mcio_config *c, int ncomp) { c->ncomp = ncomp == 2 ? 2 : 1; }
void mcio_set_measure(mcio_config *c, mcio_measure_fn fn) { c->measure_fn = fn; }

void mcio_set_reweight_goal(mcio_config *c, const double *goal) {
    free(c->reweight_goal);
    c->reweight_goal = goal ? (double *)dup_mem(goal, sizeof(double) * (size_t)(c->Ni + 1)) : NULL;
}

/* measure for the one integrand the chain sits on (mcmc/montecarlo.jl:163-170); relw has ncomp entries */
static inline void measure_one(mcio_config *c, const double *x, int i, const double *relw, const double *ud) {
    if (c->measure_fn) {
        double rw[2 * MCIO_MAXNI];
        for (int k = 0; k < c->Ni * c->ncomp; ++k) rw[k] = 0.0;
        for (int q = 0; q < c->ncomp; ++q) rw[i * c->ncomp + q] = relw[q];
        c->measure_fn(x, rw, ud, i, c->observable);
        return;
    }
    if (c->obs_bin_draw[i] >= 0) {
        const mcio_leaf *T = &c->leaf[c->draw_leaf[c->obs_bin_draw[i]]];
        int bin = (int)(x[c->obs_bin_draw[i]] - T->lower);
        if (bin < 0 || bin >= c->obs_nbin[i]) return;
        c->observable[c->obs_off[i] + bin] += relw[0];
        return;
    }
    for (int q = 0; q < c->ncomp; ++q) c->observable[c->obs_off[i] + q] += relw[q];
}

int mcio_mcmc_block(mcio_config *c, mcio_integrand_fn f, const double *ud, uint64_t seed,
                    uint32_t iteration, long block_index, long neval, long measurefreq, long nchain) {
    const int npool = c->npool, norm = c->Ni, Nd = c->Ni + 1;
    if (c->ndraw > MCIO_MAXDRAW || Nd > MCIO_MAXNI || measurefreq <= 0 || nchain < 1) return -1; /* :79 */
    const int nc = c->ncomp;
    double w[2 * MCIO_MAXNI], x[MCIO_MAXDRAW], u[MCIO_MAXDRAW];
    int kbase[64];
    int nslots = 0;
    for (int vi = 0, k = 0; vi < npool; ++vi) {
        kbase[vi] = k;
        k += c->maxdof[vi] * c->pool_width[vi];
        nslots += c->maxdof[vi];
    }
    const long steps = neval / nchain;
    /* :133; a carried chain has no start to burn in: it measures from its first step (the engine's many-chain decomposition only) */
    const long nburn = c->carry_load ? 0 : mcio_mcmc_burnin(steps, nchain, nslots, Nd, npool, c->thermal_ratio);
    const int nupd = 2 * npool + 2; /* :127-130: [changeIntegrand, swapVariable, changeVariable x 2*Nv] */
    const uint32_t st_init = stream_id_block(iteration, STREAM_MCMC_INIT, block_index), st_step = stream_id_block(iteration, STREAM_MCMC_STEP, block_index);
    int rc = 0;
    for (long ch = 0; ch < nchain; ++ch) {
        const uint64_t g = (uint64_t)ch;
        int curr = (nchain == 1) ? 0 : (int)(g % (uint64_t)Nd); /* :76 idx = 1; many chains start stratified */
        double weight[2] = {0.0, 0.0}, probability = 1.0;        /* :116 _State(curr, zero(T), 1.0) */
        int fresh = !c->carry_load;
        if (!fresh) { /* continues the previous iteration's chain: its configuration and the integrand it sat on */
            const mcio_carry *cy = c->carry;
            const long slot = carried_slot_of(c, ch, nchain, 1); /* (resampled: run_blocks, mcio_resample_chains) */
            load_carried(c, slot);
            curr = cy->curr[cy->rd][slot];
            if (curr != norm) {
                gather_x(c, x);
                f(x, w, ud);
                for (int q = 0; q < nc; ++q) weight[q] = w[nc * curr + q];
                probability = absw(c, w, curr) * c->reweight[curr];
                if (!(probability > MCIO_TINY)) {
                    fresh = 1;
                    curr = (int)(g % (uint64_t)Nd);
                }
            } else probability = c->reweight[curr];
        }
        for (long t = 0; fresh && t < 10000; ++t) {              /* :118-124 */
            /* initialize!  :190-205 (only the slots that are ever read: 1..maxdof) */
            for (int vi = 0; vi < npool; ++vi)
                for (int idx = 1; idx <= c->maxdof[vi]; ++idx) {
                    int nl = c->pool_width[vi];
                    for (int l = 0; l < nl; ++l)
                        u[l] = mcio_uniform(seed, st_init, g * 16384u + (uint64_t)t, (uint32_t)(kbase[vi] + (idx - 1) * nl + l));
                    mcio_pool_create(c, vi, idx + c->pool_offset[vi], u);
                }
            if (curr != norm) {
                gather_x(c, x);
                f(x, w, ud);
                for (int q = 0; q < nc; ++q) weight[q] = w[nc * curr + q]; /* :197 */
                probability = absw(c, w, curr) * c->reweight[curr]; /* :199 */
            } else {
                weight[0] = weight[1] = 0.0;                        /* :201 */
                probability = c->reweight[curr];                    /* :202 */
            }
            if (curr == norm || probability > MCIO_TINY) break;     /* :120-122 */
        }
        if (curr != norm && probability == 0.0) rc = -3;            /* :125-126 error(...) */
        /* holding times (the engine's diagnostic, mci_device.h mcmc_chains): step of the last change of every draw and of the
           integrand index; longest completed or still running hold */
        long last[MCIO_MAXDRAW], lastc = 0, hmax = 0;
        for (int k = 0; k < c->ndraw; ++k) last[k] = 0;
        for (long i = 1; i <= steps + nburn; ++i) {                 /* :134 */
            const uint64_t sidx = (g << 32) | (uint64_t)(i - 1);
            c->visited[curr] += 1.0;                                /* :136 */
            const int curr_old = curr;
            int moved = 0; /* 0 nothing accepted, 1 changeIntegrand, 2 changeVariable / swapVariable */
            int mv_pool = 0, mv_s1 = 0, mv_s2 = 0; /* the slot(s), 1-based within the pool, an accepted move touched */
            /* :137 rand(rng, updates); with many chains per block, chains (ch & ~63) .. (ch | 63) of a block share the
               update-type sequence (stream MCMC_GROUP): it does not depend on the chain states, so each chain is still a
               valid Markov chain and blocks stay independent */
            double uupd = mcio_uniform(seed, st_step, sidx, 0);
            if (nchain > 1) {
                const uint64_t gidx = ((uint64_t)(ch & ~63L) << 32) | (uint64_t)(i - 1);
                uupd = mcio_uniform(seed, stream_id_block(iteration, STREAM_MCMC_GROUP, block_index), gidx, 0);
            }
            int upd = (int)floor(uupd * nupd);
            if (upd >= nupd) upd = nupd - 1;
            if (upd == 0) {
                /* ---- changeIntegrand  updates.jl:1-69 ---- */
                do {
                    int j = (int)floor(mcio_uniform(seed, st_step, sidx, 1) * c->nneighbor[curr]);
                    if (j >= c->nneighbor[curr]) j = c->nneighbor[curr] - 1;
                    const int new_ = c->neighbor[curr][j];          /* :6 */
                    if (new_ == curr) break;                        /* :7 */
                    const int *cd = &c->dof[curr * npool], *nd = &c->dof[new_ * npool]; /* :9 */
                    double prop = (double)c->nneighbor[curr] / (double)c->nneighbor[new_]; /* :12 */
                    for (int vi = 0; vi < npool; ++vi) {            /* :15-26 */
                        const int off = c->pool_offset[vi], nl = c->pool_width[vi];
                        if (cd[vi] < nd[vi]) {
                            for (int pos = cd[vi] + 1; pos <= nd[vi]; ++pos) {
                                for (int l = 0; l < nl; ++l)
                                    u[l] = mcio_uniform(seed, st_step, sidx, (uint32_t)(5 + kbase[vi] + (pos - 1) * nl + l));
                                prop *= mcio_pool_create(c, vi, pos + off, u); /* :19 */
                            }
                        } else if (cd[vi] > nd[vi]) {
                            for (int pos = nd[vi] + 1; pos <= cd[vi]; ++pos) prop *= mcio_pool_remove(c, vi, pos + off); /* :23 */
                        }
                    }
                    if (prop <= 4.9406564584124654e-324) break;    /* :29-31 */
                    double neww[2] = {0.0, 0.0};                    /* :35-38 */
                    double newabs = 0.0;
                    if (new_ != norm) {
                        gather_x(c, x);
                        f(x, w, ud);
                        for (int q = 0; q < nc; ++q) neww[q] = w[nc * new_ + q];
                        newabs = absw(c, w, new_);
                    }
                    c->neval += 1;                                  /* :40 */
                    const double newp = (new_ == norm) ? c->reweight[new_] : newabs * c->reweight[new_]; /* :42-44 */
                    const double R = prop * newp / probability;     /* :46 */
                    c->propose[PA_IDX(c, 0, curr, new_)] += 1.0;    /* :48 propose[1, curr, new] */
                    if (mcio_uniform(seed, st_step, sidx, 4) < R) { /* :49 */
                        c->accept[PA_IDX(c, 0, curr, new_)] += 1.0; /* :50 */
                        moved = 1;
                        curr = new_;                                /* :51-53 */
                        weight[0] = neww[0];
                        weight[1] = neww[1];
                        probability = newp;
                    } /* else createRollback!/removeRollback! are no-ops  :55-68, sampler.jl:306,324 */
                } while (0);
            } else if (curr != norm) { /* updates.jl:73, :115 */
                const int *cd = &c->dof[curr * npool];
                int vi = (int)floor(mcio_uniform(seed, st_step, sidx, 1) * npool); /* :77, :119 */
                if (vi >= npool) vi = npool - 1;
                const int off = c->pool_offset[vi], nl = c->pool_width[vi];
                if (upd == 1) {
                    /* ---- swapVariable  updates.jl:113-147 ---- */
                    do {
                        if (cd[vi] <= 0) break;                     /* :121 */
                        int s1 = (int)floor(mcio_uniform(seed, st_step, sidx, 2) * cd[vi]) + 1; /* :122 */
                        int s2 = (int)floor(mcio_uniform(seed, st_step, sidx, 3) * cd[vi]) + 1; /* :123 */
                        if (s1 > cd[vi]) s1 = cd[vi];
                        if (s2 > cd[vi]) s2 = cd[vi];
                        if (s1 == s2) break;                        /* :124 */
                        const double prop = mcio_pool_swap(c, vi, s1 + off, s2 + off); /* :126 */
                        gather_x(c, x);
                        f(x, w, ud);                                /* :133 */
                        c->neval += 1;                              /* :135 */
                        const double newp = absw(c, w, curr) * c->reweight[curr]; /* :137 */
                        const double R = prop * newp / probability; /* :138 */
                        c->propose[PA_IDX(c, 2, curr, vi)] += 1.0;  /* :140 propose[3, curr, vi] */
                        if (mcio_uniform(seed, st_step, sidx, 4) < R) {
                            c->accept[PA_IDX(c, 2, curr, vi)] += 1.0;
                            moved = 2; mv_pool = vi; mv_s1 = s1; mv_s2 = s2;
                            for (int q = 0; q < nc; ++q) weight[q] = w[nc * curr + q];
                            probability = newp;
                        } else {
                            mcio_pool_swap(c, vi, s1 + off, s2 + off); /* :145 swapRollback! */
                        }
                    } while (0);
                } else {
                    /* ---- changeVariable  updates.jl:71-111 ---- */
                    do {
                        const mcio_leaf *v0 = &c->leaf[c->pool_leaf0[vi]];
                        if (c->pool_nleaf[vi] == 1 && v0->kind == MCIO_DISCRETE && v0->nbin == 1) break; /* :79-81 */
                        if (cd[vi] <= 0) break;                     /* :82 */
                        int slot = (int)floor(mcio_uniform(seed, st_step, sidx, 2) * cd[vi]) + 1; /* :83 */
                        if (slot > cd[vi]) slot = cd[vi];
                        for (int l = 0; l < nl; ++l)
                            u[l] = mcio_uniform(seed, st_step, sidx, (uint32_t)(5 + kbase[vi] + (slot - 1) * nl + l));
                        /* :85; a FermiK slot picks its move (scale / rotate / shift) with the otherwise unused uniform 3 */
                        const double prop = pool_shift_any(c, vi, slot + off, mcio_uniform(seed, st_step, sidx, 3), u);
                        if (prop <= 4.9406564584124654e-324) break; /* :88-90 */
                        gather_x(c, x);
                        f(x, w, ud);                                /* :92 */
                        c->neval += 1;                              /* :94 */
                        const double newp = absw(c, w, curr) * c->reweight[curr]; /* :96 */
                        const double R = prop * newp / probability; /* :97 */
                        c->propose[PA_IDX(c, 1, curr, vi)] += 1.0;  /* :99 propose[2, curr, vi] */
                        if (mcio_uniform(seed, st_step, sidx, 4) < R) {
                            c->accept[PA_IDX(c, 1, curr, vi)] += 1.0;
                            moved = 2; mv_pool = vi; mv_s1 = slot; mv_s2 = 0;
                            for (int q = 0; q < nc; ++q) weight[q] = w[nc * curr + q];
                            probability = newp;
                        } else {
                            mcio_pool_shift_rollback(c, vi, slot + off); /* :105 */
                        }
                    } while (0);
                }
            }
            if (moved == 1) { /* the slots changeIntegrand created start their first hold; the index ends one */
                for (int vi = 0; vi < npool; ++vi)
                    for (int pos = c->dof[curr_old * npool + vi] + 1; pos <= c->dof[curr * npool + vi]; ++pos)
                        for (int l = 0; l < c->pool_width[vi]; ++l) last[kbase[vi] + (pos - 1) * c->pool_width[vi] + l] = i;
                if (i - lastc > hmax) hmax = i - lastc;
                lastc = i;
            } else if (moved == 2) { /* an accepted move of a slot ends its hold (also when a Discrete redraw lands on the same value) */
                const int nl = c->pool_width[mv_pool];
                for (int which = 0; which < 2; ++which) {
                    const int sl = which == 0 ? mv_s1 : mv_s2;
                    if (sl <= 0) continue;
                    for (int l = 0; l < nl; ++l) {
                        const int k = kbase[mv_pool] + (sl - 1) * nl + l;
                        if (i - last[k] > hmax) hmax = i - last[k];
                        last[k] = i;
                    }
                }
            }
            /* ---- measurement  montecarlo.jl:144-172 ---- */
            if (i % measurefreq == 0 && i >= nburn) {
                if (curr != norm) {
                    for (int vi = 0; vi < npool; ++vi)              /* :147-154 */
                        for (int pos = 1; pos <= c->dof[curr * npool + vi]; ++pos)
                            pool_accumulate(c, vi, pos + c->pool_offset[vi], 1.0);
                    gather_x(c, x);
                    double rel[2] = {weight[0] / probability, weight[1] / probability}; /* :162 */
                    measure_one(c, x, curr, rel, ud);               /* :161-169 */
                } else {
                    c->normalization += 1.0 / c->reweight[norm];    /* :158 */
                }
            }
        }
        {   /* holds still running when the chain ends count with their length so far */
            const long tot = steps + nburn;
            if (tot - lastc > hmax) hmax = tot - lastc;
            if (curr != norm)
                for (int vi = 0; vi < npool; ++vi) {
                    const mcio_leaf *v0 = &c->leaf[c->pool_leaf0[vi]];
                    if (c->pool_nleaf[vi] == 1 && v0->kind == MCIO_DISCRETE && v0->nbin == 1) continue; /* nothing to sample, updates.jl:79-81 */
                    for (int pos = 1; pos <= c->dof[curr * npool + vi]; ++pos)
                        for (int l = 0; l < c->pool_width[vi]; ++l) {
                            const int k = kbase[vi] + (pos - 1) * c->pool_width[vi] + l;
                            if (tot - last[k] > hmax) hmax = tot - last[k];
                        }
                }
            int b = 0;
            for (long h = hmax; h > 0; h >>= 1) ++b; /* bit_width */
            if (tot < 2147483647L) c->hold_hist[b] += 1;
        }
        if (c->carry_store) store_carried(c, ch, nchain, curr);
    }
    return rc;
}

/* ref: main.jl:220-234 */
void mcio_standardize_block(long neval, long nblock, long nworker, long *nevalperblock, long *block) {
    if (nblock > nworker) nblock = (nblock / nworker) * nworker; /* :225-227 */
    else nblock = nworker;                                       /* :229 */
    *nevalperblock = neval / nblock;                             /* :232 */
    *block = nblock;
}

/* ref: main.jl:296-320 (real observables) */
void mcio_mean_std(const double *obs_sum, const double *obs_sq, long n, long block, double *mean, double *std) {
    for (long o = 0; o < n; ++o) {
        mean[o] = obs_sum[o] / (double)block; /* :317 */
        if (block > 1) {                      /* :301 */
            double v = (obs_sq[o] / (double)block - mean[o] * mean[o]) / (double)(block - 1); /* :308 */
            std[o] = v < 0.0 ? 0.0 : sqrt(v); /* :297-299 */
        } else {
            std[o] = 0.0;                     /* :311 */
        }
    }
}

/* ref: statistics.jl:186-220.  iter_mean/iter_std are [niter][stride] rows, this averages column 0
 * of the pointers given (caller offsets them).  init/max are 1-based like the reference. */
void mcio_average(const double *iter_mean, const double *iter_std, long niter, long stride, long init, long max,
                  double *mean, double *err, double *chi2) {
    (void)niter;
    if (max <= init) { /* :189-191 */
        *mean = iter_mean[0];
        *err = iter_std[0];
        *chi2 = 0.0;
        return;
    }
    double wsum = 0.0;
    for (long i = init; i <= max; ++i) { /* :217 */
        double s = iter_std[(i - 1) * stride] + 1.0e-10;
        wsum += 1.0 / (s * s);
    }
    double mea = 0.0;
    for (long i = init; i <= max; ++i) { /* :197 */
        double s = iter_std[(i - 1) * stride] + 1.0e-10;
        mea += iter_mean[(i - 1) * stride] * (1.0 / (s * s)) / wsum;
    }
    double c2 = 0.0;
    if (max > 1) /* :199-203 */
        for (long i = init; i <= max; ++i) {
            double s = iter_std[(i - 1) * stride] + 1.0e-10;
            double d = iter_mean[(i - 1) * stride] - mea;
            c2 += (1.0 / (s * s)) * d * d;
        }
    *mean = mea;
    *err = 1.0 / sqrt(wsum);                    /* :198 */
    *chi2 = c2 / (double)((max - init + 1) - 1); /* :204 */
}

/* ref: main.jl:322-346 */
void mcio_do_reweight(double *reweight, const double *visited, long nd, double gamma, const double *goal) {
    double avgstep = 0.0;
    for (long i = 0; i < nd; ++i) avgstep += visited[i]; /* :323 */
    for (long i = 0; i < nd; ++i) {                      /* :324-331 */
        if (visited[i] <= 1) reweight[i] *= pow(avgstep, gamma);
        else reweight[i] *= pow(avgstep / visited[i], gamma);
    }
    if (goal) { /* :334-337 */
        double gs = 0.0;
        for (long i = 0; i < nd; ++i) gs += goal[i];
        for (long i = 0; i < nd; ++i) reweight[i] *= goal[i] / gs;
    }
    double s = 0.0;
    for (long i = 0; i < nd; ++i) s += reweight[i];
    for (long i = 0; i < nd; ++i) reweight[i] /= s; /* :339 */
}

long mcio_packed_size(const mcio_config *c) {
    long n = 2L * c->nobs + 2 + (c->Ni + 1);
    for (int l = 0; l < c->nleaf; ++l) n += c->leaf[l].nbin;
    return n + 2L * c->npa; /* propose | accept, reduced with everything else (configuration.jl:297-298) */
}

mcio_result *mcio_result_create(int niter, int nobs, int Ni) {
    mcio_result *r = (mcio_result *)calloc(1, sizeof(mcio_result));
    r->niter = niter;
    r->nobs = nobs;
    r->Ni = Ni;
    r->iter_mean = (double *)calloc((size_t)niter * nobs, sizeof(double));
    r->iter_std = (double *)calloc((size_t)niter * nobs, sizeof(double));
    r->mean = (double *)calloc((size_t)nobs, sizeof(double));
    r->stdev = (double *)calloc((size_t)nobs, sizeof(double));
    r->chi2 = (double *)calloc((size_t)nobs, sizeof(double));
    return r;
}

void mcio_result_destroy(mcio_result *r) {
    if (!r) return;
    free(r->iter_mean); free(r->iter_std); free(r->mean); free(r->stdev); free(r->chi2);
    free(r);
}

/* One iteration's worth of blocks [block_lo, block_hi) (ref: main.jl:144-180 and _block! :236-292).
 * `c` plays summedConfig[1]: on return it holds the block-summed statistics (histograms, visited,
 * normalization, neval) and its pools/grids are unchanged.  obs_sum/obs_sq accumulate block means.
 * Block results are merged in block order, so the output does not depend on nthreads (the
 * reference's thread-order merge, main.jl:170-174, differs from this only by reassociation and by
 * (nthreads-1)*1e-10 in the histogram offsets). */
/* Carried :mcmc chains (this engine's many-chain decomposition only; mirror of k_resample_chains, mci_static_kernels.h).  The chains a
 * block stored at the end of an iteration are a sample of that iteration's target pi_k(idx, x) ~ reweight_k[idx] |f_idx(x)|; doReweight!
 * has since moved the factors (main.jl:322-346), so the next iteration's target differs from it by exactly the known ratio
 * w[idx] = reweight_{k+1}[idx] / reweight_k[idx].  The new chains therefore continue stored chains drawn with probability ~ w[curr]
 * (systematic resampling over the block's stored chains in chain order, one fixed offset: deterministic): a start population distributed like
 * the NEW target, which also takes back the visit fluctuation the new factors were computed from -- chains carried as they are start
 * over-represented exactly where the new factors say "fewer" (measured: 2 sigma per run on BASELINE configs[4]).
 *   W[j] = sum_i w[i] * #(stored chains j' <= j that ended on integrand i)     (sum over i = 0 .. Nd-1 in that order)
 *   new chain c continues the first stored chain j with W[j] > (c + u) * (W[n_old-1] / n_new),  u = (sqrt(5) - 1) / 2
 * (any offset in [0, 1) is a valid systematic resampling; 1/2 makes (c + u) n_old / n_new an integer for many chain counts -- with all
 * stored chains on one integrand the comparison is then an exact tie, decided by the last bit of w)                                  */
/* Carried :vegasmc chains: one weight per stored chain (new target / old target).  Mirror of k_resample_chains' w_chain path, association
 * included: the stored chains are cut into 256 stretches of ceil(n_old / 256); W[j] = (sum of the stretches before, added in order) +
 * (running sum along the own stretch); then the same systematic pick. */
void mcio_resample_weighted(const double *w, long n_old, long n_new, long *src) {
    double *W = (double *)malloc((size_t)n_old * sizeof(double));
    const long T = 256, per = (n_old + T - 1) / T;
    double part[256];
    for (long t = 0; t < T; ++t) {
        const long j0 = t * per < n_old ? t * per : n_old, j1 = j0 + per < n_old ? j0 + per : n_old;
        double mine = 0.0;
        for (long j = j0; j < j1; ++j) mine += w[j];
        part[t] = mine;
    }
    for (long t = 0; t < T; ++t) {
        const long j0 = t * per < n_old ? t * per : n_old, j1 = j0 + per < n_old ? j0 + per : n_old;
        double below = 0.0, run = 0.0;
        for (long q = 0; q < t; ++q) below += part[q];
        for (long j = j0; j < j1; ++j) {
            run += w[j];
            W[j] = below + run;
        }
    }
    const double step = W[n_old - 1] / (double)n_new;
    for (long c = 0; c < n_new; ++c) {
        const double target = ((double)c + 0.6180339887498949) * step;
        long lo = 0, hi = n_old - 1; /* smallest j with W[j] > target */
        while (lo < hi) {
            const long mid = (lo + hi) >> 1;
            if (W[mid] > target) hi = mid;
            else lo = mid + 1;
        }
        src[c] = lo;
    }
    free(W);
}

void mcio_resample_chains(const int *curr_old, long n_old, int nd, const double *rw_now, const double *rw_used, long n_new, long *src) {
    double w[65];
    long cnt[65]; /* (nd <= 64: the integrands of a draw are a 64-bit mask) */
    double *W = (double *)malloc((size_t)n_old * sizeof(double));
    for (int i = 0; i < nd; ++i) {
        w[i] = rw_now[i] / rw_used[i];
        cnt[i] = 0;
    }
    for (long j = 0; j < n_old; ++j) {
        cnt[curr_old[j]] += 1;
        double s = 0.0;
        for (int i = 0; i < nd; ++i) s += w[i] * (double)cnt[i];
        W[j] = s;
    }
    const double step = W[n_old - 1] / (double)n_new;
    for (long c = 0; c < n_new; ++c) {
        const double target = ((double)c + 0.6180339887498949) * step;
        long lo = 0, hi = n_old - 1; /* smallest j with W[j] > target */
        while (lo < hi) {
            const long mid = (lo + hi) >> 1;
            if (W[mid] > target) hi = mid;
            else lo = mid + 1;
        }
        src[c] = lo;
    }
    free(W);
}

static int run_blocks(mcio_config *c, int solver, mcio_integrand_fn f, const double *ud, long nevalperblock,
                      long block_lo, long block_hi, uint32_t iteration, long measurefreq, uint64_t seed,
                      int nthreads, long nchain, double *obs_sum, double *obs_sq) {
    const long nb = block_hi - block_lo;
    if (nb <= 0) return 0;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nb) nthreads = (int)nb;
    mcio_config **done = (mcio_config **)calloc((size_t)nb, sizeof(mcio_config *));
    int err = 0;
    /* carried chains: the rule of mci_api.hip mci_iteration_run */
    mcio_carry *cy = c->carry;
    const int carry_on = cy->mode != 0; /* automatic = both chain solvers */
    const int carried = solver != MCIO_VEGAS && carry_on && cy->valid && cy->solver == solver && cy->lo == block_lo && cy->hi == block_hi &&
                        cy->iteration + 1 == (long)iteration && cy->nchain > 1 && nchain > 1 &&
                        (solver != MCIO_VEGASMC || cy->ntrain_stored >= 1 || cy->ntrain_stored == cy->ntrain); /* (:vegasmc: not out of a launch on the untrained map onto a refined one) */
    const int keep = solver != MCIO_VEGAS && carry_on && nchain > 1;
    if (keep) {
        const int wr = cy->valid ? 1 - cy->cur : cy->cur;
        if (nb * nchain > cy->cap[wr]) {
            free(cy->x[wr]);
            free(cy->curr[wr]);
            free(cy->P[wr]);
            cy->cap[wr] = nb * nchain;
            cy->x[wr] = (double *)calloc((size_t)cy->cap[wr] * (size_t)(c->ndraw > 0 ? c->ndraw : 1), sizeof(double));
            cy->curr[wr] = (int *)calloc((size_t)cy->cap[wr], sizeof(int));
            cy->P[wr] = (double *)calloc((size_t)cy->cap[wr], sizeof(double));
        }
        cy->rd = cy->cur;
        cy->wr = wr;
    } else if (carried) cy->rd = cy->cur;
    cy->load_nchain = cy->nchain;
    if (carried) { /* which stored chain every new chain continues */
        if (nb * nchain > cy->src_cap) {
            free(cy->src);
            cy->src_cap = nb * nchain;
            cy->src = (long *)calloc((size_t)cy->src_cap, sizeof(long));
        }
        if (solver == MCIO_MCMC)
            for (long b = 0; b < nb; ++b)
                mcio_resample_chains(cy->curr[cy->rd] + b * cy->load_nchain, cy->load_nchain, c->Ni + 1, c->reweight, cy->rw_used, nchain, cy->src + b * nchain);
        else { /* :vegasmc: the NEW target (refined map, moved reweight factors) over the old one at every stored configuration (mirror of
                  vegasmc_carry_weights), then the weighted pick */
            double *w = (double *)malloc((size_t)cy->load_nchain * sizeof(double));
            mcio_config *cw = mcio_config_clone(c);
            free(cw->carry);
            cw->carry = cy;
            cw->carry_owner = 0;
            double pad[MCIO_MAXNI];
            for (long b = 0; b < nb; ++b) {
                for (long j = 0; j < cy->load_nchain; ++j) {
                    const long slot = b * cy->load_nchain + j;
                    load_carried(cw, slot);
                    const double r = vegasmc_target(cw, f, ud, NULL, pad) / cy->P[cy->rd][slot];
                    w[j] = (r == r && r < 1.7976931348623157e308 && r > 0.0) ? r : 0.0;
                }
                mcio_resample_weighted(w, cy->load_nchain, nchain, cy->src + b * nchain);
            }
            mcio_config_destroy(cw);
            free(w);
        }
    }
    if (keep) { /* the reweight factors this launch's chains run under */
        if (!cy->rw_used) cy->rw_used = (double *)calloc((size_t)(c->Ni + 1), sizeof(double));
        memcpy(cy->rw_used, c->reweight, (size_t)(c->Ni + 1) * sizeof(double));
    }
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (long b = 0; b < nb; ++b) {
        mcio_config *cn = mcio_config_clone(c); /* main.jl:130 deepcopy per worker; here per block */
        free(cn->carry);                         /* the clones share the parent's chain state */
        cn->carry = cy;
        cn->carry_owner = 0;
        cn->carry_load = carried;
        cn->carry_store = keep;
        cn->carry_lb = b;
        mcio_clear_statistics(cn);               /* main.jl:251 */
        int rc;
        if (solver == MCIO_VEGAS)
            rc = mcio_vegas_block(cn, f, ud, seed, iteration, block_lo + b, nevalperblock, measurefreq); /* main.jl:257 */
        else if (solver == MCIO_VEGASMC)
            rc = mcio_vegasmc_block(cn, f, ud, seed, iteration, block_lo + b, nevalperblock, measurefreq, nchain); /* main.jl:254 */
        else
            rc = mcio_mcmc_block(cn, f, ud, seed, iteration, block_lo + b, nevalperblock, measurefreq, nchain); /* main.jl:260 */
        if (rc || !(cn->normalization > 0.0)) { /* main.jl:269-271 */
#ifdef _OPENMP
#pragma omp atomic write
#endif
            err = 1;
        }
        done[b] = cn;
    }
    if (keep) {
        cy->cur = cy->wr;
        cy->valid = 1;
        cy->ntrain_stored = cy->ntrain;
        cy->solver = solver;
        cy->iteration = (long)iteration;
        cy->lo = block_lo;
        cy->hi = block_hi;
        cy->nchain = nchain;
    } else if (solver != MCIO_VEGAS) cy->valid = 0;
    mcio_clear_statistics(c); /* main.jl:149 */
    for (long b = 0; b < nb; ++b) {
        mcio_config *cn = done[b];
        mcio_add_config(c, cn); /* main.jl:273 */
        for (int o = 0; o < c->nobs; ++o) { /* main.jl:275-287 */
            double m = cn->observable[o] / cn->normalization;
            obs_sum[o] += m;
            obs_sq[o] += m * m;
        }
        mcio_config_destroy(cn);
    }
    free(done);
    return err;
}

static void pack(const mcio_config *c, const double *obs_sum, const double *obs_sq, double *out) {
    long p = 0;
    for (int o = 0; o < c->nobs; ++o) out[p++] = obs_sum[o];
    for (int o = 0; o < c->nobs; ++o) out[p++] = obs_sq[o];
    out[p++] = c->normalization;
    out[p++] = (double)c->neval;
    for (int i = 0; i < c->Ni + 1; ++i) out[p++] = c->visited[i];
    for (int l = 0; l < c->nleaf; ++l)
        for (int i = 0; i < c->leaf[l].nbin; ++i) out[p++] = c->leaf[l].hist[i];
    for (int v = 0; v < c->npa; ++v) out[p++] = c->propose[v];
    for (int v = 0; v < c->npa; ++v) out[p++] = c->accept[v];
}

int mcio_iteration(mcio_config *c, int solver, mcio_integrand_fn f, const double *ud,
                   long nevalperblock, long block_lo, long block_hi, uint32_t iteration,
                   long measurefreq, uint64_t seed, int nthreads, long nchain, double *packed_out) {
    double *obs_sum = (double *)calloc((size_t)c->nobs, sizeof(double));
    double *obs_sq = (double *)calloc((size_t)c->nobs, sizeof(double));
    int rc = run_blocks(c, solver, f, ud, nevalperblock, block_lo, block_hi, iteration, measurefreq, seed, nthreads,
                        nchain, obs_sum, obs_sq);
    if (packed_out) pack(c, obs_sum, obs_sq, packed_out);
    free(obs_sum);
    free(obs_sq);
    return rc;
}

/* ref: main.jl:71-218 (single process: mpi_nprocs() == 1) */
int mcio_integrate(mcio_config *c, int solver, mcio_integrand_fn f, const double *ud, long neval,
                   int niter, long block, int ignore, int adapt, double gamma, long measurefreq,
                   uint64_t seed, int nthreads, long nchain, mcio_result *out) {
    long nevalperblock;
    if (!(neval > block)) return -2; /* :222 */
    mcio_standardize_block(neval, block, 1, &nevalperblock, &block); /* :121 */
    double *obs_sum = (double *)calloc((size_t)c->nobs, sizeof(double));
    double *obs_sq = (double *)calloc((size_t)c->nobs, sizeof(double));
    int rc = 0;
    out->neval = 0;
    for (int iter = 0; iter < niter; ++iter) { /* :142 */
        for (int o = 0; o < c->nobs; ++o) obs_sum[o] = obs_sq[o] = 0.0; /* :144-148 */
        rc |= run_blocks(c, solver, f, ud, nevalperblock, 0, block, (uint32_t)iter, measurefreq, seed, nthreads, nchain,
                         obs_sum, obs_sq); /* :152-174 */
        out->neval += c->neval;
        if (solver == MCIO_VEGASMC || solver == MCIO_MCMC) mcio_do_reweight(c->reweight, c->visited, c->Ni + 1, gamma, c->reweight_goal); /* :183 */
        if (adapt) mcio_train(c); /* :193-198 */
        mcio_mean_std(obs_sum, obs_sq, c->nobs, block, out->iter_mean + (size_t)iter * c->nobs,
                      out->iter_std + (size_t)iter * c->nobs); /* :203 */
    }
    for (int o = 0; o < c->nobs; ++o) /* :211 -> statistics.jl:24-55 */
        mcio_average(out->iter_mean + o, out->iter_std + o, niter, c->nobs, ignore + 1, niter, &out->mean[o],
                     &out->stdev[o], &out->chi2[o]);
    free(obs_sum);
    free(obs_sq);
    return rc;
}
