/*
 * mci_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the VEGAS / VegasMC sampling path of
 * numericalEFT/MCIntegration.jl v0.4.2 (reference mounted at /root/reference).
 * Every function cites the reference file:line it follows.
 *
 * WHO MAY USE THIS: only tests/, __graft_entry__.smoke() and the `cpu_baseline`
 * leg of bench.py.  The product (mcintegration.jl_amd/, include/mci.h) never
 * links, imports or calls anything in oracle/.
 *
 * PINNING STATUS
 *   - pinned against every deterministic known-answer the reference's own tests
 *     hold for this path: Dist.locate (test/utility.jl:2-9), _maxdof
 *     (test/utility.jl:14-15), probability/padding invariant
 *     (test/utility.jl:30-55), _mean_std (test/statistics.jl:14-46), doReweight!
 *     fixed point (test/mpi_test.jl:148-169), reduce semantics
 *     (test/mpi_test.jl:73-146); hand-derived golden vectors for smooth/rescale/
 *     train!/map-draw in tests/golden/; analytic k-sigma targets of
 *     test/montecarlo.jl:298-387 and test/bubble.jl.
 *   - same-seed, stream-dependent values are "parity unpinned": the reference
 *     draws from Julia's Random.MersenneTwister (stdlib, unpinned version), Julia
 *     is absent from this image, and no reference test fixes a seed.  The oracle
 *     uses a Philox4x32-10 counter RNG instead, so agreement with the Julia
 *     reference is statistical (k-sigma), by the reference's own standard
 *     (test/runtests.jl:4-9).
 */
#ifndef MCI_ORACLE_H
#define MCI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCIO_TINY 4.940656458412465e-274 /* src/MCIntegration.jl:11  eps(0.0)*1e50 */

enum { MCIO_CONTINUOUS = 0, MCIO_DISCRETE = 1, MCIO_FERMIK = 2 };
enum { MCIO_VEGAS = 0, MCIO_VEGASMC = 1, MCIO_MCMC = 2 };
/* how Continuous/Discrete prob[idx] is maintained per draw */
enum {
    MCIO_PROB_CREATE = 0, /* prob = 1/(N*dx)        sampler.jl:303, vegas/montecarlo.jl:128-129 */
    MCIO_PROB_SHIFT = 1   /* prob *= dx_old/dx_new  sampler.jl:383-384 (what Vegas.montecarlo runs) */
};

/* user integrand: x = flat draws (see mcio_config), w = Ni*ncomp outputs (ncomp = 2: (re, im) pairs), ud = userdata */
typedef void (*mcio_integrand_fn)(const double *x, double *w, const double *ud);
/* user measure (vegas/montecarlo.jl:156-161; mcmc/montecarlo.jl:166-169): rw = relative weights [Ni*ncomp],
   idx = -1 (vegas, vegasmc: all integrands) or the current integrand (mcmc: only rw[idx*ncomp ..] is valid),
   obs = the flat observable array it accumulates into */
typedef void (*mcio_measure_fn)(const double *x, const double *rw, const double *ud, int idx, double *obs);

typedef struct {
    int kind;           /* MCIO_CONTINUOUS | MCIO_DISCRETE */
    int pool;           /* index into config.var this leaf belongs to */
    double lower, upper;
    int npts;           /* continuous: number of grid points (ninc, default 1000) */
    int nbin;           /* continuous: npts-1 ; discrete: upper-lower+1 */
    double alpha;
    int adapt;
    double *grid;         /* continuous [npts]   variable.jl:94 */
    double *hist;         /* [nbin]              variable.jl:96, :279 */
    double *accumulation; /* discrete [nbin+1]   variable.jl:280 */
    double *distribution; /* discrete [nbin]     variable.jl:281 */
    int P;                /* pool length incl. cache slot  variable.jl:139 */
    double *data;         /* [P] */
    long *gidx;           /* [P] 1-based like the reference */
    double *prob;         /* [P] */
    /* FermiK{D} (variable.jl:1-20; :mcmc only): lower = kF, upper = dk, npts = D, alpha = maxK; data is [P+1][D] */
    int width;            /* x entries (and uniforms of create!) per slot: D for FermiK, 1 otherwise */
} mcio_leaf;

typedef struct {
    int nleaf, npool, Ni; /* Ni = user integrands; Nd = Ni+1 (configuration.jl:153) */
    mcio_leaf *leaf;      /* leaves of one pool are contiguous */
    int *pool_leaf0;      /* [npool] first leaf of pool */
    int *pool_nleaf;      /* [npool] 1 for Continuous/Discrete, >1 for CompositeVar */
    int *pool_offset;     /* [npool] var.offset */
    double **pool_prob;   /* [npool][P] CompositeVar.prob (variable.jl:399); aliases leaf prob when nleaf==1 */
    double *pool_prob_cache; /* [npool] CompositeVar._prob_cache */
    int *dof;             /* [(Ni+1)*npool], last row = 0 */
    int *maxdof;          /* [npool] configuration.jl:229-236 */
    int ndraw;            /* sum_vi maxdof[vi]*pool_nleaf[vi] : uniforms per Vegas sample */
    int *draw_leaf;       /* [ndraw] leaf drawn at flat position k */
    int *draw_slot;       /* [ndraw] 1-based slot idx (without offset) */
    /* observables: integrand i owns obs[obs_off[i] .. +obs_nbin[i]); obs_bin_draw[i] = flat draw
       index of the Discrete draw that selects the bin (example/bubble.jl:81-84) or -1 */
    int nobs;
    int *obs_off, *obs_nbin, *obs_bin_draw;
    double *observable;   /* [nobs] */
    double normalization;
    long neval;
    double *reweight;     /* [Ni+1] */
    double *visited;      /* [Ni+1] */
    double *propose;      /* [3][Ni+1][pam] row-major = propose[update, integrand, target] of configuration.jl:185 (0-based):
                             changeIntegrand [0][curr][new] (mcmc/updates.jl:48), changeVariable [1][curr][vi] (:100; vegasmc
                             [1][0][vi], vegas_mc/updates.jl:90), swapVariable [2][curr][vi] (:138) */
    double *accept;       /* same shape (configuration.jl:186) */
    int prob_mode;
    int npa;              /* 3 * (Ni+1) * pam */
    int pam;              /* max(Ni+1, npool): the last dimension */
    int rng_bits;         /* :vegas sample stream: 52 (default) or 32 random bits per draw (mci_set_rng_bits) */
    int *nneighbor;       /* [Ni+1] configuration.jl:201-227 */
    int **neighbor;       /* [Ni+1][nneighbor] 0-based integrand indices; index Ni = normalisation */
    double thermal_ratio; /* mcmc/montecarlo.jl:77 (default 0.1) */
    double *reweight_goal; /* [Ni+1] or NULL  main.jl:81, :334-337 */
    int ncomp;             /* 1: Float64 weights; 2: ComplexF64 (`type` kwarg, configuration.jl:108) stored (re, im) */
    mcio_measure_fn measure_fn; /* NULL = default / bin-by-Discrete measure */
    int *pool_width;       /* [npool] x entries per slot: number of leaves, or D for a FermiK pool */
    int *draw_comp;        /* [ndraw] component within the leaf's slot (FermiK), 0 otherwise */
    unsigned long long *hold_hist; /* [64] :mcmc chains by bit_width(longest holding time); the engine's own diagnostic
                                      (include/mci.h mci_get_hold_histogram), restated here so that it can be checked */
    /* carried chains (mirror of include/mci.h mci_set_chain_carry; this engine's many-chain decomposition only): the state is shared
       with the per-block clones run_blocks makes; carry_load / carry_store / carry_lb are set per launch and block */
    struct mcio_carry *carry;
    int carry_owner;
    int carry_load, carry_store;
    long carry_lb;
} mcio_config;

/* end configurations of the last chain-solver launch: x[buf][k * cap + local block * nchain + ch], curr[buf][...] (:mcmc) */
typedef struct mcio_carry {
    int mode;          /* -1 automatic (default) / 1: the NEXT iteration of the same chain solver over the same blocks continues the chains; 0 off */
    double *x[2];
    int *curr[2];
    long cap[2];
    int cur, valid, solver;
    long lo, hi, nchain, iteration;
    int rd, wr;        /* buffers of the launch in flight */
    long load_nchain;
    /* :mcmc (mirror of k_resample_chains, mci_static_kernels.h): the reweight factors the stored chains ran under, and for the launch in
     * flight the stored chain every new chain continues, src[local block * nchain + ch] (index within the block) */
    double *rw_used;
    long *src;
    long src_cap;
    /* :vegasmc (mirror of BatchArgs::store_P / vegasmc_carry_weights): the chain's target density config.probability at every stored
     * configuration, P[buf][local block * nchain + ch]; the next launch resamples the stored chains with probability ~ new target / old */
    double *P[2];
    /* :vegasmc chains are carried only out of a launch that ran on a map train! had refined at least once: chains of the automatic
     * length have not reached their target on the untrained map of a heavy-tailed integrand (log(x)/sqrt(x): the first iteration of a
     * cold call 14 sigma per run off), and a population that is no sample of the old target cannot be resampled into one of the new
     * -- or on a map that has not been refined since (adapt = false: the target is the one the chains are walking towards)
     * (mirror of mci_problem::ntrain / chain_ntrain) */
    long ntrain, ntrain_stored;
} mcio_carry;

typedef struct {
    int niter, nobs, Ni;
    double *iter_mean; /* [niter*nobs] */
    double *iter_std;  /* [niter*nobs] */
    double *mean;      /* [nobs] */
    double *stdev;     /* [nobs] */
    double *chi2;      /* [nobs] reduced chi2 */
    long neval;
} mcio_result;

void mcio_resample_chains(const int *curr_old, long n_old, int nd, const double *rw_now, const double *rw_used, long n_new, long *src);
/* ... with one weight per stored chain (mirror of k_resample_chains' w_chain path: the same association of the running sums) */
void mcio_resample_weighted(const double *w, long n_old, long n_new, long *src);
void mcio_set_chain_carry(mcio_config *c, int mode); /* -1 automatic (:vegasmc) | 0 off | 1 :vegasmc and :mcmc */

/* ---- RNG ---- */
void mcio_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void mcio_philox4x32_r(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds);
void mcio_set_rng_rounds(int rounds); /* 10 (default) | 7: every stream of this process, mirror of mci_set_rng_rounds */
int mcio_get_rng_rounds(void);
/* uniform in [0,1) for (seed, stream, index, draw k) -- the stream contract shared with the HIP path */
double mcio_uniform(uint64_t seed, uint32_t stream, uint64_t index, uint32_t k);
double mcio_uniform32(uint64_t seed, uint32_t stream, uint64_t index, uint32_t k); /* the opt-in 32-bit stream of :vegas */

/* ---- src/distribution/common.jl ---- */
long mcio_locate(const double *acc, long n, double p);            /* :8-36, 1-based result, -1 if outside */
void mcio_smooth(const double *dist, long n, double factor, double *out); /* :43-54 */
double mcio_sum16(const double *v, long n);                        /* Julia sum() of a histogram-length vector: fixed 16-partial order */
double mcio_sum_julia(const double *v, long n);                    /* ... of any length: pairwise halves above 1024 elements */
int mcio_rescale(double *dist, long n, double alpha);             /* :67-82, in place; !=0 on assert failure */

/* ---- src/distribution/variable.jl ---- */
int mcio_train_continuous(double *grid, long npts, double *hist, double alpha); /* :206-239 */
int mcio_train_discrete(double *hist, long K, double alpha, double *distribution, double *accumulation); /* :369-382 */

/* ---- config ---- */
mcio_config *mcio_config_create(int nleaf, const int *kind, const int *pool, const double *lower,
                                const double *upper, const int *npts, const double *alpha,
                                const int *adapt, const int *pool_offset, int npool, int Ni,
                                const int *dof /* [Ni*npool] */, const int *obs_nbin /* [Ni] or NULL */,
                                const int *obs_bin_draw /* [Ni] or NULL */);
void mcio_config_destroy(mcio_config *c);
mcio_config *mcio_config_clone(const mcio_config *c);
int mcio_set_grid(mcio_config *c, int leaf, const double *grid, int npts);
int mcio_set_distribution(mcio_config *c, int leaf, const double *dist);
void mcio_maxdof(const int *dof, int nd, int npool, int *out);    /* configuration.jl:229-236 */
void mcio_clear_statistics(mcio_config *c);                       /* configuration.jl:238-250 */
void mcio_add_config(mcio_config *c, const mcio_config *ic);      /* configuration.jl:252-262 */
void mcio_train(mcio_config *c);                                  /* main.jl:194-195 */

/* ---- sampler.jl ---- */
double mcio_create(mcio_config *c, int leaf, int idx, double u);  /* :293-305, :13-22 (idx 1-based incl. offset) */
double mcio_shift(mcio_config *c, int leaf, int idx, double u);   /* :336-386, :57-71 */
void mcio_shift_rollback(mcio_config *c, int leaf, int idx);      /* :388-393, :73-77 */
double mcio_pool_shift(mcio_config *c, int vi, int idx, const double *u); /* :431-440 ; u = pool_nleaf uniforms */
double mcio_pool_create(mcio_config *c, int vi, int idx, const double *u);
void mcio_pool_shift_rollback(mcio_config *c, int vi, int idx);   /* :441-446 */
double mcio_remove(mcio_config *c, int leaf, int idx);            /* :318-323, :36-40 */
double mcio_pool_remove(mcio_config *c, int vi, int idx);         /* :422-428 */
double mcio_pool_swap(mcio_config *c, int vi, int idx1, int idx2); /* :395-408, :86-97, :448-455 (its own rollback) */
double mcio_total_probability(const mcio_config *c);              /* variable.jl:587-599 */
double mcio_probability(const mcio_config *c, int i);             /* variable.jl:606-619 */
double mcio_padding_probability(const mcio_config *c, int i);     /* variable.jl:628-641 */

/* ---- solvers ---- */
/* one Vegas block (vegas/montecarlo.jl:72-191).  Sample n of the block uses RNG index
   block_index*neval + n, stream = iteration. */
int mcio_vegas_block(mcio_config *c, mcio_integrand_fn f, const double *ud, uint64_t seed,
                     uint32_t iteration, long block_index, long neval, long measurefreq);
/* one VegasMC block (vegas_mc/montecarlo.jl:112-241) run as `nchain` independent chains of
   neval/nchain steps; nchain=1 is the reference's single chain. */
int mcio_vegasmc_block(mcio_config *c, mcio_integrand_fn f, const double *ud, uint64_t seed,
                       uint32_t iteration, long block_index, long neval, long measurefreq,
                       long nchain);

/* one MCMC block (mcmc/montecarlo.jl:72-184 with mcmc/updates.jl:1-147) run as `nchain` independent chains of
   neval/nchain measured steps each (+ burn-in, mcio_mcmc_burnin); nchain=1 is the reference's single chain. */
int mcio_mcmc_block(mcio_config *c, mcio_integrand_fn f, const double *ud, uint64_t seed,
                    uint32_t iteration, long block_index, long neval, long measurefreq, long nchain);
long mcio_mcmc_burnin(long steps, long nchain, int nslots, int Nd, int npool, double thermal_ratio);
int mcio_set_neighbor(mcio_config *c, const int *offsets /* [Ni+2] */, const int *list); /* configuration.jl:201-227 */
void mcio_set_thermal_ratio(mcio_config *c, double r);
void mcio_set_ncomp(mcio_config *c, int ncomp);              /* before any run; resizes nothing: obs_nbin already counts doubles */
void mcio_set_measure(mcio_config *c, mcio_measure_fn fn);
void mcio_set_reweight_goal(mcio_config *c, const double *goal /* [Ni+1] or NULL */);

/* ---- main.jl / statistics.jl ---- */
void mcio_standardize_block(long neval, long nblock, long nworker, long *nevalperblock, long *block); /* main.jl:220-234 */
void mcio_mean_std(const double *obs_sum, const double *obs_sq, long n, long block, double *mean, double *std); /* main.jl:296-320 */
void mcio_average(const double *iter_mean, const double *iter_std, long niter, long stride, long init, long max,
                  double *mean, double *err, double *chi2);       /* statistics.jl:186-220 (1-based init,max) */
void mcio_do_reweight(double *reweight, const double *visited, long nd, double gamma, const double *goal); /* main.jl:322-346 */

/* integrate loop (main.jl:71-218). nthreads>1 mirrors parallel=:thread (main.jl:153-158).
   block_lo/block_hi select the blocks this worker runs (MPI-rank analogue, main.jl:152-166);
   pass 0,-1 for all. If packed_out != NULL the loop stops after ONE iteration's sampling and writes
   [obsSum(nobs) | obsSqSum(nobs) | normalization | neval | visited(Ni+1) | histograms] without training. */
int mcio_integrate(mcio_config *c, int solver, mcio_integrand_fn f, const double *ud, long neval,
                   int niter, long block, int ignore, int adapt, double gamma, long measurefreq,
                   uint64_t seed, int nthreads, long nchain, mcio_result *out);
int mcio_iteration(mcio_config *c, int solver, mcio_integrand_fn f, const double *ud,
                   long nevalperblock, long block_lo, long block_hi, uint32_t iteration,
                   long measurefreq, uint64_t seed, int nthreads, long nchain, double *packed_out);
long mcio_packed_size(const mcio_config *c);
mcio_result *mcio_result_create(int niter, int nobs, int Ni);
void mcio_result_destroy(mcio_result *r);

/* built-in integrands (independent C restatements of the catalog used by tests/bench) */
mcio_integrand_fn mcio_builtin(const char *name);

#ifdef __cplusplus
}
#endif
#endif
