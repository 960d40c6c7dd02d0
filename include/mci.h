/*
 * mci.h -- C ABI of the MI355X-native VEGAS / VegasMC sampling engine (libmci_hip.so).
 *
 * This is the drop-in boundary for ONE path of numericalEFT/MCIntegration.jl (v0.4.2): the
 * per-iteration sample batch behind `integrate(...; solver=:vegas|:vegasmc|:mcmc)`.  The reference has no
 * FFI of its own; the seam this library replaces is the solver dispatch inside `_block!`
 * (reference src/main.jl:253-264) together with the iteration loop around it
 * (src/main.jl:142-207).  Every entry point below names the reference code it stands in for.
 * A Julia `ccall` binding (mcintegration.jl_amd/julia/MCIntegrationHIP.jl) and a Python ctypes
 * binding (mcintegration.jl_amd/_lib.py) consume exactly these symbols; see INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns an int status (0 = MCI_OK)
 * and leaves a message retrievable with mci_last_error(); the caller owns all input arrays (they
 * are copied), the library owns device buffers until *_destroy.  One mci_ctx per host thread /
 * process / GPU; calls on one mci_problem are not re-entrant.  All arithmetic is IEEE fp64.
 */
#ifndef MCI_H
#define MCI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCI_OK 0
#define MCI_ERR_INVALID 1       /* bad argument (reference: @assert / error() in the ctor paths) */
#define MCI_ERR_HIP 2           /* HIP runtime / hiprtc failure */
#define MCI_ERR_COMPILE 3       /* integrand source failed to compile */
#define MCI_ERR_NORMALIZATION 4 /* "Block normalization ... is not positively defined!"  main.jl:269-271 */
#define MCI_ERR_HISTOGRAM 5     /* "histogram should be all finite / positive"  variable.jl:212-213, common.jl:71,79 */
#define MCI_ERR_COMM 6          /* RCCL failure */
#define MCI_ERR_NO_DEVICE 7     /* no HIP device: the product path never falls back to the CPU */

enum { MCI_CONTINUOUS = 0, MCI_DISCRETE = 1, MCI_FERMIK = 2 }; /* Dist.Continuous variable.jl:87-99 / Dist.Discrete :272-284 / Dist.FermiK :1-20 */
enum { MCI_VEGAS = 0, MCI_VEGASMC = 1, MCI_MCMC = 2 }; /* solver=:vegas main.jl:256 / :vegasmc :253 / :mcmc :259 */
/* not a solver: names the persistent :vegas kernel (mci_set_persistent) for mci_compile_solver / mci_kernel_code_object */
enum { MCI_VEGAS_PERSISTENT = 3 };
/* ... and the chain solvers' kernels with several lanes per chain (mci_set_chain_speculation), code objects of their own */
enum { MCI_VEGASMC_LANES = 5, MCI_MCMC_LANES = 6 };

typedef struct mci_ctx mci_ctx;
typedef struct mci_problem mci_problem;

/* One leaf variable: a `Continuous(lower, upper; alpha, adapt, ninc, grid)` (variable.jl:137-153) or a
 * `Discrete(lower, upper; distribution, alpha, adapt)` (variable.jl:299-325).  Leaves that share `pool`
 * and number more than one form a `CompositeVar` (variable.jl:397-427). */
typedef struct {
    int32_t kind;       /* MCI_CONTINUOUS | MCI_DISCRETE */
    int32_t pool;       /* index into Configuration.var */
    double lower, upper;
    int32_t npoints;    /* continuous: grid points `ninc` (default 1000 => 999 increments); discrete: ignored */
    double alpha;       /* learning rate (default 2.0) */
    int32_t adapt;
    const double *init; /* optional: continuous grid[npoints] | discrete distribution[upper-lower+1]; NULL = default */
    /* MCI_FERMIK = `FermiK(dim, kF, dk, maxK)` (variable.jl:11-19; sampler.jl:109-281): lower = kF, upper = dk,
       npoints = dim (2 | 3), alpha = maxK.  A slot holds `dim` consecutive x entries; no adaptive map; solver = :mcmc only
       (the reference's own restriction, test/bubble_FermiK.jl:2). */
} mci_leaf_desc;

/* `Configuration(; var, dof, obs, ...)` (configuration.jl:105-194). */
typedef struct {
    int32_t nleaf;
    const mci_leaf_desc *leaves; /* leaves of one pool must be contiguous */
    int32_t npool;
    int32_t nintegrand;          /* N user integrands; the normalisation integrand (configuration.jl:153) is implicit */
    const int32_t *dof;          /* [nintegrand*npool] row-major: dof[i][vi] */
    const int32_t *obs_nbin;     /* [nintegrand] or NULL(=ncomp each): number of doubles observable i holds (`obs`
                                    kwarg; a ComplexF64 entry counts 2) */
    const int32_t *obs_bin_draw; /* [nintegrand] or NULL(=-1): flat draw index of the Discrete draw that selects the
                                    bin of observable i -- the `measure` of example/bubble.jl:81-84; -1 = default
                                    measure (vegas/montecarlo.jl:151-153) */
    /* `neighbor` kwarg (configuration.jl:201-227), used by solver=:mcmc: CSR lists of 0-based integrand indices,
       index nintegrand = the normalisation integrand.  neighbor_offsets[nintegrand+2]; both NULL = the default
       chain 1-2-...-N with the normalisation attached to the first integrand (configuration.jl:203-208). */
    const int32_t *neighbor_offsets;
    const int32_t *neighbor_list;
    int32_t ncomp;               /* `type` kwarg (configuration.jl:108): 0/1 = Float64 weights, 2 = ComplexF64 -- the
                                    integrand writes (re, im) pairs w[2i], w[2i+1]; abs() is the modulus; real and
                                    imaginary parts are separate statistics columns (main.jl:279,284,302-305) */
} mci_problem_desc;

/* `integrate` keyword arguments (main.jl:71-90). */
typedef struct {
    int32_t solver;      /* MCI_VEGAS | MCI_VEGASMC | MCI_MCMC */
    int64_t neval;       /* evaluations per iteration (summed over all ranks) */
    int32_t niter;
    int64_t block;       /* statistical blocks per iteration, all ranks (standardised like main.jl:220-234) */
    int32_t ignore;      /* iterations excluded from the final average; <0 = (adapt ? 1 : 0) */
    int32_t adapt;
    double gamma;        /* reweight learning rate (vegasmc, mcmc: doReweight!, main.jl:322-346) */
    int64_t measurefreq;
    uint64_t seed;
    int64_t nchain;      /* vegasmc, mcmc: independent chains per block (1 = the reference's single chain); 0 = auto */
    int32_t first_iteration; /* RNG stream offset so that a resumed run (config=res.config) draws fresh numbers */
    double thermal_ratio;    /* mcmc: burn-in steps = floor(steps * thermal_ratio), mcmc/montecarlo.jl:77,:133 (default 0.1) */
    const double *reweight_goal; /* [nintegrand+1] or NULL: main.jl:81, :334-337 */
} mci_integrate_args;

/* `Result` (statistics.jl:16-63).  All arrays are caller-allocated. */
typedef struct {
    int32_t niter, nobs;
    double *iter_mean; /* [niter*nobs] per-iteration mean  (main.jl:203) */
    double *iter_std;  /* [niter*nobs] per-iteration std */
    double *mean;      /* [nobs] inverse-variance weighted average (statistics.jl:186-220) */
    double *stdev;     /* [nobs] */
    double *chi2;      /* [nobs] reduced chi2 */
    int64_t neval;     /* evaluations actually performed, all iterations, this rank's view after reduction */
    double seconds;    /* wall time of the iteration loop (kernels + train + all-reduce) */
    double *visited;   /* [nintegrand+1] or NULL: config.visited of the last iteration (configuration.jl:46), read with the statistics */
    int32_t correlated; /* out: 1 = the iterations of this call continued each other's chains (mci_set_chain_carry), so `stdev` is the
                           block-lineage error (mci_lineage_sums) instead of statistics.jl:198, which assumes independent iterations;
                           `mean` and `chi2` are the reference's either way */
    int32_t warmup;     /* out: launches of this call that were run again instead of being counted (automatic :mcmc chain lengths:
                           their chains were too short for the holding times they measured, mci_mcmc_launch_valid) */
} mci_result;

/* ---- context: HIP device + stream (+ RCCL communicator); replaces MPI.Init, main.jl:113-114 ---- */
int mci_ctx_create(int32_t device, mci_ctx **out);
int mci_ctx_destroy(mci_ctx *ctx);
const char *mci_last_error(void);             /* thread-local message of the last failing call */
int mci_device_count(int32_t *count);
void *mci_ctx_stream(mci_ctx *ctx);           /* hipStream_t, for callers that order their own work after ours */

/* RCCL over xGMI: replaces MPIreduce/MPIbcast (utility/parallel.jl:25-99; call sites main.jl:177-188,
 * configuration.jl:264-321).  Rank 0 creates the 128-byte id and ships it to the other ranks by any
 * means it has (torch.distributed store, MPI.bcast, a file). */
int mci_comm_unique_id(void *id128);
int mci_comm_init(mci_ctx *ctx, int32_t rank, int32_t nranks, const void *id128);
int mci_comm_rank(const mci_ctx *ctx, int32_t *rank, int32_t *nranks);
/* v[0..n) <- sum over the ranks of the problem's communicator (MPI.Allreduce, utility/parallel.jl:25-60), through the library's
 * stream; identity without a communicator.  For the few host-side sums of a run (the lineage sums of mci_lineage_sums). */
int mci_comm_sum(mci_problem *prob, double *v, int32_t n);

/* ---- problem = Configuration + integrand ---- */
int mci_problem_create(mci_ctx *ctx, const mci_problem_desc *desc, mci_problem **out);
int mci_problem_destroy(mci_problem *prob);
/* The integrand closure (vegas/montecarlo.jl:140-144) as a HIP C++ function body:
 *     const double* x  -- flat draws in the reference's draw order (pool, slot, leaf), 0-based
 *     double*       w  -- nintegrand outputs
 *     const double* ud -- userdata (configuration.jl:113)
 * JIT-compiled with hiprtc for gfx950 together with the hand-written kernels. */
int mci_set_integrand_source(mci_problem *prob, const char *body, const double *userdata, int32_t nuserdata);
/* Slow path for closures that cannot be expressed as device source ("batch callback"): per launch the library
 * hands the host callback every draw of the batch, draw-major x[k*n + i] (so that x[k] is a contiguous vector over
 * the n samples -- what a vectorised `(x, c) -> ...` wants), the callback writes w[q*n + i] for the nw =
 * nintegrand*ncomp outputs and returns 0; the sample kernel then regenerates the same draws around those weights.
 * solver = MCI_VEGASMC (the reference's default, main.jl:72; the closure sits inside the Markov step, vegas_mc/updates.jl:67-75):
 * the chains of a launch advance in lock step, ONE kernel launch and ONE callback per step, n = the chains of the launch
 * (nblocks * nchain), x = the configurations they propose; same streams and arithmetic as the device-source chains, so both
 * give the same results.  solver = MCI_MCMC the same way (one launch and one callback per step; the library asks this form for
 * every integrand and keeps the one the chain needs -- mci_set_integrand_host_indexed is the reference's own :mcmc form).
 * The callback WRITES its weights into the array it is handed: at this boundary the integrand is always the reference's in-place form
 * `integrand(var, weights, config)` (inplace = true; main.jl:26, vegas/montecarlo.jl:140-141, vegas_mc/updates.jl:67-70); a binding
 * offers the returning form `integrand(var, config)` by copying what the closure returns into `w`, and picks between the two -- and the
 * :mcmc form -- by solver and `inplace` flag like the reference (main.jl:26-28), not by the closure's parameter count.
 * PCIe- and host-bound by construction. */
typedef int (*mci_host_integrand_fn)(const double *x, double *w, int64_t n, int32_t ndraw, int32_t nw, void *user);
int mci_set_integrand_host(mci_problem *prob, mci_host_integrand_fn fn, void *user);
/* The same for solver = MCI_MCMC, whose closure takes the integrand index first -- `integrand(idx, var, config)`
 * (mcmc/montecarlo.jl:34-36, mcmc/updates.jl:35-38) -- and is asked for ONE integrand per configuration: per Markov step the
 * callback gets idx[i] (0-based; -1: chain i needs no evaluation this step, leave w alone) next to x[k*n + i] and writes the
 * ncomp components w[q*n + i] of integrand idx[i].  One launch and one callback per step; a chain whose start configuration has
 * zero weight redraws it (mcmc/montecarlo.jl:118-124) and lags one step behind per retry.  Setting it replaces a plain host
 * integrand and vice versa; under MCI_VEGAS / MCI_VEGASMC the library calls it with every integrand in turn. */
typedef int (*mci_host_integrand_idx_fn)(const int32_t *idx, const double *x, double *w, int64_t n, int32_t ndraw, int32_t ncomp, void *user);
int mci_set_integrand_host_indexed(mci_problem *prob, mci_host_integrand_idx_fn fn, void *user);
/* The `measure` callback (vegas/montecarlo.jl:156-161, mcmc/montecarlo.jl:166-169) as a HIP C++ function body:
 *     const double* x, ud as above; const double* rw -- relative weights [nintegrand*ncomp];
 *     const int idx -- -1 (vegas, vegasmc) or the integrand an mcmc chain sits on (only its rw is non-zero);
 *     obs_add(k, v) -- accumulate v into flat observable k (0 <= k < sum obs_nbin)
 * NULL restores the default measure (vegas/montecarlo.jl:151-153) / the declarative obs_bin_draw one. */
int mci_set_measure_source(mci_problem *prob, const char *body);
/* Slow path for `measure` closures that must stay on the host (vegas/montecarlo.jl:156-161), the counterpart of
 * mci_set_integrand_host: after each launch the library calls the callback ONCE PER BLOCK with that block's n records --
 * draw-major configurations x[k*stride + i] and relative weights relw[q*stride + i] -- and the callback accumulates the block's
 * observables into obs[nobs] (zeroed; flat over the `obs` kwarg); they then go through the same block merge as device-side
 * observables.  What a record is:
 *   MCI_VEGAS    every MEASURED sample of the block (those with ne % measurefreq == 0, vegas/montecarlo.jl:148: n = neval_per_block /
 *                measurefreq records, the skipped samples are squeezed out before the call), relw = weights[q] *
 *                padding_probability * jac (:152);
 *   MCI_VEGASMC  every MEASURED step of every chain of the block (vegas_mc/montecarlo.jl:213-227: steps j * measurefreq past the
 *                burn-in), chain-major, relw = weights[q] * padding_probability / probability (:220);
 *   MCI_MCMC     the same for mcmc/montecarlo.jl:143-169: relw is zero except for the integrand the chain sits on (:162); a chain
 *                on the normalization integrand calls no measure (:157-159) and leaves an all-zero record.
 * The measure runs after the launch, not inside the step loop: a measure that only reads (var, weights) -- the reference's
 * contract -- cannot tell.  fn = NULL restores the device-side measure. */
typedef int (*mci_host_measure_fn)(const double *x, const double *relw, int64_t n, int64_t stride, int32_t ndraw, int32_t nw,
                                   int64_t block, double *obs, int32_t nobs, void *user);
int mci_set_measure_host(mci_problem *prob, mci_host_measure_fn fn, void *user);
/* The `:mcmc` form `measure(idx, var, obs, relative_weight, config)` (mcmc/montecarlo.jl:166-169): idx[i] = the integrand record i
 * belongs to (0-based; -1: no measure call for this record, skip it), relw[q*stride + i] the ncomp components of ITS relative
 * weight.  Under MCI_VEGAS / MCI_VEGASMC the library calls it with every integrand in turn (idx constant).  Setting it replaces
 * a plain host measure and vice versa. */
typedef int (*mci_host_measure_idx_fn)(const int32_t *idx, const double *x, const double *relw, int64_t n, int64_t stride, int32_t ndraw,
                                       int32_t ncomp, int64_t block, double *obs, int32_t nobs, void *user);
int mci_set_measure_host_indexed(mci_problem *prob, mci_host_measure_idx_fn fn, void *user);
int mci_compile(mci_problem *prob);          /* JIT or kernel-cache load of the vegas kernel; implicit on first run */
int mci_compile_solver(mci_problem *prob, int32_t solver); /* same for one solver's kernel (one code object each) */
/* path of the kernel-cache file (gfx950 code object) the solver's kernel was loaded from -- the analogue of asking Julia
 * for `@code_native` of the specialised `montecarlo` method (vegas/montecarlo.jl:72-75); diagnostics read its ISA from it */
int mci_kernel_code_object(mci_problem *prob, int32_t solver, char *buf, int32_t n);
int mci_set_launch(mci_problem *prob, int32_t threads_per_workgroup, int32_t workgroups_per_block);
int mci_problem_info(const mci_problem *prob, int32_t *ndraw, int32_t *nobs, int64_t *packed_size,
                     int32_t *table_mode, int64_t *lds_bytes);
/* interleaved copies of the LDS histograms the :vegas sample kernel keeps (1 = the plain layout): the placement rule's choice
 * before the kernel is compiled, what the compiled kernel uses afterwards (DESIGN.md "Histogram copies"); diagnostics and
 * bench.py's roofline price the kernel's ds_add_f64 with the matching access pattern */
int mci_get_histogram_copies(const mci_problem *prob, int32_t *copies);

/* ---- one iteration, step by step (what mci_integrate runs; also the testing seam) ---- */
/* blocks [block_lo, block_hi) of `_block!` (main.jl:236-292) on this GPU; leaves the local packed buffer
 * [obsSum(nobs) | obsSqSum(nobs) | normalization | neval | visited(N+1) | histograms | propose | accept] on the device */
int mci_iteration_run(mci_problem *prob, int32_t solver, int64_t neval_per_block, int64_t block_lo,
                      int64_t block_hi, int32_t iteration, uint64_t seed, int64_t measurefreq, int64_t nchain,
                      double thermal_ratio);
/* sum the packed buffer over ranks: MPIreduceConfig!+MPIbcastConfig! (configuration.jl:264-321) as ONE
 * ncclAllReduce; no-op without a communicator */
int mci_iteration_reduce(mci_problem *prob);
/* doReweight! (main.jl:322-346, vegasmc only), train! + clearStatistics (main.jl:190-199), and the
 * iteration's (mean, std) = _mean_std (main.jl:296-320).  mean/std may be NULL. */
int mci_iteration_finish(mci_problem *prob, int32_t solver, int64_t block_total, int32_t adapt, double gamma,
                         double *mean, double *std);
/* wait for the stream and raise what the device flagged since the last check: the reference's error() / @assert of
 * main.jl:269-271 (block normalization), variable.jl:212-213 and common.jl:71,79 (histogram / distribution not finite or
 * not positive), mcmc/montecarlo.jl:125-126 (no non-zero start found) */
int mci_check_status(mci_problem *prob);
/* the whole loop (main.jl:142-218) */
int mci_integrate(mci_problem *prob, const mci_integrate_args *args, mci_result *result);

/* ---- state access: res.config.var[i].grid etc. (docs/src/index.md:129) and external reducers ---- */
/* :mcmc diagnostic of the last launch (summed over the ranks when a communicator is attached: every rank sizes its chains from
 * the same histogram): out64[b] = number of chains whose longest holding time h (steps during which a live slot, or the
 * integrand index, did not change; holds still running at the end count) has bit_width(h) == b.  An automatic chain length follows
 * 2^(top occupied b) of the launch before it (mci_mcmc_auto_chains): the host waits for that launch's sample kernel -- not for its
 * merge and train! -- before it sizes the next one; a fixed lag, so a run stays reproducible. */
int mci_get_hold_histogram(mci_problem *prob, uint64_t *out64);
/* Was the last automatic :mcmc launch long enough for the holding times it measured itself (chain length >= 16 x the longest hold, 4 x
 * for carried chains)?  Waits for that launch's sample kernel.  *warm: some launch of this problem has been.  Until then the chain
 * lengths escalate and mci_integrate runs an iteration again instead of counting it (mci_result.warmup); a caller that drives
 * mci_iteration_run itself does the same with this call (run the iteration again as iteration + 16384 * attempt: the chains go on,
 * the Philox streams are new).  After the first valid launch nothing is repeated: no selection on what an iteration measured. */
int mci_mcmc_launch_valid(mci_problem *prob, int32_t *valid, int32_t *warm, int64_t *chain_len, int64_t *hold_max);
/* take the last finished iteration out of the iteration log and the block log again (it stays in the map, the reweight factors and the
 * carried chains): what a warm-up launch that is run again amounts to */
int mci_iteration_discard(mci_problem *prob);
/* the statistics head [obsSum|obsSqSum|normalization|neval|visited] of the last `nrows` finished
 * iterations (oldest first), nstat = 2*nobs+2+N+1 doubles per row: the per-iteration history that
 * `Result.iterations` is built from (statistics.jl:24-33), kept on the device until asked for */
int mci_get_iteration_log(mci_problem *prob, int32_t nrows, double *out);
/* Chain solvers: the block means m_b = observable_b / normalization_b (main.jl:275-280) of the last `rows` iterations (oldest first),
 * out[rows][nblocks][nobs] with nblocks = this rank's blocks; *carried = how many of the logged iterations continued the chains of the
 * one before.  The log starts over with mci_reset_block_log, with every mci_integrate call, and when the block range changes. */
int mci_get_block_means(mci_problem *prob, int32_t rows, double *out, int64_t *nblocks, int32_t *carried);
int mci_reset_block_log(mci_problem *prob);
/* make room for `rows` more iterations in that log now (it grows on demand otherwise, synchronising the stream when it does) */
int mci_reserve_iteration_log(mci_problem *prob, int32_t rows);
int mci_get_packed(mci_problem *prob, double *out, int64_t n);
int mci_set_packed(mci_problem *prob, const double *in, int64_t n);
void *mci_packed_device_ptr(mci_problem *prob);
int mci_get_grid(mci_problem *prob, int32_t leaf, double *out, int32_t n);
int mci_set_grid(mci_problem *prob, int32_t leaf, const double *grid, int32_t n);
int mci_get_distribution(mci_problem *prob, int32_t leaf, double *distribution, double *accumulation, int32_t k);
int mci_set_distribution(mci_problem *prob, int32_t leaf, const double *distribution, int32_t k);
int mci_get_reweight(mci_problem *prob, double *out, int32_t n);
int mci_set_reweight(mci_problem *prob, const double *in, int32_t n);
int mci_set_reweight_goal(mci_problem *prob, const double *goal, int32_t n); /* main.jl:81; NULL/0 clears */
/* config.propose / config.accept of the last iteration (configuration.jl:185-186; the numbers behind report(config),
 * configuration.jl:345-464), n = 3 * (N+1) * max(N+1, npool) entries each, row-major [update][integrand][target], 0-based:
 * changeIntegrand [0][curr][new] (mcmc/updates.jl:48,50), changeVariable [1][curr][vi] (mcmc/updates.jl:100,102; vegasmc
 * [1][0][vi], vegas_mc/updates.jl:90,92), swapVariable [2][curr][vi] (mcmc/updates.jl:138,140); clearStatistics! offsets
 * included (1e-8 / 1e-10 per block config, configuration.jl:247-248).  They are the tail of the packed buffer, so after
 * mci_iteration_reduce they hold the sum over all ranks, like MPIreduceConfig! (configuration.jl:297-298). */
int mci_get_acceptance(mci_problem *prob, double *propose, double *accept, int32_t n);
/* resume across processes: what train!/doReweight! have learned -- grids, distributions, reweight --
 * as a small self-describing binary file ("MCISTATE", version 1).  The reference keeps this state only in memory
 * (`config = res.config`, docs/src/index.md:129). */
int mci_save_state(mci_problem *prob, const char *path);
int mci_load_state(mci_problem *prob, const char *path);
/* Opt-in cheaper uniform stream of solver = :vegas: bits = 52 (default) draws every uniform with 52 random mantissa bits, the resolution
 * of Julia's rand(Float64) (sampler.jl:296), two draws per Philox4x32-10 block; bits = 32 uses one 32-bit word per draw (the top 32
 * mantissa bits, y on a 2^-32 lattice), four draws per block -- half the generator work per sample.  Same counter scheme otherwise
 * (draw k -> block k >> 2, word k & 3); the chain solvers are not affected.  Mirrored in the oracle (mcio_set_rng_bits). */
int mci_set_rng_bits(mci_problem *prob, int32_t bits);
/* Opt-in cheaper generator for EVERY stream of the problem (all three solvers): rounds = 10 (default) is Philox4x32-10, the
 * Random123 default; rounds = 7 is Philox4x32-7, the fewest rounds its authors found to pass BigCrush ("crush-resistant", Salmon et
 * al., SC'11) -- 30 % less generator work, a different (equally valid) stream.  Same keys and counters; pinned on the Random123
 * known-answer vectors for 7 rounds (tests/golden) and mirrored in the oracle (mcio_set_rng_rounds). */
int mci_set_rng_rounds(mci_problem *prob, int32_t rounds);
/* How train!(Continuous) walks the smoothed histogram to place the new grid points (variable.jl:227-234):
 *   1  the reference's serial recurrence -- the same additions and subtractions on the same operands in the same order, starting from
 *      sums in ONE fixed association of the family Julia's @simd sum() belongs to (csrc/mci_train.h sum16; the reference itself has no
 *      single order across CPUs): its decisions (does this bin yield a new grid point) are taken from
 *      the prefix-scan form, one lane walks the additions and subtractions alone, every decision is checked against the exact record
 *      (+14 us per iteration at ninc = 1000 on MI355X); a decision that does not hold sends the walk through mode 2;
 *   2  the recurrence with its compares and branches on one lane in hand-written ISA (+38 us): what 1 falls back to -- bit-identical results;
 *   0  the same walk as a fixed-order prefix scan + one bisection per grid point (agrees with the recurrence to 1e-12 of
 *      the variable's range per train! step, i.e. whole runs agree to ~1e-4 instead of ~1e-6);
 *  -1  automatic (default): 1 when the iteration's sample launch on this rank is >= 2^26 samples (the walk then costs ~1 % or less), else 0.
 * (No environment variable stands behind it: the library reads MCI_KERNEL_CACHE and MCI_JIT_FLAGS and nothing else.) */
int mci_set_train_walk(mci_problem *prob, int32_t mode);
/* Deterministic mode: with on = 1 a fixed seed gives BIT-IDENTICAL results run to run (histograms, grids, every iteration's mean
 * and error), like the reference's sequential loop under `MersenneTwister(seed)` (configuration.jl:190, vegas/montecarlo.jl:117-187).
 * By default the order in which the waves of a workgroup add to its LDS histogram follows the hardware's wave schedule, results
 * agree run to run at rounding level only, and train! carries those last bits on.  In this mode every solver's kernel keeps one
 * copy of the workgroup's LDS histograms and observables PER WAVE (the largest of 512 / 256 / 128 / 64 threads whose copies fit a
 * CU's LDS); a wave's adds are in program order and the copies are summed in a fixed order, as all cross-workgroup merges already
 * are.  Costs the bank-conflict relief of the interleaved copies (headline configuration: see DESIGN.md); layouts whose histograms
 * need several LDS tiles (more than ~9 independent 999-bin grids) are refused.  Results still depend on the launch geometry
 * (mci_set_launch) and on the rank count, as the summation order does. */
int mci_set_deterministic(mci_problem *prob, int32_t on);
/* Carried chains (this engine's many-chain decomposition only; nchain = 1, the reference's chain, always starts afresh like
 * montecarlo.jl:151-153 / mcmc/montecarlo.jl:118-124 do at every block): with mode -1 (default) or 1 a chain-solver launch that is the
 * NEXT iteration of the same solver over the same blocks continues the previous launch's chains instead of drawing new starts and
 * burning them in.  :vegasmc: chain (block, ch) starts from the configuration chain (block, ch mod previous nchain) ended with, its
 * bins and probabilities looked up again on the refined map.  :mcmc: a chain's state is (integrand index, configuration) and
 * doReweight! moves the weight of every index between iterations (main.jl:322-346), so the stored chains of a block -- a sample of
 * the finished iteration's target -- are resampled (systematic, deterministic) with probability ~ reweight_new[index] /
 * reweight_old[index] into a sample of the new one.  Such a launch keeps the reference's `ne >= neval/100` (vegas_mc/montecarlo.jl:213)
 * under :vegasmc and burns nothing in under :mcmc (floor(steps * thermal_ratio), mcmc/montecarlo.jl:133, is the burn-in of a chain
 * that starts somewhere; a continued chain measures from its first step); automatic chain counts are then sized for the duplicates
 * of a stored chain to part before they are copied again (:vegasmc two burn-in floors, :mcmc 4 x the longest measured holding time)
 * instead of for start-up bias (DESIGN.md "Chains").  mode 0: every launch starts its chains afresh.
 * Consecutive iterations of carried chains are correlated, which the reference's combination of iterations (statistics.jl:186-220)
 * does not expect: mci_integrate reports the block-lineage error for such runs (mci_result.correlated, mci_lineage_sums).
 * Mirrored in the oracle (mcio_set_chain_carry, mcio_resample_chains). */
int mci_set_chain_carry(mci_problem *prob, int32_t mode);
/* chains per block of the last chain-solver launch and whether it continued the launch before it */
int mci_last_chain_launch(const mci_problem *prob, int64_t *nchain, int32_t *carried);
/* For a caller that runs the iteration loop itself (mci_iteration_run / _reduce / _finish; mci_integrate does this on its own): does
 * the estimate of the launches that follow enter the final average (iteration >= ignore, main.jl:82, :211)?  The first iteration of
 * the default call does not, and its automatic :vegasmc chains stay short; a launch that counts on a map train! has never refined
 * (adapt = false, or ignore = 0) runs its fresh chains 8 x as long, because chains of the usual length have not reached their target
 * on the untrained map (DESIGN.md "Chains", profiles/r05_bias.txt A5 / A6).  Default 0. */
int mci_set_iteration_counted(mci_problem *prob, int32_t counted);
/* Several lanes per chain.  The reference's chain is one sequential loop on one core (vegas_mc/montecarlo.jl:184-232,
 * mcmc/montecarlo.jl:134-172); a chain-solver launch with few chains -- the reference's default call runs 16, one per block -- would
 * leave all but a few lanes of the GPU idle.  Such a launch gives every chain a GROUP of lanes (a power of two up to 64) that steps it
 * speculatively: the uniforms of a step are addressed by (chain, step), so the lanes evaluate the proposals of the next steps along a
 * tree of accept / reject outcomes at once, a ballot of the accept tests picks the way the chain actually takes, the lanes on it do
 * their steps' bookkeeping (propose / accept counters, histogram, measurement) and the last of them hands its configuration to the
 * group (csrc/mci_spec.h).  The chain is the SAME chain -- same law, same uniforms, same arithmetic per step: sums differ by
 * reassociation only, and nchain = 1 stays the reference's chain.
 * lanes: -1 (default) automatic -- the largest group that keeps the launch within one wave per SIMD; 1: one lane per chain always;
 * 2 .. 64: that group size.  accept in (0, 1): ONE tree, built for that acceptance -- the group's lanes are the `lanes` most probable
 * nodes of the outcome tree of a chain whose steps change its configuration with that probability (-> 0: the reject chain, up to
 * `lanes` steps per trip through a run of rejections; 1/2: the complete binary tree, log2(lanes) steps per trip whatever happens);
 * <= 0 (default): a family of trees (:vegasmc 0.03 .. 0.93, :mcmc 0.03 .. 0.8) -- every group measures, every eight trips, the fraction
 * of its chain's steps that changed the configuration and moves to the tree built for the nearest acceptance, so a chain in a sticky
 * state runs down the reject chain and a chain on a well-adapted map down the accept edges.  max_accepts: the most accept edges on a
 * way through a tree (:mcmc builds its proposals once per accept level); < 0: the solver's default (:vegasmc 12, :mcmc 2, 3 on its
 * trees for chains that accept most steps).
 * Launches with a host integrand and the deterministic mode keep one lane per chain. */
int mci_set_chain_speculation(mci_problem *prob, int32_t lanes, double accept, int32_t max_accepts);
/* Warm-up of the automatic :mcmc chain length (DESIGN.md "Chains"): launches the last mci_integrate call ran AGAIN with longer chains
 * instead of counting (mci_result.warmup) have trained the map, moved the reweight factors and advanced the chains, but their
 * evaluations are in neither mci_result.neval nor the estimate: this is how many there were, on this rank.  A launch is accepted on
 * the holding times IT measured (it is long enough for its own holds), and from the first accepted launch on nothing is repeated or
 * left out again -- the selection acts on the warm-up only, never on a counted iteration's value. */
int mci_last_integrate_discarded(const mci_problem *prob, int64_t *neval, int32_t *launches);
/* A several-lanes-per-chain code object proves itself before it is trusted: the first launch through one that has never run on a device
 * (a fresh compile, a pre-filled cache entry) is preceded by <= 2 blocks x <= 512 steps through it AND through the lane-per-chain kernel
 * of the same problem -- both step the reference's chain (vegas_mc/montecarlo.jl:198-211, mcmc/montecarlo.jl:134-172) on the same streams --
 * and their packed buffers are compared (statistics 1e-9, histograms and propose / accept tables 1e-8).  status: 0 nothing launched yet,
 * 1 verified (now, or earlier: a marker file next to the code object in the kernel cache), -1 the check FAILED -- a miscompiled code
 * object; one warning on stderr, and this problem keeps one lane per chain --, -2 the unit did not compile (automatic lanes: one lane per
 * chain; forced lanes: MCI_ERR_COMPILE).  solver: MCI_VEGASMC | MCI_MCMC. */
int mci_chain_speculation_status(const mci_problem *prob, int32_t solver, int32_t *status);
/* lanes per chain of the last chain-solver launch (1: one lane per chain) and the accept levels of its tree */
int mci_last_chain_speculation(const mci_problem *prob, int32_t *lanes, int32_t *max_accepts);
/* the tree mci_set_chain_speculation(lanes, accept, max_accepts) stands for, node by node ([lanes] each; NULL: not wanted): the step
 * offset of the node's proposal, the nearest ancestor it hangs below by an accept edge (-1: none), its number of accept edges, and
 * the lanes that must have accepted / rejected for the node to be on the chain's path (bit = lane).  Lanes are numbered
 * ancestors-first.  No device needed. */
int mci_speculation_tree(int32_t lanes, double accept, int32_t max_accepts, int32_t *depth, int32_t *anc, int32_t *nacc, uint64_t *needacc,
                         uint64_t *needrej);
/* JIT or kernel-cache load of a chain solver's several-lanes-per-chain kernel (its own code object; implicit on first use) */
int mci_compile_chain_speculation(mci_problem *prob, int32_t solver);
/* Persistent :vegas iterations.  The reference's loop (main.jl:142-207) at the reference's own default size (neval = 1e4, main.jl:76)
 * is launch-bound on a GPU: a microsecond of sampling per iteration behind two dependent kernel launches.  With mode -1 (default) a
 * single-rank mci_integrate call of solver MCI_VEGAS at measurefreq == 1 over ONE Continuous variable type whose iterations are that
 * small (samples x draws < 2^19, at most 7 draws per sample) runs ALL its iterations as one launch of at most 128 co-resident sampling workgroups + one statistics
 * workgroup (csrc/mci_train.h vegas_persist): sample -> histograms merged with global atomics -> one grid-wide wait -> every sampling
 * workgroup runs train! on its OWN copy of the map (same arithmetic on the same numbers: the copies stay bit-identical) while the
 * statistics workgroup merges the blocks -> next iteration.  Same Philox streams and the same arithmetic as the launch chain (sums
 * differ by reassociation only).  Needs tables and histograms in LDS (table mode 0), device-source integrand and measure, the
 * prefix-scan walk, no forced launch geometry and no kernel timing; anything else, mode 0, and every call through mci_iteration_run
 * take the launch chain.  mode 1: every call the layout allows, whatever its size, and the kernel is compiled on the spot (mode -1
 * compiles its larger translation unit on a thread of its own once the process has made 256 such calls, and takes the launch chain
 * until the code object is there -- in the kernel cache, where every later process finds it).  A grid-wide wait that runs out of time
 * (2 s: the workgroups were not all resident, a device shared with other long-running kernels) does not lose the call: mci_integrate
 * runs the same iterations again through the launch chain (the map is only written back after the last turn), and later calls take
 * the launch chain. */
int mci_set_persistent(mci_problem *prob, int32_t mode);
/* whether the last mci_integrate ran as one persistent launch */
int mci_last_integrate_persistent(const mci_problem *prob, int32_t *persistent);
/* Dist.train! on the histograms currently in the packed buffer (variable.jl:206-239, :369-382) */
int mci_train(mci_problem *prob);
/* the adaptive map alone (sampler.jl:293-305, :13-22) + integrand: first `n` samples of block `block_index`
 * of `iteration`, written to host arrays x[n*ndraw], jac[n], w[n*nintegrand] */
int mci_sample_dump(mci_problem *prob, int32_t iteration, uint64_t seed, int64_t neval_per_block,
                    int64_t block_index, int64_t n, double *x, double *jac, double *w);
/* HIP-event durations (ms, oldest first) of the last `n` sampling-kernel launches, recorded on the
 * library's stream around every launch (ring of 512), and the last launch geometry */
int mci_kernel_times_ms(mci_problem *prob, float *ms, int32_t n, int32_t *got, int32_t *workgroups, int32_t *threads);
/* The events cost ~5.5 us of idle queue each -- a third of a launch-bound iteration (neval = 1e4), nothing next to millions of
 * samples: mode -1 (default) records them for launches of >= 2^20 samples, 0 never, 1 always (mci_kernel_times_ms returns the
 * recorded launches only). */
int mci_set_kernel_timing(mci_problem *prob, int32_t mode);
/* the shader clock (MHz) the sample loops of the last n timed MCI_VEGAS launches ran at, oldest first: the first wave of workgroup 0
 * reads s_memtime (shader cycles) and s_memrealtime (the constant-rate reference, hipDeviceAttributeWallClockRate) around its loop.
 * What the roofline of bench.py prices data-sheet issue cycles with: the chip clocks to its power budget, not to its 2.4 GHz peak. */
int mci_kernel_clocks(mci_problem *prob, double *mhz, int32_t n, int32_t *got);
/* HIP-event durations (ms, oldest first, ring of 64) of the per-iteration ncclAllReduce of mci_iteration_reduce on this rank,
 * recorded under the same rule: the time a rank spends in the one exchange step of the path (main.jl:177-188) -- its own wait for
 * the slowest rank's sample pass plus the latency of an all-reduce of `packed_size` doubles */
int mci_comm_times_ms(mci_problem *prob, float *ms, int32_t n, int32_t *got);
/* ncclAllReduce calls the library has issued on this context so far and the element count of the last one.  An iteration of ANY
 * solver is ONE collective (mci_iteration_reduce): [statistics | histograms | propose | accept] and, behind an :mcmc launch that
 * measured its holding times for the automatic chain length, the 64 counts of their histogram as exact doubles (the reference
 * reduces its statistics, histograms and tables one array at a time, configuration.jl:264-299).  A run of carried chains ends with
 * one more small one (mci_comm_sum: the block-lineage sums). */
int mci_comm_collectives(const mci_ctx *ctx, int64_t *calls, int64_t *last_count);
/* For an EXTERNAL reducer of the packed buffer (what comm.py's TorchDistComm is): the number of doubles to sum over the ranks --
 * mci_problem_info's packed_size + the 64 holding-time counts behind it; mci_get_packed / mci_set_packed take either size -- and
 * the call that tells the library the sum has happened (device buffer: after the collective was queued on the library's stream;
 * host path: after mci_set_packed), so that every rank sizes its next :mcmc chains from the summed counts. */
int mci_reduce_size(const mci_problem *prob, int64_t *n);
int mci_external_reduce_done(mci_problem *prob);

/* ---- host-side statistics of the path (pure functions, no GPU needed) ---- */
void mci_standardize_block(int64_t neval, int64_t nblock, int64_t nworker, int64_t *nevalperblock,
                           int64_t *block);                                          /* main.jl:220-234 */
/* first measured step of a VegasMC chain of `steps` steps: the reference's `ne >= neval/100`
 * (vegas_mc/montecarlo.jl:213) for its single chain (nchain = 1); with nchain > 1 independent chains per block
 * additionally >= min(steps/2, 64*nslots) so that each short chain forgets its start (nslots = sum of maxdof) */
double mci_chain_burnin(int64_t steps, int64_t nchain, int32_t nslots);
/* burn-in steps an MCMC chain runs before its `steps` measured ones: floor(steps*thermal_ratio)
 * (mcmc/montecarlo.jl:133); with nchain > 1 at least 64*nslots + 16*(npool+1)*nd */
int64_t mci_mcmc_burnin(int64_t steps, int64_t nchain, int32_t nslots, int32_t nd, int32_t npool, double thermal_ratio);
/* chains per block of an :mcmc launch with nchain = 0 ("automatic"; the reference has no counterpart: it runs one chain
 * per block).  hold_max = the longest holding time the launch before measured (mci_get_hold_histogram), hold_len = the chain length
 * (measured steps) of that launch (0: no growth cap), carried = the launch continues that launch's chains.  hold_max = 0 (nothing
 * measured yet): chains of 4096 steps or 2 burn-in floors; otherwise 16*hold_max (fresh) / 4*hold_max (carried) but at most 2*hold_len
 * -- a hold longer than a quarter of the chain that measured it is censored by that chain, so the length escalates from launch to
 * launch until the holds fit -- and never fewer than 8 / 1 burn-in floors; capped so that one GPU gets at most 131072 chains */
int64_t mci_mcmc_auto_chains(int64_t nevalperblock, int64_t nblocks, int32_t nslots, int32_t nd, int32_t npool,
                             int64_t hold_max, int64_t hold_len, int32_t carried);
void mci_maxdof(const int32_t *dof, int32_t nd, int32_t npool, int32_t *out);        /* configuration.jl:229-236 */
void mci_mean_std(const double *obs_sum, const double *obs_sq, int64_t n, int64_t block, double *mean,
                  double *std);                                                      /* main.jl:296-320 */
void mci_average(const double *iter_mean, const double *iter_std, int64_t stride, int64_t init, int64_t max,
                 double *mean, double *err, double *chi2);                           /* statistics.jl:186-220 */
/* Error of the weighted average when consecutive iterations are correlated (carried chains) but the blocks are independent: block b's
 * lineage is its own weighted average over the iterations, m_b = sum_i w_i * block_means[i][b] with the weights of statistics.jl:197,
 * :217 (w_i ~ 1/(iter_std[i] + 1e-10)^2, normalised); sum[o] = sum_b m_b, sumsq[o] = sum_b m_b^2 over THIS rank's blocks -- summed
 * over the ranks they go through mci_mean_std (the reference's own _mean_std, main.jl:296-320, over lineages instead of blocks):
 * the same mean as mci_average and the scatter of the lineages as its error.  block_means[niter][nblocks][nobs], iter_std[niter][nobs];
 * init / max 1-based like mci_average. */
void mci_lineage_sums(const double *block_means, int64_t niter, int64_t nblocks, int64_t nobs, const double *iter_std,
                      int64_t init, int64_t max, double *sum, double *sumsq);
void mci_do_reweight(double *reweight, const double *visited, int64_t nd, double gamma,
                     const double *goal);                                            /* main.jl:322-346 */
const char *mci_version(void);
/* the number in front of the dot: it changes whenever a struct of this header changes its layout (4: mci_result grew `correlated`
 * and `warmup`; 5: entry points only).  A caller built against another header must not pass structs. */
int32_t mci_abi_version(void);
#define MCI_ABI_VERSION 5

#ifdef __cplusplus
}
#endif
#endif /* MCI_H */
