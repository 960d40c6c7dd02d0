// mci_api.hip -- host core of libmci_hip.so: the C ABI of include/mci.h.
//
// Owns: the Configuration analogue (src/configuration.jl:105-194), the device-resident state (grids,
// distributions, histograms, packed statistics), the per-iteration launch chain
//     sample batch (JIT, mci_device.h) -> merge (k_hist_stage1, k_finalize) -> RCCL all-reduce -> k_train
// and the iteration loop + Result statistics (src/main.jl:142-218, :296-320, src/statistics.jl:186-220).
// There is NO CPU fallback: without a HIP device every compute entry point fails with MCI_ERR_NO_DEVICE.
#include "../../include/mci.h"
#include "mci_debug.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <link.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "mci_device.h" // BatchArgs / DumpArgs (the templates themselves are instantiated by the JIT)
#include "mci_jit.h"
#include "mci_static_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(x)                                                                                           \
    do {                                                                                                    \
        hipError_t e_ = (x);                                                                                \
        if (e_ != hipSuccess) return fail(MCI_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- RCCL, loaded lazily so that single-GPU use never touches it -----------------------------------
struct Id128 { char b[128]; }; // ncclUniqueId (rccl.h:43)
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128 /* by value */, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
const int kNcclFloat64 = 8, kNcclSum = 0; // ncclDouble, ncclSum (rccl.h)

// If the host process already carries an RCCL (PyTorch-ROCm bundles its own and resolves it through its rpath),
// bind to THAT copy: two RCCL instances in one process would each open their own IPC/proxy state on the same GPUs.
int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *out) {
    const char *n = info->dlpi_name;
    if (n && strstr(n, "librccl.so")) {
        *(std::string *)out = n;
        return 1;
    }
    return 0;
}

int rccl_load() {
    if (g_rccl.h) return MCI_OK;
    void *h = nullptr;
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    if (!loaded.empty()) h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(MCI_ERR_COMM, "cannot load librccl.so: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void **, int, Id128, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(MCI_ERR_COMM, "librccl.so lacks the nccl* entry points");
    g_rccl.h = h;
    return MCI_OK;
}

} // namespace

struct mci_ctx {
    int device = -1;
    bool offline = false; // compile-only context (no GPU): lets build() pre-fill the kernel cache
    hipStream_t stream = nullptr;
    void *comm = nullptr;
    int rank = 0, nranks = 1;
    long long collectives = 0, last_count = 0; // ncclAllReduce calls issued on this context so far | elements of the last one (mci_comm_collectives)
};

namespace {
// train! stages one leaf in LDS: train_lds_doubles(nbin) + nbin doubles in k_finish (~4.5 per bin; + the serial walk's slots where they fit) next to ~2 KiB of static LDS
// -> the largest grid one workgroup can refine
const int64_t kTrainLdsMax = 160 * 1024 - 4096;
const int kMaxLeafBins = 4400;
struct Leaf {
    int kind, pool, npts, nbin, adapt, eoff, doff, boff;
    double lower, upper, alpha;
    int width = 1; // x entries per slot: D for a FermiK leaf
};
} // namespace

struct mci_problem {
    mci_ctx *ctx = nullptr;
    std::vector<Leaf> leaves;
    int npool = 0, ni = 0;
    std::vector<int> dof, maxdof, pool_leaf0, pool_nleaf;
    mcijit::ProblemShape shape;
    int nstat = 0;
    int64_t packed_n = 0;
    int64_t lds_bytes = 0;
    int64_t lds_bytes_k1 = 0; // split-all sample pass: fixed part + edge cache
    // host mirrors of the tables (uploaded at create / set_*)
    std::vector<double> h_edges, h_dacc, h_ddist, h_reweight, h_ud;
    // device
    double *d_edges = nullptr, *d_dacc = nullptr, *d_ddist = nullptr, *d_reweight = nullptr, *d_ud = nullptr;
    double *d_part_cols = nullptr, *d_part_hist = nullptr, *d_ghist = nullptr, *d_stage1 = nullptr, *d_packed = nullptr;
    double *d_scratch = nullptr, *d_iterlog = nullptr, *d_dump = nullptr;
    int *d_status = nullptr;
    mci::LeafDev *d_leaves = nullptr;
    int64_t cap_wg = 0, cap_blocks = 0, cap_iter = 0, cap_dump = 0;
    // kernels
    // one code object per solver, JIT-compiled (or loaded from the kernel cache) the first time the solver runs;
    // the vegas module also holds the sample-dump kernel
    // kernel slots (kslot): :vegas for measurefreq == 1 | :vegasmc | :mcmc | :vegas for any measurefreq | sample dump
    //                      | :vegasmc with several lanes per chain | :mcmc with several lanes per chain (mci_spec.h)
    static const int kSlots = 7;
    hipModule_t module[kSlots] = {};
    hipFunction_t f_solver[kSlots] = {}, f_dump = nullptr;
    bool compiled[kSlots] = {};
    std::string code_object[kSlots]; // kernel-cache file each slot's code object was loaded from / written to
    // Several lanes per chain (mci_spec.h, mci_set_chain_speculation): lanes -1 automatic (as many as the launch's chains leave idle),
    // 1 never, 2..64 forced; the acceptance the speculation tree is built for (<= 0: the solver's default) and the most accept edges
    // on a way through it (-1: the solver's default); the tree of the last such launch on the device
    int spec_lanes = -1, spec_maxacc = -1;
    double spec_accept = 0.0;
    mci::SpecNode *d_spec_tab = nullptr;
    int spec_tab_lanes = 0, spec_tab_limit = -2, spec_tab_maxacc = 0;
    int spec_ntree = 0, spec_first = 0; // trees on the device, the one a group starts on
    float spec_accepts[8] = {};          // the acceptance each of them was built for
    double spec_tab_accept = -1.0;
    int last_spec_lanes = 1, last_spec_maxacc = 0; // of the last chain launch (1: one lane per chain)
    int64_t last_discarded_neval = 0;              // evaluations of the warm-up launches the last mci_integrate ran again instead of counting
    int32_t last_discarded_launches = 0;
    static const int64_t kSpecFill = 65536;        // lanes a launch of few chains spreads over: one wave on each of the 1024 SIMDs
    bool vegas_planned = false, vegas_keys = false; // the :vegas plan (workgroup size, histogram copies, VGPR round keys) stands for both variants
    std::vector<double> h_goal; // reweight_goal (main.jl:81); empty = none
    double *d_goal = nullptr;
    int npa = 0;                    // 3 * (ni+1) * max(ni+1, npool): entries of config.propose (configuration.jl:185)
    double *d_part_pa = nullptr;    // [rows][2*npa] per-workgroup propose | accept tables of the chain solvers
    int64_t cap_pa = 0;
    unsigned long long *d_hold = nullptr; // [64] :mcmc holding-time histogram of the last launch (this rank), see mci_get_hold_histogram
    int64_t hold_max = 0;                 // upper edge of its top occupied bucket; 0: no :mcmc launch seen yet
    // split vegas pass (NTILE > 1): per-sample histogram weights and 16-bit bins of the tiles >= 1
    double *d_tile_w = nullptr;
    uint32_t *d_tile_bins = nullptr;
    int64_t cap_tile = 0;
    int ntdraw = 0; // draws whose histogram lives in a tile >= 1
    hipFunction_t f_tiles[2] = {nullptr, nullptr}; // replay kernel of the two :vegas variants
    // second merge stage (partials -> packed), launched lazily: a single-rank mci_iteration_finish fuses it with
    // the refinement (k_finish); anything else that looks at `packed` first flushes it (k_finalize)
    mci::MergeArgs merge{};
    bool merge_pending = false;
    bool has_fermik = false; // FermiK variables: solver = :mcmc only
    // host integrand ("batch callback"): draws dumped SoA -> callback -> weights uploaded -> accumulate kernel
    mci_host_integrand_fn host_fn = nullptr;
    mci_host_integrand_idx_fn host_idx_fn = nullptr; // the `integrand(idx, var, config)` form (mcmc/montecarlo.jl:34-36)
    int32_t *h_hidx = nullptr;                       // pinned: which integrand the host evaluates per chain (:mcmc)
    int64_t cap_hidx = 0;
    std::vector<double> h_tmp;                       // all-integrands <-> one-integrand adaptation of the two callback forms
    void *host_user = nullptr;
    double *d_hx = nullptr, *d_hw = nullptr, *h_hx = nullptr, *h_hw = nullptr; // device / pinned host
    int64_t cap_host = 0;
    // chain state between the per-step launches of a chain solver with a host integrand (BatchArgs::HostStep)
    void *d_hstep = nullptr;
    int64_t cap_hstep = 0; // chains
    // host measure ("batch callback"): draws + relative weights of the launch -> host closure per block -> block observables
    mci_host_measure_fn hmeas_fn = nullptr;
    mci_host_measure_idx_fn hmeas_idx_fn = nullptr; // the `measure(idx, var, obs, relative_weight, config)` form (mcmc/montecarlo.jl:166-169)
    void *hmeas_user = nullptr;
    double *d_mx = nullptr, *d_mrelw = nullptr, *h_mx = nullptr, *h_mrelw = nullptr, *d_mobs = nullptr;
    int32_t *d_midx = nullptr, *h_midx = nullptr;   // chain solvers: the integrand index of every record (:mcmc), -1 = no record
    int64_t cap_hmeas = 0, cap_mobs = 0;
    std::vector<double> h_mtmp;                     // callback form != record form: rows regrouped here
    std::vector<int32_t> h_mitmp;
    int threads = 256, wg_per_block = 0; // 0 = auto
    bool threads_explicit = false;       // mci_set_launch named a workgroup size
    // Plain-layout :vegas kernels of light integrands are compiled for workgroups of up to 512 threads (they need <= 128 registers anyway),
    // and mid-size launches -- one workgroup per CU, 2^19 <= samples x draws, samples < 2^22: the sizes the reference's own tests and
    // examples run -- use them: twice the lanes behind the same 256 prologues, epilogues and partial rows (tools/midsize_sweep.py,
    // profiles/r05_latency.txt: -7 .. -11 % per iteration on 2-D and 6-D integrands at 3e5 .. 3e6 samples)
    bool vegas_wide = false;
    // :vegas kernels whose tables take more than half of a CU's LDS (one workgroup per CU: 16 or 32 independent grids) pick their
    // workgroup size from the compiled code: the largest of 1024 / 768 / 512 threads (4 / 3 / 2 waves per SIMD) at which the sample
    // pass shows no scratch (128 / 168 / 256 registers).  threads_vegas = 0: the vegas kernel follows `threads`
    int threads_vegas = 0;
    bool vegas_plan_a = false; // the ladder is active (no explicit size was asked for)
    // histogram copies of the :vegas sample kernel (mci_device.h hslot): what the placement rule picked (shape.hcopy is what the
    // compiled kernel uses: the rule's choice, or 1 when that kernel needs more than 128 VGPRs and two 512-thread workgroups
    // would not share a CU)
    int kernel_timing = -1;       // mci_set_kernel_timing
    bool time_this_launch = true;
    bool ev_valid[512] = {};      // one per slot of the event ring (kEvRing)
    int hcopy_auto = 1, hcopy_rule = 1; // in force | what the placement rule picked at create
    // deterministic mode (mci_set_deterministic): every solver's kernel keeps one histogram / observable copy per wave; the workgroup
    // size each was compiled for (the largest of 512 / 256 / 128 / 64 threads whose copies fit the CU's LDS)
    bool deterministic = false;
    int threads_det[3] = {0, 0, 0};
    bool hcopy_plan = false; // the rule also picked the workgroup size (512 threads) for the :vegas kernel
    // refinement walk of train! (variable.jl:227-234): -1 automatic -- the reference's serial recurrence whenever the sample
    // launch before it is long enough to hide its ~14 us per iteration (>= kSerialWalkSamples samples or chain steps on this
    // rank: 1 % of the headline iteration), the prefix-scan form below that; mci_set_train_walk / MCI_TRAIN_SERIAL=1 | 0 force one
    int train_serial = -1;
    bool debug_wrong_decision = false; // csrc/mci_debug.h: the serial walk's slots with one planted wrong decision (TrainArgs::serial_walk == 3)
    int64_t last_samples = 0; // samples (vegas) or chain steps of the last sample launch on this rank
    static const int64_t kSerialWalkSamples = (int64_t)1 << 26;
    bool train_lds_raised = false; // k_train / k_finish allowed more than 64 KiB of dynamic LDS (large grids)
    // HIP events around the per-iteration ncclAllReduce (mci_comm_times_ms), recorded under the same rule as the sample launch's
    std::vector<hipEvent_t> cevs;
    bool cev_valid[64] = {};
    int64_t reduces = 0;
    static const int kCevRing = 64;
    // :mcmc automatic chain length: the holding-time histogram of launch k is copied to pinned host memory behind the launch (after
    // an all-reduce over the ranks, so that every rank sizes its chains from the SAME histogram) and is looked at when launch k + 1
    // is sized: the host waits for the sample kernel of launch k (not for its merge / train!, which run while launch k + 1 is
    // queued) -- ~10 us of idle queue per iteration, nothing next to a chain launch; the lag is fixed, so a run is reproducible
    unsigned long long *h_hold = nullptr;   // pinned [64]
    double *h_hold_d = nullptr;             // pinned [64]: the histogram summed over the ranks, as it comes out of the packed all-reduce
    bool hold_from_packed = false;          // the histogram in flight is the summed one (h_hold_d), not this rank's own (h_hold)
    bool hold_deferred = false;             // a communicator is set: the launch's histogram is published behind its packed all-reduce
    bool hold_ext_pending = false;          // no communicator: this rank's counts were published; an external reducer may still sum them (mci_external_reduce_done)
    hipEvent_t hold_ev = nullptr;
    bool hold_inflight = false;
    int64_t hold_launches = 0;              // :mcmc launches that recorded a histogram
    int64_t hold_len = 0;                   // measured steps per chain of the launch `hold_max` comes from
    int64_t hold_len_inflight = 0;          // ... of the launch whose histogram is in flight
    bool hold_carried_inflight = false;     // that launch continued the chains of the one before (8 x its holds instead of 16 x)
    // Warm-up of the automatic :mcmc chain length: until a launch has run chains long enough for the holds IT measured
    // (mcmc_launch_valid), lengths escalate and mci_integrate repeats an iteration instead of counting it; afterwards a launch is
    // sized from the larger of the last two launches' holds (the longest hold of a launch is an extreme value: it moves by a bucket
    // from launch to launch) and nothing is ever repeated or left out again (no selection on what an iteration measured)
    bool mcmc_warm = false;
    bool hold_valid = false;                // the launch `hold_max` comes from was long enough for its own holds
    bool hold_measured = false;             // the last :mcmc launch measured its holding times at all (not with a host integrand)
    int64_t hold_prev = 0;                  // hold_max of the launch before that, once warm
    // per-block means of the chain solvers' iterations (MergeArgs::block_means): rows [blk_rows][blk_stride = local blocks * nobs];
    // what the block-lineage error of a run of carried chains is computed from (mci_lineage_sums)
    double *d_blocklog = nullptr;
    int64_t cap_blocklog = 0, blk_rows = 0, blk_stride = 0, blk_lo = -1;
    int blk_carried = 0;                    // rows of the log whose launch continued the chains of the one before
    // Carried chains (BatchArgs::carry_x): end configurations of the last chain launch, two buffers (read one, write the other),
    // and what that launch was -- an iteration continues it when it is the NEXT iteration of the same solver over the same blocks
    double *d_chain_x[2] = {nullptr, nullptr};
    double *d_chain_P[2] = {nullptr, nullptr}; // :vegasmc: the target density at every stored configuration (BatchArgs::store_P)
    double *d_carry_w = nullptr;               // :vegasmc: new target / old target of the stored chains (mci_vegasmc_carry_weights)
    int64_t cap_carry_w = 0;
    hipFunction_t f_carryw[2] = {nullptr, nullptr}; // that kernel in the lane-per-chain | several-lanes-per-chain code object of :vegasmc
    int *d_chain_curr[2] = {nullptr, nullptr};
    int64_t chain_cap[2] = {0, 0};
    int chain_cur = 0;           // buffer the last launch wrote
    bool chain_valid = false;
    int chain_solver = -1, chain_iteration = -1;
    int64_t chain_lo = 0, chain_hi = 0, chain_nchain = 0;
    int chain_carry = -1;        // mci_set_chain_carry: -1 automatic / 1 (the rule above), 0 never
    // :vegasmc chains are carried only out of a launch that ran on a map train! had refined at least once: chains of the automatic
    // length have not reached their target on the UNTRAINED map of a heavy-tailed integrand (log(x)/sqrt(x): the first iteration of a cold
    // call is 14 sigma per run off), and a population that is no sample of the old target cannot be resampled into one of the new --
    // carried out of iteration 1 the second iteration was 4 sigma per run-iteration off, started afresh 1.2 (profiles/r05_bias.txt A4)
    int64_t ntrain = 0, chain_ntrain = 0; // train! steps of this problem so far | ... when the stored chains were launched
    bool launch_counted = false;          // mci_integrate | mci_set_iteration_counted: the iteration being launched enters the final estimate (it >= ignore)
    // :mcmc: the reweight factors the stored chains ran under, and which stored chain every chain of the launch in flight continues
    // (k_resample_chains: the stored chains resampled to the target doReweight! has moved since)
    double *d_reweight_used = nullptr, *d_carry_W = nullptr;
    int *d_carry_src = nullptr;
    int64_t cap_carry_src = 0, cap_carry_W = 0;
    bool last_carried = false;   // the last chain launch continued the one before it
    // last launch
    unsigned long long *d_clocks = nullptr; // [kEvRing][2] shader-clock | reference-clock ticks of the timed :vegas launches' sample loops
    std::vector<hipEvent_t> evs; // ring of (start, stop) pairs around the sampling kernel, one pair per launch
    int64_t launches = 0;
    static const int kEvRing = 512;
    static_assert(sizeof(ev_valid) / sizeof(ev_valid[0]) == kEvRing, "one validity flag per event-ring slot");
    int last_wg = 0, last_threads = 0, last_nblocks = 0;
    int64_t last_nchain = 0; // chains per block of the last chain-solver launch
    int log_row = 0;
    double *h_log = nullptr;  // pinned: mci_integrate's read-back of the iteration log (+ the status word behind it)
    size_t cap_hlog = 0;
    // persistent :vegas iterations (mci_train.h vegas_persist; mci_set_persistent): its own code object -- the plain layout at
    // `threads` -- and the two grid-wide counters, which only grow (the host keeps their values)
    hipModule_t module_persist = nullptr;
    hipFunction_t f_persist = nullptr;
    bool persist_compiled = false, persist_failed = false;
    std::string persist_code_object;
    int persist_threads = 256;    // its workgroup size: 512 for the hand-pipelined loops (8..16 draws), else `threads`
    // its translation unit takes twice as long to compile as the plain sample kernel (train! comes with it): in automatic mode a code
    // object that is not in the kernel cache is compiled on a thread of its own while the calls go through the launch chain
    struct PersistJob;
    PersistJob *persist_job = nullptr;
    unsigned long long *d_persist = nullptr; // [0] arrived | done << 40, [2] gave up
    double *d_edges_backup = nullptr;        // the map a persistent launch started from (restored when it stalls)
    unsigned long long persist_arrive = 0, persist_done = 0;
    unsigned long long persist_spin_ticks = 200000000ull; // ticks of the 100 MHz wall clock a grid-wide wait may take: 2 s (mci_debug_persist_spin_ticks)
    int persistent = -1;          // -1 automatic (launch-bound :vegas calls of mci_integrate), 0 never, 1 whenever the layout allows
    bool last_persistent = false; // the last mci_integrate ran as one persistent launch
    static const int kGroups = mci::kMergeGroups;
    static const int64_t kChainFill = 131072; // chains per GPU that keep 2 waves on each of the 1024 SIMDs
    // automatic :mcmc chain lengths (mci_mcmc_auto_chains): measured steps per chain while nothing has been measured | how much longer
    // than the chains that measured the holds a launch's chains may be.  (MCI_MCMC_PILOT / MCI_MCMC_GROW: experiment knobs)
    static int64_t kMcmcPilotSteps, kMcmcGrow;
    static int64_t kMcmcCarryHolds, kMcmcCarryHalfFloors; // carried chains: length in longest holds | minimum length in HALF burn-in floors
};

// A repeated iteration (the warm-up of automatic :mcmc chain lengths, mci_integrate) draws from the Philox streams of iteration
// i + kRepeatStride * attempt: the iteration index has 17 bits (DESIGN.md "RNG streams"), runs of fewer than 16384 iterations leave the upper ones free
static const int kRepeatStride = 16384, kMaxRepeats = 7;

// (process-wide; csrc/mci_debug.h mci_debug_mcmc_policy moves them for A/B campaigns -- tools/mcmc_policy.py, profiles/r04_mcmc_policy.txt)
int64_t mci_problem::kMcmcPilotSteps = 4096;
int64_t mci_problem::kMcmcGrow = 2;
int64_t mci_problem::kMcmcCarryHolds = 4;
int64_t mci_problem::kMcmcCarryHalfFloors = 2;

// Layout decisions of mci_problem_create that tests and A/B tools force (csrc/mci_debug.h mci_debug_override): process-wide, consulted
// by the NEXT mci_problem_create.  The library itself reads two environment variables and no others: MCI_KERNEL_CACHE (where code objects
// are cached) and MCI_JIT_FLAGS (extra hiprtc options), mci_jit.h.
namespace {
struct Override { bool on = false; int64_t v = 0; };
struct Overrides { Override table_mode, hist_tile_bins, no_split_all, l1_phase, train_walk, hist_copies, fresh_floors, fresh_burnin_pct; } g_over;
Override *override_slot(const char *key) {
    if (!key) return nullptr;
    if (!strcmp(key, "table_mode")) return &g_over.table_mode;
    if (!strcmp(key, "hist_tile_bins")) return &g_over.hist_tile_bins;
    if (!strcmp(key, "no_split_all")) return &g_over.no_split_all;
    if (!strcmp(key, "l1_phase")) return &g_over.l1_phase;
    if (!strcmp(key, "train_walk")) return &g_over.train_walk;
    if (!strcmp(key, "hist_copies")) return &g_over.hist_copies;
    if (!strcmp(key, "fresh_floors")) return &g_over.fresh_floors;
    if (!strcmp(key, "fresh_burnin_pct")) return &g_over.fresh_burnin_pct;
    return nullptr;
}
} // namespace

static void persist_job_drop(mci_problem *p);
namespace { void persist_orphans_join(); }
// counters [0..2] of the persistent :vegas kernel + (MCI_PERSIST_TRACE builds) the phase stamps of three workgroups over eight turns
static const size_t kPersistWords = 8 + 3 * 8 * 8 + 16;

namespace {

int upload(mci_problem *p) {
    if (p->ctx->offline) return MCI_OK;
    auto up = [&](double *&d, const std::vector<double> &h) -> int {
        size_t n = h.size() ? h.size() : 1;
        if (!d) HIPCHK(hipMalloc((void **)&d, n * sizeof(double)));
        if (h.size()) HIPCHK(hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, p->ctx->stream));
        return MCI_OK;
    };
    int rc;
    if ((rc = up(p->d_edges, p->h_edges))) return rc;
    if ((rc = up(p->d_dacc, p->h_dacc))) return rc;
    if ((rc = up(p->d_ddist, p->h_ddist))) return rc;
    if ((rc = up(p->d_reweight, p->h_reweight))) return rc;
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

int ensure_capacity(mci_problem *p, int64_t nwg, int64_t nblocks) {
    const auto &s = p->shape;
    if (nwg > p->cap_wg) {
        if (p->d_part_cols) (void)hipFree(p->d_part_cols);
        if (p->d_part_hist) (void)hipFree(p->d_part_hist);
        p->d_part_cols = p->d_part_hist = nullptr;
        HIPCHK(hipMalloc((void **)&p->d_part_cols, (size_t)nwg * s.ncols * sizeof(double)));
        if (s.table_mode == 0 || s.table_mode == 3) HIPCHK(hipMalloc((void **)&p->d_part_hist, (size_t)nwg * (s.nbin ? s.nbin : 1) * sizeof(double)));
        p->cap_wg = nwg;
    }
    if (nblocks > p->cap_blocks) {
        if (p->d_scratch) (void)hipFree(p->d_scratch);
        p->d_scratch = nullptr;
        HIPCHK(hipMalloc((void **)&p->d_scratch, (size_t)nblocks * s.ncols * sizeof(double)));
        p->cap_blocks = nblocks;
    }
    return MCI_OK;
}

int check_status(mci_problem *p) {
    int st = 0;
    HIPCHK(hipMemcpyAsync(&st, p->d_status, sizeof(int), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    if (!st) return MCI_OK;
    HIPCHK(hipMemsetAsync(p->d_status, 0, sizeof(int), p->ctx->stream));
    if (st & mci::ST_PERSIST_STALL) { // a grid-wide wait of the persistent :vegas launch ran out of time: its counters are void
        // (mci_integrate recovers by itself and never gets here with this bit; this is the message of a stall somebody else finds)
        HIPCHK(hipMemsetAsync(p->d_persist, 0, 3 * sizeof(unsigned long long), p->ctx->stream));
        HIPCHK(hipMemsetAsync(p->d_ghist, 0, 3 * (size_t)(p->shape.nbin ? p->shape.nbin : 1) * sizeof(double), p->ctx->stream));
        p->persist_arrive = p->persist_done = 0;
        p->persist_failed = true; // (later calls take the launch-per-iteration path)
        return fail(MCI_ERR_HIP, "the persistent :vegas launch stalled (is the device shared with other long-running kernels?); "
                                 "the iterations of this call are void -- later calls launch per iteration (mci_set_persistent(prob, 0))");
    }
    if (st & mci::ST_MCMC_INIT) return fail(MCI_ERR_INVALID, "Cannot find the variables that makes the integrand nonzero!"); // mcmc/montecarlo.jl:126
    if (st & mci::ST_NORMALIZATION) return fail(MCI_ERR_NORMALIZATION, "Block normalization is not positively defined!");
    if (st & mci::ST_HIST_NONFINITE) return fail(MCI_ERR_HISTOGRAM, "histogram should be all finite");
    if (st & mci::ST_HIST_NONPOSITIVE) return fail(MCI_ERR_HISTOGRAM, "histogram should be all positive and non-zero");
    return fail(MCI_ERR_HISTOGRAM, "distribution is not all finite");
}

// after a stalled persistent :vegas launch: status word, grid-wide counters and the three histogram buffers back to their idle state
int persist_recover(mci_problem *p) {
    hipStream_t st = p->ctx->stream;
    HIPCHK(hipMemsetAsync(p->d_status, 0, sizeof(int), st));
    HIPCHK(hipMemsetAsync(p->d_persist, 0, 3 * sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(p->d_ghist, 0, 3 * (size_t)(p->shape.nbin ? p->shape.nbin : 1) * sizeof(double), st));
    p->persist_arrive = p->persist_done = 0;
    p->persist_failed = true; // (later calls take the launch-per-iteration path)
    p->merge_pending = false;
    return MCI_OK;
}

// :mcmc holding-time histogram of the launch just queued -> pinned host memory, behind the launch on the stream.  One process: this
// rank's counts, straight from the kernel's buffer (the host later waits for the sample kernel only).  With a communicator every rank
// must size its next chains from the SAME histogram: the 64 counts ride in the iteration's ONE all-reduce -- k_finalize appends them
// to `packed` as exact doubles (MergeArgs::hold), mci_iteration_reduce sums packed_n + 64 doubles and publishes the tail
// (hold_publish_reduced) -- so the launch only notes what it measured with.
int hold_publish(mci_problem *p, int64_t chain_len, bool carried) {
    hipStream_t st = p->ctx->stream;
    if (!p->h_hold) {
        HIPCHK(hipHostMalloc((void **)&p->h_hold, 64 * sizeof(unsigned long long), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&p->h_hold_d, 64 * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&p->hold_ev, hipEventDisableTiming));
    }
    p->hold_len_inflight = chain_len;
    p->hold_carried_inflight = carried;
    p->hold_launches += 1;
    if (p->ctx->comm) {
        p->hold_deferred = true;
        return MCI_OK;
    }
    if (p->hold_inflight) HIPCHK(hipEventSynchronize(p->hold_ev)); // (a histogram nobody looked at)
    HIPCHK(hipMemcpyAsync(p->h_hold, p->d_hold, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(p->hold_ev, st));
    p->hold_inflight = true;
    p->hold_from_packed = false;
    p->hold_ext_pending = true;
    return MCI_OK;
}

// ... behind the all-reduce of `packed` (the library's, or an external reducer's: mci_external_reduce_done): the summed counts
int hold_publish_reduced(mci_problem *p) {
    hipStream_t st = p->ctx->stream;
    if (p->hold_inflight) HIPCHK(hipEventSynchronize(p->hold_ev));
    HIPCHK(hipMemcpyAsync(p->h_hold_d, p->d_packed + p->packed_n, 64 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(p->hold_ev, st));
    p->hold_inflight = true;
    p->hold_from_packed = true;
    p->hold_deferred = false;
    return MCI_OK;
}

// before an :mcmc launch with an automatic chain count is sized: take in the histogram of the launch before it.  The host waits for
// that launch's sample kernel here (its merge and train! are still running or queued: the next launch is queued behind them while they
// run); what the two-launch lag of the rounds before cost is in profiles/r03_c5_kernel_stats.txt (two more launches sized from the
// untrained map's holding times: 324 ms of a cold BASELINE configs[4] call).
int hold_consume(mci_problem *p) {
    if (!p->hold_inflight) return MCI_OK;
    HIPCHK(hipEventSynchronize(p->hold_ev));
    p->hold_inflight = false;
    int top = -1;
    for (int b = 0; b < 64; ++b)
        if (p->hold_from_packed ? p->h_hold_d[b] > 0.5 : p->h_hold[b] != 0ull) top = b;
    if (top >= 0) {
        p->hold_prev = p->mcmc_warm ? p->hold_max : 0;
        p->hold_max = (int64_t)1 << top; // bucket b holds bit_width(h) == b, i.e. h < 2^b
        p->hold_len = p->hold_len_inflight;
        // was that launch long enough for what it measured itself?  (the rule its successor is sized by, mci_mcmc_auto_chains)
        p->hold_valid = p->hold_len >= (p->hold_carried_inflight ? mci_problem::kMcmcCarryHolds : 16) * p->hold_max;
        if (p->hold_valid) p->mcmc_warm = true;
    }
    return MCI_OK;
}

void drop_modules(mci_problem *p) {
    p->vegas_planned = p->vegas_keys = p->vegas_wide = false;
    p->f_dump = nullptr;
    for (int k = 0; k < mci_problem::kSlots; ++k) {
        p->compiled[k] = false;
        if (p->module[k]) {
            (void)hipModuleUnload(p->module[k]);
            p->module[k] = nullptr;
        }
    }
    p->persist_compiled = p->persist_failed = false;
    persist_job_drop(p);
    p->f_persist = nullptr;
    if (p->module_persist) {
        (void)hipModuleUnload(p->module_persist);
        p->module_persist = nullptr;
    }
}

} // namespace

static int flush_merge(mci_problem *p);
static int comm_sum_host(mci_problem *p, double *v, int n);

// room for `rows` rows of [blk_stride] doubles in the block log (grows with a copy and a stream synchronisation; mci_integrate reserves
// its iterations before the loop)
static int grow_block_log(mci_problem *p, int64_t rows) {
    const int64_t need = rows * p->blk_stride;
    if (need <= p->cap_blocklog) return MCI_OK;
    int64_t ncap = p->cap_blocklog ? p->cap_blocklog : 4096;
    while (ncap < need) ncap *= 2;
    double *n = nullptr;
    HIPCHK(hipMalloc((void **)&n, (size_t)ncap * sizeof(double)));
    if (p->d_blocklog) {
        HIPCHK(hipMemcpyAsync(n, p->d_blocklog, (size_t)p->cap_blocklog * sizeof(double), hipMemcpyDeviceToDevice, p->ctx->stream));
        HIPCHK(hipStreamSynchronize(p->ctx->stream));
        (void)hipFree(p->d_blocklog);
    }
    p->d_blocklog = n;
    p->cap_blocklog = ncap;
    return MCI_OK;
}

extern "C" {

const char *mci_last_error(void) { return g_err.c_str(); }
// "mci-hip <abi>.<revision>": <abi> changes whenever a struct of include/mci.h changes its layout (mci_result grew `correlated` and
// `warmup` in ABI 4; ABI 5 adds entry points only) -- a caller built against another header compares it before passing structs
const char *mci_version(void) { return "mci-hip 5.0 (gfx950)"; }
int32_t mci_abi_version(void) { return 5; }

int mci_device_count(int32_t *count) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return MCI_OK;
}

int mci_ctx_create(int32_t device, mci_ctx **out) {
    if (!out) return fail(MCI_ERR_INVALID, "out is NULL");
    mci_ctx *c = new mci_ctx();
    if (device < 0) { // offline / compile-only
        c->offline = true;
        *out = c;
        return MCI_OK;
    }
    mcijit::warm_up_async(); // (the compiler loads while the HIP runtime initialises the device below)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        mcijit::warm_up_join();
        delete c;
        return fail(MCI_ERR_NO_DEVICE, "no HIP device visible: the MI355X path has no CPU fallback");
    }
    if (device >= n) {
        delete c;
        return fail(MCI_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    }
    c->device = device;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c;
    return MCI_OK;
}

int mci_ctx_destroy(mci_ctx *c) {
    if (!c) return MCI_OK;
    mcijit::warm_up_join();
    persist_orphans_join();
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MCI_OK;
}

void *mci_ctx_stream(mci_ctx *c) { return c ? (void *)c->stream : nullptr; }

int mci_comm_unique_id(void *id128) {
    int rc = rccl_load();
    if (rc) return rc;
    int r = g_rccl.GetUniqueId(id128);
    if (r) return fail(MCI_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return MCI_OK;
}

int mci_comm_init(mci_ctx *c, int32_t rank, int32_t nranks, const void *id128) {
    if (!c || c->offline) return fail(MCI_ERR_INVALID, "communicator needs an online context");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(MCI_ERR_INVALID, "bad rank %d / %d", rank, nranks);
    int rc = rccl_load();
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    Id128 id;
    memcpy(id.b, id128, 128);
    int r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r) return fail(MCI_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    c->rank = rank;
    c->nranks = nranks;
    return MCI_OK;
}

int mci_comm_rank(const mci_ctx *c, int32_t *rank, int32_t *nranks) {
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return MCI_OK;
}

// ---------------------------------------------------------------------------------------------------
// Configuration(; var, dof, obs)   reference src/configuration.jl:105-194
// ---------------------------------------------------------------------------------------------------
int mci_problem_create(mci_ctx *ctx, const mci_problem_desc *d, mci_problem **out) {
    if (!ctx || !d || !out) return fail(MCI_ERR_INVALID, "NULL argument");
    if (d->nleaf < 1 || d->npool < 1 || d->nintegrand < 1) return fail(MCI_ERR_INVALID, "At least one integrand is required."); // :163
    if (d->nintegrand > 31) return fail(MCI_ERR_INVALID, "at most 31 integrands are supported");
    mci_problem *p = new mci_problem();
    p->ctx = ctx;
    p->npool = d->npool;
    p->ni = d->nintegrand;
    const int Nd = p->ni + 1;
    p->dof.assign((size_t)Nd * p->npool, 0); // last row: normalisation integrand, dof = 0   :153
    for (int i = 0; i < p->ni * p->npool; ++i) {
        if (d->dof[i] < 0) { delete p; return fail(MCI_ERR_INVALID, "dof must be non-negative"); }
        p->dof[i] = d->dof[i];
    }
    p->maxdof.assign(p->npool, 0);
    mci_maxdof(p->dof.data(), Nd, p->npool, p->maxdof.data()); // :155
    p->pool_leaf0.assign(p->npool, -1);
    p->pool_nleaf.assign(p->npool, 0);
    auto &s = p->shape;
    int eoff = 0, aoff = 0, doff = 0, boff = 0;
    for (int l = 0; l < d->nleaf; ++l) {
        const mci_leaf_desc &ld = d->leaves[l];
        if (ld.pool < 0 || ld.pool >= p->npool) { delete p; return fail(MCI_ERR_INVALID, "leaf %d: pool out of range", l); }
        if (p->pool_leaf0[ld.pool] < 0) p->pool_leaf0[ld.pool] = l;
        else if (p->pool_leaf0[ld.pool] + p->pool_nleaf[ld.pool] != l) { delete p; return fail(MCI_ERR_INVALID, "leaves of pool %d are not contiguous", ld.pool); }
        p->pool_nleaf[ld.pool] += 1;
        Leaf L{};
        L.kind = ld.kind;
        L.pool = ld.pool;
        L.lower = ld.lower;
        L.upper = ld.upper;
        L.alpha = ld.alpha;
        L.adapt = ld.adapt ? 1 : 0;
        if (ld.kind == MCI_CONTINUOUS) {
            if (!(ld.upper > ld.lower + 2 * 2.220446049250313e-16)) { delete p; return fail(MCI_ERR_INVALID, "leaf %d: upper > lower required", l); } // variable.jl:140
            L.npts = ld.npoints > 0 ? ld.npoints : 1000; // variable.jl:137
            if (L.npts < 2) { delete p; return fail(MCI_ERR_INVALID, "leaf %d: at least 2 grid points", l); }
            L.nbin = L.npts - 1;                          // variable.jl:147
            L.eoff = eoff;
            L.doff = 0;
            for (int i = 0; i < L.npts; ++i) {
                double v;
                if (ld.init) v = ld.init[i];
                else { // collect(LinRange(lower, upper, ninc))
                    const double t = (double)i / (double)(L.npts - 1);
                    v = (1.0 - t) * ld.lower + t * ld.upper;
                    if (i == 0) v = ld.lower;
                    if (i == L.npts - 1) v = ld.upper;
                }
                p->h_edges.push_back(v);
            }
            eoff += L.npts;
        } else if (ld.kind == MCI_DISCRETE) {
            const int K = (int)(ld.upper - ld.lower) + 1;
            if (K < 1) { delete p; return fail(MCI_ERR_INVALID, "leaf %d: upper >= lower required", l); } // variable.jl:304
            L.npts = K;
            L.nbin = K; // variable.jl:305
            L.eoff = aoff;
            L.doff = doff;
            std::vector<double> dist(K);
            double sum = 0.0;
            for (int i = 0; i < K; ++i) {
                dist[i] = ld.init ? ld.init[i] : 1.0;
                if (!(dist[i] >= 0.0)) { delete p; return fail(MCI_ERR_INVALID, "distribution should be all non-negative!"); } // variable.jl:309
                sum += dist[i];
            }
            double run = 0.0;
            p->h_dacc.push_back(0.0); // variable.jl:313-314
            for (int i = 0; i < K; ++i) {
                dist[i] /= sum; // variable.jl:312
                run += dist[i];
                p->h_ddist.push_back(dist[i]);
                p->h_dacc.push_back(run);
            }
            aoff += K + 1;
            doff += K;
        } else if (ld.kind == MCI_FERMIK) { // FermiK(dim, kF, dk, maxK)  variable.jl:11-19: lower = kF, upper = dk, npoints = dim
            if (ld.npoints != 2 && ld.npoints != 3) { delete p; return fail(MCI_ERR_INVALID, "leaf %d: FermiK has 2 or 3 dimensions", l); }
            if (!(ld.lower > 0.0) || !(ld.upper > 0.0)) { delete p; return fail(MCI_ERR_INVALID, "leaf %d: FermiK needs kF > 0 and dk > 0", l); }
            L.npts = ld.npoints;
            L.width = ld.npoints;
            L.nbin = 1;   // histogram = [0.0]  variable.jl:19
            L.adapt = 0;  // train!(Var) = nothing  variable.jl:557
            L.eoff = 0;
            L.doff = 0;
            p->has_fermik = true;
        } else {
            delete p;
            return fail(MCI_ERR_INVALID, "leaf %d: unknown kind %d", l, ld.kind);
        }
        if (L.nbin > kMaxLeafBins) {
            delete p;
            return fail(MCI_ERR_INVALID, "leaf %d: %d increments; train! refines a grid inside one CU's LDS, at most %d increments per variable", l, L.nbin, kMaxLeafBins);
        }
        L.boff = boff;
        boff += L.nbin;
        p->leaves.push_back(L);
    }
    for (int v = 0; v < p->npool; ++v) {
        if (p->pool_nleaf[v] == 0) { delete p; return fail(MCI_ERR_INVALID, "pool %d has no variable", v); }
        if (p->pool_nleaf[v] > 1)
            for (int l = 0; l < p->pool_nleaf[v]; ++l)
                if (p->leaves[p->pool_leaf0[v] + l].kind == MCI_FERMIK) { delete p; return fail(MCI_ERR_INVALID, "pool %d: FermiK cannot be part of a CompositeVar", v); }
    }
    // flat draw order: pool, slot, leaf  (vegas/montecarlo.jl:122-131, sampler.jl:431-440)
    s.nleaf = d->nleaf;
    s.ni = p->ni;
    s.npool = p->npool;
    for (int v = 0; v < p->npool; ++v) {
        s.pool_first_draw.push_back((int)s.draw_leaf.size());
        s.pool_maxdof.push_back(p->maxdof[v]);
        int width = 0; // x entries per slot: one per leaf, D for a FermiK pool
        for (int l = 0; l < p->pool_nleaf[v]; ++l) width += p->leaves[p->pool_leaf0[v] + l].width;
        s.pool_nleaf.push_back(width);
        for (int idx = 0; idx < p->maxdof[v]; ++idx)
            for (int l = 0; l < p->pool_nleaf[v]; ++l)
                for (int j = 0; j < p->leaves[p->pool_leaf0[v] + l].width; ++j) {
                    s.draw_leaf.push_back(p->pool_leaf0[v] + l);
                    s.draw_pool.push_back(v);
                    s.draw_slot.push_back(idx);
                }
    }
    s.ndraw = (int)s.draw_leaf.size();
    if (s.ndraw < 1 || s.ndraw > 64) { delete p; return fail(MCI_ERR_INVALID, "1..64 draws per sample supported, got %d", s.ndraw); }
    s.own_mask.assign(Nd, 0ull);
    s.cover_mask.assign(s.ndraw, 0ull);
    for (int i = 0; i < p->ni; ++i)
        for (int k = 0; k < s.ndraw; ++k)
            if (s.draw_slot[k] < p->dof[(size_t)i * p->npool + s.draw_pool[k]]) {
                s.own_mask[i] |= 1ull << k;
                s.cover_mask[k] |= 1ull << i;
            }
    s.dof = p->dof;
    if (d->ncomp != 0 && d->ncomp != 1 && d->ncomp != 2) { delete p; return fail(MCI_ERR_INVALID, "ncomp must be 1 (Float64) or 2 (ComplexF64)"); }
    s.ncomp = d->ncomp == 2 ? 2 : 1;
    { // neighbor graph of the integrands (mcmc): configuration.jl:201-227, 0-based, index ni = normalisation
        std::vector<std::vector<int>> nb(Nd);
        if (d->neighbor_offsets && d->neighbor_list) {
            for (int i = 0; i < Nd; ++i) {
                const int b = d->neighbor_offsets[i], e = d->neighbor_offsets[i + 1];
                if (e <= b) { delete p; return fail(MCI_ERR_INVALID, "%d elements are expected for neighbor", Nd); } // :226
                for (int j = b; j < e; ++j) {
                    if (d->neighbor_list[j] < 0 || d->neighbor_list[j] >= Nd) { delete p; return fail(MCI_ERR_INVALID, "neighbor %d of integrand %d out of range", d->neighbor_list[j], i); }
                    nb[i].push_back(d->neighbor_list[j]);
                }
            }
        } else { // :203-208
            for (int i = 0; i < Nd; ++i) nb[i] = {i - 1, i + 1};
            if (Nd == 2) nb[0] = {1};
            else nb[0] = {Nd - 1, 1};
            nb[Nd - 1] = {0};
            if (Nd >= 3) nb[Nd - 2] = {Nd - 3};
        }
        s.nbmax = 1;
        for (auto &v : nb) s.nbmax = (int)v.size() > s.nbmax ? (int)v.size() : s.nbmax;
        s.nneighbor.clear();
        s.neighbor.assign((size_t)Nd * s.nbmax, 0);
        for (int i = 0; i < Nd; ++i) {
            s.nneighbor.push_back((int)nb[i].size());
            for (int j = 0; j < s.nbmax; ++j) s.neighbor[(size_t)i * s.nbmax + j] = j < (int)nb[i].size() ? nb[i][j] : i;
        }
    }
    s.nobs = 0;
    for (int i = 0; i < p->ni; ++i) {
        const int nb = d->obs_nbin ? d->obs_nbin[i] : s.ncomp;
        const int bd = d->obs_bin_draw ? d->obs_bin_draw[i] : -1;
        if (nb < 1 || (bd >= s.ndraw)) { delete p; return fail(MCI_ERR_INVALID, "observable %d: bad shape", i); }
        if (bd >= 0 && p->leaves[s.draw_leaf[bd]].kind != MCI_DISCRETE) { delete p; return fail(MCI_ERR_INVALID, "observable %d: bin draw must be a Discrete draw", i); }
        if (bd >= 0 && s.ncomp != 1) { delete p; return fail(MCI_ERR_INVALID, "observable %d: binned observables are real", i); }
        s.obs_off.push_back(s.nobs);
        s.obs_nbin.push_back(nb);
        s.obs_bin_draw.push_back(bd);
        s.nobs += nb;
    }
    s.ncols = s.nobs + 2 + Nd;
    p->npa = 3 * Nd * (Nd > p->npool ? Nd : p->npool);
    s.nedge = eoff;
    s.ndacc = aoff;
    s.nddist = doff;
    s.nbin = boff;
    for (auto &L : p->leaves) {
        s.leaf_kind.push_back(L.kind);
        s.leaf_nbin.push_back(L.nbin);
        s.leaf_eoff.push_back(L.eoff);
        s.leaf_doff.push_back(L.doff);
        s.leaf_boff.push_back(L.boff);
        s.leaf_adapt.push_back(L.adapt);
        s.leaf_lower.push_back(L.lower);
        s.leaf_upper.push_back(L.upper);
    }
    // table placement (DESIGN.md "data layout"): keep >= 2 workgroups per CU when everything is in LDS.
    // PAIR_TABLE stores (g[i], g[i+1]-g[i]) per bin (16 B, one ds_read_b128 per draw) when that still fits.
    {
        int npair = 0;
        for (auto &L : p->leaves) {
            s.leaf_poff.push_back(npair);
            if (L.kind == MCI_CONTINUOUS) npair += 2 * L.nbin;
        }
        s.npair = npair;
        const int64_t fixed = (int64_t)(s.ndacc + s.nddist + s.nobs + 16 * s.ncols + 2 * p->npa) * 8;
        const int64_t e1 = (int64_t)s.nedge * 8, e2 = (int64_t)npair * 8, hb = (int64_t)s.nbin * 8;
        // lim0: >= 2 workgroups of 256 threads per CU; lim1: one 1024-thread workgroup owning the CU's LDS
        const int64_t lim0 = 80 * 1024, lim1 = 160 * 1024 - 1024;
        int mode = 3, pair = 0;
        if (fixed + e2 + hb <= lim0) { mode = 0; pair = 1; }
        else if (fixed + e1 + hb <= lim0) { mode = 0; pair = 0; }
        else if (fixed + e2 + hb <= lim1) { mode = 0; pair = 1; }
        else if (fixed + e1 + hb <= lim1) { mode = 0; pair = 0; }
        if (g_over.table_mode.on) { // test / diagnostic override (mci_debug_override)
            const int m = (int)g_over.table_mode.v;
            if (m == 1 && fixed + e1 <= lim1) { mode = 1; pair = (fixed + e2 <= lim1) ? 1 : 0; }
            if (m == 2) { mode = 2; pair = 0; }
            if (m == 3) { mode = 3; pair = 0; }
        }
        if (g_over.train_walk.on) p->train_serial = g_over.train_walk.v == 2 ? 2 : g_over.train_walk.v != 0 ? 1 : 0; // (= mci_set_train_walk on every new problem)
        // histogram tiles: contiguous leaves, each tile's bins fit the LDS left over
        s.leaf_tile.assign(p->leaves.size(), 0);
        s.tile_boff.assign(1, 0);
        s.tile_nbin.assign(1, s.nbin);
        if (mode == 3) {
            int64_t budget = (lim1 - fixed) / 8; // doubles
            if (g_over.hist_tile_bins.on) budget = g_over.hist_tile_bins.v;
            s.tile_boff.clear();
            s.tile_nbin.clear();
            // as few tiles as the budget allows, filled evenly: the replay kernel's time follows its LARGEST tile
            // (C4: 19 + 13 grids 2.70 ms, 16 + 16 grids 2.23 ms)
            int64_t fill = budget;
            {
                int64_t ntile_min = 1, acc = 0;
                for (const Leaf &L : p->leaves) {
                    if (acc + L.nbin > budget) { ntile_min += 1; acc = 0; }
                    acc += L.nbin;
                }
                const int64_t even = ((int64_t)s.nbin + ntile_min - 1) / ntile_min;
                int64_t mx = 0;
                for (const Leaf &L : p->leaves) mx = L.nbin > mx ? L.nbin : mx;
                fill = even + mx - 1 < budget ? even + mx - 1 : budget; // a tile closes once it holds >= `even` bins
                if (g_over.hist_tile_bins.on) fill = budget;
            }
            int cur = -1;
            for (size_t l = 0; l < p->leaves.size(); ++l) {
                const Leaf &L = p->leaves[l];
                if (L.nbin > budget) { delete p; return fail(MCI_ERR_INVALID, "leaf %zu: %d bins do not fit the LDS histogram", l, L.nbin); }
                if (cur < 0 || s.tile_nbin[cur] + L.nbin > fill) {
                    s.tile_boff.push_back(L.boff);
                    s.tile_nbin.push_back(0);
                    cur += 1;
                }
                s.leaf_tile[l] = cur;
                s.tile_nbin[cur] += L.nbin;
            }
        }
        s.ntile = (int)s.tile_nbin.size();
        // several tiles under :vegas -> "split-all": the sample pass keeps no histogram at all and uses the LDS for the edges
        // of as many leading grids as fit (they stop being L2 gathers); every tile is replayed by mci_vegas_tiles.  Measured on
        // C4 (32 grids): 10.3 -> see profiles; the override no_split_all = 1 restores "tile 0 in the sample pass" for A/B runs.
        s.split_all = (s.ntile > 1 && !(g_over.no_split_all.on && g_over.no_split_all.v != 0)) ? 1 : 0;
        s.leaf_ecoff.assign(p->leaves.size(), -1);
        s.ec_doubles = 0;
        if (s.split_all) {
            int64_t budget = (lim1 - fixed) / 8;
            for (size_t l = 0; l < p->leaves.size(); ++l) {
                const Leaf &L = p->leaves[l];
                if (L.kind != MCI_CONTINUOUS || s.ec_doubles + L.nbin + 1 > budget) continue;
                s.leaf_ecoff[l] = s.ec_doubles;
                s.ec_doubles += L.nbin + 1;
            }
        }
        p->lds_bytes_k1 = fixed + (int64_t)s.ec_doubles * 8;
        p->ntdraw = 0;
        if (s.ntile > 1)
            for (int k = 0; k < s.ndraw; ++k) {
                const Leaf &L = p->leaves[s.draw_leaf[k]];
                if (L.adapt && s.cover_mask[k] && s.leaf_tile[s.draw_leaf[k]] >= (s.split_all ? 0 : 1)) {
                    p->ntdraw += 1;
                    if (L.nbin > 65536) { delete p; return fail(MCI_ERR_INVALID, "leaf %d: more than 65536 bins with tiled histograms", s.draw_leaf[k]); }
                }
            }
        s.htile = 0;
        for (int v : s.tile_nbin) s.htile = v > s.htile ? v : s.htile;
        s.table_mode = mode;
        s.pair_table = pair;
        const bool hist_lds = (mode == 0 || mode == 3);
        p->lds_bytes = fixed + (mode <= 1 ? (pair ? e2 : e1) : 0) + (hist_lds ? (int64_t)s.htile * 8 : 0);
        // Interleaved histogram copies for the :vegas sample kernel (mci_device.h hslot): fewer LDS bank conflicts of the
        // random-address ds_add_f64.  Rule: tables in LDS (mode 0), as many copies (<= 8) as leave room for TWO 512-thread
        // workgroups per CU (4 waves per SIMD when the kernel needs <= 128 VGPRs; compile_solver checks).  Measured on C2
        // (tools/hcopy_sweep.sh, kernel ms per 1e8 samples): 1 copy x 256 threads 1.715 | 4 x 512 1.663 | 8 x 512 1.625 |
        // 16 x 1024 (one workgroup per CU) 1.662 | 8 x 1024 1.694.  The override hist_copies forces a count (1 = off).
        s.hcopy = 1;
        {
            int hc = 1;
            const int64_t one = (int64_t)s.htile * 8;
            int nadd = 0; // ds_add_f64 per sample
            for (int k = 0; k < s.ndraw; ++k) nadd += (p->leaves[s.draw_leaf[k]].adapt && s.cover_mask[k]) ? 1 : 0;
            if (mode == 0 && s.ntile == 1 && p->lds_bytes <= lim0 && nadd >= 4) // (a 1-D integrand runs 4 % slower with 512 threads and gains nothing)
                while (hc < 8 && p->lds_bytes + one * (2 * hc - 1) <= lim0) hc *= 2;
            if (g_over.hist_copies.on) { // diagnostic override
                hc = (int)g_over.hist_copies.v;
                while (hc > 1 && (!hist_lds || s.ntile != 1 || (hc & (hc - 1)) || p->lds_bytes + one * (hc - 1) > lim1)) hc >>= 1;
                if (hc < 1) hc = 1;
            }
            s.hcopy = p->hcopy_auto = p->hcopy_rule = hc;
        }
        const int64_t hcopy_bytes = (int64_t)s.htile * 8 * (s.hcopy - 1);
        // one tile, grids gathered from L2 (10 .. 18 independent grids): the LDS left next to the histogram caches the edges of the
        // leading grids for the :vegas sample pass
        if (mode == 3 && s.ntile == 1) {
            const int64_t budget = (lim1 - p->lds_bytes - hcopy_bytes) / 8;
            for (size_t l = 0; l < p->leaves.size(); ++l) {
                const Leaf &L = p->leaves[l];
                if (L.kind != MCI_CONTINUOUS || s.ec_doubles + L.nbin + 1 > budget) continue;
                s.leaf_ecoff[l] = s.ec_doubles;
                s.ec_doubles += L.nbin + 1;
            }
            p->lds_bytes_k1 = p->lds_bytes + (int64_t)s.ec_doubles * 8;
        }
        // Split-all pass (several histogram tiles, e.g. 32 grids): the grids gathered from global memory are walked dimension-major by
        // all waves of a workgroup in step, so that the CU's L1 sees one or two 8 KB tables at a time (draw_gather_phase).  Measured
        // on C4 (tools/ab_c2.py): 7.26 -> 6.95 ms per 1e8 samples; with one tile (16 grids, histogram in the pass) the barriers
        // cost more than the locality buys (2.77 -> 3.47 ms), so it stays off there.  The override l1_phase = 0 | 1 forces it.
        s.l1_phase = (mode == 3 && s.split_all) ? 1 : 0;
        if (g_over.l1_phase.on) s.l1_phase = (mode >= 2 && g_over.l1_phase.v > 0) ? 1 : 0; // (test / diagnostic override: 0 = natural draw order)
        // one big workgroup per CU owns its LDS
        if (p->lds_bytes > lim0) p->threads = 512; // measured (tools/c4_sweep.py): 2 waves/SIMD beat 1 fat and 4 spilling ones
        // ... and as many waves as its registers allow.  With the bins packed as they are drawn and the phased trips unconditional the
        // 32-grid Genz pass needs 146 VGPRs with the gather phase (209 before): 768 threads, 6.97 -> 6.45 ms per 1e8 samples; the 16-grid
        // Gaussian (histogram in the pass, 104 VGPRs) runs 1024 threads: 2.78 -> 2.44 ms (tools/c4_abenv.sh).  compile_solver walks the
        // ladder 1024 -> 768 -> 512 until the code object shows no scratch.
        if (p->lds_bytes > lim0) {
            p->vegas_plan_a = true;
            p->threads_vegas = 1024;
        }
        if (s.hcopy > 1 && !p->vegas_plan_a) { // two 512-thread workgroups per CU (the rule above)
            p->hcopy_plan = true;
            p->threads_vegas = 512;
        }
    }
    p->nstat = 2 * s.nobs + 2 + Nd;
    p->packed_n = p->nstat + s.nbin + 2 * p->npa; // [statistics | histograms | propose | accept]
    p->h_reweight.assign(Nd, 1.0 / Nd); // configuration.jl:110,172-173
    s.body = "w[0] = 1.0;";
    if (!ctx->offline) {
        HIPCHK(hipSetDevice(ctx->device));
        int rc = upload(p);
        if (rc) { delete p; return rc; }
        // (+ 64: the :mcmc holding-time histogram rides behind the tables in the all-reduce, hold_publish)
        HIPCHK(hipMalloc((void **)&p->d_packed, (size_t)(p->packed_n + 64) * sizeof(double)));
        HIPCHK(hipMemset(p->d_packed, 0, (size_t)(p->packed_n + 64) * sizeof(double)));
        // (three buffers: the persistent :vegas kernel rotates through them, mci_train.h vegas_persist; everything else uses the first)
        HIPCHK(hipMalloc((void **)&p->d_ghist, 3 * (size_t)(s.nbin ? s.nbin : 1) * sizeof(double)));
        HIPCHK(hipMemset(p->d_ghist, 0, 3 * (size_t)(s.nbin ? s.nbin : 1) * sizeof(double)));
        HIPCHK(hipMalloc((void **)&p->d_stage1, (size_t)mci_problem::kGroups * (s.nbin ? s.nbin : 1) * sizeof(double)));
        HIPCHK(hipMalloc((void **)&p->d_status, 4 * sizeof(int))); // [0] ST_* bits | [1], [2] serial walks of train! as slots, in the general form (mci_debug_walk_counts)
        HIPCHK(hipMemset(p->d_status, 0, 4 * sizeof(int)));
        std::vector<mci::LeafDev> ld;
        for (auto &L : p->leaves) ld.push_back({L.kind, L.nbin, L.eoff, L.doff, L.boff, L.adapt, L.alpha});
        HIPCHK(hipMalloc((void **)&p->d_leaves, ld.size() * sizeof(mci::LeafDev)));
        HIPCHK(hipMemcpy(p->d_leaves, ld.data(), ld.size() * sizeof(mci::LeafDev), hipMemcpyHostToDevice));
        p->evs.resize(2 * mci_problem::kEvRing);
        for (auto &e : p->evs) HIPCHK(hipEventCreate(&e));
    }
    *out = p;
    return MCI_OK;
}

int mci_problem_destroy(mci_problem *p) {
    if (!p) return MCI_OK;
    if (!p->ctx->offline) {
        (void)hipStreamSynchronize(p->ctx->stream);
        for (void *q : {(void *)p->d_edges, (void *)p->d_dacc, (void *)p->d_ddist, (void *)p->d_reweight, (void *)p->d_ud,
                        (void *)p->d_part_cols, (void *)p->d_part_hist, (void *)p->d_ghist, (void *)p->d_stage1,
                        (void *)p->d_packed, (void *)p->d_scratch, (void *)p->d_iterlog, (void *)p->d_dump,
                        (void *)p->d_status, (void *)p->d_leaves})
            if (q) (void)hipFree(q);
        for (int k = 0; k < mci_problem::kSlots; ++k)
            if (p->module[k]) (void)hipModuleUnload(p->module[k]);
        if (p->module_persist) (void)hipModuleUnload(p->module_persist);
        if (p->d_persist) (void)hipFree(p->d_persist);
    }
    persist_job_drop(p);
    if (!p->ctx->offline) {
        if (p->d_goal) (void)hipFree(p->d_goal);
        if (p->d_part_pa) (void)hipFree(p->d_part_pa);
        if (p->d_hold) (void)hipFree(p->d_hold);
        for (int b = 0; b < 2; ++b) {
            if (p->d_chain_x[b]) (void)hipFree(p->d_chain_x[b]);
            if (p->d_chain_curr[b]) (void)hipFree(p->d_chain_curr[b]);
        }
        if (p->d_reweight_used) (void)hipFree(p->d_reweight_used);
        if (p->d_carry_W) (void)hipFree(p->d_carry_W);
        if (p->d_carry_src) (void)hipFree(p->d_carry_src);
        if (p->d_spec_tab) (void)hipFree(p->d_spec_tab);
        for (int b = 0; b < 2; ++b)
            if (p->d_chain_P[b]) (void)hipFree(p->d_chain_P[b]);
        if (p->d_carry_w) (void)hipFree(p->d_carry_w);
        if (p->d_clocks) (void)hipFree(p->d_clocks);
        if (p->d_edges_backup) (void)hipFree(p->d_edges_backup);
        if (p->h_hold) (void)hipHostFree(p->h_hold);
        if (p->h_hold_d) (void)hipHostFree(p->h_hold_d);
        if (p->h_log) (void)hipHostFree(p->h_log);
        if (p->hold_ev) (void)hipEventDestroy(p->hold_ev);
        if (p->d_blocklog) (void)hipFree(p->d_blocklog);
        for (auto &e : p->cevs) (void)hipEventDestroy(e);
        if (p->d_hx) (void)hipFree(p->d_hx);
        if (p->d_hstep) (void)hipFree(p->d_hstep);
        if (p->h_hidx) (void)hipHostFree(p->h_hidx);
        if (p->d_hw) (void)hipFree(p->d_hw);
        if (p->h_hx) (void)hipHostFree(p->h_hx);
        if (p->h_hw) (void)hipHostFree(p->h_hw);
        if (p->d_tile_w) (void)hipFree(p->d_tile_w);
        if (p->d_tile_bins) (void)hipFree(p->d_tile_bins);
        if (p->d_mx) (void)hipFree(p->d_mx);
        if (p->d_mrelw) (void)hipFree(p->d_mrelw);
        if (p->d_mobs) (void)hipFree(p->d_mobs);
        if (p->h_mx) (void)hipHostFree(p->h_mx);
        if (p->h_mrelw) (void)hipHostFree(p->h_mrelw);
        if (p->d_midx) (void)hipFree(p->d_midx);
        if (p->h_midx) (void)hipHostFree(p->h_midx);
        for (auto &e : p->evs) (void)hipEventDestroy(e);
    }
    delete p;
    return MCI_OK;
}

int mci_set_integrand_source(mci_problem *p, const char *body, const double *ud, int32_t nud) {
    if (!p || !body) return fail(MCI_ERR_INVALID, "NULL argument");
    p->shape.body = body;
    p->shape.host_integrand = 0;
    p->host_fn = nullptr;
    p->h_ud.assign(ud, ud + (nud > 0 ? nud : 0));
    drop_modules(p);
    if (!p->ctx->offline) {
        if (p->d_ud) (void)hipFree(p->d_ud);
        p->d_ud = nullptr;
        HIPCHK(hipMalloc((void **)&p->d_ud, (p->h_ud.size() ? p->h_ud.size() : 1) * sizeof(double)));
        if (p->h_ud.size()) HIPCHK(hipMemcpy(p->d_ud, p->h_ud.data(), p->h_ud.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return MCI_OK;
}

int mci_set_integrand_host(mci_problem *p, mci_host_integrand_fn fn, void *user) {
    if (!p || !fn) return fail(MCI_ERR_INVALID, "NULL argument");
    p->host_fn = fn;
    p->host_user = user;
    p->host_idx_fn = nullptr;
    p->shape.host_integrand = 1;
    p->shape.body = "";
    p->h_ud.clear();
    drop_modules(p);
    if (!p->ctx->offline && !p->d_ud) HIPCHK(hipMalloc((void **)&p->d_ud, sizeof(double)));
    return MCI_OK;
}

int mci_set_integrand_host_indexed(mci_problem *p, mci_host_integrand_idx_fn fn, void *user) {
    if (!p || !fn) return fail(MCI_ERR_INVALID, "NULL argument");
    p->host_idx_fn = fn;
    p->host_fn = nullptr;
    p->host_user = user;
    p->shape.host_integrand = 1;
    p->shape.body = "";
    p->h_ud.clear();
    drop_modules(p);
    if (!p->ctx->offline && !p->d_ud) HIPCHK(hipMalloc((void **)&p->d_ud, sizeof(double)));
    return MCI_OK;
}

// The host closure over n configurations x[k*n + i].  idx == NULL: every integrand, w[(j*ncomp + q)*n + i] (vegas, vegasmc);
// idx != NULL: integrand idx[i] only, w[q*n + i] (mcmc).  Either callback form serves either request.
static int eval_host_integrand(mci_problem *p, const int32_t *idx, const double *x, double *w, int64_t n) {
    const auto &s = p->shape;
    const int nw = s.ni * s.ncomp, nc = s.ncomp;
    int hrc = 0;
    if (!idx) {
        memset(w, 0, (size_t)n * nw * sizeof(double));
        if (p->host_fn) hrc = p->host_fn(x, w, n, s.ndraw, nw, p->host_user);
        else {
            std::vector<int32_t> which((size_t)n);
            for (int j = 0; j < s.ni && !hrc; ++j) {
                std::fill(which.begin(), which.end(), j);
                hrc = p->host_idx_fn(which.data(), x, w + (size_t)j * nc * n, n, s.ndraw, nc, p->host_user);
            }
        }
    } else if (p->host_idx_fn) {
        memset(w, 0, (size_t)n * nc * sizeof(double));
        hrc = p->host_idx_fn(idx, x, w, n, s.ndraw, nc, p->host_user);
    } else {
        p->h_tmp.assign((size_t)n * nw, 0.0);
        hrc = p->host_fn(x, p->h_tmp.data(), n, s.ndraw, nw, p->host_user);
        for (int q = 0; q < nc; ++q)
            for (int64_t i = 0; i < n; ++i) w[(size_t)q * n + i] = idx[i] >= 0 ? p->h_tmp[((size_t)idx[i] * nc + q) * n + i] : 0.0;
    }
    if (hrc) return fail(MCI_ERR_INVALID, "the host integrand failed (%d)", hrc);
    return MCI_OK;
}

int mci_set_measure_source(mci_problem *p, const char *body) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    p->shape.measure_body = body ? body : "";
    p->shape.host_measure = 0;
    p->hmeas_fn = nullptr;
    p->hmeas_idx_fn = nullptr;
    drop_modules(p);
    return MCI_OK;
}

int mci_set_measure_host(mci_problem *p, mci_host_measure_fn fn, void *user) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    p->hmeas_fn = fn;
    p->hmeas_idx_fn = nullptr;
    p->hmeas_user = user;
    p->shape.host_measure = fn ? 1 : 0;
    if (fn) p->shape.measure_body = "";
    drop_modules(p);
    return MCI_OK;
}

int mci_set_measure_host_indexed(mci_problem *p, mci_host_measure_idx_fn fn, void *user) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    p->hmeas_idx_fn = fn;
    p->hmeas_fn = nullptr;
    p->hmeas_user = user;
    p->shape.host_measure = fn ? 1 : 0;
    if (fn) p->shape.measure_body = "";
    drop_modules(p);
    return MCI_OK;
}

int mci_set_launch(mci_problem *p, int32_t threads, int32_t wg_per_block) {
    if (threads > 0) {
        if (threads % 64 || threads > 1024) return fail(MCI_ERR_INVALID, "threads per workgroup must be a multiple of 64, <= 1024");
        p->threads_explicit = true;
        if (threads != p->threads || p->threads_vegas) {
            p->threads = threads;
            p->vegas_plan_a = false; // an explicit size: the vegas kernel follows it
            p->hcopy_plan = false;
            p->threads_vegas = 0;
            // histogram copies are sized for two 512-thread workgroups per CU: smaller workgroups would leave the CU half empty
            p->hcopy_auto = threads >= 512 || g_over.hist_copies.on ? p->hcopy_rule : 1;
            drop_modules(p);
        }
    }
    if (wg_per_block >= 0) p->wg_per_block = wg_per_block;
    return MCI_OK;
}

// What the histogram-copy rule asks of the :vegas kernel the next time it is compiled: copies and workgroup size.  With BOTH opt-in
// streams on (32 bits per draw, seven rounds) the loop is bound by its LDS pipe again, and sixteen copies -- conflict-free, one
// 1024-thread workgroup per CU -- beat eight: 84.4 against 77.5 Gsamples/s on the headline configuration; with one opt-in or none
// eight copies in two 512-thread workgroups win (bench.py: rounds 7: 73.6 against 70.2, 32 bits: 75.5 against 76.2, default: 66.7
// against 61.7).
static int planned_hcopy(const mci_problem *p, int *threads) {
    const auto &s = p->shape;
    int hc = p->hcopy_auto, t = 512;
    if (p->hcopy_plan && hc >= 8 && s.rng_bits == 32 && s.rng_rounds == 7 && p->lds_bytes + (int64_t)s.htile * 8 * 15 <= 159 * 1024) {
        hc = 16;
        t = 1024;
    }
    if (threads) *threads = t;
    return hc;
}

// dynamic LDS of the :vegas sample kernel: the tables (+ the edge cache of the many-grid plans) + its histogram copies
static int64_t vegas_lds(const mci_problem *p) {
    const auto &s = p->shape;
    return (s.ec_doubles > 0 ? p->lds_bytes_k1 : p->lds_bytes) + (int64_t)s.htile * 8 * (s.hcopy - 1);
}

// launches of at most this many partial rows flush their histograms with global atomics (mci_iteration_run)
static const int64_t kAtomicRows = 256;
static bool atomic_rows_ok(const mci_problem *p) { return !p->deterministic && kAtomicRows > 0; }

// workgroup size / dynamic LDS of a solver's sample kernel
static int solver_threads(const mci_problem *p, int solver) {
    if (p->deterministic && p->threads_det[solver]) return p->threads_det[solver];
    return solver == MCI_VEGAS && p->threads_vegas ? p->threads_vegas : p->threads;
}
static int64_t det_lds(const mci_problem *p, int threads) { // deterministic mode: tables + (threads / 64) histogram and observable copies
    const auto &s = p->shape;
    return p->lds_bytes + ((int64_t)s.htile + s.nobs) * 8 * (threads / 64 - 1);
}
static int64_t solver_lds(const mci_problem *p, int solver) {
    if (p->deterministic) return det_lds(p, solver_threads(p, solver));
    return solver == MCI_VEGAS ? vegas_lds(p) : p->lds_bytes;
}

// ---- JIT of the sample-batch kernels ------------------------------------------------------------------------------------
// Kernel slots: 0 :vegas for measurefreq == 1 (the reference's default, main.jl:84: the loop without the carried remainder),
// 1 :vegasmc, 2 :mcmc, 3 :vegas for any measurefreq -- each its own code object, compiled the first time it is needed (a new
// integrand pays for the loop it runs, not for both).  The sample-dump kernel is a fifth, equally lazy one.
enum { kSlotVegasAny = 3, kSlotDump = 4, kSlotVegasmcSpec = 5, kSlotMcmcSpec = 6 };
static int kslot(int solver, int64_t measurefreq) { return solver == MCI_VEGAS && measurefreq != 1 ? kSlotVegasAny : solver; }
static int slot_solver(int slot) { return slot == kSlotVegasAny ? MCI_VEGAS : slot == kSlotVegasmcSpec ? MCI_VEGASMC : slot == kSlotMcmcSpec ? MCI_MCMC : slot; }

namespace {
struct Candidate { // one hiprtc job
    std::string src;
    int threads = 256;
    std::vector<char> code;
    std::string log, path;
    bool cached = false;
    int rc = 0;
    long vgprs() const { return mcijit::kernel_vgprs(code, "mci_vegas_batch"); }
    long scratch() const { return mcijit::kernel_scratch_bytes(code, "mci_vegas_batch"); }
};
// the candidates of a plan are independent translation units: compiled side by side (hiprtc is re-entrant), so a plan that has to
// look at two or three of them before it knows which one runs costs the latency of the slowest, not their sum
void compile_all(std::vector<Candidate *> &cs) {
    std::vector<std::thread> th;
    for (size_t i = 1; i < cs.size(); ++i)
        th.emplace_back([c = cs[i]] { c->rc = mcijit::compile(c->src, c->threads, c->code, c->log, c->cached, &c->path); });
    if (!cs.empty()) cs[0]->rc = mcijit::compile(cs[0]->src, cs[0]->threads, cs[0]->code, cs[0]->log, cs[0]->cached, &cs[0]->path);
    for (auto &t : th) t.join();
}
} // namespace

static int load_slot(mci_problem *p, int slot, Candidate &c, int64_t lds) {
    if (mcijit::max_static_lds_bytes(c.code) != 0) // (mci_device.h draw_leaf: the pair table is addressed from LDS address 0)
        return fail(MCI_ERR_COMPILE, "the code object declares static LDS (%ld bytes): the sample kernels expect their dynamic segment at LDS address 0",
                    mcijit::max_static_lds_bytes(c.code));
    p->code_object[slot] = c.path;
    if (p->ctx->offline) return MCI_OK;
    static const char *const names[mci_problem::kSlots] = {"mci_vegas_batch", "mci_vegasmc_chains", "mci_mcmc_chains", "mci_vegas_batch", "mci_sample_dump",
                                                            "mci_vegasmc_spec", "mci_mcmc_spec"};
    HIPCHK(hipSetDevice(p->ctx->device));
    if (hipModuleLoadData(&p->module[slot], c.code.data()) != hipSuccess) {
        // a cached code object that does not load (truncated by a crash, foreign file): drop it and compile afresh, once
        if (!c.cached) return fail(MCI_ERR_HIP, "hipModuleLoadData failed for a freshly compiled code object");
        unlink(c.path.c_str());
        if (mcijit::compile(c.src, c.threads, c.code, c.log, c.cached, &c.path)) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", c.log.c_str());
        HIPCHK(hipModuleLoadData(&p->module[slot], c.code.data()));
    }
    HIPCHK(hipModuleGetFunction(&p->f_solver[slot], p->module[slot], names[slot]));
    if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)p->f_solver[slot], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (slot == MCI_VEGASMC || slot == kSlotVegasmcSpec) {
        hipFunction_t &fw = p->f_carryw[slot == MCI_VEGASMC ? 0 : 1];
        HIPCHK(hipModuleGetFunction(&fw, p->module[slot], "mci_vegasmc_carry_weights"));
        if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)fw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (slot_solver(slot) == MCI_VEGAS && slot != kSlotDump && p->shape.ntile > 1) {
        HIPCHK(hipModuleGetFunction(&p->f_tiles[slot == kSlotVegasAny ? 1 : 0], p->module[slot], "mci_vegas_tiles"));
        if (p->lds_bytes > 64 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)p->f_tiles[slot == kSlotVegasAny ? 1 : 0], hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    }
    return MCI_OK;
}

// the map + integrand alone (mci_sample_dump, host integrands): its own small code object
static int ensure_dump(mci_problem *p) {
    if (p->compiled[kSlotDump]) return MCI_OK;
    Candidate c;
    mcijit::ProblemShape sh = p->shape;
    sh.hcopy = 1;
    sh.det = 0;
    c.src = mcijit::generate_source(sh, MCI_VEGAS, mcijit::kUnitDump);
    c.threads = 256;
    c.rc = mcijit::compile(c.src, c.threads, c.code, c.log, c.cached, &c.path);
    if (c.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", c.log.c_str());
    int rc = load_slot(p, kSlotDump, c, p->lds_bytes);
    if (rc) return rc;
    p->f_dump = p->f_solver[kSlotDump];
    p->compiled[kSlotDump] = true;
    return MCI_OK;
}

static int compile_solver(mci_problem *p, int slot) {
    if (slot < 0 || slot > kSlotVegasAny) return fail(MCI_ERR_INVALID, "Solver %d is not supported!", slot); // main.jl:263
    if (p->compiled[slot]) return MCI_OK;
    const int solver = slot_solver(slot);
    const int unit = slot == MCI_VEGAS ? mcijit::kUnitVegasMf1 : mcijit::kUnitSolver;
    if (p->shape.measure_body.empty() && !p->shape.host_measure) // vegas/montecarlo.jl:104, mcmc/montecarlo.jl:84
        for (int i = 0; i < p->ni; ++i)
            if (p->shape.obs_bin_draw[i] < 0 && p->shape.obs_nbin[i] != p->shape.ncomp)
                return fail(MCI_ERR_INVALID, "the default measure can only handle observable as Vector with %d scalar elements!", p->ni);
    if (p->deterministic) {
        // one copy of the LDS histograms (and observables) per wave, as many waves as fit: 512 / 256 / 128 / 64 threads
        if (p->shape.ntile > 1 || p->shape.table_mode == 1 || p->shape.table_mode == 2 || p->shape.ec_doubles > 0)
            return fail(MCI_ERR_INVALID, "deterministic mode keeps one copy of the workgroup's histograms per wave in LDS: %d bins (%d tile(s)) do not fit",
                        p->shape.nbin, p->shape.ntile);
        int T = solver == MCI_VEGAS ? 512 : (p->threads < 512 ? p->threads : 512); // (the chain kernels need ~200 registers: 256 threads)
        while (T > 64 && det_lds(p, T) > 159 * 1024) T >>= 1;
        if (det_lds(p, T) > 159 * 1024) return fail(MCI_ERR_INVALID, "deterministic mode: the tables do not fit one CU's LDS");
        p->threads_det[solver] = T;
        p->shape.det = 1;
        p->shape.hcopy = T / 64;
        Candidate c;
        c.src = mcijit::generate_source(p->shape, solver, unit);
        c.threads = T;
        c.rc = mcijit::compile(c.src, c.threads, c.code, c.log, c.cached, &c.path);
        if (c.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", c.log.c_str());
        if (int rc = load_slot(p, slot, c, det_lds(p, T))) return rc;
        p->compiled[slot] = true;
        return MCI_OK;
    }
    p->shape.det = 0;
    static const char *const kVgprKeys = "#define MCI_PIPE_VGPR_KEYS 1\n";
    Candidate chosen;
    if (solver != MCI_VEGAS) {
        chosen.src = mcijit::generate_source(p->shape, solver, unit);
        chosen.threads = p->threads;
        chosen.rc = mcijit::compile(chosen.src, chosen.threads, chosen.code, chosen.log, chosen.cached, &chosen.path);
        if (chosen.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", chosen.log.c_str());
    } else if (p->vegas_planned) {
        // the other measurefreq variant of a kernel whose plan (workgroup size, histogram copies, round keys) stands
        chosen.src = (p->vegas_keys ? std::string(kVgprKeys) : std::string()) + mcijit::generate_source(p->shape, solver, unit);
        chosen.threads = p->threads_vegas ? p->threads_vegas : p->vegas_wide ? 512 : p->threads;
        chosen.rc = mcijit::compile(chosen.src, chosen.threads, chosen.code, chosen.log, chosen.cached, &chosen.path);
        if (chosen.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", chosen.log.c_str());
        if (p->vegas_keys && (chosen.vgprs() > 128 || chosen.scratch() != 0)) { // (this variant carries a few registers more)
            chosen.src = mcijit::generate_source(p->shape, solver, unit);
            chosen.rc = mcijit::compile(chosen.src, chosen.threads, chosen.code, chosen.log, chosen.cached, &chosen.path);
            if (chosen.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", chosen.log.c_str());
        }
    } else {
        const bool hcopy_plan = p->hcopy_plan && !g_over.hist_copies.on;
        int tcopy = 512;
        p->shape.hcopy = planned_hcopy(p, &tcopy);
        if (p->hcopy_plan) p->threads_vegas = tcopy;
        const int T0 = p->threads_vegas ? p->threads_vegas : p->threads;
        // (light integrands: a launch bound of 512 threads costs the plain layout nothing -- see vegas_wide; anything that would need scratch
        // or more than 128 registers under it is compiled for the default size instead)
        const bool try_wide = p->threads == 256 && !p->threads_explicit && !p->deterministic && p->shape.ndraw <= 8 && !p->shape.host_integrand;
        if (hcopy_plan) {
            // Histogram copies pay when the kernel runs four or five waves per SIMD either way (81..128 VGPRs: two 512-thread workgroups
            // share a CU).  More registers: two such workgroups no longer fit.  Fewer: the plain layout runs six or more waves per SIMD
            // in 256-thread workgroups and the 80 KB of copies would cap it at four (C5 :vegas, 78 VGPRs: 1.88 ms per 1e8 samples plain,
            // 2.21 ms with 8 copies; profiles/r02_ablation.txt).  And up to 128 VGPRs registers are free on the copy plan: the pipelined
            // sample loop (mci_device.h draw_sample_pipe) asks for its Philox round keys in VGPRs (20 registers; the all-VGPR v_bitop3_b32
            // issues faster than the form with an SGPR key: C2 1.358 -> 1.331 ms per 1e8 samples) unless that crosses the line.
            // Candidates, compiled side by side: [copies + VGPR keys], [plain layout]; [copies, SGPR keys] only if the first is too fat.
            Candidate keys, plain, nokeys;
            const std::string with_copies = mcijit::generate_source(p->shape, solver, unit);
            keys.src = kVgprKeys + with_copies;
            keys.threads = nokeys.threads = T0;
            nokeys.src = with_copies;
            mcijit::ProblemShape sh = p->shape;
            sh.hcopy = 1;
            plain.src = mcijit::generate_source(sh, solver, unit);
            plain.threads = try_wide ? 512 : p->threads;
            std::vector<Candidate *> both = {&keys, &plain};
            compile_all(both);
            if (keys.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", keys.log.c_str());
            if (plain.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", plain.log.c_str());
            Candidate *copy = &keys;
            p->vegas_keys = true;
            if (keys.vgprs() > 128 || keys.scratch() != 0) {
                nokeys.rc = mcijit::compile(nokeys.src, nokeys.threads, nokeys.code, nokeys.log, nokeys.cached, &nokeys.path);
                if (nokeys.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", nokeys.log.c_str());
                copy = &nokeys;
                p->vegas_keys = false;
            }
            if (copy->vgprs() > 128 || copy->vgprs() <= 80) { // the plain layout
                p->shape.hcopy = 1;
                p->threads_vegas = 0;
                p->vegas_keys = false;
                p->vegas_wide = try_wide && plain.scratch() == 0 && plain.vgprs() <= 128;
                if (try_wide && !p->vegas_wide) {
                    plain.threads = p->threads;
                    plain.rc = mcijit::compile(plain.src, plain.threads, plain.code, plain.log, plain.cached, &plain.path);
                    if (plain.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", plain.log.c_str());
                }
                chosen = std::move(plain);
            } else chosen = std::move(*copy);
        } else if (p->vegas_plan_a) {
            // many-grid plans (one workgroup per CU owns the LDS): the largest of 1024 / 768 / 512 threads at which the sample pass shows
            // no scratch -- the rungs compiled side by side
            Candidate rung[3];
            const std::string src = mcijit::generate_source(p->shape, solver, unit);
            const int ts[3] = {1024, 768, 512};
            std::vector<Candidate *> all;
            for (int i = 0; i < 3; ++i) {
                rung[i].src = src;
                rung[i].threads = ts[i];
                if (ts[i] <= T0) all.push_back(&rung[i]);
            }
            compile_all(all);
            size_t pick = all.size() - 1;
            for (size_t i = 0; i < all.size(); ++i) {
                if (all[i]->rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", all[i]->log.c_str());
                if (all[i]->scratch() == 0) { pick = i; break; }
            }
            p->threads_vegas = all[pick]->threads;
            chosen = std::move(*all[pick]);
        } else {
            chosen.src = mcijit::generate_source(p->shape, solver, unit);
            chosen.threads = try_wide ? 512 : T0;
            chosen.rc = mcijit::compile(chosen.src, chosen.threads, chosen.code, chosen.log, chosen.cached, &chosen.path);
            if (chosen.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", chosen.log.c_str());
            p->vegas_wide = try_wide && chosen.scratch() == 0 && chosen.vgprs() <= 128;
            if (try_wide && !p->vegas_wide) {
                chosen.threads = T0;
                chosen.rc = mcijit::compile(chosen.src, chosen.threads, chosen.code, chosen.log, chosen.cached, &chosen.path);
                if (chosen.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", chosen.log.c_str());
            }
        }
        p->vegas_planned = true;
    }
    int64_t lds = p->lds_bytes;
    if (solver == MCI_VEGAS) {
        lds = vegas_lds(p);
        if (p->shape.ec_doubles > 0 && p->lds_bytes_k1 > lds) lds = p->lds_bytes_k1;
        if (p->lds_bytes > lds) lds = p->lds_bytes;
    }
    if (int rc = load_slot(p, slot, chosen, lds)) return rc;
    p->compiled[slot] = true;
    return MCI_OK;
}

// ---- several lanes per chain (mci_spec.h) ---------------------------------------------------------------------------------
// the chain solver's kernel with a group of lanes per chain: its own code object (slots 5, 6), compiled when a launch first asks for it
static int compile_spec(mci_problem *p, int solver) {
    const int slot = solver == MCI_VEGASMC ? kSlotVegasmcSpec : kSlotMcmcSpec;
    if (p->compiled[slot]) return MCI_OK;
    if (p->shape.measure_body.empty() && !p->shape.host_measure) // vegas/montecarlo.jl:104, mcmc/montecarlo.jl:84
        for (int i = 0; i < p->ni; ++i)
            if (p->shape.obs_bin_draw[i] < 0 && p->shape.obs_nbin[i] != p->shape.ncomp)
                return fail(MCI_ERR_INVALID, "the default measure can only handle observable as Vector with %d scalar elements!", p->ni);
    p->shape.det = 0;
    Candidate c;
    c.src = mcijit::generate_source(p->shape, solver, mcijit::kUnitSpec);
    c.threads = 256; // (a launch of few chains runs one wave per SIMD: up to 512 registers per lane)
    c.rc = mcijit::compile(c.src, c.threads, c.code, c.log, c.cached, &c.path, mcijit::kHdrSpec);
    if (c.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", c.log.c_str());
    if (int rc = load_slot(p, slot, c, p->lds_bytes)) return rc;
    p->compiled[slot] = true;
    return MCI_OK;
}

// The speculation tree of a group of `lanes` lanes: the `lanes` most probable nodes of the accept / reject tree of a chain whose
// steps change its configuration with probability `accept` (greedy: the most probable frontier node next; ties go to the older
// candidate), with at most `limit` accept edges on any way from the root (limit < 0: no bound).  accept -> 0 gives the reject chain,
// accept = 1/2 the complete binary tree.  Nodes are numbered in the order they are taken: ancestors first.
static const int kSpecMaxLevels = 12; // (:mcmc exchanges configurations once per accept level: mci_spec.h spec_wave_max counts below 64)
static void spec_build(int lanes, double accept, int limit, std::vector<mci::SpecNode> &tab, int *maxacc) {
    struct Cand { double prob; int parent; bool via_acc; long seq; };
    std::vector<Cand> front;
    front.push_back({1.0, -1, false, 0});
    long seq = 1;
    tab.clear();
    *maxacc = 0;
    while ((int)tab.size() < lanes && !front.empty()) {
        size_t best = 0;
        for (size_t i = 1; i < front.size(); ++i)
            if (front[i].prob > front[best].prob || (front[i].prob == front[best].prob && front[i].seq < front[best].seq)) best = i;
        const Cand cd = front[best];
        front.erase(front.begin() + (long)best);
        mci::SpecNode nd{};
        if (cd.parent < 0) {
            nd.depth = 0;
            nd.anc = -1;
            nd.nacc = 0;
            nd.needacc = nd.needrej = nd.accdepth = 0ull;
        } else {
            const mci::SpecNode &pn = tab[(size_t)cd.parent];
            nd.depth = pn.depth + 1;
            nd.anc = cd.via_acc ? cd.parent : pn.anc;
            nd.nacc = pn.nacc + (cd.via_acc ? 1 : 0);
            nd.needacc = pn.needacc | (cd.via_acc ? 1ull << cd.parent : 0ull);
            nd.needrej = pn.needrej | (cd.via_acc ? 0ull : 1ull << cd.parent);
            nd.accdepth = pn.accdepth | (cd.via_acc ? 1ull << pn.depth : 0ull);
        }
        const int me = (int)tab.size();
        tab.push_back(nd);
        if (nd.nacc > *maxacc) *maxacc = nd.nacc;
        front.push_back({cd.prob * (1.0 - accept), me, false, seq++});
        if ((limit < 0 || nd.nacc + 1 <= limit) && nd.nacc + 1 <= kSpecMaxLevels) front.push_back({cd.prob * accept, me, true, seq++});
    }
    int deepest = 0;
    for (auto &nd : tab) deepest = nd.depth > deepest ? nd.depth : deepest;
    unsigned long long any = 0ull;
    for (auto &nd : tab) any |= nd.accdepth;
    for (auto &nd : tab) {
        nd.levels = *maxacc | (deepest << 8);
        nd.anydepth = any;
    }
}

// The trees of the next launch on the device (rebuilt when lanes / acceptance / limit change).  accept > 0: that one tree.  accept <= 0
// (the default): the solver's family of trees, one per assumed acceptance -- a group starts on `first` and moves, every few trips, to the
// tree built for the acceptance its chain has shown (mci_spec.h spec_adapt).  :vegasmc proposals do not depend on the configuration they
// start from, an accept level costs one exchange: unbounded; :mcmc runs mcmc_propose once per level: at most `limit` (default 2, 3 on the
// trees for chains that accept most steps).
static int spec_upload(mci_problem *p, int solver, int lanes, double accept, int limit) {
    const double key = accept > 0.0 ? accept : -(double)(solver + 1);
    if (p->d_spec_tab && p->spec_tab_lanes == lanes && p->spec_tab_accept == key && p->spec_tab_limit == limit) return MCI_OK;
    static const double fam_vegasmc[7] = {0.03, 0.12, 0.3, 0.5, 0.7, 0.85, 0.93}, fam_mcmc[6] = {0.03, 0.1, 0.2, 0.35, 0.55, 0.8};
    std::vector<mci::SpecNode> all;
    p->spec_ntree = 0;
    p->spec_tab_maxacc = 0;
    auto add = [&](double acc, int lim) {
        std::vector<mci::SpecNode> tab;
        int maxacc = 0;
        spec_build(lanes, acc, lim, tab, &maxacc);
        all.insert(all.end(), tab.begin(), tab.end());
        p->spec_accepts[p->spec_ntree++] = (float)acc;
        if (maxacc > p->spec_tab_maxacc) p->spec_tab_maxacc = maxacc;
    };
    if (accept > 0.0) {
        add(accept, limit);
        p->spec_first = 0;
    } else if (solver == MCI_VEGASMC) {
        for (double acc : fam_vegasmc) add(acc, limit);
        p->spec_first = 3;
    } else {
        for (double acc : fam_mcmc) add(acc, limit >= 0 ? limit : (acc >= 0.5 ? 3 : 2));
        p->spec_first = 3;
    }
    if (!p->d_spec_tab) HIPCHK(hipMalloc((void **)&p->d_spec_tab, 8 * 64 * sizeof(mci::SpecNode)));
    // (pageable source: the copy has left `all` when the call returns)
    HIPCHK(hipMemcpyAsync(p->d_spec_tab, all.data(), all.size() * sizeof(mci::SpecNode), hipMemcpyHostToDevice, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    p->spec_tab_lanes = lanes;
    p->spec_tab_accept = key;
    p->spec_tab_limit = limit;
    return MCI_OK;
}

int mci_set_chain_speculation(mci_problem *p, int32_t lanes, double accept, int32_t max_accepts) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (lanes != -1 && (lanes < 1 || lanes > 64 || (lanes & (lanes - 1)))) return fail(MCI_ERR_INVALID, "lanes per chain: -1 (automatic), 1 (one lane per chain) or a power of two up to 64");
    if (accept >= 1.0) return fail(MCI_ERR_INVALID, "the acceptance a speculation tree is built for lies in (0, 1); <= 0: the solver's default");
    p->spec_lanes = lanes;
    p->spec_accept = accept > 0.0 ? accept : 0.0;
    p->spec_maxacc = max_accepts < 0 ? -1 : max_accepts;
    return MCI_OK;
}

int mci_last_integrate_discarded(const mci_problem *p, int64_t *neval, int32_t *launches) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (neval) *neval = p->last_discarded_neval;
    if (launches) *launches = p->last_discarded_launches;
    return MCI_OK;
}

int mci_last_chain_speculation(const mci_problem *p, int32_t *lanes, int32_t *max_accepts) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (lanes) *lanes = p->last_spec_lanes;
    if (max_accepts) *max_accepts = p->last_spec_maxacc;
    return MCI_OK;
}

int mci_speculation_tree(int32_t lanes, double accept, int32_t max_accepts, int32_t *depth, int32_t *anc, int32_t *nacc, uint64_t *needacc, uint64_t *needrej) {
    if (lanes < 1 || lanes > 64 || !(accept > 0.0 && accept < 1.0)) return fail(MCI_ERR_INVALID, "speculation tree: 1..64 lanes, acceptance in (0, 1)");
    std::vector<mci::SpecNode> tab;
    int maxacc = 0;
    spec_build(lanes, accept, max_accepts, tab, &maxacc);
    for (int i = 0; i < lanes; ++i) {
        if (depth) depth[i] = tab[(size_t)i].depth;
        if (anc) anc[i] = tab[(size_t)i].anc;
        if (nacc) nacc[i] = tab[(size_t)i].nacc;
        if (needacc) needacc[i] = tab[(size_t)i].needacc;
        if (needrej) needrej[i] = tab[(size_t)i].needrej;
    }
    return MCI_OK;
}

int mci_compile_chain_speculation(mci_problem *p, int32_t solver) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (solver != MCI_VEGASMC && solver != MCI_MCMC) return fail(MCI_ERR_INVALID, "several lanes per chain: solver MCI_VEGASMC or MCI_MCMC");
    if (p->shape.host_integrand) return fail(MCI_ERR_INVALID, "a host integrand keeps one lane per chain");
    return compile_spec(p, solver);
}

int mci_compile(mci_problem *p) { return compile_solver(p, MCI_VEGAS); }

int mci_kernel_code_object(mci_problem *p, int32_t solver, char *buf, int32_t n) {
    if (!p || !buf || n < 1) return fail(MCI_ERR_INVALID, "NULL argument");
    if (solver == MCI_VEGAS_PERSISTENT) {
        if (!p->persist_compiled) return fail(MCI_ERR_INVALID, "the persistent :vegas kernel has not been compiled yet");
        snprintf(buf, (size_t)n, "%s", p->persist_code_object.c_str());
        return MCI_OK;
    }
    if (solver == MCI_VEGASMC_LANES || solver == MCI_MCMC_LANES) {
        const int sl = solver == MCI_VEGASMC_LANES ? kSlotVegasmcSpec : kSlotMcmcSpec;
        if (!p->compiled[sl]) return fail(MCI_ERR_INVALID, "the several-lanes-per-chain kernel has not been compiled yet");
        snprintf(buf, (size_t)n, "%s", p->code_object[sl].c_str());
        return MCI_OK;
    }
    if (solver < 0 || solver > 2) return fail(MCI_ERR_INVALID, "Solver %d is not supported!", solver);
    const int slot = (solver == MCI_VEGAS && !p->compiled[solver] && p->compiled[kSlotVegasAny]) ? kSlotVegasAny : solver;
    if (!p->compiled[slot]) return fail(MCI_ERR_INVALID, "solver %d has not been compiled yet", solver);
    snprintf(buf, (size_t)n, "%s", p->code_object[slot].c_str());
    return MCI_OK;
}

int mci_set_rng_bits(mci_problem *p, int32_t bits) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (bits != 52 && bits != 32) return fail(MCI_ERR_INVALID, "rng bits must be 52 (default: the resolution of rand(Float64)) or 32");
    if (p->shape.rng_bits != bits) {
        p->shape.rng_bits = bits;
        drop_modules(p);
    }
    return MCI_OK;
}

int mci_set_rng_rounds(mci_problem *p, int32_t rounds) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (rounds != 10 && rounds != 7) return fail(MCI_ERR_INVALID, "Philox4x32 rounds must be 10 (default) or 7 (the fewest that pass BigCrush)");
    if (p->shape.rng_rounds != rounds) {
        p->shape.rng_rounds = rounds;
        drop_modules(p);
    }
    return MCI_OK;
}

int mci_set_train_walk(mci_problem *p, int32_t mode) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (mode < -1 || mode > 2) return fail(MCI_ERR_INVALID, "train walk mode must be -1 (automatic), 0 (prefix scan), 1 (serial recurrence) or 2 (serial recurrence, general form only)");
    p->train_serial = mode;
    return MCI_OK;
}

// csrc/mci_debug.h
int mci_debug_plant_wrong_decision(mci_problem *p, int32_t on) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    p->debug_wrong_decision = on != 0;
    return MCI_OK;
}

int mci_debug_override(const char *key, int64_t value, int32_t on) {
    Override *o = override_slot(key);
    if (!o) return fail(MCI_ERR_INVALID, "no such override: %s", key ? key : "(null)");
    o->on = on != 0;
    o->v = value;
    return MCI_OK;
}

int mci_debug_mcmc_policy(int64_t pilot_steps, int64_t grow, int64_t carry_holds, int64_t carry_half_floors) {
    if (pilot_steps > 0) mci_problem::kMcmcPilotSteps = pilot_steps;
    if (grow > 0) mci_problem::kMcmcGrow = grow;
    if (carry_holds > 0) mci_problem::kMcmcCarryHolds = carry_holds;
    if (carry_half_floors > 0) mci_problem::kMcmcCarryHalfFloors = carry_half_floors;
    return MCI_OK;
}

int mci_debug_persist_spin_ticks(mci_problem *p, unsigned long long ticks) {
    if (!p || ticks == 0) return fail(MCI_ERR_INVALID, "bad argument");
    p->persist_spin_ticks = ticks;
    return MCI_OK;
}

int mci_set_deterministic(mci_problem *p, int32_t on) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    const bool want = on != 0;
    if (want != p->deterministic) {
        p->deterministic = want;
        p->shape.det = want ? 1 : 0;
        drop_modules(p);
    }
    return MCI_OK;
}

int mci_set_chain_carry(mci_problem *p, int32_t mode) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (mode < -1 || mode > 1) return fail(MCI_ERR_INVALID, "chain carry mode must be -1 (automatic) or 1 (many-chain launches of :vegasmc and :mcmc continue the chains of the iteration before) or 0 (every launch starts its chains afresh)");
    p->chain_carry = mode;
    if (mode == 0) p->chain_valid = false;
    return MCI_OK;
}

int mci_set_iteration_counted(mci_problem *p, int32_t counted) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    p->launch_counted = counted != 0;
    return MCI_OK;
}

int mci_set_persistent(mci_problem *p, int32_t mode) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (mode < -1 || mode > 1) return fail(MCI_ERR_INVALID, "persistent mode must be -1 (automatic: launch-bound :vegas calls), 0 (one launch chain per iteration) or 1 (whenever the layout allows)");
    p->persistent = mode;
    return MCI_OK;
}

// development aid (tools/persist_trace.py): the raw counter / stamp words of the persistent kernel
int mci_debug_persist_words(mci_problem *p, unsigned long long *out, int32_t n) {
    if (!p || !out || !p->d_persist) return fail(MCI_ERR_INVALID, "no persistent launch yet");
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    HIPCHK(hipMemcpy(out, p->d_persist, (size_t)(n < (int)kPersistWords ? n : (int)kPersistWords) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MCI_OK;
}

// development aid (tools/fuzz_layouts.py --walk): how many serial walks of train! ran as slots with given decisions, how many in the general form
int mci_debug_walk_counts(mci_problem *p, int64_t *out) {
    if (!p || !out || !p->d_status) return fail(MCI_ERR_INVALID, "NULL argument");
    int h[2] = {0, 0};
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    HIPCHK(hipMemcpy(h, p->d_status + 1, sizeof(h), hipMemcpyDeviceToHost));
    out[0] = h[0];
    out[1] = h[1];
    return MCI_OK;
}

int mci_last_integrate_persistent(const mci_problem *p, int32_t *persistent) {
    if (!p || !persistent) return fail(MCI_ERR_INVALID, "NULL argument");
    *persistent = p->last_persistent ? 1 : 0;
    return MCI_OK;
}

int mci_last_chain_launch(const mci_problem *p, int64_t *nchain, int32_t *carried) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (nchain) *nchain = p->last_nchain;
    if (carried) *carried = p->last_carried ? 1 : 0;
    return MCI_OK;
}

int mci_check_status(mci_problem *p) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    return check_status(p);
}
static int compile_persist(mci_problem *p, bool background);
static bool persist_layout_ok(const mci_problem *p);
int mci_compile_solver(mci_problem *p, int32_t solver) {
    if (solver == MCI_VEGAS_PERSISTENT) { // the persistent :vegas kernel (mci_set_persistent), for layouts that allow it
        if (!persist_layout_ok(p)) return fail(MCI_ERR_INVALID, "this layout has no persistent :vegas kernel (mci_set_persistent)");
        return compile_persist(p, false);
    }
    if (solver == MCI_VEGASMC_LANES || solver == MCI_MCMC_LANES) return mci_compile_chain_speculation(p, solver == MCI_VEGASMC_LANES ? MCI_VEGASMC : MCI_MCMC);
    if (solver < 0 || solver > 2) return fail(MCI_ERR_INVALID, "Solver %d is not supported!", solver); // main.jl:263
    return compile_solver(p, solver);
}

int mci_get_histogram_copies(const mci_problem *p, int32_t *copies) {
    if (!p || !copies) return fail(MCI_ERR_INVALID, "NULL argument");
    *copies = (p->compiled[MCI_VEGAS] || p->compiled[kSlotVegasAny]) ? p->shape.hcopy : planned_hcopy(p, nullptr);
    return MCI_OK;
}

int mci_problem_info(const mci_problem *p, int32_t *ndraw, int32_t *nobs, int64_t *packed_size, int32_t *table_mode, int64_t *lds_bytes) {
    if (ndraw) *ndraw = p->shape.ndraw;
    if (nobs) *nobs = p->shape.nobs;
    if (packed_size) *packed_size = p->packed_n;
    if (table_mode) *table_mode = p->shape.table_mode;
    if (lds_bytes) *lds_bytes = p->lds_bytes;
    return MCI_OK;
}

// ---------------------------------------------------------------------------------------------------
// one iteration
// ---------------------------------------------------------------------------------------------------
int mci_iteration_run(mci_problem *p, int32_t solver, int64_t nevalperblock, int64_t block_lo, int64_t block_hi,
                      int32_t iteration, uint64_t seed, int64_t measurefreq, int64_t nchain, double thermal_ratio) {
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context: no device to run on");
    if (solver != MCI_VEGAS && solver != MCI_VEGASMC && solver != MCI_MCMC) return fail(MCI_ERR_INVALID, "Solver %d is not supported!", solver); // main.jl:263
    const bool auto_chains = nchain <= 0; // (the holding times of an :mcmc launch are handed to the host only when the next one may size its chains from them)
    if (measurefreq <= 0) return fail(MCI_ERR_INVALID, "measurefreq must be positive"); // vegas/montecarlo.jl:77
    const int64_t nblocks = block_hi - block_lo;
    if (nblocks < 1 || nevalperblock < 1) return fail(MCI_ERR_INVALID, "empty iteration");
    if (p->has_fermik && solver != MCI_MCMC) return fail(MCI_ERR_INVALID, "FermiK variables work with solver=:mcmc only"); // test/bubble_FermiK.jl:2,:133
    const int kern = kslot(solver, measurefreq);
    // (a chain solver's lane-per-chain kernel is compiled once the launch is known to run one lane per chain: a launch of few chains
    // runs the several-lanes-per-chain kernel instead, mci_spec.h, and pays for that code object only)
    int rc = (solver == MCI_VEGAS || p->deterministic || p->shape.host_integrand || p->spec_lanes == 1) ? compile_solver(p, kern) : MCI_OK;
    if (rc) return rc;
    if (solver == MCI_VEGAS && p->shape.host_integrand && (rc = ensure_dump(p))) return rc;
    if ((rc = flush_merge(p))) return rc; // a previous batch nobody looked at: merge it (resets the global histogram)
    HIPCHK(hipSetDevice(p->ctx->device));
    const auto &s = p->shape;
    int T = solver_threads(p, solver);
    // mid-size :vegas launches of a plain-layout kernel compiled for it: 512-thread workgroups (mci_problem::vegas_wide)
    if (solver == MCI_VEGAS && p->vegas_wide && !p->threads_vegas && p->wg_per_block <= 0 && nblocks * nevalperblock < ((int64_t)1 << 22) &&
        nblocks * nevalperblock * p->shape.ndraw >= ((int64_t)1 << 19))
        T = 512;
    int64_t units = nevalperblock; // lanes of useful work per block
    if (solver != MCI_VEGAS && (block_hi > 4096 || iteration >= 131072 || iteration < 0))
        return fail(MCI_ERR_INVALID, "chain solvers address a chain by (block < 4096, iteration < 131072): got block_hi=%lld, iteration=%d",
                    (long long)block_hi, (int)iteration);
    double burnin = 0.0;
    int64_t nburn = 0;
    // Does this launch continue the chains of the previous one?  (the next iteration of the same solver over the same blocks;
    // decided before the chains are sized -- carried chains start from configurations that are already distributed like
    // the chain's target, so they neither need the many-chain burn-in floors nor their length as a safety margin against start-up bias)
    // (:mcmc: a chain's state includes the integrand index, whose weight doReweight! moves between iterations -- the stored chains are
    // resampled to the moved target first, k_resample_chains below.  Chains carried as they were started over-represented exactly where
    // the new factors say "fewer": 2 sigma per run low on the 12-D member of BASELINE configs[4], profiles/r03_chain_carry.txt.)
    // (:vegasmc: not out of a launch on the untrained map onto a refined one -- chains of the automatic length have not reached their
    // target there, and no resampling turns them into a sample of the new one, profiles/r05_bias.txt A4; while the map stays as it is
    // -- adapt = false -- they go on towards the same target)
    const bool carry_on = p->chain_carry != 0;
    const bool may_carry = solver != MCI_VEGAS && carry_on && p->chain_valid && p->chain_solver == solver &&
                           p->chain_lo == block_lo && p->chain_hi == block_hi && p->chain_nchain > 1 &&
                           (solver != MCI_VEGASMC || p->chain_ntrain >= 1 || p->chain_ntrain == p->ntrain) &&
                           ((p->chain_iteration & (kRepeatStride - 1)) + 1 == (iteration & (kRepeatStride - 1)) ||                        // the next iteration
                            ((p->chain_iteration & (kRepeatStride - 1)) == (iteration & (kRepeatStride - 1)) && iteration > p->chain_iteration)); // ... or the same one again (mci_integrate, warm-up)
    if (solver == MCI_VEGASMC) {
        int nslots = 0; // (pool, slot) pairs changeVariable can pick (updates.jl:50,:58)
        for (int v = 0; v < p->npool; ++v) nslots += p->maxdof[v];
        if (nchain <= 0) { // auto: as many chains as keep 2 waves per SIMD busy (kChainFill lanes per GPU, tools/chain_sweep.py),
            // but never shorter than 8 burn-in floors.  Short chains under-sample the sticky high-|f|/q states of
            // singular integrands: measured on 1/(1 - cos x cos y cos z) at 2e9 steps, 381-step chains are 6 sigma low,
            // 763-step chains are within 1.4 sigma (tools/chain_bias_c1.py).
            // Carried chains are stationary from their first step: two floors per iteration let them settle on the refined map.
            const int64_t fl = 64 * (int64_t)nslots > 128 ? 64 * (int64_t)nslots : 128;
            // A launch on a map train! has never refined whose estimate COUNTS (mci_integrate with ignore = 0: adapt = false, main.jl:82)
            // runs chains 8 x as long: on the untrained map chains of 8 floors have not reached their target -- 3.4 sigma per run low on
            // the 12-D member of BASELINE configs[4], 5 on 1/(1 - cos x cos y cos z), with every iteration counted; with 64 floors
            // within errors (profiles/r05_bias.txt A5, A6).  The default call ignores that iteration and keeps the short ones.
            const int64_t fresh = g_over.fresh_floors.on ? g_over.fresh_floors.v : (p->launch_counted && p->ntrain == 0) ? 64 : 8;
            nchain = nevalperblock / ((may_carry ? 2 : fresh) * fl);
            const int64_t cap = mci_problem::kChainFill / nblocks > 64 ? mci_problem::kChainFill / nblocks : 64;
            if (nchain > cap) nchain = cap;
            if (nchain < 1) nchain = 1;
        }
        if (nchain > nevalperblock) return fail(MCI_ERR_INVALID, "nchain=%lld exceeds the %lld steps of a block", (long long)nchain, (long long)nevalperblock);
        // (carried chains keep the reference's own `ne >= neval/100` only, vegas_mc/montecarlo.jl:213)
        burnin = mci_chain_burnin(nevalperblock / nchain, (may_carry && nchain > 1) ? 1 : nchain, nslots);
        if (g_over.fresh_burnin_pct.on && !may_carry && nchain > 1 && auto_chains) { // (experiment: tools/run_r05_floors.sh)
            const double b = (double)(nevalperblock / nchain) * (double)g_over.fresh_burnin_pct.v / 100.0;
            if (b > burnin) burnin = b;
        }
        units = nchain;
    } else if (solver == MCI_MCMC) {
        int nslots = 0;
        for (int v = 0; v < p->npool; ++v) nslots += p->maxdof[v];
        if (!(thermal_ratio >= 0.0)) return fail(MCI_ERR_INVALID, "thermal_ratio must be non-negative");
        if (nchain <= 0) { // auto: LONG chains.  The walk over (integrand, variables) mixes slowly when |f|/q is heavy-tailed:
            // on the bubble diagram 1e3-step chains are 2.7 % (55 sigma) off at 2e9 steps and need ~1e5 burn-in steps each
            // to lose that bias (tools/bubble_mcmc_bias.py); only chains much longer than the mixing time are safe, which
            // is what the reference's one-chain-per-block gives.  More chains: raise `block` (the reference's own knob) or
            // pass nchain explicitly for integrands known to mix fast (C5: 10 Gsteps/s at nchain = 4096).
            // From the second :mcmc launch of a problem on, the length follows what the previous launch measured: 16 x the
            // longest time any chain's slot (or integrand index) went without changing (mci_mcmc_auto_chains).
            // Carried chains (resampled to the moved target, k_resample_chains) start from stationary configurations AND a stationary
            // integrand index: nothing to burn in.  What their length still has to cover is the longest holding time: a population
            // grows by duplication (a launch of more chains than the one before continues every stored chain several times), and the
            // copies of a chain must have gone their own ways before they are copied again -- 4 x the longest hold instead of the
            // 16 x (+ burn-in) of fresh chains.  profiles/r03_chain_carry.txt: carried chains of two burn-in floors on 1/(1 - cos^3)
            // keep their few ancestors' view of its sticky states for many iterations (-4.8 sigma pooled over 64 seeds); at 2, 4
            // and 16 x the hold the pooled deviations are those of fresh chains.  profiles/r04_mcmc_policy.txt D: 4 x against the 8 x of
            // round 3 on 384-512 seeds (same pulls, same scatter / error; 2 x: the error bars start to fall short).
            // The holds are those of the launch BEFORE this one (hold_consume waits for its sample kernel); a first launch, with nothing
            // measured, runs pilot-length chains, and a launch's chains are at most kMcmcGrow times as long as those that measured the
            // holds (mci_mcmc_auto_chains).
            if ((rc = hold_consume(p))) return rc;
            // (once warm: the larger of the last two launches' holds, and no growth cap -- both were measured by chains that held them)
            const int64_t hold_eff = p->mcmc_warm && p->hold_prev > p->hold_max ? p->hold_prev : p->hold_max;
            nchain = mci_mcmc_auto_chains(nevalperblock, nblocks, nslots, p->ni + 1, p->npool, hold_eff, p->mcmc_warm && p->hold_valid ? 0 : p->hold_len,
                                          may_carry ? 1 : 0);
        }
        if (nchain > nevalperblock) return fail(MCI_ERR_INVALID, "nchain=%lld exceeds the %lld steps of a block", (long long)nchain, (long long)nevalperblock);
        // (carried chains have no start to burn in: floor(steps * thermal_ratio), mcmc/montecarlo.jl:133, is the burn-in of a chain that
        // begins at a random configuration; a chain that continues a stationary one measures from its first step)
        nburn = (may_carry && nchain > 1) ? 0 : mci_mcmc_burnin(nevalperblock / nchain, nchain, nslots, p->ni + 1, p->npool, thermal_ratio);
        units = nchain;
    } else {
        nchain = 1;
    }
    // Several lanes per chain (mci_spec.h): a launch whose chains leave most of the chip idle gives every chain a group of G lanes that
    // step it speculatively -- the same chain, G <= 64 proposals evaluated per trip.  Automatic: the largest G that keeps the launch
    // within one wave per SIMD (kSpecFill lanes).  Host integrands keep the lock-step launches; the deterministic mode one lane per chain.
    int G = 1, spec_maxacc = 0;
    if (solver != MCI_VEGAS && !s.host_integrand && !p->deterministic && p->spec_lanes != 1) {
        if (p->spec_lanes > 1) G = p->spec_lanes;
        else {
            G = 64;
            while (G > 1 && nblocks * nchain * G > mci_problem::kSpecFill) G >>= 1;
            // (groups of 2 and 4 lanes lose: a trip costs more than a lane-per-chain step and advances barely more -- BASELINE configs[4],
            // 24400 pilot chains: 32.3 ms with 2 lanes per chain against 21.8; the bubble diagram 3.5 | 2.15 | 1.1 us per step at 4 | 16 | 64
            // lanes against 5.6 with one, profiles/r05_spec.txt)
            if (G < 8) G = 1;
        }
    }
    int T_launch = T;
    if (G > 1) {
        // the trees: the one built for the acceptance that was given, else the solver's family (spec_upload)
        if ((rc = compile_spec(p, solver))) return rc;
        if ((rc = spec_upload(p, solver, G, p->spec_accept, p->spec_maxacc))) return rc;
        spec_maxacc = p->spec_tab_maxacc;
        units = nchain * G;
        T_launch = units >= 256 ? 256 : (int)((units + 63) / 64) * 64;
    }
    if (G == 1 && (rc = compile_solver(p, kern))) return rc;
    p->last_spec_lanes = G;
    p->last_spec_maxacc = spec_maxacc;
    int wpb = p->wg_per_block;
    if (G > 1) {
        if (wpb <= 0) wpb = (int)((2048 + nblocks - 1) / nblocks);
        const int64_t maxw = (units + T_launch - 1) / T_launch;
        if (wpb > maxw) wpb = (int)maxw;
        if (wpb < 1) wpb = 1;
    } else
    if (wpb <= 0) { // 256 CUs x 8..16 workgroups in the grid, never a workgroup without work
        // measured on C2 (workgroup-count sweep): 16 workgroups per CU even out the tail once a launch is long
        // enough that the extra partial rows (merged by k_hist_stage1) do not matter
        // (only while a workgroup's tables are cheap to stage: C3 with 66 KB per workgroup lost 15 % at 4096)
        // (... counted in 256-thread workgroups: the 512-thread workgroups of the histogram-copy plan take half as many -- warm
        // tools/ab_c2.py, C2: 1024 / 2048 / 4096 / 8192 workgroups 1.509 / 1.504 / 1.515 / 1.551 ms per iteration)
        const int64_t big = T >= 1024 ? 1024 : T >= 512 ? 2048 : 4096;
        int64_t target = (units * nblocks >= (int64_t)1 << 25 && p->lds_bytes <= 32 * 1024) ? big : 2048;
        // :vegas launches of up to a few million samples: a workgroup's prologue and epilogue (tables staged, histogram zeroed and
        // flushed) cost what ~50 samples per thread cost, so the grid shrinks to one workgroup per CU (tools/latency.py, us per
        // iteration at neval = 1e6: 2048 workgroups 39.9, 512: 27.7, 256: 26.9; C2 at 1e6: 64.8 -> 43.9).  Longer launches keep the
        // full grid: a grid between 256 and 512 workgroups leaves half of the CUs' second slot empty (C2 at 1e7: 320 workgroups
        // 271.7 us, 2048: 210.1)
        if (solver == MCI_VEGAS && units * nblocks < ((int64_t)1 << 22) && target > 256) target = 256;
        // ... and light launches (samples x draws below 2^19: a 2-D integrand at neval = 1e5) to a quarter of the CUs: their prologues and
        // epilogues weigh more than a few more samples per lane (tools/latency.py, x^2 + y^2 at 1e5: 22.0 -> 18.6 us per iteration; the
        // 16-D Gaussian at 1e5 keeps the full 256: 23.4 against 25.9 us)
        if (solver == MCI_VEGAS && units * nblocks * s.ndraw < ((int64_t)1 << 19) && target > 64) target = 64;
        wpb = (int)((target + nblocks - 1) / nblocks);
        const int64_t maxw = (units + T - 1) / T;
        if (wpb > maxw) wpb = (int)maxw;
        if (wpb < 1) wpb = 1;
    }
    const bool hist_lds = (s.table_mode == 0 || s.table_mode == 3);
    // Few partial rows (launch-bound :vegas iterations): no partial histograms, no first merge launch -- the workgroups add their
    // non-zero bins to the merged histogram directly (global f64 atomics; the order of those adds follows the hardware, so the
    // deterministic mode keeps the fixed-order merge).  tools/latency.py, us per iteration: x^2 + y^2 at neval = 1e4 22.7 -> 17-19,
    // 1e5 23.8 -> 18.6, 1e6 26.5 -> 24.4; 16-D Gaussian at 1e5 27.3 -> 23.4, 1e6 41.8 -> 37.2.
    // NTILE > 1 histogram tiles.  vegas: ONE sample pass (tile 0) parks weights + bins per sample, mci_vegas_tiles
    // replays them for the other tiles.  Chain solvers: NTILE workgroups per row, each recomputing the chain and
    // keeping one tile.
    const bool split = solver == MCI_VEGAS && s.ntile > 1;
    if (!split && wpb * s.ntile > 4096 / nblocks && s.ntile > 1) wpb = (int)(4096 / nblocks / s.ntile) > 0 ? (int)(4096 / nblocks / s.ntile) : 1;
    const int64_t nrows = nblocks * wpb;   // partial rows: one per (block, slice)
    const bool atomic_flush = solver == MCI_VEGAS && hist_lds && s.ntile == 1 && atomic_rows_ok(p) && nrows <= kAtomicRows && !s.host_integrand;
    const int64_t nwg = split ? nrows : nrows * s.ntile;
    if ((rc = ensure_capacity(p, nrows, nblocks))) return rc;
    if (solver != MCI_VEGAS && nrows > p->cap_pa) {
        if (p->d_part_pa) (void)hipFree(p->d_part_pa);
        p->d_part_pa = nullptr;
        p->cap_pa = 0;
        HIPCHK(hipMalloc((void **)&p->d_part_pa, (size_t)nrows * 2 * p->npa * sizeof(double)));
        p->cap_pa = nrows;
    }
    if (split) {
        const int64_t nsamp = nblocks * nevalperblock;
        if (nsamp > p->cap_tile) {
            if (p->d_tile_w) (void)hipFree(p->d_tile_w);
            if (p->d_tile_bins) (void)hipFree(p->d_tile_bins);
            p->d_tile_w = nullptr;
            p->d_tile_bins = nullptr;
            p->cap_tile = 0;
            HIPCHK(hipMalloc((void **)&p->d_tile_w, (size_t)nsamp * s.ni * sizeof(double)));
            HIPCHK(hipMalloc((void **)&p->d_tile_bins, (size_t)nsamp * ((p->ntdraw + 1) / 2 > 0 ? (p->ntdraw + 1) / 2 : 1) * sizeof(uint32_t)));
            p->cap_tile = nsamp;
        }
    }
    mci::BatchArgs a{};
    a.edges = p->d_edges;
    a.dacc = p->d_dacc;
    a.ddist = p->d_ddist;
    a.reweight = p->d_reweight;
    a.ud = p->d_ud;
    a.part_cols = p->d_part_cols;
    a.part_hist = p->d_part_hist;
    a.ghist = p->d_ghist;
    a.part_pa = p->d_part_pa;
    a.seed = seed;
    a.iteration = (mci::u32)iteration;
    a.neval_per_block = nevalperblock;
    a.block_lo = block_lo;
    a.wg_per_block = wpb;
    a.measurefreq = measurefreq;
    a.nchain = nchain;
    a.burnin = burnin;
    a.nburn = nburn;
    // (three buffers from 64 rows on: x^2 + y^2 at neval = 1e6, 256 rows: see tools/latency.py)
    const int ghist_buffers = atomic_flush ? (nrows > 64 ? 3 : 1) : 0;
    a.hist_atomic = ghist_buffers;
    if (solver != MCI_VEGAS) {
        const bool carried = may_carry && nchain > 1;
        const bool keep = carry_on && nchain > 1;
        if (carried) {
            a.carry_x = p->d_chain_x[p->chain_cur];
            a.carry_curr = p->d_chain_curr[p->chain_cur];
            a.carry_nchain = p->chain_nchain;
            a.carry_cap = p->chain_cap[p->chain_cur];
        }
        if (carried) { // which stored chain each chain continues: the stored ones resampled to the moved target
            if (nblocks * nchain > p->cap_carry_src) {
                if (p->d_carry_src) (void)hipFree(p->d_carry_src);
                p->d_carry_src = nullptr;
                p->cap_carry_src = 0;
                HIPCHK(hipMalloc((void **)&p->d_carry_src, (size_t)(nblocks * nchain) * sizeof(int)));
                p->cap_carry_src = nblocks * nchain;
            }
            if (nblocks * p->chain_nchain > p->cap_carry_W) {
                if (p->d_carry_W) (void)hipFree(p->d_carry_W);
                p->d_carry_W = nullptr;
                p->cap_carry_W = 0;
                HIPCHK(hipMalloc((void **)&p->d_carry_W, (size_t)(nblocks * p->chain_nchain) * sizeof(double)));
                p->cap_carry_W = nblocks * p->chain_nchain;
            }
            mci::ResampleArgs ra{};
            ra.curr_old = p->d_chain_curr[p->chain_cur];
            ra.n_old = p->chain_nchain;
            ra.n_new = nchain;
            ra.nd = p->ni + 1;
            ra.rw_now = p->d_reweight;
            ra.rw_used = p->d_reweight_used;
            ra.src = p->d_carry_src;
            ra.W = p->d_carry_W;
            if (solver == MCI_VEGASMC) {
                // :vegasmc: the target itself moved with the map and the reweight factors -- pi_new / pi_old at every stored configuration
                // (the chain kernel's own code object evaluates it: relocate, integrand, paddings), then the same systematic resampling
                const int64_t total = nblocks * p->chain_nchain;
                if (total > p->cap_carry_w) {
                    if (p->d_carry_w) (void)hipFree(p->d_carry_w);
                    p->d_carry_w = nullptr;
                    p->cap_carry_w = 0;
                    HIPCHK(hipMalloc((void **)&p->d_carry_w, (size_t)total * sizeof(double)));
                    p->cap_carry_w = total;
                }
                a.carry_P = p->d_chain_P[p->chain_cur];
                a.carry_w = p->d_carry_w;
                a.carry_total = total;
                mci::BatchArgs wa = a; // (edges, tables, reweight, userdata and the carry fields; everything else unused)
                struct Scratch { // (freed on every way out of this block, the failing ones included)
                    double *p = nullptr;
                    ~Scratch() { if (p) (void)hipFree(p); }
                } cw; // a host closure: evaluated at the stored configurations here, one more callback per iteration
                double *&d_cw = cw.p;
                if (s.host_integrand) {
                    const int nw = s.ni * s.ncomp;
                    std::vector<double> hx((size_t)total * s.ndraw), hw((size_t)total * nw);
                    for (int k = 0; k < s.ndraw; ++k)
                        HIPCHK(hipMemcpyAsync(hx.data() + (size_t)k * total, a.carry_x + (size_t)k * a.carry_cap, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
                    HIPCHK(hipStreamSynchronize(p->ctx->stream));
                    if ((rc = eval_host_integrand(p, nullptr, hx.data(), hw.data(), total))) return rc;
                    HIPCHK(hipMalloc((void **)&d_cw, hw.size() * sizeof(double)));
                    HIPCHK(hipMemcpyAsync(d_cw, hw.data(), hw.size() * sizeof(double), hipMemcpyHostToDevice, p->ctx->stream));
                    HIPCHK(hipStreamSynchronize(p->ctx->stream)); // (`hw` leaves scope)
                    wa.host_w = d_cw;
                }
                void *wargs[] = {&wa};
                const int64_t wgrid = (total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048;
                const int tw = G > 1 ? 256 : (T < 256 ? T : 256); // (within the launch bound its code object was compiled for)
                HIPCHK(hipModuleLaunchKernel(p->f_carryw[G > 1 ? 1 : 0], (unsigned)wgrid, 1, 1, (unsigned)tw, 1, 1, (unsigned)p->lds_bytes, p->ctx->stream, wargs, nullptr));
                if (d_cw) HIPCHK(hipStreamSynchronize(p->ctx->stream)); // (the kernel has read it before `cw` lets go of it)
                ra.w_chain = p->d_carry_w;
            }
            hipLaunchKernelGGL(mci::k_resample_chains, dim3((unsigned)nblocks), dim3(256), 0, p->ctx->stream, ra);
            HIPCHK(hipGetLastError());
            a.carry_src = p->d_carry_src;
        }
        if (keep && solver == MCI_MCMC) { // the reweight factors this launch's chains run under (doReweight! moves them behind it)
            if (!p->d_reweight_used) HIPCHK(hipMalloc((void **)&p->d_reweight_used, (size_t)(p->ni + 1) * sizeof(double)));
            HIPCHK(hipMemcpyAsync(p->d_reweight_used, p->d_reweight, (size_t)(p->ni + 1) * sizeof(double), hipMemcpyDeviceToDevice, p->ctx->stream));
        }
        if (keep) {
            const int wb = p->chain_valid ? 1 - p->chain_cur : p->chain_cur;
            const int64_t need = nblocks * nchain;
            if (need > p->chain_cap[wb]) {
                if (p->d_chain_x[wb]) (void)hipFree(p->d_chain_x[wb]);
                if (p->d_chain_curr[wb]) (void)hipFree(p->d_chain_curr[wb]);
                if (p->d_chain_P[wb]) (void)hipFree(p->d_chain_P[wb]);
                p->d_chain_x[wb] = nullptr;
                p->d_chain_curr[wb] = nullptr;
                p->d_chain_P[wb] = nullptr;
                p->chain_cap[wb] = 0;
                HIPCHK(hipMalloc((void **)&p->d_chain_x[wb], (size_t)need * s.ndraw * sizeof(double)));
                HIPCHK(hipMalloc((void **)&p->d_chain_curr[wb], (size_t)need * sizeof(int)));
                HIPCHK(hipMalloc((void **)&p->d_chain_P[wb], (size_t)need * sizeof(double)));
                p->chain_cap[wb] = need;
            }
            a.store_x = p->d_chain_x[wb];
            a.store_curr = p->d_chain_curr[wb];
            a.store_P = solver == MCI_VEGASMC ? p->d_chain_P[wb] : nullptr;
            a.store_cap = p->chain_cap[wb];
            p->chain_cur = wb;
            p->chain_valid = true;
            p->chain_ntrain = p->ntrain;
            p->chain_solver = solver;
            p->chain_iteration = iteration;
            p->chain_lo = block_lo;
            p->chain_hi = block_hi;
            p->chain_nchain = nchain;
        } else {
            p->chain_valid = false;
        }
        p->last_carried = carried;
    }
    if (solver == MCI_MCMC && !s.host_integrand && nevalperblock / nchain + nburn < ((int64_t)1 << 31) - 1) {
        if (!p->d_hold) HIPCHK(hipMalloc((void **)&p->d_hold, 64 * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(p->d_hold, 0, 64 * sizeof(unsigned long long), p->ctx->stream));
        a.hold_hist = p->d_hold;
    }
    if (G > 1) {
        a.spec_tab = p->d_spec_tab;
        a.spec_lanes = G;
        a.spec_maxacc = spec_maxacc;
        a.spec_ntree = p->spec_ntree;
        a.spec_first = p->spec_first;
        for (int k = 0; k < 8; ++k) a.spec_accept[k] = p->spec_accepts[k];
    }
    a.status = p->d_status;
    a.tile_w = p->d_tile_w;
    a.tile_bins = p->d_tile_bins;
    a.tile_stride = nblocks * nevalperblock;
    a.nrows = nrows;
    // Split-all :vegas: the replay partitions a block's parked samples on its own.  Every replay workgroup zeroes and flushes a whole LDS
    // tile (C4: 128 KB) and every row it writes is read again by the merge, so it runs ~2 workgroups per CU and tile pair instead of one
    // per sample-pass row (C4: 512 instead of 2048 workgroups, 67 instead of 262 MB of partial histograms written and read back:
    // k_hist_stage1 100 -> 12.6 us, profiles/r04_c4_kernel_stats.txt).  The partition only decides which workgroup adds a sample to the
    // histogram: sums differ by reassociation.
    int64_t hist_rows = nrows;
    if (split && s.split_all) {
        int64_t rwpb = 512 / (nblocks * s.ntile);
        if (rwpb > wpb) rwpb = wpb;
        if (rwpb < 1) rwpb = 1;
        a.tiles_wpb = (int)rwpb;
        a.tiles_rows = hist_rows = nblocks * rwpb;
    }
    if (s.host_integrand) {
        // "batch callback": the closure cannot run on the device, so the draws of this launch go to the host (SoA,
        // x[k*n + i]), the callback fills w[q*n + i], and the sample kernel regenerates the same draws (same Philox
        // indices) around the uploaded weights.  PCIe + host bound by construction; solver = :vegas only.
        if (solver != MCI_VEGAS && s.ntile > 1) return fail(MCI_ERR_INVALID, "a host integrand under a chain solver needs the histograms in one LDS tile");
        // :vegas -- the draws of the whole launch; chain solvers -- one configuration per chain and Markov step (below)
        const int64_t n = solver == MCI_VEGAS ? nblocks * nevalperblock : nblocks * nchain;
        if ((double)n * (double)(s.ndraw + s.ni * s.ncomp) * 8.0 > 8.0 * 1024 * 1024 * 1024)
            return fail(MCI_ERR_INVALID, "a host integrand over %lld configurations of %d doubles per launch (more than 8 GiB): lower neval or "
                                         "give the integrand as device source (mci_set_integrand_source)", (long long)n, s.ndraw + s.ni * s.ncomp);
        if (n > p->cap_host) {
            if (p->d_hx) (void)hipFree(p->d_hx);
            if (p->d_hw) (void)hipFree(p->d_hw);
            if (p->h_hx) (void)hipHostFree(p->h_hx);
            if (p->h_hw) (void)hipHostFree(p->h_hw);
            p->d_hx = p->d_hw = p->h_hx = p->h_hw = nullptr;
            p->cap_host = 0;
            HIPCHK(hipMalloc((void **)&p->d_hx, (size_t)n * s.ndraw * sizeof(double)));
            HIPCHK(hipMalloc((void **)&p->d_hw, (size_t)n * s.ni * s.ncomp * sizeof(double)));
            HIPCHK(hipHostMalloc((void **)&p->h_hx, (size_t)n * s.ndraw * sizeof(double), hipHostMallocDefault));
            HIPCHK(hipHostMalloc((void **)&p->h_hw, (size_t)n * s.ni * s.ncomp * sizeof(double), hipHostMallocDefault));
            p->cap_host = n;
        }
        if (solver == MCI_VEGAS) {
        mci::DumpArgs d{};
        d.edges = p->d_edges;
        d.dacc = p->d_dacc;
        d.ddist = p->d_ddist;
        d.ud = p->d_ud;
        d.x = p->d_hx;
        d.soa = 1;
        d.seed = seed;
        d.iteration = (mci::u32)iteration;
        d.first_index = block_lo * nevalperblock;
        d.n = n;
        void *dargs[] = {&d};
        const unsigned dgrid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipStream_t hs = p->ctx->stream;
        HIPCHK(hipModuleLaunchKernel(p->f_dump, dgrid, 1, 1, 256, 1, 1, (unsigned)p->lds_bytes, hs, dargs, nullptr));
        HIPCHK(hipMemcpyAsync(p->h_hx, p->d_hx, (size_t)n * s.ndraw * sizeof(double), hipMemcpyDeviceToHost, hs));
        HIPCHK(hipStreamSynchronize(hs));
        if ((rc = eval_host_integrand(p, nullptr, p->h_hx, p->h_hw, n))) return rc;
        HIPCHK(hipMemcpyAsync(p->d_hw, p->h_hw, (size_t)n * s.ni * s.ncomp * sizeof(double), hipMemcpyHostToDevice, hs));
        }
        a.host_w = p->d_hw;
    }
    // host measure: records per block, rows of relative weights per record, measured-step window of a chain (BatchArgs::hm_*)
    int64_t hm_n = 0, hm_first = 0, hm_count = 0;
    int hm_rows = 0;
    if (s.host_measure) {
        const int nw = s.ni * s.ncomp;
        if (solver == MCI_VEGAS) {
            hm_n = nevalperblock;
            hm_rows = nw;
        } else {
            // a chain measures at steps j * measurefreq: :vegasmc from `burnin` on (vegas_mc/montecarlo.jl:213), :mcmc from nburn on
            // (mcmc/montecarlo.jl:143) -- the same comparisons the kernels make
            const int64_t mfq = measurefreq > 0 ? measurefreq : 1;
            const int64_t last = solver == MCI_VEGASMC ? nevalperblock / nchain : nevalperblock / nchain + nburn;
            hm_first = 1;
            if (solver == MCI_VEGASMC) {
                hm_first = (int64_t)(burnin / (double)mfq);
                if (hm_first < 1) hm_first = 1;
                while (hm_first > 1 && (double)((hm_first - 1) * mfq) >= burnin) --hm_first;
                while ((double)(hm_first * mfq) < burnin) ++hm_first;
            } else if (nburn > 0) {
                hm_first = (nburn + mfq - 1) / mfq;
                if (hm_first < 1) hm_first = 1;
            }
            hm_count = last / mfq - hm_first + 1;
            if (hm_count < 0) hm_count = 0;
            hm_n = nchain * hm_count;
            hm_rows = solver == MCI_MCMC ? s.ncomp : nw;
        }
        const int64_t n = nblocks * hm_n > 0 ? nblocks * hm_n : 1;
        // every record crosses PCIe and sits in pinned host memory: refuse launches whose records would not reasonably fit
        if ((double)n * (double)(s.ndraw + nw + 1) * 8.0 > 8.0 * 1024 * 1024 * 1024)
            return fail(MCI_ERR_INVALID, "a host measure over %lld records of %d doubles per launch (more than 8 GiB): lower neval, raise measurefreq "
                                         "or give the measure as device source (mci_set_measure_source)", (long long)n, s.ndraw + nw);
        if (n > p->cap_hmeas) {
            if (p->d_mx) (void)hipFree(p->d_mx);
            if (p->d_mrelw) (void)hipFree(p->d_mrelw);
            if (p->d_midx) (void)hipFree(p->d_midx);
            if (p->h_mx) (void)hipHostFree(p->h_mx);
            if (p->h_mrelw) (void)hipHostFree(p->h_mrelw);
            if (p->h_midx) (void)hipHostFree(p->h_midx);
            p->d_mx = p->d_mrelw = p->h_mx = p->h_mrelw = nullptr;
            p->d_midx = p->h_midx = nullptr;
            p->cap_hmeas = 0;
            HIPCHK(hipMalloc((void **)&p->d_mx, (size_t)n * s.ndraw * sizeof(double)));
            HIPCHK(hipMalloc((void **)&p->d_mrelw, (size_t)n * nw * sizeof(double)));
            HIPCHK(hipMalloc((void **)&p->d_midx, (size_t)n * sizeof(int32_t)));
            HIPCHK(hipHostMalloc((void **)&p->h_mx, (size_t)n * s.ndraw * sizeof(double), hipHostMallocDefault));
            HIPCHK(hipHostMalloc((void **)&p->h_mrelw, (size_t)n * nw * sizeof(double), hipHostMallocDefault));
            HIPCHK(hipHostMalloc((void **)&p->h_midx, (size_t)n * sizeof(int32_t), hipHostMallocDefault));
            p->cap_hmeas = n;
        }
        if (nblocks * s.nobs > p->cap_mobs) {
            if (p->d_mobs) (void)hipFree(p->d_mobs);
            p->d_mobs = nullptr;
            HIPCHK(hipMalloc((void **)&p->d_mobs, (size_t)nblocks * s.nobs * sizeof(double)));
            p->cap_mobs = nblocks * s.nobs;
        }
        a.host_mx = p->d_mx;
        a.host_relw = p->d_mrelw;
        a.host_midx = p->d_midx;
        a.hm_first = hm_first;
        a.hm_count = hm_count;
        a.hm_stride = nblocks * hm_n;
        if (solver != MCI_VEGAS) { // a chain on the normalization integrand leaves no record (:mcmc): preset "none"
            HIPCHK(hipMemsetAsync(p->d_mx, 0, (size_t)n * s.ndraw * sizeof(double), p->ctx->stream));
            HIPCHK(hipMemsetAsync(p->d_mrelw, 0, (size_t)n * hm_rows * sizeof(double), p->ctx->stream));
            HIPCHK(hipMemsetAsync(p->d_midx, 0xFF, (size_t)n * sizeof(int32_t), p->ctx->stream));
        }
    }
    void *args[] = {&a};
    hipFunction_t f = p->f_solver[G > 1 ? (solver == MCI_VEGASMC ? kSlotVegasmcSpec : kSlotMcmcSpec) : kern];
    hipStream_t st = p->ctx->stream;
    const int slot = (int)(p->launches % mci_problem::kEvRing);
    // HIP events around the sample launch (mci_kernel_times_ms): each record is a barrier packet with a signal, ~5.5 us of idle
    // queue -- a third of a launch-bound iteration (neval = 1e4: 36 -> 25 us), nothing next to a launch of millions of samples.
    // mci_set_kernel_timing: -1 (default) = launches of >= 2^20 samples, 0 = never, 1 = always
    p->time_this_launch = p->kernel_timing > 0 || (p->kernel_timing < 0 && nblocks * nevalperblock >= ((int64_t)1 << 20));
    if (p->time_this_launch && solver == MCI_VEGAS) { // ... and the clock the sample loop ran at (mci_kernel_clocks)
        if (!p->d_clocks) {
            HIPCHK(hipMalloc((void **)&p->d_clocks, (size_t)2 * mci_problem::kEvRing * sizeof(unsigned long long)));
            HIPCHK(hipMemsetAsync(p->d_clocks, 0, (size_t)2 * mci_problem::kEvRing * sizeof(unsigned long long), st));
        }
        a.clock_out = p->d_clocks + 2 * slot;
    }
    if (p->time_this_launch) HIPCHK(hipEventRecord(p->evs[2 * slot], st));
    if (solver != MCI_VEGAS && s.host_integrand) {
        // The closure sits inside the Markov step (vegas_mc/updates.jl:67-75, mcmc/updates.jl:35-38): the chains of this launch advance
        // in lock step, one kernel launch per step; each hands the host the nc configurations to evaluate and takes their weights back
        // (vegasmc_host_step, mcmc_host_step).  PCIe- and host-bound by construction: two copies, one callback and one launch per step.
        const int64_t nc = nblocks * nchain, steps = nevalperblock / nchain;
        const int nw = s.ni * s.ncomp, nd = s.ndraw;
        if (nc >= ((int64_t)1 << 31) || steps + nburn >= ((int64_t)1 << 31) - 1) return fail(MCI_ERR_INVALID, "too many chains or steps for the host-closure path");
        if (nc > p->cap_hstep) {
            if (p->d_hstep) (void)hipFree(p->d_hstep);
            p->d_hstep = nullptr;
            p->cap_hstep = 0;
            // doubles: cx, cprob, pprob [nd] each; cw [nw]; cprobability, pprop, puacc, cwabs; ints: cbin, pbin [nd] each; pvi, ccurr, cit, ctr, pnew, put, hidx; done
            HIPCHK(hipMalloc(&p->d_hstep, (size_t)nc * ((3 * nd + nw + 4) * sizeof(double) + (2 * nd + 7) * sizeof(int)) + 16));
            p->cap_hstep = nc;
        }
        if (nc > p->cap_hidx) {
            if (p->h_hidx) (void)hipHostFree(p->h_hidx);
            p->h_hidx = nullptr;
            p->cap_hidx = 0;
            HIPCHK(hipHostMalloc((void **)&p->h_hidx, (size_t)(nc + 1) * sizeof(int32_t), hipHostMallocDefault));
            p->cap_hidx = nc;
        }
        {
            double *dp = (double *)p->d_hstep;
            a.hs.cx = dp; dp += (size_t)nd * nc;
            a.hs.cprob = dp; dp += (size_t)nd * nc;
            a.hs.pprob = dp; dp += (size_t)nd * nc;
            a.hs.cw = dp; dp += (size_t)nw * nc;
            a.hs.cprobability = dp; dp += nc;
            a.hs.pprop = dp; dp += nc;
            a.hs.puacc = dp; dp += nc;
            a.hs.cwabs = dp; dp += nc;
            int *ip = (int *)dp;
            a.hs.cbin = ip; ip += (size_t)nd * nc;
            a.hs.pbin = ip; ip += (size_t)nd * nc;
            a.hs.pvi = ip; ip += nc;
            a.hs.ccurr = ip; ip += nc;
            a.hs.cit = ip; ip += nc;
            a.hs.ctr = ip; ip += nc;
            a.hs.pnew = ip; ip += nc;
            a.hs.put = ip; ip += nc;
            a.hs.hidx = ip; ip += nc; // (hidx[nc] = done: one copy brings both back)
            a.hs.done = ip;
        }
        a.hs.hx = p->d_hx;
        a.hs.nc = nc;
        a.hs.steps = steps;
        // the step launches ADD to the partial rows
        HIPCHK(hipMemsetAsync(p->d_part_cols, 0, (size_t)nrows * s.ncols * sizeof(double), st));
        if (hist_lds && s.nbin > 0) HIPCHK(hipMemsetAsync(p->d_part_hist, 0, (size_t)nrows * s.nbin * sizeof(double), st));
        HIPCHK(hipMemsetAsync(p->d_part_pa, 0, (size_t)nrows * 2 * p->npa * sizeof(double), st));
        HIPCHK(hipMemsetAsync(a.hs.done, 0, sizeof(int), st));
        if (solver == MCI_VEGASMC) {
            for (int64_t ne = 0; ne <= steps + 1; ++ne) {
                a.hs.ne = ne;
                HIPCHK(hipModuleLaunchKernel(f, (unsigned)nwg, 1, 1, (unsigned)T, 1, 1, (unsigned)solver_lds(p, solver), st, args, nullptr));
                if (ne > steps) break;
                HIPCHK(hipMemcpyAsync(p->h_hx, p->d_hx, (size_t)nc * nd * sizeof(double), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if ((rc = eval_host_integrand(p, nullptr, p->h_hx, p->h_hw, nc))) return rc;
                HIPCHK(hipMemcpyAsync(p->d_hw, p->h_hw, (size_t)nc * nw * sizeof(double), hipMemcpyHostToDevice, st));
            }
        } else {
            // every chain counts its own steps (a start that has to be redrawn costs a launch): launch until all of them are through
            const int64_t limit = steps + nburn + 2 + 10000; // (mcmc/montecarlo.jl:118: at most 10000 tries of the start)
            for (int64_t ne = 0;; ++ne) {
                a.hs.ne = ne;
                HIPCHK(hipModuleLaunchKernel(f, (unsigned)nwg, 1, 1, (unsigned)T, 1, 1, (unsigned)solver_lds(p, solver), st, args, nullptr));
                HIPCHK(hipMemcpyAsync(p->h_hidx, a.hs.hidx, (size_t)(nc + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpyAsync(p->h_hx, p->d_hx, (size_t)nc * nd * sizeof(double), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if (p->h_hidx[nc] >= nc) break;
                if (ne > limit) return fail(MCI_ERR_INVALID, "host-closure :mcmc chains did not finish (%d of %lld)", (int)p->h_hidx[nc], (long long)nc);
                if ((rc = eval_host_integrand(p, p->h_hidx, p->h_hx, p->h_hw, nc))) return rc;
                HIPCHK(hipMemcpyAsync(p->d_hw, p->h_hw, (size_t)nc * s.ncomp * sizeof(double), hipMemcpyHostToDevice, st));
            }
        }
    } else
    HIPCHK(hipModuleLaunchKernel(f, (unsigned)nwg, 1, 1, (unsigned)T_launch, 1, 1, (unsigned)solver_lds(p, solver), st, args, nullptr));
    if (solver == MCI_MCMC) p->hold_measured = a.hold_hist != nullptr;
    // (an explicit chain count: nobody sizes a launch from this one's holds, and the host keeps queueing launches back to back)
    if (a.hold_hist && auto_chains && (rc = hold_publish(p, nevalperblock / nchain, solver != MCI_VEGAS && p->last_carried))) return rc;
    if (split)
        HIPCHK(hipModuleLaunchKernel(p->f_tiles[kern == kSlotVegasAny ? 1 : 0], (unsigned)(((hist_rows + 7) / 8) * 8 * (s.ntile - (s.split_all ? 0 : 1))), 1, 1, (unsigned)T, 1, 1, (unsigned)p->lds_bytes, st, args, nullptr));
    if (p->time_this_launch) HIPCHK(hipEventRecord(p->evs[2 * slot + 1], st));
    p->ev_valid[slot] = p->time_this_launch;
    p->launches += 1;
    if (s.host_measure) {
        // the closure cannot run on the device: this launch's (measured) configurations and relative weights go to the host
        // (draw-major, like the host integrand path), the callback accumulates block b's observables from block b's records, and
        // they join the block's partial row before the merge.  PCIe- and host-bound by construction.
        const int64_t n = nblocks * hm_n;
        const int nw = s.ni * s.ncomp, nc = s.ncomp;
        std::vector<double> obs((size_t)nblocks * s.nobs, 0.0);
        if (n > 0) {
            HIPCHK(hipMemcpyAsync(p->h_mx, p->d_mx, (size_t)n * s.ndraw * sizeof(double), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(p->h_mrelw, p->d_mrelw, (size_t)n * hm_rows * sizeof(double), hipMemcpyDeviceToHost, st));
            if (solver == MCI_MCMC) HIPCHK(hipMemcpyAsync(p->h_midx, p->d_midx, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            const double *relw = p->h_mrelw;
            if (solver == MCI_MCMC && p->hmeas_fn) { // plain form: every integrand's row, zero except the one the chain sat on
                p->h_mtmp.assign((size_t)n * nw, 0.0);
                for (int64_t i = 0; i < n; ++i)
                    if (p->h_midx[i] >= 0)
                        for (int q = 0; q < nc; ++q) p->h_mtmp[(size_t)(p->h_midx[i] * nc + q) * n + i] = p->h_mrelw[(size_t)q * n + i];
                relw = p->h_mtmp.data();
            }
            if (solver != MCI_MCMC && p->hmeas_idx_fn) p->h_mitmp.resize((size_t)hm_n);
            // :vegas calls `measure` for the samples with (ne % measurefreq == 0) only (vegas/montecarlo.jl:148-165): the records the
            // cadence skips are squeezed out on the host, so that a measure which is not linear in the weights (a visit count, a
            // per-call bin count) sees exactly the calls the reference makes
            const bool squeeze = solver == MCI_VEGAS && measurefreq > 1;
            const int64_t keep = squeeze ? nevalperblock / measurefreq : hm_n;
            std::vector<double> sx, sw;
            if (squeeze) {
                sx.resize((size_t)(keep > 0 ? keep : 1) * s.ndraw);
                sw.resize((size_t)(keep > 0 ? keep : 1) * nw);
            }
            for (int64_t b = 0; b < nblocks; ++b) {
                const int64_t off = b * hm_n;
                double *ob = obs.data() + (size_t)b * s.nobs;
                int hrc = 0;
                if (squeeze) {
                    for (int k = 0; k < s.ndraw; ++k)
                        for (int64_t j = 0; j < keep; ++j) sx[(size_t)k * keep + j] = p->h_mx[(size_t)k * n + off + (j + 1) * measurefreq - 1];
                    for (int q = 0; q < nw; ++q)
                        for (int64_t j = 0; j < keep; ++j) sw[(size_t)q * keep + j] = relw[(size_t)q * n + off + (j + 1) * measurefreq - 1];
                    if (p->hmeas_fn) hrc = p->hmeas_fn(sx.data(), sw.data(), keep, keep, s.ndraw, nw, block_lo + b, ob, s.nobs, p->hmeas_user);
                    else {
                        p->h_mitmp.resize((size_t)(keep > 0 ? keep : 1));
                        for (int j = 0; j < s.ni && !hrc; ++j) {
                            std::fill(p->h_mitmp.begin(), p->h_mitmp.end(), (int32_t)j);
                            hrc = p->hmeas_idx_fn(p->h_mitmp.data(), sx.data(), sw.data() + (size_t)j * nc * keep, keep, keep, s.ndraw, nc, block_lo + b, ob,
                                                  s.nobs, p->hmeas_user);
                        }
                    }
                } else if (p->hmeas_fn) {
                    hrc = p->hmeas_fn(p->h_mx + off, relw + off, hm_n, n, s.ndraw, nw, block_lo + b, ob, s.nobs, p->hmeas_user);
                } else if (solver == MCI_MCMC) {
                    hrc = p->hmeas_idx_fn(p->h_midx + off, p->h_mx + off, relw + off, hm_n, n, s.ndraw, nc, block_lo + b, ob, s.nobs, p->hmeas_user);
                } else { // indexed form under :vegas / :vegasmc: every integrand in turn
                    for (int j = 0; j < s.ni && !hrc; ++j) {
                        std::fill(p->h_mitmp.begin(), p->h_mitmp.end(), (int32_t)j);
                        hrc = p->hmeas_idx_fn(p->h_mitmp.data(), p->h_mx + off, relw + (size_t)j * nc * n + off, hm_n, n, s.ndraw, nc, block_lo + b, ob,
                                              s.nobs, p->hmeas_user);
                    }
                }
                if (hrc) return fail(MCI_ERR_INVALID, "the host measure failed (%d)", hrc);
            }
        }
        HIPCHK(hipMemcpyAsync(p->d_mobs, obs.data(), obs.size() * sizeof(double), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(mci::k_add_host_obs, dim3((unsigned)((nblocks * s.nobs + 255) / 256)), dim3(256), 0, st, p->d_mobs, (int)nblocks, s.nobs, s.ncols, wpb,
                           p->d_part_cols);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st)); // `obs` leaves scope
    }
    p->last_samples = nblocks * nevalperblock;
    p->last_wg = (int)nwg;
    p->last_threads = T_launch;
    p->last_nblocks = (int)nblocks;
    if (solver != MCI_VEGAS) p->last_nchain = nchain;
    // merge: block sums -> packed
    const int nb256 = (s.nbin + 255) / 256;
    // (reading a few partial rows directly in the second stage instead -- no first-stage launch when an iteration is launch-bound --
    // was measured at neval = 1e4: k_finish grows by what the launch took, 26 us per iteration either way)
    if (hist_lds && s.nbin > 0 && !atomic_flush)
        hipLaunchKernelGGL(mci::k_hist_stage1, dim3(nb256, mci_problem::kGroups), dim3(256), 0, st, p->d_part_hist, (int)hist_rows, s.nbin,
                           (int)mci_problem::kGroups, p->d_stage1);
    HIPCHK(hipGetLastError());
    mci::MergeArgs &m = p->merge;
    m.part_cols = p->d_part_cols;
    m.ncols = s.ncols;
    m.nobs = s.nobs;
    m.ni = s.ni;
    m.nblocks = (int)nblocks;
    m.wg_per_block = wpb;
    m.stage1 = p->d_stage1;
    m.ngroup = (int)mci_problem::kGroups;
    m.ghist = p->d_ghist;
    m.use_ghist = (hist_lds && !atomic_flush) ? 0 : atomic_flush ? ghist_buffers : 1;
    m.nbin = s.nbin;
    m.packed = p->d_packed;
    m.status = p->d_status;
    m.scratch = p->d_scratch;
    m.part_pa = solver != MCI_VEGAS ? p->d_part_pa : nullptr;
    m.npa = p->npa;
    m.nrows = (int)nrows;
    m.block_means = nullptr;
    m.hold = a.hold_hist; // (:mcmc: the 64 counts follow the tables in `packed`, so that ONE all-reduce carries them; NULL: zeros)
    if (solver != MCI_VEGAS) { // the chain solvers keep every block's mean of every iteration (one row of the block log)
        const int64_t stride = nblocks * s.nobs;
        if (stride != p->blk_stride || block_lo != p->blk_lo) {
            p->blk_rows = 0;
            p->blk_carried = 0;
            p->blk_stride = stride;
            p->blk_lo = block_lo;
        }
        if ((rc = grow_block_log(p, p->blk_rows + 1))) return rc;
        m.block_means = p->d_blocklog + (size_t)p->blk_rows * stride;
        p->blk_rows += 1;
        p->blk_carried += p->last_carried ? 1 : 0;
    }
    p->merge_pending = true;
    return MCI_OK;
}

// partials -> packed, if the last mci_iteration_run has not been merged yet
static int flush_merge(mci_problem *p) {
    if (!p->merge_pending) return MCI_OK;
    p->merge_pending = false;
    HIPCHK(hipSetDevice(p->ctx->device));
    const int nb256 = (p->shape.nbin + 255) / 256;
    hipLaunchKernelGGL(mci::k_finalize, dim3(nb256 + 1 + (2 * p->npa + 3) / 4), dim3(256), 0, p->ctx->stream, p->merge);
    HIPCHK(hipGetLastError());
    return MCI_OK;
}

int mci_iteration_reduce(mci_problem *p) {
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    if (!p->ctx->comm) return MCI_OK; // no communicator: single process (mpi_nprocs() == 1)
    int rc = flush_merge(p);
    if (rc) return rc;
    // HIP events around the collective under the sample launch's rule (mci_set_kernel_timing): what a rank waits for here is the
    // slowest rank's sample pass plus the latency of one small all-reduce (mci_comm_times_ms)
    const bool timed = p->time_this_launch;
    const int slot = (int)(p->reduces % mci_problem::kCevRing);
    if (timed) {
        if (p->cevs.empty()) {
            p->cevs.resize(2 * mci_problem::kCevRing);
            for (auto &e : p->cevs) HIPCHK(hipEventCreate(&e));
        }
        HIPCHK(hipEventRecord(p->cevs[2 * slot], p->ctx->stream));
    }
    // ONE collective per iteration whatever the solver: [statistics | histograms | propose | accept] and, behind an :mcmc launch that
    // measured its holding times, the 64 counts of their histogram (exact in doubles)
    const size_t count = (size_t)p->packed_n + (p->hold_deferred ? 64 : 0);
    int r = g_rccl.AllReduce(p->d_packed, p->d_packed, count, kNcclFloat64, kNcclSum, p->ctx->comm, p->ctx->stream);
    if (r) return fail(MCI_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    p->ctx->collectives += 1;
    p->ctx->last_count = (long long)count;
    if (p->hold_deferred && (rc = hold_publish_reduced(p))) return rc; // the summed holding-time counts -> pinned host memory
    if (timed) HIPCHK(hipEventRecord(p->cevs[2 * slot + 1], p->ctx->stream));
    p->cev_valid[slot] = timed;
    p->reduces += 1;
    return MCI_OK;
}

int mci_comm_collectives(const mci_ctx *c, int64_t *calls, int64_t *last_count) {
    if (!c) return fail(MCI_ERR_INVALID, "NULL argument");
    if (calls) *calls = c->collectives;
    if (last_count) *last_count = c->last_count;
    return MCI_OK;
}

// An external reducer (comm.py TorchDistComm) has summed mci_reduce_size() doubles of `packed` over the ranks: what the library does
// behind its own all-reduce -- the summed :mcmc holding-time counts go to the host, every rank sizes its next chains from them
int mci_external_reduce_done(mci_problem *p) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    if (!p->hold_ext_pending) return MCI_OK;
    p->hold_ext_pending = false;
    HIPCHK(hipSetDevice(p->ctx->device));
    return hold_publish_reduced(p);
}

int mci_reduce_size(const mci_problem *p, int64_t *n) {
    if (!p || !n) return fail(MCI_ERR_INVALID, "NULL argument");
    *n = p->packed_n + 64;
    return MCI_OK;
}

int mci_comm_times_ms(mci_problem *p, float *ms, int32_t n, int32_t *got) {
    if (!p || !ms || !got) return fail(MCI_ERR_INVALID, "NULL argument");
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    int64_t have = p->reduces < mci_problem::kCevRing ? p->reduces : mci_problem::kCevRing;
    if (have > n) have = n;
    int32_t k = 0;
    for (int64_t i = 0; i < have; ++i) { // oldest first
        const int slot = (int)((p->reduces - have + i) % mci_problem::kCevRing);
        if (!p->cev_valid[slot]) continue;
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, p->cevs[2 * slot], p->cevs[2 * slot + 1]));
        ms[k++] = t;
    }
    *got = k;
    return MCI_OK;
}

static int launch_train(mci_problem *p, int do_train, int do_reweight, double gamma, double *log_row) {
    const auto &s = p->shape;
    int maxn = 1;
    for (auto &L : p->leaves) maxn = L.nbin > maxn ? L.nbin : maxn;
    mci::TrainArgs a{};
    a.leaves = p->d_leaves;
    a.nleaf = s.nleaf;
    a.packed = p->d_packed;
    a.nstat = p->nstat;
    a.edges = p->d_edges;
    a.dacc = p->d_dacc;
    a.ddist = p->d_ddist;
    a.iter_log_row = log_row;
    a.reweight = p->d_reweight;
    a.goal = p->h_goal.empty() ? nullptr : p->d_goal;
    a.nd = s.ni + 1;
    a.do_reweight = do_reweight;
    a.gamma = gamma;
    a.do_train = do_train;
    if (do_train) p->ntrain += 1;
    a.serial_walk = p->train_serial >= 0 ? p->train_serial : (p->last_samples == 0 || p->last_samples >= mci_problem::kSerialWalkSamples) ? 1 : 0;
    if (p->debug_wrong_decision && a.serial_walk == 1) a.serial_walk = 3;
    a.status = p->d_status;
    a.maxn = maxn;
    // d | sg | wa (train_leaf) | the serial walk's slots and their record, where they fit (grids of up to ~2700 increments), else k_finish's merged histogram alone
    a.spare = (size_t)(mci::train_lds_doubles(maxn) + mci::train_spare_doubles(maxn)) * sizeof(double) <= (size_t)kTrainLdsMax ? 1 : 0;
    const size_t sm = (size_t)(mci::train_lds_doubles(maxn) + (a.spare ? mci::train_spare_doubles(maxn) : maxn)) * sizeof(double);
    // two bins per thread for the default 999-bin grids: the rescale (a pow and a log per bin) and the second merge stage are the
    // latency chains of a lone workgroup; with four bins per thread (256 threads) a launch-bound iteration took 24.7 us, with two
    // 22.2, with one (1024 threads) 22.3 (tools/latency.py, neval = 1e4)
    const unsigned tt = maxn > 256 ? 512u : 256u;
    if (sm > 64 * 1024 && !p->train_lds_raised) { // grids of more than ~1600 increments
        HIPCHK(hipFuncSetAttribute((const void *)mci::k_train, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrainLdsMax));
        HIPCHK(hipFuncSetAttribute((const void *)mci::k_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrainLdsMax));
        p->train_lds_raised = true;
    }
    if (p->merge_pending) { // nothing looked at `packed` since the sample batch: merge + refine in one launch
        p->merge_pending = false;
        hipLaunchKernelGGL(mci::k_finish, dim3(s.nleaf + 1 + (2 * p->npa + 3) / 4), dim3(tt), sm, p->ctx->stream, p->merge, a);
    } else {
        hipLaunchKernelGGL(mci::k_train, dim3(s.nleaf + 1), dim3(tt), sm, p->ctx->stream, a);
    }
    HIPCHK(hipGetLastError());
    return MCI_OK;
}

// room for `rows` more iterations in the device-side iteration log (it grows by itself, with a stream synchronisation each time:
// a caller that must not synchronise inside a timed loop reserves first)
static int grow_iteration_log(mci_problem *p, int64_t need) {
    if (need <= p->cap_iter) return MCI_OK;
    int64_t ncap = p->cap_iter ? p->cap_iter : 64;
    while (ncap < need) ncap *= 2;
    double *n = nullptr;
    HIPCHK(hipMalloc((void **)&n, (size_t)ncap * p->nstat * sizeof(double)));
    if (p->d_iterlog) {
        HIPCHK(hipMemcpyAsync(n, p->d_iterlog, (size_t)p->cap_iter * p->nstat * sizeof(double), hipMemcpyDeviceToDevice, p->ctx->stream));
        HIPCHK(hipStreamSynchronize(p->ctx->stream));
        (void)hipFree(p->d_iterlog);
    }
    p->d_iterlog = n;
    p->cap_iter = ncap;
    return MCI_OK;
}

int mci_reserve_iteration_log(mci_problem *p, int32_t rows) {
    if (!p || rows < 0) return fail(MCI_ERR_INVALID, "bad argument");
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    HIPCHK(hipSetDevice(p->ctx->device));
    return grow_iteration_log(p, (int64_t)p->log_row + rows);
}

int mci_iteration_finish(mci_problem *p, int32_t solver, int64_t block_total, int32_t adapt, double gamma, double *mean, double *std) {
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    HIPCHK(hipSetDevice(p->ctx->device));
    const auto &s = p->shape;
    if (int grc = grow_iteration_log(p, (int64_t)p->log_row + 1)) return grc;
    double *row = p->d_iterlog + (size_t)p->log_row * p->nstat;
    // doReweight! runs for the chain solvers whether or not the grid adapts (main.jl:183 is outside the `if adapt`)
    int rc = launch_train(p, adapt ? 1 : 0, (solver == MCI_VEGASMC || solver == MCI_MCMC) ? 1 : 0, gamma, row);
    if (rc) return rc;
    p->log_row += 1;
    if (mean || std) {
        std::vector<double> h(p->nstat);
        HIPCHK(hipMemcpyAsync(h.data(), row, (size_t)p->nstat * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
        if ((rc = check_status(p))) return rc; // synchronises
        std::vector<double> m(s.nobs), e(s.nobs);
        mci_mean_std(h.data(), h.data() + s.nobs, s.nobs, block_total, m.data(), e.data());
        if (mean) memcpy(mean, m.data(), s.nobs * sizeof(double));
        if (std) memcpy(std, e.data(), s.nobs * sizeof(double));
    }
    return MCI_OK;
}

int mci_train(mci_problem *p) {
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    int rc = flush_merge(p);
    if (rc) return rc;
    rc = launch_train(p, 1, 0, 1.0, nullptr);
    if (rc) return rc;
    return check_status(p);
}

// ---------------------------------------------------------------------------------------------------
// persistent :vegas iterations: all `niter` iterations of a launch-bound mci_integrate call as ONE launch (mci_train.h vegas_persist)
// ---------------------------------------------------------------------------------------------------
namespace {
// LDS of the persistent kernel (bytes): the sample loop's carve (plain layout) or the refinement's (scratch, merged histogram, scan
// scratch), whichever is larger -- each is dead while the other runs --, and behind them (map_off, doubles) the workgroup's own copy of
// the map and the flag words (mci_train.h PersistArgs)
int64_t persist_lds(const mci_problem *p, int *map_off) {
    const int N = p->leaves.empty() ? 1 : p->leaves[0].nbin;
    const int64_t a = (p->lds_bytes + 7) / 8, b = (int64_t)mci::train_lds_doubles(N) + N + 256;
    const int64_t off = ((a > b ? a : b) + 1) & ~(int64_t)1;
    if (map_off) *map_off = (int)off;
    return (off + (N + 2) + 4) * 8;
}
// Structural conditions of the persistent kernel: ONE Continuous leaf (every sampling workgroup refines its own copy of the map), tables
// and histograms in LDS in one tile, device-source integrand and measure, everything within the 64 KiB every workgroup may ask for.
bool persist_layout(const mci_problem *p) {
    const auto &s = p->shape;
    if (p->deterministic || s.table_mode != 0 || s.ntile != 1 || s.host_integrand || s.host_measure || s.nbin <= 0 || s.ec_doubles > 0) return false;
    if (s.nleaf != 1 || p->leaves.size() != 1 || p->leaves[0].kind != 0) return false;
    return persist_lds(p, nullptr) <= 64 * 1024;
}
// Which calls run persistently, and on how many workgroups per block: one rank, :vegas at measurefreq == 1, the prefix-scan walk, no
// forced geometry or timing, a grid that is co-resident next to another one like it (<= 128 sampling workgroups + the statistics one).
// Automatic mode adds: launches of samples x draws < 2^19 per iteration over at most 7 draws per sample (tools/latency.py and sweeps of
// sizes and dimensions on the final code, us per iteration by the library's clock, persistent | launch chain: 2-D 11.5 | 12.8 at neval =
// 1e4, 13.7 | 13.9 at 1e5, 15.5 | 16.4 at 2e5, 19.6 | 19.2 at 5e5; 4-D 13.1 | 15.2 at 1e4, 16.6 | 19.7 at 1.2e5; 6-D 13.5 | 16.3 at 1e4,
// 17.1 | 17.6 at 8e4; 16-D 18.3 | 15.8 at 1e4 -- from 8 draws on the launch chain runs the hand-pipelined loop on its tuned layout,
// histogram copies and 512-thread workgroups, which this kernel's plain 256-thread layout does not match).
bool persist_plan(const mci_problem *p, const mci_integrate_args *a, int64_t nevalperblock, int64_t nblocks, int *wpb_out) {
    const auto &s = p->shape;
    if (p->persistent == 0 || p->persist_failed) return false;
    if (a->solver != MCI_VEGAS || a->measurefreq != 1 || a->niter < 1) return false;
    if (p->ctx->nranks != 1) return false; // (a one-rank communicator's all-reduce is the identity)
    if (!persist_layout(p)) return false;
    if (p->wg_per_block > 0 || p->kernel_timing > 0 || p->train_serial >= 1) return false;
    const int64_t work = nevalperblock * nblocks * s.ndraw;
    if (p->persistent < 0 && (work >= ((int64_t)1 << 19) || s.ndraw > 7)) return false;
    const int T = p->threads;
    int64_t target = work < ((int64_t)1 << 19) ? 64 : 128;
    int64_t wpb = (target + nblocks - 1) / nblocks;
    const int64_t maxw = (nevalperblock + T - 1) / T;
    if (wpb > maxw) wpb = maxw;
    if (wpb < 1) wpb = 1;
    while (wpb > 1 && wpb * nblocks > 128) --wpb;
    if (wpb * nblocks > 255) return false;
    if (wpb_out) *wpb_out = (int)wpb;
    return true;
}
} // namespace

static bool persist_layout_ok(const mci_problem *p) { return persist_layout(p); }

// one hiprtc job on a thread of its own (the thread touches nothing but this record)
struct mci_problem::PersistJob {
    Candidate c;
    std::thread th;
    std::atomic<bool> done{false};
};
// A problem that goes away (or changes its kernels) while its job is still compiling does not wait for it: the job moves to a
// process-wide list -- its code object still lands in the kernel cache, where the next problem with that kernel finds it -- and the
// list is joined when a context is destroyed and at exit (a thread inside hiprtc must not outlive the process's static objects).
namespace {
std::mutex g_orphan_mu;
std::vector<mci_problem::PersistJob *> g_orphans;
void persist_orphans_join() {
    std::vector<mci_problem::PersistJob *> mine;
    {
        std::lock_guard<std::mutex> g(g_orphan_mu);
        mine.swap(g_orphans);
    }
    for (auto *j : mine) {
        if (j->th.joinable()) j->th.join();
        delete j;
    }
}
} // namespace
static void persist_job_drop(mci_problem *p) {
    if (!p->persist_job) return;
    mci_problem::PersistJob *j = p->persist_job;
    p->persist_job = nullptr;
    if (j->done.load(std::memory_order_acquire)) {
        if (j->th.joinable()) j->th.join();
        delete j;
        return;
    }
    static std::once_flag once;
    std::call_once(once, [] { atexit(persist_orphans_join); });
    std::lock_guard<std::mutex> g(g_orphan_mu);
    g_orphans.push_back(j);
}
// MCI_OK with p->persist_compiled set: the kernel is loaded.  MCI_OK without: not yet (background == true and the code object is
// still being compiled) -- the caller takes the launch chain this time.
static int compile_persist(mci_problem *p, bool background) {
    if (p->persist_compiled) return MCI_OK;
    Candidate local, *c = &local;
    if (p->persist_job) {
        if (!p->persist_job->done.load(std::memory_order_acquire)) {
            if (background) return MCI_OK;
            p->persist_job->th.join(); // (a caller that insists)
        }
        if (p->persist_job->th.joinable()) p->persist_job->th.join();
        local = std::move(p->persist_job->c);
        delete p->persist_job;
        p->persist_job = nullptr;
    } else {
        mcijit::ProblemShape sh = p->shape;
        sh.hcopy = 1;
        sh.det = 0;
        c->src = mcijit::generate_source(sh, MCI_VEGAS, mcijit::kUnitVegasPersist, p->leaves[0].alpha);
        // (512 threads for the hand-pipelined loops of 8..16 draws -- what the launch chain runs them at -- was tried: 22.5 instead of 18.3 us
        // per iteration of the 16-D Gaussian at neval = 1e4, against 15.8 as a launch chain; the automatic rule stops at 7 draws)
        c->threads = p->threads;
        c->rc = mcijit::compile(c->src, c->threads, c->code, c->log, c->cached, &c->path, mcijit::kHdrTrain, /*cache_only=*/background);
        if (c->rc == -1) { // not in the kernel cache: compile it behind the caller's back ...
            // ... once this process has made kPersistAfterCalls launch-bound calls of this kernel (by this problem or others with the same
            // shape and integrand): the persistent launch saves ~40 us per default-size call and its translation unit costs 0.8 s of hiprtc
            // -- on a thread of its own, but comgr serialises compiles, so another new kernel compiled meanwhile queues behind it (measured:
            // 0.69 instead of 0.24 s, tools/cold_start.py).  A loop of hundreds of small calls gets it (and every later process finds it in
            // the kernel cache); a script that makes a few calls never pays.
            {
                static const int kPersistAfterCalls = 256;
                static std::mutex mu;
                static std::map<uint64_t, int> asked;
                std::lock_guard<std::mutex> g(mu);
                if (++asked[mcijit::fnv1a(c->src)] < kPersistAfterCalls) return MCI_OK;
            }
            p->persist_job = new mci_problem::PersistJob;
            p->persist_job->c = std::move(local);
            mci_problem::PersistJob *j = p->persist_job;
            j->th = std::thread([j] {
                j->c.rc = mcijit::compile(j->c.src, j->c.threads, j->c.code, j->c.log, j->c.cached, &j->c.path, mcijit::kHdrTrain);
                j->done.store(true, std::memory_order_release);
            });
            return MCI_OK;
        }
    }
    if (c->rc) {
        p->persist_failed = true; // (the launch chain's own compile reports what is wrong with the integrand)
        return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", c->log.c_str());
    }
    if (mcijit::max_static_lds_bytes(c->code) != 0 || mcijit::kernel_scratch_bytes(c->code, "mci_vegas_persist") != 0) {
        p->persist_failed = true; // (not an error of the call: it takes the launch chain)
        return fail(MCI_ERR_COMPILE, "the persistent :vegas kernel came out with static LDS or scratch");
    }
    p->persist_code_object = c->path;
    p->persist_threads = c->threads;
    if (p->ctx->offline) {
        p->persist_compiled = true;
        return MCI_OK;
    }
    HIPCHK(hipSetDevice(p->ctx->device));
    if (hipModuleLoadData(&p->module_persist, c->code.data()) != hipSuccess) {
        p->persist_failed = true;
        if (c->cached) unlink(c->path.c_str()); // a cached code object that does not load (truncated by a crash, foreign file)
        return fail(MCI_ERR_HIP, "hipModuleLoadData failed for the persistent :vegas code object");
    }
    HIPCHK(hipModuleGetFunction(&p->f_persist, p->module_persist, "mci_vegas_persist"));
    if (!p->d_persist) {
        HIPCHK(hipMalloc((void **)&p->d_persist, kPersistWords * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(p->d_persist, 0, kPersistWords * sizeof(unsigned long long), p->ctx->stream));
        p->persist_arrive = p->persist_done = 0;
    }
    p->persist_compiled = true;
    return MCI_OK;
}

// queue the one launch that runs iterations first_iteration .. first_iteration + niter - 1 over blocks [lo, hi)
static int persist_launch(mci_problem *p, const mci_integrate_args *ia, int64_t nevalperblock, int64_t lo, int64_t hi, int wpb) {
    const auto &s = p->shape;
    const int64_t nblocks = hi - lo, nrows = nblocks * wpb;
    int rc;
    if ((rc = flush_merge(p))) return rc; // (a batch nobody looked at resets the global histogram when it is merged)
    HIPCHK(hipSetDevice(p->ctx->device));
    if ((rc = ensure_capacity(p, 2 * nrows, nblocks))) return rc; // (the partial rows are double-buffered by the turn's parity)
    if ((rc = grow_iteration_log(p, (int64_t)p->log_row + ia->niter))) return rc;
    const int T = p->persist_threads;
    mci::BatchArgs a{};
    a.edges = p->d_edges;
    a.dacc = p->d_dacc;
    a.ddist = p->d_ddist;
    a.reweight = p->d_reweight;
    a.ud = p->d_ud;
    a.part_cols = p->d_part_cols;
    a.part_hist = p->d_part_hist;
    a.ghist = p->d_ghist;
    a.seed = ia->seed;
    a.iteration = (mci::u32)ia->first_iteration;
    a.neval_per_block = nevalperblock;
    a.block_lo = lo;
    a.wg_per_block = wpb;
    a.measurefreq = 1;
    a.nchain = 1;
    a.hist_atomic = 1;
    a.status = p->d_status;
    a.tile_stride = nblocks * nevalperblock;
    a.nrows = nrows;
    mci::PersistArgs f{};
    mci::MergeArgs &m = f.m;
    m.part_cols = p->d_part_cols;
    m.ncols = s.ncols;
    m.nobs = s.nobs;
    m.ni = s.ni;
    m.nblocks = (int)nblocks;
    m.wg_per_block = wpb;
    m.stage1 = p->d_stage1;
    m.ngroup = (int)mci_problem::kGroups;
    m.ghist = p->d_ghist;
    m.use_ghist = 1;
    m.nbin = s.nbin;
    m.packed = p->d_packed;
    m.status = p->d_status;
    m.scratch = p->d_scratch;
    m.part_pa = nullptr;
    m.npa = p->npa;
    m.nrows = (int)nrows;
    mci::TrainArgs &t = f.t;
    t.leaves = p->d_leaves;
    t.nleaf = s.nleaf;
    t.packed = p->d_packed;
    t.nstat = p->nstat;
    t.edges = p->d_edges;
    t.dacc = p->d_dacc;
    t.ddist = p->d_ddist;
    t.iter_log_row = p->d_iterlog + (size_t)p->log_row * p->nstat;
    t.reweight = p->d_reweight;
    t.goal = nullptr;
    t.nd = s.ni + 1;
    t.do_reweight = 0; // (:vegas: main.jl:183 runs doReweight! for the chain solvers only)
    t.gamma = ia->gamma;
    t.do_train = ia->adapt ? 1 : 0;
    t.serial_walk = 0;
    t.status = p->d_status;
    t.maxn = p->leaves[0].nbin;
    f.niter = ia->niter;
    const int64_t lds = persist_lds(p, &f.map_off);
    f.ctr = p->d_persist;
    if (p->persist_arrive > (1ull << 39)) { // (the arrive count owns 40 bits of the counter word: start over long before it spills)
        HIPCHK(hipMemsetAsync(p->d_persist, 0, 3 * sizeof(unsigned long long), p->ctx->stream));
        p->persist_arrive = p->persist_done = 0;
    }
    f.arrive0 = p->persist_arrive;
    f.done0 = p->persist_done;
    f.spin_ticks = p->persist_spin_ticks; // 2 s of the 100 MHz wall clock per wait
    void *args[] = {&a, &f};
    // The map the call starts from, kept aside: workgroup 0 writes the refined map back as soon as ITS last turn is through, and another
    // workgroup can still run out of time after that -- the fall-back to the launch chain (mci_integrate) restores this copy instead
    // of trusting that `edges` was not touched (8 KB, device to device, behind nothing: ~2 us of a 0.17 ms call)
    if (!p->d_edges_backup) HIPCHK(hipMalloc((void **)&p->d_edges_backup, (p->h_edges.size() ? p->h_edges.size() : 1) * sizeof(double)));
    if (p->h_edges.size()) HIPCHK(hipMemcpyAsync(p->d_edges_backup, p->d_edges, p->h_edges.size() * sizeof(double), hipMemcpyDeviceToDevice, p->ctx->stream));
    // nrows sampling workgroups + the statistics workgroup
    HIPCHK(hipModuleLaunchKernel(p->f_persist, (unsigned)nrows + 1, 1, 1, (unsigned)T, 1, 1, (unsigned)lds, p->ctx->stream, args, nullptr));
    p->persist_arrive += (unsigned long long)(ia->niter + 1) * (unsigned long long)nrows; // (+ one "finished reading" per workgroup at the end)
    p->persist_done += (unsigned long long)ia->niter;
    p->time_this_launch = false;
    p->merge_pending = false;
    p->merge = m; // (what `packed` was merged from, for the record)
    p->merge.part_cols = p->d_part_cols + (size_t)((ia->niter - 1) & 1) * (size_t)nrows * s.ncols;
    p->last_samples = nblocks * nevalperblock;
    p->last_wg = (int)nrows;
    p->last_threads = T;
    p->last_nblocks = (int)nblocks;
    p->log_row += ia->niter;
    return MCI_OK;
}

// sum over the ranks of a few host doubles (the lineage sums of a run): through a device scratch word of the communicator's stream
static int comm_sum_host(mci_problem *p, double *v, int n) {
    if (!p->ctx->comm) return MCI_OK;
    double *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, (size_t)n * sizeof(double)));
    hipStream_t st = p->ctx->stream;
    hipError_t e = hipMemcpyAsync(d, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st);
    int r = e == hipSuccess ? g_rccl.AllReduce(d, d, (size_t)n, kNcclFloat64, kNcclSum, p->ctx->comm, st) : 0;
    p->ctx->collectives += 1;
    p->ctx->last_count = n;
    if (e == hipSuccess && !r) e = hipMemcpyAsync(v, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && !r) e = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (r) return fail(MCI_ERR_COMM, "ncclAllReduce (lineage sums): %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    if (e != hipSuccess) return fail(MCI_ERR_HIP, "lineage sums: %s", hipGetErrorString(e));
    return MCI_OK;
}

// integrate  (reference src/main.jl:71-218)
int mci_integrate(mci_problem *p, const mci_integrate_args *a, mci_result *res) {
    if (!p || !a || !res) return fail(MCI_ERR_INVALID, "NULL argument");
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context: no device to run on");
    const auto &s = p->shape;
    if (res->niter < a->niter || res->nobs != s.nobs) return fail(MCI_ERR_INVALID, "result buffers too small");
    if (!(a->neval > a->block)) return fail(MCI_ERR_INVALID, "neval=%lld should be larger than nblock = %lld", (long long)a->neval, (long long)a->block); // main.jl:222
    int64_t nevalperblock, block;
    mci_standardize_block(a->neval, a->block, p->ctx->nranks, &nevalperblock, &block); // main.jl:121
    const int64_t per = block / p->ctx->nranks;                                         // main.jl:122
    const int64_t lo = per * p->ctx->rank, hi = lo + per;
    if (a->solver != MCI_VEGAS && a->solver != MCI_VEGASMC && a->solver != MCI_MCMC) return fail(MCI_ERR_INVALID, "Solver %d is not supported!", a->solver); // main.jl:263
    // launch-bound :vegas calls: the whole loop below as one persistent launch (same iterations, same Philox streams)
    int wpb_persist = 0, rc = 0;
    bool persist = persist_plan(p, a, nevalperblock, hi - lo, &wpb_persist);
    // (automatic mode: a code object that is not in the kernel cache yet is compiled on a thread of its own, and until it is there the
    // calls go through the launch chain -- a new integrand's first call costs what it did, 0.2 s, not the 0.8 s of the larger unit)
    if (persist) (void)compile_persist(p, p->persistent < 0);
    persist = persist && p->persist_compiled;
    if (!persist && (rc = compile_solver(p, kslot(a->solver, a->measurefreq)))) return rc;
    if ((rc = mci_set_reweight_goal(p, a->reweight_goal, a->reweight_goal ? p->ni + 1 : 0))) return rc;
    const int ignore = a->ignore >= 0 ? a->ignore : (a->adapt ? 1 : 0);
    const size_t nlog = (size_t)a->niter * p->nstat; // the pinned landing place of the statistics (+ the status word), sized outside the timed loop
    if (nlog + 1 > p->cap_hlog) {
        size_t ncap = p->cap_hlog ? p->cap_hlog : (size_t)64 * p->nstat + 1;
        while (ncap < nlog + 1) ncap *= 2;
        if (p->h_log) (void)hipHostFree(p->h_log);
        p->h_log = nullptr;
        p->cap_hlog = 0;
        HIPCHK(hipHostMalloc((void **)&p->h_log, ncap * sizeof(double), hipHostMallocDefault));
        p->cap_hlog = ncap;
    }
    int64_t blk_row0 = -1; // this call's first row of the block log (chain solvers): the log starts over with every call
    if (a->solver != MCI_VEGAS) {
        if ((rc = flush_merge(p))) return rc; // (a pending merge writes its row of the old log)
        p->blk_rows = 0;
        p->blk_carried = 0;
        p->blk_stride = (hi - lo) * s.nobs;
        p->blk_lo = lo;
        blk_row0 = 0;
        if ((rc = grow_block_log(p, a->niter))) return rc;
    }
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    const int row0 = p->log_row;
    auto t0 = std::chrono::steady_clock::now();
    double *h = p->h_log;
    int *hstatus = reinterpret_cast<int *>(p->h_log + nlog);
    int res_warmup = 0;
    p->last_discarded_neval = 0;
    p->last_discarded_launches = 0;
    for (int attempt = 0;; ++attempt) {
        p->last_persistent = persist;
        if (persist && (rc = persist_launch(p, a, nevalperblock, lo, hi, wpb_persist))) return rc;
        // (a hipGraph replay of this chain was measured and dropped: 37.6 against 34.8 us per launch-bound iteration for the eager
        // asynchronous launches on ROCm 7.0 / MI355X, profiles/r02_ablation.txt)
        for (int it = 0; it < a->niter && !persist; ++it) { // main.jl:142
            for (int attempt = 0;; ++attempt) {
                const int32_t iter = a->first_iteration + it + kRepeatStride * attempt;
                p->launch_counted = it >= ignore;
                if ((rc = mci_iteration_run(p, a->solver, nevalperblock, lo, hi, iter, a->seed, a->measurefreq, a->nchain, a->thermal_ratio))) return rc;
                if ((rc = mci_iteration_reduce(p))) return rc;                                   // main.jl:177-188
                if ((rc = mci_iteration_finish(p, a->solver, block, a->adapt, a->gamma, nullptr, nullptr))) return rc; // main.jl:183-199
                // Warm-up of the automatic :mcmc chain length (mci_mcmc_auto_chains): an iteration whose chains turned out too short for
                // the holding times they measured is not counted -- it has trained the map and moved the reweight factors, its chains
                // go on -- and runs again with longer chains (the Philox streams of iteration + kRepeatStride * attempt), until the
                // first launch that is long enough; from then on nothing is repeated.  The first iteration of a call that ignores it
                // anyway (main.jl:82) is let through as it is.
                if (a->solver != MCI_MCMC || a->nchain > 0 || p->mcmc_warm || (it == 0 && ignore >= 1) || attempt >= kMaxRepeats ||
                    a->first_iteration + it >= kRepeatStride || p->last_nchain <= 1 || !p->hold_inflight)
                    break;
                int32_t valid = 0;
                if ((rc = mci_mcmc_launch_valid(p, &valid, nullptr, nullptr, nullptr))) return rc;
                if (valid) break;
                if ((rc = mci_iteration_discard(p))) return rc; // (the repeat overwrites this attempt's rows of the iteration log and of the block log)
                res_warmup += 1;
                p->last_discarded_neval += nevalperblock * (hi - lo);
                p->last_discarded_launches += 1;
            }
        }
        p->launch_counted = false;
        // the statistics of all iterations and the status word come back behind the last kernel in ONE synchronisation, into pinned memory (a
        // pageable destination goes through a staging copy: ~15 us of a 0.2 ms default-size call)
        HIPCHK(hipMemcpyAsync(h, p->d_iterlog + (size_t)row0 * p->nstat, nlog * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
        HIPCHK(hipMemcpyAsync(hstatus, p->d_status, sizeof(int), hipMemcpyDeviceToHost, p->ctx->stream));
        HIPCHK(hipStreamSynchronize(p->ctx->stream));
        if (persist && attempt == 0 && (*hstatus & mci::ST_PERSIST_STALL)) {
            // A grid-wide wait of the persistent launch ran out of time (its workgroups were not all resident: a device shared with another
            // long-running kernel).  Nothing of the call is lost: the map the call started from is restored from the copy persist_launch
            // took (workgroup 0 may have written its refined map back before another workgroup gave up), counters, histogram buffers and
            // the status word are reset, the iteration log is rewound, and the same iterations run through the launch chain (as every
            // later call of this problem does).
            if (p->d_edges_backup && p->h_edges.size())
                HIPCHK(hipMemcpyAsync(p->d_edges, p->d_edges_backup, p->h_edges.size() * sizeof(double), hipMemcpyDeviceToDevice, p->ctx->stream));
            if ((rc = persist_recover(p))) return rc;
            p->log_row = row0;
            persist = false;
            if ((rc = compile_solver(p, kslot(a->solver, a->measurefreq)))) return rc;
            continue;
        }
        break;
    }
    if (*hstatus && (rc = check_status(p))) return rc; // (reads it again, clears it, names the failure)
    auto t1 = std::chrono::steady_clock::now();
    res->seconds = std::chrono::duration<double>(t1 - t0).count();
    res->neval = 0;
    for (int it = 0; it < a->niter; ++it) { // main.jl:203
        const double *row = h + (size_t)it * p->nstat;
        mci_mean_std(row, row + s.nobs, s.nobs, block, res->iter_mean + (size_t)it * s.nobs, res->iter_std + (size_t)it * s.nobs);
        res->neval += (int64_t)row[2 * s.nobs + 1];
        if (res->visited && it == a->niter - 1) memcpy(res->visited, row + 2 * s.nobs + 2, (size_t)(s.ni + 1) * sizeof(double));
    }
    for (int o = 0; o < s.nobs; ++o) // main.jl:211 -> statistics.jl:24-55
        mci_average(res->iter_mean + o, res->iter_std + o, s.nobs, ignore + 1, a->niter, &res->mean[o], &res->stdev[o], &res->chi2[o]);
    // Carried chains: consecutive iterations are not independent, which statistics.jl:186-220 assumes -- but the blocks are (a block's
    // chains descend from that block's chains only), so the error comes from the scatter of the blocks' weighted averages over the run
    // (mci_lineage_sums + the reference's own _mean_std over them); same weights, same mean.
    res->correlated = 0;
    res->warmup = res_warmup;
    if (a->solver != MCI_VEGAS && blk_row0 >= 0 && p->blk_rows - blk_row0 == a->niter && p->blk_carried > 0 && a->niter > ignore + 1) {
        const int64_t nb = hi - lo;
        std::vector<double> bm((size_t)a->niter * nb * s.nobs), sums(2 * (size_t)s.nobs);
        HIPCHK(hipMemcpyAsync(bm.data(), p->d_blocklog + (size_t)blk_row0 * p->blk_stride, bm.size() * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
        HIPCHK(hipStreamSynchronize(p->ctx->stream));
        mci_lineage_sums(bm.data(), a->niter, nb, s.nobs, res->iter_std, ignore + 1, a->niter, sums.data(), sums.data() + s.nobs);
        if ((rc = comm_sum_host(p, sums.data(), (int)sums.size()))) return rc;
        std::vector<double> lm(s.nobs), le(s.nobs);
        mci_mean_std(sums.data(), sums.data() + s.nobs, s.nobs, block, lm.data(), le.data());
        // (a column that is identically zero -- the imaginary part of a real integrand -- keeps the reference's 1e-10-regularised error,
        // statistics.jl:192-198, instead of an exact 0)
        for (int o = 0; o < s.nobs; ++o) res->stdev[o] = le[o] > 0.0 ? le[o] : res->stdev[o];
        res->correlated = 1;
    }
    return MCI_OK;
}

// ---------------------------------------------------------------------------------------------------
// state access
// ---------------------------------------------------------------------------------------------------
int mci_get_iteration_log(mci_problem *p, int32_t nrows, double *out) {
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    if (nrows < 1 || nrows > p->log_row) return fail(MCI_ERR_INVALID, "only %d iterations are logged", p->log_row);
    HIPCHK(hipMemcpyAsync(out, p->d_iterlog + (size_t)(p->log_row - nrows) * p->nstat, (size_t)nrows * p->nstat * sizeof(double),
                          hipMemcpyDeviceToHost, p->ctx->stream));
    return check_status(p); // synchronises; surfaces normalization / histogram errors of the logged iterations
}

int mci_get_packed(mci_problem *p, double *out, int64_t n) {
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    if (n != p->packed_n && n != p->packed_n + 64) // (+ 64: with the :mcmc holding-time counts an external reducer sums too, mci_reduce_size)
        return fail(MCI_ERR_INVALID, "packed size is %lld", (long long)p->packed_n);
    if (int rc = flush_merge(p)) return rc;
    HIPCHK(hipMemcpyAsync(out, p->d_packed, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

int mci_set_packed(mci_problem *p, const double *in, int64_t n) {
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    if (n != p->packed_n && n != p->packed_n + 64) // (+ 64: with the :mcmc holding-time counts an external reducer sums too, mci_reduce_size)
        return fail(MCI_ERR_INVALID, "packed size is %lld", (long long)p->packed_n);
    if (int rc = flush_merge(p)) return rc;
    HIPCHK(hipMemcpyAsync(p->d_packed, in, (size_t)n * sizeof(double), hipMemcpyHostToDevice, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

void *mci_packed_device_ptr(mci_problem *p) {
    if (!p || p->ctx->offline || flush_merge(p)) return nullptr;
    return (void *)p->d_packed;
}

int mci_get_grid(mci_problem *p, int32_t leaf, double *out, int32_t n) {
    if (leaf < 0 || leaf >= (int)p->leaves.size() || p->leaves[leaf].kind != MCI_CONTINUOUS) return fail(MCI_ERR_INVALID, "leaf %d is not Continuous", leaf);
    const Leaf &L = p->leaves[leaf];
    if (n != L.npts) return fail(MCI_ERR_INVALID, "grid has %d points", L.npts);
    if (p->ctx->offline) {
        memcpy(out, p->h_edges.data() + L.eoff, n * sizeof(double));
        return MCI_OK;
    }
    HIPCHK(hipMemcpyAsync(out, p->d_edges + L.eoff, n * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

int mci_set_grid(mci_problem *p, int32_t leaf, const double *grid, int32_t n) {
    if (leaf < 0 || leaf >= (int)p->leaves.size() || p->leaves[leaf].kind != MCI_CONTINUOUS) return fail(MCI_ERR_INVALID, "leaf %d is not Continuous", leaf);
    const Leaf &L = p->leaves[leaf];
    if (n != L.npts) return fail(MCI_ERR_INVALID, "grid has %d points (the number of points is fixed at creation)", L.npts);
    for (int i = 1; i < n; ++i)
        if (!(grid[i] > grid[i - 1])) return fail(MCI_ERR_INVALID, "grid must be strictly increasing");
    memcpy(p->h_edges.data() + L.eoff, grid, n * sizeof(double));
    if (p->ctx->offline) return MCI_OK;
    HIPCHK(hipMemcpyAsync(p->d_edges + L.eoff, grid, n * sizeof(double), hipMemcpyHostToDevice, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

int mci_get_distribution(mci_problem *p, int32_t leaf, double *dist, double *acc, int32_t k) {
    if (leaf < 0 || leaf >= (int)p->leaves.size() || p->leaves[leaf].kind != MCI_DISCRETE) return fail(MCI_ERR_INVALID, "leaf %d is not Discrete", leaf);
    const Leaf &L = p->leaves[leaf];
    if (k != L.nbin) return fail(MCI_ERR_INVALID, "distribution has %d entries", L.nbin);
    if (p->ctx->offline) {
        if (dist) memcpy(dist, p->h_ddist.data() + L.doff, k * sizeof(double));
        if (acc) memcpy(acc, p->h_dacc.data() + L.eoff, (k + 1) * sizeof(double));
        return MCI_OK;
    }
    if (dist) HIPCHK(hipMemcpyAsync(dist, p->d_ddist + L.doff, k * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    if (acc) HIPCHK(hipMemcpyAsync(acc, p->d_dacc + L.eoff, (k + 1) * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

int mci_set_distribution(mci_problem *p, int32_t leaf, const double *dist, int32_t k) {
    if (leaf < 0 || leaf >= (int)p->leaves.size() || p->leaves[leaf].kind != MCI_DISCRETE) return fail(MCI_ERR_INVALID, "leaf %d is not Discrete", leaf);
    const Leaf &L = p->leaves[leaf];
    if (k != L.nbin) return fail(MCI_ERR_INVALID, "distribution has %d entries", L.nbin);
    double sum = 0.0;
    for (int i = 0; i < k; ++i) {
        if (!(dist[i] >= 0.0)) return fail(MCI_ERR_INVALID, "distribution should be all non-negative!");
        sum += dist[i];
    }
    double run = 0.0;
    p->h_dacc[L.eoff] = 0.0;
    for (int i = 0; i < k; ++i) {
        p->h_ddist[L.doff + i] = dist[i] / sum;
        run += p->h_ddist[L.doff + i];
        p->h_dacc[L.eoff + i + 1] = run;
    }
    if (p->ctx->offline) return MCI_OK;
    HIPCHK(hipMemcpyAsync(p->d_ddist + L.doff, p->h_ddist.data() + L.doff, k * sizeof(double), hipMemcpyHostToDevice, p->ctx->stream));
    HIPCHK(hipMemcpyAsync(p->d_dacc + L.eoff, p->h_dacc.data() + L.eoff, (k + 1) * sizeof(double), hipMemcpyHostToDevice, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

int mci_get_reweight(mci_problem *p, double *out, int32_t n) {
    if (n != p->ni + 1) return fail(MCI_ERR_INVALID, "reweight has %d entries", p->ni + 1);
    if (p->ctx->offline) {
        memcpy(out, p->h_reweight.data(), n * sizeof(double));
        return MCI_OK;
    }
    HIPCHK(hipMemcpyAsync(out, p->d_reweight, n * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

int mci_set_reweight(mci_problem *p, const double *in, int32_t n) {
    if (n != p->ni + 1) return fail(MCI_ERR_INVALID, "Wrong reweight vector size! Note that the last element in reweight vector is for the normalization diagram."); // configuration.jl:174
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        if (!(in[i] > 0)) return fail(MCI_ERR_INVALID, "All reweight factors should be positive."); // configuration.jl:175
        s += in[i];
    }
    for (int i = 0; i < n; ++i) p->h_reweight[i] = in[i] / s; // configuration.jl:173
    if (p->ctx->offline) return MCI_OK;
    HIPCHK(hipMemcpyAsync(p->d_reweight, p->h_reweight.data(), n * sizeof(double), hipMemcpyHostToDevice, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

int mci_get_acceptance(mci_problem *p, double *propose, double *accept, int32_t n) {
    if (n != p->npa) return fail(MCI_ERR_INVALID, "propose/accept have %d entries (3 x %d x %d)", p->npa, p->ni + 1, p->npa / (3 * (p->ni + 1)));
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    std::vector<double> h(2 * (size_t)p->npa);
    if (int rc = flush_merge(p)) return rc;
    HIPCHK(hipMemcpyAsync(h.data(), p->d_packed + p->nstat + p->shape.nbin, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    if (propose) memcpy(propose, h.data(), (size_t)p->npa * sizeof(double));
    if (accept) memcpy(accept, h.data() + p->npa, (size_t)p->npa * sizeof(double));
    return MCI_OK;
}

int mci_set_reweight_goal(mci_problem *p, const double *goal, int32_t n) {
    if (!goal || n == 0) {
        p->h_goal.clear();
        return MCI_OK;
    }
    if (n != p->ni + 1) return fail(MCI_ERR_INVALID, "reweight_goal has %d entries", p->ni + 1);
    p->h_goal.assign(goal, goal + n);
    if (p->ctx->offline) return MCI_OK;
    if (!p->d_goal) HIPCHK(hipMalloc((void **)&p->d_goal, (size_t)n * sizeof(double)));
    HIPCHK(hipMemcpyAsync(p->d_goal, p->h_goal.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

// ---------------------------------------------------------------------------------------------------
// resume across processes: the reference keeps trained state only in memory (`config = res.config`,
// docs/src/index.md:129) and defines no file format; this is a small self-describing binary dump of what
// `train!` and `doReweight!` have learned: grids, distributions, reweight.
//   "MCISTATE" | u32 version | u32 nleaf | u32 ni | per leaf: u32 kind, u32 n | f64 reweight[ni+1] |
//   per leaf: f64 grid[n]  (Continuous)  or  f64 distribution[n]  (Discrete)
// ---------------------------------------------------------------------------------------------------
int mci_save_state(mci_problem *p, const char *path) {
    if (!p || !path) return fail(MCI_ERR_INVALID, "NULL argument");
    std::vector<double> rw(p->ni + 1);
    int rc = mci_get_reweight(p, rw.data(), p->ni + 1);
    if (rc) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) return fail(MCI_ERR_INVALID, "cannot open %s for writing", path);
    const uint32_t hdr[3] = {1u, (uint32_t)p->leaves.size(), (uint32_t)p->ni};
    bool ok = fwrite("MCISTATE", 1, 8, f) == 8 && fwrite(hdr, sizeof(uint32_t), 3, f) == 3;
    for (auto &L : p->leaves) { // (a FermiK leaf has nothing trained: header entry only, n = 0)
        const uint32_t kn[2] = {(uint32_t)L.kind, (uint32_t)(L.kind == MCI_CONTINUOUS ? L.npts : L.kind == MCI_DISCRETE ? L.nbin : 0)};
        ok = ok && fwrite(kn, sizeof(uint32_t), 2, f) == 2;
    }
    ok = ok && fwrite(rw.data(), sizeof(double), rw.size(), f) == rw.size();
    for (size_t l = 0; l < p->leaves.size() && ok; ++l) {
        const Leaf &L = p->leaves[l];
        if (L.kind == MCI_FERMIK) continue;
        const int n = L.kind == MCI_CONTINUOUS ? L.npts : L.nbin;
        std::vector<double> v(n);
        rc = L.kind == MCI_CONTINUOUS ? mci_get_grid(p, (int)l, v.data(), n) : mci_get_distribution(p, (int)l, v.data(), nullptr, n);
        if (rc) { fclose(f); return rc; }
        ok = fwrite(v.data(), sizeof(double), (size_t)n, f) == (size_t)n;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? MCI_OK : fail(MCI_ERR_INVALID, "short write to %s", path);
}

int mci_load_state(mci_problem *p, const char *path) {
    if (!p || !path) return fail(MCI_ERR_INVALID, "NULL argument");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(MCI_ERR_INVALID, "cannot open %s", path);
    char magic[8];
    uint32_t hdr[3];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "MCISTATE", 8) || fread(hdr, sizeof(uint32_t), 3, f) != 3 || hdr[0] != 1u) {
        fclose(f);
        return fail(MCI_ERR_INVALID, "%s is not a version-1 MCISTATE file", path);
    }
    if (hdr[1] != p->leaves.size() || hdr[2] != (uint32_t)p->ni) {
        fclose(f);
        return fail(MCI_ERR_INVALID, "%s holds %u variables / %u integrands, the problem has %zu / %d", path, hdr[1], hdr[2], p->leaves.size(), p->ni);
    }
    for (size_t l = 0; l < p->leaves.size(); ++l) {
        uint32_t kn[2];
        const Leaf &L = p->leaves[l];
        if (fread(kn, sizeof(uint32_t), 2, f) != 2 || kn[0] != (uint32_t)L.kind ||
            kn[1] != (uint32_t)(L.kind == MCI_CONTINUOUS ? L.npts : L.kind == MCI_DISCRETE ? L.nbin : 0)) {
            fclose(f);
            return fail(MCI_ERR_INVALID, "%s: variable %zu does not match the problem (kind / number of grid points)", path, l);
        }
    }
    std::vector<double> rw(p->ni + 1);
    bool ok = fread(rw.data(), sizeof(double), rw.size(), f) == rw.size();
    std::vector<std::vector<double>> tabs(p->leaves.size());
    for (size_t l = 0; l < p->leaves.size() && ok; ++l) {
        const Leaf &L = p->leaves[l];
        tabs[l].resize(L.kind == MCI_CONTINUOUS ? L.npts : L.kind == MCI_DISCRETE ? L.nbin : 0);
        ok = fread(tabs[l].data(), sizeof(double), tabs[l].size(), f) == tabs[l].size();
    }
    fclose(f);
    if (!ok) return fail(MCI_ERR_INVALID, "%s is truncated", path);
    int rc = mci_set_reweight(p, rw.data(), p->ni + 1);
    for (size_t l = 0; l < p->leaves.size() && !rc; ++l)
        if (p->leaves[l].kind != MCI_FERMIK)
        rc = p->leaves[l].kind == MCI_CONTINUOUS ? mci_set_grid(p, (int)l, tabs[l].data(), (int)tabs[l].size())
                                                 : mci_set_distribution(p, (int)l, tabs[l].data(), (int)tabs[l].size());
    return rc;
}

int mci_sample_dump(mci_problem *p, int32_t iteration, uint64_t seed, int64_t nevalperblock, int64_t block_index, int64_t n,
                    double *x, double *jac, double *w) {
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    if (n < 1 || n > nevalperblock) return fail(MCI_ERR_INVALID, "n must be in 1..neval_per_block");
    int rc = ensure_dump(p);
    if (rc) return rc;
    HIPCHK(hipSetDevice(p->ctx->device));
    const auto &s = p->shape;
    const int64_t per = s.ndraw + 1 + s.ni * s.ncomp;
    if (n * per > p->cap_dump) {
        if (p->d_dump) (void)hipFree(p->d_dump);
        p->d_dump = nullptr;
        HIPCHK(hipMalloc((void **)&p->d_dump, (size_t)(n * per) * sizeof(double)));
        p->cap_dump = n * per;
    }
    mci::DumpArgs a{};
    a.edges = p->d_edges;
    a.dacc = p->d_dacc;
    a.ddist = p->d_ddist;
    a.ud = p->d_ud;
    a.x = p->d_dump;
    a.jac = p->d_dump + n * s.ndraw;
    a.w = a.jac + n;
    a.seed = seed;
    a.iteration = (mci::u32)iteration;
    a.first_index = block_index * nevalperblock;
    a.n = n;
    void *args[] = {&a};
    const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    HIPCHK(hipModuleLaunchKernel(p->f_dump, grid, 1, 1, 256, 1, 1, (unsigned)p->lds_bytes, p->ctx->stream, args, nullptr));
    if (x) HIPCHK(hipMemcpyAsync(x, a.x, (size_t)n * s.ndraw * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    if (jac) HIPCHK(hipMemcpyAsync(jac, a.jac, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    if (w) HIPCHK(hipMemcpyAsync(w, a.w, (size_t)n * s.ni * s.ncomp * sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK;
}

int mci_set_kernel_timing(mci_problem *p, int32_t mode) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    p->kernel_timing = mode < 0 ? -1 : mode > 0 ? 1 : 0;
    return MCI_OK;
}

int mci_kernel_times_ms(mci_problem *p, float *ms, int32_t n, int32_t *got, int32_t *wg, int32_t *threads) {
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    int64_t have = p->launches < mci_problem::kEvRing ? p->launches : mci_problem::kEvRing;
    if (have > n) have = n;
    int64_t k = 0;
    for (int64_t i = 0; i < have; ++i) { // oldest first; launches that ran without events (mci_set_kernel_timing) are skipped
        const int slot = (int)((p->launches - have + i) % mci_problem::kEvRing);
        if (!p->ev_valid[slot]) continue;
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, p->evs[2 * slot], p->evs[2 * slot + 1]));
        ms[k++] = t;
    }
    have = k;
    if (got) *got = (int32_t)have;
    if (wg) *wg = p->last_wg;
    if (threads) *threads = p->last_threads;
    return MCI_OK;
}

// shader clock of the last n timed :vegas launches (oldest first), MHz: ticks of s_memtime (shader cycles) over ticks of s_memrealtime
// (the device's constant-rate reference, hipDeviceAttributeWallClockRate) across the sample loop of workgroup 0's first wave
int mci_kernel_clocks(mci_problem *p, double *mhz, int32_t n, int32_t *got) {
    if (!p || !mhz || !got) return fail(MCI_ERR_INVALID, "NULL argument");
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    *got = 0;
    if (!p->d_clocks) return MCI_OK;
    HIPCHK(hipSetDevice(p->ctx->device));
    int khz = 0;
    HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, p->ctx->device));
    std::vector<unsigned long long> h((size_t)2 * mci_problem::kEvRing);
    HIPCHK(hipMemcpyAsync(h.data(), p->d_clocks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    int64_t have = p->launches < mci_problem::kEvRing ? p->launches : mci_problem::kEvRing;
    if (have > n) have = n;
    int32_t k = 0;
    for (int64_t i = 0; i < have; ++i) {
        const int slot = (int)((p->launches - have + i) % mci_problem::kEvRing);
        if (!p->ev_valid[slot] || !h[(size_t)2 * slot + 1]) continue;
        mhz[k++] = (double)h[(size_t)2 * slot] / (double)h[(size_t)2 * slot + 1] * (double)khz * 1.0e-3;
    }
    *got = k;
    return MCI_OK;
}

// ---------------------------------------------------------------------------------------------------
// host-side statistics (pure functions)
// ---------------------------------------------------------------------------------------------------
void mci_standardize_block(int64_t neval, int64_t nblock, int64_t nworker, int64_t *nevalperblock, int64_t *block) {
    (void)neval;
    if (nblock > nworker) nblock = (nblock / nworker) * nworker; // main.jl:225-227
    else nblock = nworker;                                       // main.jl:229
    *nevalperblock = neval / nblock;                             // main.jl:232
    *block = nblock;
}

double mci_chain_burnin(int64_t steps, int64_t nchain, int32_t nslots) {
    double thr = (double)steps / 100.0; // vegas_mc/montecarlo.jl:213  `ne >= neval / 100`
    if (nchain > 1) {                   // many short chains: every chain must forget its start (DESIGN.md "chains")
        double fl = 64.0 * (double)nslots;
        if (fl > (double)steps / 2.0) fl = (double)steps / 2.0;
        if (fl > thr) thr = fl;
    }
    return thr;
}

int64_t mci_mcmc_burnin(int64_t steps, int64_t nchain, int32_t nslots, int32_t nd, int32_t npool, double thermal_ratio) {
    int64_t nburn = (int64_t)floor((double)steps * thermal_ratio); // mcmc/montecarlo.jl:133
    if (nchain > 1) { // many short chains: every chain must forget its start (DESIGN.md "chains")
        int64_t fl = 64 * (int64_t)nslots + 16 * (int64_t)(npool + 1) * nd;
        if (fl > nburn) nburn = fl;
    }
    return nburn;
}

int64_t mci_mcmc_auto_chains(int64_t nevalperblock, int64_t nblocks, int32_t nslots, int32_t nd, int32_t npool, int64_t hold_max,
                             int64_t hold_len, int32_t carried) {
    // Chain length (measured steps) of an automatic :mcmc launch.  hold_max = the longest time any chain's slot (or integrand index)
    // went without changing in the launch before (upper edge of the top occupied bucket), hold_len = the chain length of that launch
    // (0: no growth cap -- the holds were measured by chains that were long enough for them).
    //   nothing measured (hold_max = 0): pilot-length chains, kMcmcPilotSteps or 2 burn-in floors -- the first iteration trains the map
    //     and is ignored by default (main.jl:82); its holds are those of the UNTRAINED map, up to 2^13 steps on BASELINE configs[4]
    //     where the trained map holds for 2^8: chains sized for them (131072 steps in the rounds before) cost 0.74 s of a cold call
    //   fresh chains: 16 x hold_max, never fewer than 8 burn-in floors.  Calibration (profiles/r01_chain_bias.txt): on the bubble
    //     diagram chains of 1-2 x that holding time are ~1e-3 off, chains of 8 x are unbiased at the 5e-4 level of the measurement
    //   carried chains (stationary starts): 4 x hold_max, never fewer than 1 floor (profiles/r03_chain_carry.txt, r04_mcmc_policy.txt D)
    //   at most kMcmcGrow x hold_len: a hold longer than a quarter of the chain that measured it is censored by that chain's
    //     length -- what it says is "longer", not how long -- so the length escalates by that factor per launch until the
    //     measured holds fit (heavy-tailed integrands: 2^14..2^15 steps on the bubble diagram and on 1/(1 - cos^3)) instead of
    //     jumping to 8-16 x a number the untrained map inflated.  The launches on the way are warm-up: mci_integrate repeats them
    //     (mci_mcmc_launch_valid) -- together they cost less than the first launch that is long enough, a geometric series
    const int64_t fl = 64 * (int64_t)nslots + 16 * (int64_t)(npool + 1) * nd;
    int64_t len, floor_len = carried ? mci_problem::kMcmcCarryHalfFloors * fl / 2 : 8 * fl;
    if (hold_max <= 0) {
        len = mci_problem::kMcmcPilotSteps;
        floor_len = 2 * fl;
    } else {
        len = (carried ? mci_problem::kMcmcCarryHolds : 16) * hold_max;
        if (hold_len > 0 && len > mci_problem::kMcmcGrow * hold_len) len = mci_problem::kMcmcGrow * hold_len;
    }
    if (len < floor_len) len = floor_len;
    int64_t nchain = nevalperblock / len;
    const int64_t cap = mci_problem::kChainFill / (nblocks > 0 ? nblocks : 1) > 64 ? mci_problem::kChainFill / (nblocks > 0 ? nblocks : 1) : 64;
    if (nchain > cap) nchain = cap;
    if (nchain < 1) nchain = 1;
    return nchain;
}

int mci_get_block_means(mci_problem *p, int32_t rows, double *out, int64_t *nblocks, int32_t *carried) {
    if (!p || rows < 0 || (rows > 0 && !out)) return fail(MCI_ERR_INVALID, "bad argument");
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    if (rows > p->blk_rows) return fail(MCI_ERR_INVALID, "the block log holds %lld iterations, %d asked for", (long long)p->blk_rows, (int)rows);
    HIPCHK(hipSetDevice(p->ctx->device));
    int rc = flush_merge(p);
    if (rc) return rc;
    if (nblocks) *nblocks = p->shape.nobs > 0 ? p->blk_stride / p->shape.nobs : 0;
    if (carried) *carried = p->blk_carried;
    if (rows > 0) {
        HIPCHK(hipMemcpyAsync(out, p->d_blocklog + (size_t)(p->blk_rows - rows) * p->blk_stride, (size_t)rows * p->blk_stride * sizeof(double),
                              hipMemcpyDeviceToHost, p->ctx->stream));
        HIPCHK(hipStreamSynchronize(p->ctx->stream));
    }
    return MCI_OK;
}

int mci_comm_sum(mci_problem *p, double *v, int32_t n) {
    if (!p || !v || n < 0) return fail(MCI_ERR_INVALID, "bad argument");
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    HIPCHK(hipSetDevice(p->ctx->device));
    return comm_sum_host(p, v, n);
}

int mci_reset_block_log(mci_problem *p) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    p->blk_rows = 0;
    p->blk_carried = 0;
    return MCI_OK;
}

int mci_mcmc_launch_valid(mci_problem *p, int32_t *valid, int32_t *warm, int64_t *chain_len, int64_t *hold_max) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    HIPCHK(hipSetDevice(p->ctx->device));
    int rc = hold_consume(p); // (waits for the last :mcmc launch's sample kernel if its histogram is still in flight)
    if (rc) return rc;
    if (valid) *valid = (p->hold_valid || !p->hold_measured) ? 1 : 0; // (nothing measured: nothing to hold the launch against)
    if (warm) *warm = p->mcmc_warm ? 1 : 0;
    if (chain_len) *chain_len = p->hold_len;
    if (hold_max) *hold_max = p->hold_max;
    return MCI_OK;
}

int mci_iteration_discard(mci_problem *p) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (p->log_row < 1) return fail(MCI_ERR_INVALID, "no finished iteration to discard");
    p->log_row -= 1;
    if (p->blk_rows > 0) {
        p->blk_rows -= 1;
        p->blk_carried -= (p->last_carried && p->blk_carried > 0) ? 1 : 0;
    }
    return MCI_OK;
}

int mci_get_hold_histogram(mci_problem *p, uint64_t *out64) {
    if (!p || !out64) return fail(MCI_ERR_INVALID, "NULL argument");
    if (p->ctx->offline) return fail(MCI_ERR_NO_DEVICE, "offline context");
    memset(out64, 0, 64 * sizeof(uint64_t));
    if (!p->d_hold) return MCI_OK;
    HIPCHK(hipSetDevice(p->ctx->device));
    HIPCHK(hipMemcpyAsync(out64, p->d_hold, 64 * sizeof(uint64_t), hipMemcpyDeviceToHost, p->ctx->stream));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    return MCI_OK; // (diagnostic read; the automatic chain length follows its own copies, hold_publish / hold_consume)
}

void mci_maxdof(const int32_t *dof, int32_t nd, int32_t npool, int32_t *out) {
    for (int v = 0; v < npool; ++v) {
        int m = 0;
        for (int i = 0; i < nd; ++i) m = dof[(size_t)i * npool + v] > m ? dof[(size_t)i * npool + v] : m;
        out[v] = m;
    }
}

void mci_mean_std(const double *obs_sum, const double *obs_sq, int64_t n, int64_t block, double *mean, double *std) {
    for (int64_t o = 0; o < n; ++o) {
        const double m = obs_sum[o] / (double)block; // main.jl:317
        mean[o] = m;
        if (block > 1) {
            const double v = (obs_sq[o] / (double)block - m * m) / (double)(block - 1); // main.jl:308
            std[o] = v < 0.0 ? 0.0 : sqrt(v);                                           // main.jl:297-299
        } else {
            std[o] = 0.0; // main.jl:311
        }
    }
}

void mci_average(const double *iter_mean, const double *iter_std, int64_t stride, int64_t init, int64_t max, double *mean,
                 double *err, double *chi2) {
    if (max <= init) { // statistics.jl:189-191
        *mean = iter_mean[0];
        *err = iter_std[0];
        *chi2 = 0.0;
        return;
    }
    double wsum = 0.0, mea = 0.0, c2 = 0.0;
    for (int64_t i = init; i <= max; ++i) { // statistics.jl:217
        const double sd = iter_std[(i - 1) * stride] + 1.0e-10;
        wsum += 1.0 / (sd * sd);
    }
    for (int64_t i = init; i <= max; ++i) { // statistics.jl:197
        const double sd = iter_std[(i - 1) * stride] + 1.0e-10;
        mea += iter_mean[(i - 1) * stride] * (1.0 / (sd * sd)) / wsum;
    }
    for (int64_t i = init; i <= max; ++i) { // statistics.jl:200
        const double sd = iter_std[(i - 1) * stride] + 1.0e-10;
        const double dlt = iter_mean[(i - 1) * stride] - mea;
        c2 += (1.0 / (sd * sd)) * dlt * dlt;
    }
    *mean = mea;
    *err = 1.0 / sqrt(wsum);                      // statistics.jl:198
    *chi2 = c2 / (double)((max - init + 1) - 1);  // statistics.jl:204
}

void mci_lineage_sums(const double *block_means, int64_t niter, int64_t nblocks, int64_t nobs, const double *iter_std, int64_t init,
                      int64_t max, double *sum, double *sumsq) {
    (void)niter;
    for (int64_t o = 0; o < nobs; ++o) {
        double wsum = 0.0; // the weights of statistics.jl:217, :197
        for (int64_t i = init; i <= max; ++i) {
            const double sd = iter_std[(i - 1) * nobs + o] + 1.0e-10;
            wsum += 1.0 / (sd * sd);
        }
        double s1 = 0.0, s2 = 0.0;
        for (int64_t b = 0; b < nblocks; ++b) {
            double mb = 0.0; // this block's lineage: its weighted average over the iterations
            for (int64_t i = init; i <= max; ++i) {
                const double sd = iter_std[(i - 1) * nobs + o] + 1.0e-10;
                mb += block_means[((i - 1) * nblocks + b) * nobs + o] * (1.0 / (sd * sd)) / wsum;
            }
            s1 += mb;
            s2 += mb * mb;
        }
        sum[o] = s1;
        sumsq[o] = s2;
    }
}

void mci_do_reweight(double *reweight, const double *visited, int64_t nd, double gamma, const double *goal) {
    double avgstep = 0.0;
    for (int64_t i = 0; i < nd; ++i) avgstep += visited[i]; // main.jl:323
    for (int64_t i = 0; i < nd; ++i) {                      // main.jl:324-331
        if (visited[i] <= 1) reweight[i] *= pow(avgstep, gamma);
        else reweight[i] *= pow(avgstep / visited[i], gamma);
    }
    if (goal) { // main.jl:334-337
        double gs = 0.0;
        for (int64_t i = 0; i < nd; ++i) gs += goal[i];
        for (int64_t i = 0; i < nd; ++i) reweight[i] *= goal[i] / gs;
    }
    double s = 0.0;
    for (int64_t i = 0; i < nd; ++i) s += reweight[i];
    for (int64_t i = 0; i < nd; ++i) reweight[i] /= s; // main.jl:339
}

} // extern "C"
