// mci_host_types.h -- part of the ONE translation unit mci_api.hip (included there, in order; not a stand-alone header):
// the RCCL loader, mci_ctx, the problem's host-side state (mci_problem), the process-wide test overrides and the helpers every section uses (upload, capacity, status, the :mcmc holding-time hand-over).
namespace {

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
    // The parked stream of a many-grid :vegas problem (GBs) outlives the problem: destroying it hands the buffer to the context, the
    // next problem that needs one takes it.  The driver wipes VRAM on release, and a hipMalloc that lands on pages still being wiped
    // waits for them: a second engine right after a first one's 4.8 GB were freed took 380 ms for a 5 ms launch
    // (profiles/r06_other_configs.txt) -- the pattern of every sweep that builds a Configuration per call.  Freed by mci_ctx_destroy.
    std::mutex spare_mu;
    void *spare = nullptr;
    size_t spare_bytes = 0;
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
    // A several-lanes-per-chain code object proves itself before it is trusted (mci_host_jit.h spec_self_check): until a code object has
    // reproduced the lane-per-chain kernel's packed sums on THIS device (a marker file next to it in the kernel cache remembers that it did),
    // its first launch is preceded by a two-block, 512-step run through both kernels.  [0] :vegasmc, [1] :mcmc --
    // spec_state: 0 not looked at yet, 1 verified, -1 the check failed (one lane per chain from now on), -2 the unit did not compile
    // (automatic lanes: one lane per chain from now on)
    int spec_state[2] = {0, 0};
    bool spec_need_check[2] = {false, false}; // the loaded code object has no marker yet
    bool in_self_check = false;
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
    double *d_tile_w = nullptr;      // (one allocation: the weights, then -- 256-byte aligned -- the packed bins)
    uint32_t *d_tile_bins = nullptr;
    size_t tile_bytes = 0;           // its size
    int64_t cap_tile = 0;
    int64_t last_split_chunks = 0, last_split_bytes = 0; // chunks of the last many-grid :vegas launch | bytes of parked stream it held at a time
    int ntdraw = 0; // draws whose histogram lives in a tile >= 1
    int tdraw_words = 0; // 32-bit words of packed bins per parked sample (mci_device.h tdraw_words)
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
    bool clock_valid[512] = {};   // ... and: the launch of that slot stamped its shader clock (timed one-tile :vegas launches only, mci_kernel_clocks)
    int hcopy_auto = 1, hcopy_rule = 1; // in force | what the placement rule picked at create
    // deterministic mode (mci_set_deterministic): every solver's kernel keeps one histogram / observable copy per wave; the workgroup
    // size each was compiled for (the largest of 512 / 256 / 128 / 64 threads whose copies fit the CU's LDS)
    bool deterministic = false;
    int threads_det[3] = {0, 0, 0};
    bool hcopy_plan = false; // the rule also picked the workgroup size (512 threads) for the :vegas kernel
    // refinement walk of train! (variable.jl:227-234): -1 automatic -- the reference's serial recurrence whenever the sample
    // launch before it is long enough to hide its ~14 us per iteration (>= kSerialWalkSamples samples or chain steps on this
    // rank: 1 % of the headline iteration), the prefix-scan form below that; mci_set_train_walk forces one
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
struct Overrides { Override table_mode, hist_tile_bins, no_split_all, l1_phase, train_walk, hist_copies, fresh_floors, fresh_burnin_pct, spec_self_check, split_chunk; } g_over;
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
    if (!strcmp(key, "spec_self_check")) return &g_over.spec_self_check;
    if (!strcmp(key, "split_chunk")) return &g_over.split_chunk;
    return nullptr;
}
} // namespace

static void persist_job_drop(mci_problem *p);
namespace { void persist_orphans_join(); }
// counters [0..2] of the persistent :vegas kernel + (MCI_PERSIST_TRACE builds) the phase stamps of three workgroups over eight turns
static const size_t kPersistWords = 8 + 3 * 8 * 8 + 16;

namespace {

// the parked stream's buffer: from the context's spare one if that is big enough (and not more than twice as big), else hipMalloc
int tile_alloc(mci_problem *p, size_t bytes) {
    mci_ctx *c = p->ctx;
    void *base = nullptr;
    {
        std::lock_guard<std::mutex> g(c->spare_mu);
        if (c->spare && c->spare_bytes >= bytes && c->spare_bytes <= 2 * bytes + ((size_t)64 << 20)) {
            base = c->spare;
            p->tile_bytes = c->spare_bytes;
            c->spare = nullptr;
            c->spare_bytes = 0;
        }
    }
    if (!base) {
        HIPCHK(hipMalloc(&base, bytes));
        p->tile_bytes = bytes;
    }
    p->d_tile_w = (double *)base;
    return MCI_OK;
}
// ... and back: the context keeps the largest buffer it has been handed (work queued on the context's one stream is ordered behind
// the kernels that used it), anything else is freed
void tile_release(mci_problem *p) {
    if (!p->d_tile_w) return;
    mci_ctx *c = p->ctx;
    void *drop = p->d_tile_w;
    {
        std::lock_guard<std::mutex> g(c->spare_mu);
        if (p->tile_bytes > c->spare_bytes) {
            drop = c->spare;
            c->spare = p->d_tile_w;
            c->spare_bytes = p->tile_bytes;
        }
    }
    if (drop) (void)hipFree(drop);
    p->d_tile_w = nullptr;
    p->d_tile_bins = nullptr;
    p->tile_bytes = 0;
    p->cap_tile = 0;
}

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
    // (no wait for a copy still in flight -- this rank's own counts published behind the sample kernel when an external reducer sums the
    // packed buffer: the summed counts go to ANOTHER pinned buffer, and the event is simply recorded again behind them)
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

