// mci_host_integrate.h -- part of the ONE translation unit mci_api.hip (included there, in order; not a stand-alone header):
// the persistent :vegas launch and the iteration loop mci_integrate (src/main.jl:142-218).
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

