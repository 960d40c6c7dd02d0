// mci_host_iteration.h -- part of the ONE translation unit mci_api.hip (included there, in order; not a stand-alone header):
// one iteration: mci_iteration_run (sample launch plans of the three solvers), reduce (the ONE all-reduce), finish (train!, doReweight!, statistics).
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
        if (g_over.fresh_burnin_pct.on && !may_carry && nchain > 1 && auto_chains) { // (experiment: tools/run_batch.sh r05_floors)
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
    if (G > 1 && p->spec_state[solver - 1] < 0) G = 1; // (its code object failed its self-check, or did not compile: one lane per chain)
    if (G > 1) {
        rc = compile_spec(p, solver);
        if (rc == MCI_ERR_COMPILE && p->spec_lanes == -1) {
            // automatic lanes: a unit that does not compile (up to 512 VGPRs, many bpermutes; a backend switch a later compiler may
            // refuse) must not take the solver down with it -- the lane-per-chain kernel steps the same chains
            fprintf(stderr, "mci: the several-lanes-per-chain kernel of this problem did not compile; one lane per chain instead\n%s\n", mci_last_error());
            p->spec_state[solver - 1] = -2;
            G = 1;
        } else if (rc) return rc;
    }
    if (G > 1 && !p->in_self_check && ((p->spec_need_check[solver - 1] && !(g_over.spec_self_check.on && g_over.spec_self_check.v == 0)) ||
                                       (g_over.spec_self_check.on && g_over.spec_self_check.v == 1 && p->spec_state[solver - 1] == 0))) {
        if ((rc = spec_self_check(p, solver, G, nevalperblock, block_lo, block_hi, iteration, seed, measurefreq, thermal_ratio))) return rc;
        if (p->spec_state[solver - 1] < 0) G = 1;
    }
    if (G > 1) {
        // the trees: the one built for the acceptance that was given, else the solver's family (spec_upload)
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
    // Many-grid launches park (weights, bins) of every sample for the replay.  The stream is bounded whatever neval is -- the reference's
    // loop allocates nothing per sample (vegas/montecarlo.jl:117-187) -- by running the launch in chunks of a block's samples: sample pass
    // -> replay per chunk, same Philox indices, the partial rows of a later chunk added to those before it (BatchArgs::chunk_lo).  A chunk
    // is at most 2^27 samples over all blocks and at most 7.5 GB of parked stream (C4, 48 B per sample: all of neval = 1e8 in one
    // chunk as before, neval = 1e10 in 75); host closures read the whole launch's stream and keep the one chunk (they are refused above 8 GiB).
    int64_t chunk_len = nevalperblock, nchunks = 1;
    if (split) {
        const int64_t words = p->tdraw_words > 0 ? p->tdraw_words : 1;
        const int64_t bytes = (int64_t)s.ni * 8 + words * 4;
        if (!s.host_integrand && !s.host_measure) {
            int64_t cap = (int64_t)1 << 27;
            if (cap * bytes > (int64_t)7500000000) cap = (int64_t)7500000000 / bytes;
            if (g_over.split_chunk.on && g_over.split_chunk.v > 0) cap = g_over.split_chunk.v;
            int64_t per = (cap / nblocks) & ~(int64_t)3; // (a multiple of four: the replay reads four consecutive samples per lane as 16-byte loads)
            if (per < 4) per = 4;
            if (per < chunk_len) {
                chunk_len = per;
                nchunks = (nevalperblock + chunk_len - 1) / chunk_len;
            }
        }
        const int64_t nsamp = nblocks * chunk_len;
        if (nsamp > p->cap_tile) {
            tile_release(p);
            const size_t wbytes = (((size_t)nsamp * s.ni * sizeof(double)) + 255) & ~(size_t)255; // (the bins start 256-byte aligned: 16-byte loads)
            if ((rc = tile_alloc(p, wbytes + (size_t)nsamp * words * sizeof(uint32_t)))) return rc;
            p->d_tile_bins = (uint32_t *)((char *)p->d_tile_w + wbytes);
            p->cap_tile = nsamp;
        }
        p->last_split_chunks = nchunks;
        p->last_split_bytes = nsamp * bytes;
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
    a.tile_stride = nblocks * chunk_len;
    a.chunk_lo = 0;
    a.chunk_hi = nevalperblock;
    a.chunk_len = chunk_len;
    a.accum = 0;
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
    for (int64_t c = 0; c < nchunks; ++c) { // (one trip, except for a many-grid launch whose parked stream is bounded: sample pass -> replay per chunk)
        if (nchunks > 1) {
            a.chunk_lo = c * chunk_len;
            a.chunk_hi = a.chunk_lo + chunk_len < nevalperblock ? a.chunk_lo + chunk_len : nevalperblock;
            a.accum = c > 0 ? 1 : 0;
        }
        HIPCHK(hipModuleLaunchKernel(f, (unsigned)nwg, 1, 1, (unsigned)T_launch, 1, 1, (unsigned)solver_lds(p, solver), st, args, nullptr));
        if (split && c + 1 < nchunks)
            HIPCHK(hipModuleLaunchKernel(p->f_tiles[kern == kSlotVegasAny ? 1 : 0], (unsigned)(((hist_rows + 7) / 8) * 8 * (s.ntile - (s.split_all ? 0 : 1))), 1, 1, (unsigned)T, 1, 1, (unsigned)p->lds_bytes, st, args, nullptr));
    }
    if (solver == MCI_MCMC) p->hold_measured = a.hold_hist != nullptr;
    // (an explicit chain count: nobody sizes a launch from this one's holds, and the host keeps queueing launches back to back)
    if (a.hold_hist && auto_chains && (rc = hold_publish(p, nevalperblock / nchain, solver != MCI_VEGAS && p->last_carried))) return rc;
    if (split) // (the replay of the one chunk, or of the last one)
        HIPCHK(hipModuleLaunchKernel(p->f_tiles[kern == kSlotVegasAny ? 1 : 0], (unsigned)(((hist_rows + 7) / 8) * 8 * (s.ntile - (s.split_all ? 0 : 1))), 1, 1, (unsigned)T, 1, 1, (unsigned)p->lds_bytes, st, args, nullptr));
    if (p->time_this_launch) HIPCHK(hipEventRecord(p->evs[2 * slot + 1], st));
    p->ev_valid[slot] = p->time_this_launch;
    p->clock_valid[slot] = a.clock_out != nullptr && !split && s.ntile == 1; // (what the kernel stamps: mci_device.h vegas_batch `stamp`)
    p->launches += 1;
    if (s.host_measure) {
        // the closure cannot run on the device: this launch's (measured) configurations and relative weights go to the host
        // (draw-major, like the host integrand path), the callback accumulates block b's observables from block b's records, and
        // they join the block's partial row before the merge.  PCIe- and host-bound by construction.
        const int64_t n = nblocks * hm_n;
        const int nw = s.ni * s.ncomp, nc = s.ncomp;
        std::vector<double> obs((size_t)nblocks * s.nobs, 0.0);
        // (the self-check of a new several-lanes-per-chain code object, spec_self_check, never calls the USER's closure: its two small
        // launches compare histograms, normalisation, visits and acceptance tables; the observables stay zero in both)
        if (n > 0 && !p->in_self_check) {
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

