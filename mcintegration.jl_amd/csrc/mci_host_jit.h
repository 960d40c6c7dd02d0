// mci_host_jit.h -- part of the ONE translation unit mci_api.hip (included there, in order; not a stand-alone header):
// the kernel slots and their JIT, the speculation trees of the several-lanes-per-chain kernels, the per-problem setters.
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
        if (!p->threads_vegas && p->vegas_wide && (chosen.vgprs() > 128 || chosen.scratch() != 0)) {
            // (the 512-thread launch bound of a light integrand's plain layout was checked on the FIRST variant only: this one does not
            // fit it -- both variants run 256-thread workgroups from here on, which the first one's code object allows)
            p->vegas_wide = false;
            chosen.threads = p->threads;
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
    Candidate c, lane;
    c.src = mcijit::generate_source(p->shape, solver, mcijit::kUnitSpec);
    c.threads = 256; // (a launch of few chains runs one wave per SIMD: up to 512 registers per lane)
    // In the cache, with the marker of a passed self-check next to it: nothing else to do.  Otherwise the lane-per-chain unit the check
    // compares it with is compiled NEXT to it (hiprtc is re-entrant): the check costs the slower of the two compilations, not their sum.
    const bool cached = mcijit::compile(c.src, c.threads, c.code, c.log, c.cached, &c.path, mcijit::kHdrSpec, /*cache_only=*/true) == 0;
    const bool verified = cached && access((c.path + ".ok").c_str(), F_OK) == 0;
    std::thread side;
    if (!verified && !p->compiled[solver] && !p->ctx->offline && !(g_over.spec_self_check.on && g_over.spec_self_check.v == 0)) {
        lane.src = mcijit::generate_source(p->shape, solver, mcijit::kUnitSolver);
        lane.threads = p->threads;
        side = std::thread([&lane] { lane.rc = mcijit::compile(lane.src, lane.threads, lane.code, lane.log, lane.cached, &lane.path); });
    }
    if (!cached) {
        c.rc = mcijit::compile(c.src, c.threads, c.code, c.log, c.cached, &c.path, mcijit::kHdrSpec);
        if (c.rc == 2) {
            // (the unit is built with a backend switch, mci_jit.h: a compiler that does not know it any more gets the unit without it --
            // the self-check below is what stands between such an object and the user's histogram)
            std::string log2;
            Candidate d;
            d.src = c.src;
            d.threads = c.threads;
            d.rc = mcijit::compile(d.src, d.threads, d.code, log2, d.cached, &d.path, mcijit::kHdrSpec, false, /*no_exec_mask_flag=*/true);
            if (d.rc == 0) {
                d.log = c.log;
                c = std::move(d);
            }
        }
    }
    if (side.joinable()) side.join();
    if (c.rc) return fail(MCI_ERR_COMPILE, "integrand failed to compile for gfx950:\n%s", c.log.c_str());
    if (int rc = load_slot(p, slot, c, p->lds_bytes)) return rc;
    p->compiled[slot] = true;
    p->spec_need_check[solver - 1] = !verified;
    if (verified && p->spec_state[solver - 1] == 0) p->spec_state[solver - 1] = 1;
    return MCI_OK;
}

// A NEW several-lanes-per-chain code object proves itself before it is trusted.  Every user integrand is a new translation unit, and
// one of ~1000 campaign layouts came out of ROCm 7.2's compiler with the right chains and its histogram adds in the wrong bins
// (profiles/r05_fuzz.txt: right estimates, a map adapting to noise, no error).  Both chain kernels of a problem are product kernels and
// step the SAME chain (same (chain, step)-addressed uniforms; nchain = 1 is the reference's chain, vegas_mc/montecarlo.jl:198-211), so
// the first launch through a code object without a marker is preceded by <= 2 blocks x <= 512 steps through it and through the
// lane-per-chain kernel, and the two packed buffers are compared: statistics to 1e-9, histogram and propose / accept tables to 1e-8
// (relative to the larger entry, with the section's largest entry as the floor).  Agreement: a marker file next to the code object,
// never checked again.  Disagreement: one warning, status -1, and the problem keeps one lane per chain.  The launch that triggered
// the check then runs as if nothing had happened: everything a launch leaves behind on the host side is put back.
static int spec_self_check(mci_problem *p, int solver, int G, int64_t nevalperblock, int64_t block_lo, int64_t block_hi, int32_t iteration,
                           uint64_t seed, int64_t measurefreq, double thermal_ratio) {
    const int slot = solver == MCI_VEGASMC ? kSlotVegasmcSpec : kSlotMcmcSpec;
    const int64_t nb = block_hi - block_lo < 2 ? block_hi - block_lo : 2, npb = nevalperblock < 512 ? nevalperblock : 512;
    const int64_t mf = measurefreq * 4 <= npb ? measurefreq : 1;
    struct Saved {
        int spec_lanes, kernel_timing, chain_cur, chain_solver, chain_iteration, last_wg, last_threads, last_nblocks, last_spec_lanes, last_spec_maxacc, blk_carried;
        bool chain_valid, last_carried, hold_measured, time_this_launch;
        int64_t chain_lo, chain_hi, chain_nchain, chain_ntrain, last_samples, last_nchain, launches, blk_rows, blk_stride, blk_lo;
    } sv = {p->spec_lanes, p->kernel_timing, p->chain_cur, p->chain_solver, p->chain_iteration, p->last_wg, p->last_threads, p->last_nblocks, p->last_spec_lanes,
            p->last_spec_maxacc, p->blk_carried, p->chain_valid, p->last_carried, p->hold_measured, p->time_this_launch, p->chain_lo, p->chain_hi, p->chain_nchain,
            p->chain_ntrain, p->last_samples, p->last_nchain, p->launches, p->blk_rows, p->blk_stride, p->blk_lo};
    int rc = flush_merge(p);
    if (rc) return rc;
    int h_status[4] = {0, 0, 0, 0};
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    HIPCHK(hipMemcpy(h_status, p->d_status, sizeof(h_status), hipMemcpyDeviceToHost));
    std::vector<double> got[2];
    p->in_self_check = true;
    p->kernel_timing = 0;
    for (int pass = 0; pass < 2 && !rc; ++pass) {
        p->spec_lanes = pass == 0 ? G : 1;
        got[pass].assign((size_t)p->packed_n, 0.0);
        rc = mci_iteration_run(p, solver, npb, block_lo, block_lo + nb, iteration, seed, mf, 1, thermal_ratio);
        if (!rc) rc = mci_get_packed(p, got[pass].data(), p->packed_n); // (merges the launch and waits for it)
    }
    p->in_self_check = false;
    p->spec_lanes = sv.spec_lanes; p->kernel_timing = sv.kernel_timing; p->chain_cur = sv.chain_cur; p->chain_solver = sv.chain_solver;
    p->chain_iteration = sv.chain_iteration; p->last_wg = sv.last_wg; p->last_threads = sv.last_threads; p->last_nblocks = sv.last_nblocks;
    p->last_spec_lanes = sv.last_spec_lanes; p->last_spec_maxacc = sv.last_spec_maxacc; p->blk_carried = sv.blk_carried; p->chain_valid = sv.chain_valid;
    p->last_carried = sv.last_carried; p->hold_measured = sv.hold_measured; p->time_this_launch = sv.time_this_launch; p->chain_lo = sv.chain_lo;
    p->chain_hi = sv.chain_hi; p->chain_nchain = sv.chain_nchain; p->chain_ntrain = sv.chain_ntrain; p->last_samples = sv.last_samples;
    p->last_nchain = sv.last_nchain; p->launches = sv.launches; p->blk_rows = sv.blk_rows; p->blk_stride = sv.blk_stride; p->blk_lo = sv.blk_lo;
    if (p->d_status) (void)hipMemcpy(p->d_status, h_status, sizeof(h_status), hipMemcpyHostToDevice); // (what the two small launches flagged is theirs)
    if (rc) return rc;
    const size_t nstat = (size_t)(2 * p->shape.nobs + 2 + p->ni + 1);
    double worst = 0.0;
    long bad = 0, first_bad = -1;
    for (int sec = 0; sec < 2; ++sec) {
        const size_t lo = sec ? nstat : 0, hi = sec ? (size_t)p->packed_n : nstat;
        const double tol = sec ? 1e-8 : 1e-9;
        double top = 0.0;
        for (size_t i = lo; i < hi; ++i) top = std::fmax(top, std::fmax(std::fabs(got[0][i]), std::fabs(got[1][i])));
        for (size_t i = lo; i < hi; ++i) {
            const double a = got[0][i], b = got[1][i];
            const double d = std::fabs(a - b), lim = tol * (std::fmax(std::fabs(a), std::fabs(b)) + 1e-3 * top);
            const bool ok = (std::isfinite(a) && std::isfinite(b)) ? d <= lim : (std::isnan(a) == std::isnan(b) && (std::isnan(a) || a == b));
            if (!ok) {
                if (first_bad < 0) first_bad = (long)i;
                ++bad;
                if (top > 0.0 && d / top > worst) worst = d / top;
            }
        }
    }
    p->spec_need_check[solver - 1] = false;
    if (bad == 0) {
        p->spec_state[solver - 1] = 1;
        if (FILE *f = fopen((p->code_object[slot] + ".ok").c_str(), "w")) { // (the marker: this code object has reproduced the lane-per-chain kernel on a device)
            fprintf(f, "%s\n", mcijit::compiler_id().c_str());
            fclose(f);
        }
        return MCI_OK;
    }
    p->spec_state[solver - 1] = -1;
    fprintf(stderr, "mci: the several-lanes-per-chain kernel of this problem (%s, %s) does not reproduce its lane-per-chain kernel on a %lld-block, %lld-step "
                    "check: %ld of %lld packed entries differ (first at %ld, largest difference %.3g of its section's maximum).  A miscompiled code object -- "
                    "this problem keeps one lane per chain (same chains, slower for launches of few chains); mci_chain_speculation_status reports -1.\n",
            solver == MCI_VEGASMC ? ":vegasmc" : ":mcmc", p->code_object[slot].c_str(), (long long)nb, (long long)npb, bad, (long long)p->packed_n, first_bad, worst);
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

int mci_chain_speculation_status(const mci_problem *p, int32_t solver, int32_t *status) {
    if (!p || !status || (solver != MCI_VEGASMC && solver != MCI_MCMC)) return fail(MCI_ERR_INVALID, "solver MCI_VEGASMC or MCI_MCMC");
    *status = p->spec_state[solver - 1];
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

int mci_debug_split_chunks(const mci_problem *p, int64_t *chunks, int64_t *bytes) {
    if (!p) return fail(MCI_ERR_INVALID, "NULL argument");
    if (chunks) *chunks = p->last_split_chunks;
    if (bytes) *bytes = p->last_split_bytes;
    return MCI_OK;
}

int mci_debug_compiler_id(const char *set, char *out, int32_t n) {
    if (set) mcijit::compiler_id_override() = set; // ("" takes the override back)
    if (out && n > 0) snprintf(out, (size_t)n, "%s", mcijit::compiler_id().c_str());
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

