// mci_host_statistics.h -- part of the ONE translation unit mci_api.hip (included there, in order; not a stand-alone header):
// host-side statistics: pure functions (_standardize_block, burn-in lengths, _mean_std, average, doReweight!) and the block log.
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

