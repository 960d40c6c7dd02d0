// mci_host_access.h -- part of the ONE translation unit mci_api.hip (included there, in order; not a stand-alone header):
// state access (grids, distributions, reweight, packed buffer), MCISTATE files, the sample dump, kernel timings.
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
        if (!p->ev_valid[slot] || !p->clock_valid[slot] || !h[(size_t)2 * slot + 1]) continue; // (a slot whose launch did not stamp holds an older launch's words)
        mhz[k++] = (double)h[(size_t)2 * slot] / (double)h[(size_t)2 * slot + 1] * (double)khz * 1.0e-3;
    }
    *got = k;
    return MCI_OK;
}

