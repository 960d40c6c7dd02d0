// mci_host_problem.h -- part of the ONE translation unit mci_api.hip (included there, in order; not a stand-alone header):
// mci_problem_create / destroy (Configuration, src/configuration.jl:105-194), the integrand / measure setters, launch geometry.
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
        // (the kernels read a Discrete leaf's lower bound and a FermiK's kF / dk out of the source; a Continuous leaf's bounds live in its
        // edge table and stay OUT of the source, so that a sweep over a domain -- Continuous(0, beta) at many temperatures -- reuses one code object)
        s.leaf_lower.push_back(L.kind == MCI_CONTINUOUS ? 0.0 : L.lower);
        s.leaf_upper.push_back(L.kind == MCI_CONTINUOUS ? 0.0 : L.upper);
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
        int tdraw_maxbin = 2;
        if (s.ntile > 1)
            for (int k = 0; k < s.ndraw; ++k) {
                const Leaf &L = p->leaves[s.draw_leaf[k]];
                if (L.adapt && s.cover_mask[k] && s.leaf_tile[s.draw_leaf[k]] >= (s.split_all ? 0 : 1)) {
                    p->ntdraw += 1;
                    if (L.nbin > 65536) { delete p; return fail(MCI_ERR_INVALID, "leaf %d: more than 65536 bins with tiled histograms", s.draw_leaf[k]); }
                    if (s.leaf_nbin[s.draw_leaf[k]] > tdraw_maxbin) tdraw_maxbin = s.leaf_nbin[s.draw_leaf[k]];
                }
            }
        // words of packed bins a parked sample takes: mci_device.h tdraw_bits() / tdraw_words(), the same arithmetic (999-bin grids: 10 bits
        // per draw, 32 draws in 10 words -- the 16 bits per draw this used to allocate for were half as much again as the kernels touch)
        {
            int bits = 1;
            while ((1 << bits) < tdraw_maxbin) ++bits;
            p->tdraw_words = (p->ntdraw * bits + 31) / 32;
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
        tile_release(p); // (to the context: the next many-grid problem takes it)
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

