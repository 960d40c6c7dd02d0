// mci_spec.h -- the chain solvers with SEVERAL LANES PER CHAIN: a group of 2..64 lanes steps ONE Markov chain speculatively.
//
// The reference's chain is one sequential loop (vegas_mc/montecarlo.jl:184-232, mcmc/montecarlo.jl:134-172): propose from the current
// configuration, accept or reject, measure.  mci_device.h maps a chain to a LANE, which fills the chip only when statistics allow tens
// of thousands of chains; the reference's default call (16 blocks = 16 chains) or a sticky integrand under :mcmc (a few hundred long
// chains) leaves 99 % of it idle.  Here a chain owns a GROUP of G lanes, and a trip of the group advances it by several steps:
//
//   * the uniforms of step s of chain g are addressed by (g, s) (DESIGN.md "RNG streams"), so any lane can form the proposal of any
//     step once it knows the configuration that step starts from;
//   * the lanes of a group are the nodes of a SPECULATION TREE (SpecNode, built by the host): the root proposes step ne0 from the
//     trip's base configuration; a node's REJECT child proposes the next step from the same configuration, its ACCEPT child from the
//     node's own proposal.  A pure reject chain (the "linear" tree) advances 1/q steps per trip when a step changes the configuration
//     with probability q -- 64 in the sticky states that set a chain's holding times --, a complete binary tree log2(G + 1) steps
//     whatever the acceptance; the host picks the G most probable nodes for an assumed acceptance;
//   * every lane evaluates its node's proposal; a ballot of the accept tests selects the one root-to-leaf path the chain actually takes
//     (a node is on it iff all the ancestors it hangs below by an accept edge accepted and all those it hangs below by a reject edge
//     rejected: two mask compares per lane, no sequential walk); the lanes ON the path do the bookkeeping of their own step
//     (propose / accept counters, histogram, measurement) with the configuration their step ends in, and the deepest of them hands
//     its end configuration to the whole group as the next trip's base (ds_bpermute).
//
// It is the SAME chain -- same law, same uniforms, same arithmetic per step; only the order in which the steps' contributions are
// added differs -- so the oracle (one sequential chain) checks it at the tolerances of the lane-per-chain kernels
// (tests/test_hip_parity.py, tests/test_hip_spec.py).
#pragma once

namespace mci {

// (SpecNode is declared in mci_device.h next to BatchArgs: the host fills it)

__device__ __forceinline__ int lane_read(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
__device__ __forceinline__ double lane_read(double v, int src) {
    const u64 u = (u64)__double_as_longlong(v);
    const u32 lo = (u32)__builtin_amdgcn_ds_bpermute(src << 2, (int)(u32)u), hi = (u32)__builtin_amdgcn_ds_bpermute(src << 2, (int)(u32)(u >> 32));
    return __longlong_as_double((long long)(((u64)hi << 32) | (u64)lo));
}
__device__ __forceinline__ u64 lane_read(u64 v, int src) {
    const u32 lo = (u32)__builtin_amdgcn_ds_bpermute(src << 2, (int)(u32)v), hi = (u32)__builtin_amdgcn_ds_bpermute(src << 2, (int)(u32)(v >> 32));
    return ((u64)hi << 32) | (u64)lo;
}
template <class Cfg> __device__ __forceinline__ Chain<Cfg> lane_read(const Chain<Cfg> &c, int src) {
    Chain<Cfg> r;
    static_for<0, Cfg::NDRAW>([&](auto K) {
        constexpr int k = decltype(K)::value;
        r.x[k] = lane_read(c.x[k], src);
        r.prob[k] = lane_read(c.prob[k], src);
        r.bin[k] = lane_read(c.bin[k], src);
    });
    return r;
}
template <class Cfg> __device__ __forceinline__ void chain_select(Chain<Cfg> &dst, bool take, const Chain<Cfg> &src) {
    static_for<0, Cfg::NDRAW>([&](auto K) {
        constexpr int k = decltype(K)::value;
        dst.x[k] = take ? src.x[k] : dst.x[k];
        dst.prob[k] = take ? src.prob[k] : dst.prob[k];
        dst.bin[k] = take ? src.bin[k] : dst.bin[k];
    });
}

// what a group's lanes know about themselves and their group
struct SpecLane {
    int G, m, gbase;
    int maxacc;  // accept levels the wave goes through per trip: the most of its groups' trees
    u64 anydepth; // ... and the depths whose draw some lane of the wave needs (:vegasmc)
    u64 gmask;
    SpecNode nd;
    // the tree the group is on, and what its chain has done since the tree was last looked at
    int tree, trips, steps, accepts;
};
// the union of a 64-bit mask over the lanes of the wave (wave-uniform; once per change of tree)
__device__ __forceinline__ u64 spec_wave_or(u64 mine) {
    u64 m = 0ull;
    for (int bit = 0; bit < 64; ++bit)
        if (__ballot(((mine >> bit) & 1ull) != 0ull) != 0ull) m |= 1ull << bit;
    return m;
}
// the largest of a small number (< 64) over the lanes of the wave (wave-uniform): the most accept levels of the trees its groups are on
__device__ __forceinline__ int spec_wave_max(int mine) {
    int m = 0;
#pragma unroll
    for (int bit = 5; bit >= 0; --bit) {
        const int cand = m | (1 << bit);
        if (__ballot(mine >= cand) != 0ull) m = cand;
    }
    return m;
}
__device__ __forceinline__ SpecLane spec_lane(const BatchArgs &a) {
    SpecLane s;
    const int lane = (int)(threadIdx.x & 63u);
    s.G = a.spec_lanes;
    s.m = lane & (s.G - 1);
    s.gbase = lane & ~(s.G - 1);
    s.gmask = s.G >= 64 ? ~0ull : ((1ull << s.G) - 1ull);
    s.tree = a.spec_ntree > 1 ? a.spec_first : 0;
    s.nd = a.spec_tab[s.tree * s.G + s.m];
    s.maxacc = spec_wave_max(s.nd.levels & 0xff);
    s.anydepth = spec_wave_or(s.nd.anydepth);
    s.trips = s.steps = s.accepts = 0;
    return s;
}
// Every kSpecWindow trips a group looks at the fraction of its chain's steps that changed the configuration and moves to the tree built
// for the nearest acceptance (spec_accept[]).  Wave-uniform control flow; the chain does not depend on the tree.
constexpr int kSpecWindow = 8;
__device__ __forceinline__ void spec_adapt(const BatchArgs &a, SpecLane &s, int adv, u64 accm) {
    if (a.spec_ntree <= 1) return;
    s.trips += 1;
    s.steps += adv;
    s.accepts += __popcll(accm);
    const bool look = s.trips >= kSpecWindow;
    if (__ballot(look) == 0ull) return;
    int pick = s.tree;
    if (look && s.steps > 0) {
        const float q = (float)s.accepts / (float)s.steps;
        float best = 2.0f;
        for (int k = 0; k < a.spec_ntree; ++k) {
            const float d = fabsf(q - a.spec_accept[k]);
            if (d < best) { best = d; pick = k; }
        }
    }
    if (look) s.trips = s.steps = s.accepts = 0;
    const bool move = pick != s.tree;
    if (__ballot(move) == 0ull) return;
    if (move) {
        s.tree = pick;
        s.nd = a.spec_tab[pick * s.G + s.m];
    }
    s.maxacc = spec_wave_max(s.nd.levels & 0xff);
    s.anydepth = spec_wave_or(s.nd.anydepth);
}
// the chain's path through the tree from the lanes' accept tests: is this lane on it, and which is the deepest lane that is
// (the tree's lanes are numbered ancestors-first, so that is the highest one)
struct SpecPath {
    u64 okm, pathm; // group-relative lane masks: accept tests that came out true | lanes on the path
    bool onpath;
    int last;       // group-relative; 0 when the group has nothing left to do
};
__device__ __forceinline__ SpecPath spec_path(const SpecLane &s, bool valid, bool ok) {
    SpecPath p;
    p.okm = (__ballot(ok) >> s.gbase) & s.gmask;
    p.onpath = valid && (p.okm & s.nd.needacc) == s.nd.needacc && (p.okm & s.nd.needrej) == 0ull;
    p.pathm = (__ballot(p.onpath) >> s.gbase) & s.gmask;
    p.last = p.pathm ? 63 - __clzll((long long)p.pathm) : 0;
    return p;
}

// =============================================================================================
// VegasMC, G lanes per chain  (vegas_mc/montecarlo.jl:112-241, vegas_mc/updates.jl:45-106; lane-per-chain form: mci_device.h
// vegasmc_chains -- same streams, same arithmetic per step)
// =============================================================================================
template <class Cfg> __device__ __forceinline__ void vegasmc_chains_spec(const BatchArgs &a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NI = Cfg::NI, NORMI = Cfg::NI;
    const int tid = threadIdx.x, T = blockDim.x;
    double *sE = smem + Lds<Cfg>::E, *sDA = smem + Lds<Cfg>::DA, *sDD = smem + Lds<Cfg>::DD;
    double *sH = smem + Lds<Cfg>::H, *sO = smem + Lds<Cfg>::O;
    stage_tables<Cfg>(a.edges, a.dacc, a.ddist, sE, sDA, sDD);
    if constexpr (Mode<Cfg>::HIST_LDS)
        for (int i = tid; i < Cfg::HTILE * Cfg::HCOPY; i += T) sH[i] = 0.0;
    for (int i = tid; i < Cfg::NOBS * ocopy<Cfg>(); i += T) sO[i] = 0.0;
    u64 *sPA = reinterpret_cast<u64 *>(smem + Lds<Cfg>::PA);
    for (int i = tid; i < 2 * PaTable<Cfg>::N; i += T) sPA[i] = 0ull;
    __syncthreads();
    Tables<Cfg> t;
    if constexpr (Mode<Cfg>::EDGE_LDS) t.E = sE;
    else t.E = a.edges;
    t.DA = sDA;
    t.DD = sDD;

    const WorkItem wi = work_item<Cfg>(a);
    const int slice = wi.slice, tile = wi.tile;
    const i64 B = a.block_lo + wi.lb;
    const i64 steps = a.neval_per_block / a.nchain;
    const u32 bs = (u32)B << 20;
    const u32 st_init = a.iteration * 8u + STREAM_MC_INIT + bs, st_step = a.iteration * 8u + STREAM_MC_STEP + bs;
    const u32 k0 = (u32)a.seed, k1 = (u32)(a.seed >> 32);
    double rw[NI + 1];
    static_for<0, NI + 1>([&](auto I) { rw[decltype(I)::value] = a.reweight[decltype(I)::value]; });
    SpecLane sp = spec_lane(a);

    double acc[Cfg::NW];
    static_for<0, Cfg::NW>([&](auto I) { acc[decltype(I)::value] = 0.0; });
    double extra[Cfg::NCOLS - Cfg::NOBS];
    static_for<0, Cfg::NCOLS - Cfg::NOBS>([&](auto I) { extra[decltype(I)::value] = 0.0; });
    constexpr int XN = Cols<Cfg>::NORM - Cfg::NOBS, XE = Cols<Cfg>::NEVAL - Cfg::NOBS, XV = Cols<Cfg>::VISITED - Cfg::NOBS;
    u32 npr[Cfg::NPOOL], nac[Cfg::NPOOL]; // propose[2, 1, vi], accept[2, 1, vi] of the steps this lane has committed (vegas_mc/updates.jl:90-92)
    static_for<0, Cfg::NPOOL>([&](auto V) { npr[decltype(V)::value] = 0u; nac[decltype(V)::value] = 0u; });
    auto flush_pa = [&]() {
        static_for<0, Cfg::NPOOL>([&](auto V) {
            constexpr int v = decltype(V)::value;
            if (npr[v]) lds_count(&sPA[PaTable<Cfg>::idx(1, 0, v)], (u64)npr[v]);
            if (nac[v]) lds_count(&sPA[PaTable<Cfg>::N + PaTable<Cfg>::idx(1, 0, v)], (u64)nac[v]);
            npr[v] = 0u;
            nac[v] = 0u;
        });
    };
    constexpr int MAXNL = [] { int mx = 1; for (int v = 0; v < Cfg::NPOOL; ++v) mx = Cfg::pool_nleaf(v) > mx ? Cfg::pool_nleaf(v) : mx; return mx; }();

    // the block's chains are dealt to its groups: group (slice * T + tid) / G takes chains first, first + cpp, ...
    const i64 cpp = (i64)a.wg_per_block * T / sp.G, first = ((i64)slice * T + tid) / sp.G;
    const i64 npass = (a.nchain + cpp - 1) / cpp;
    for (i64 pass = 0; pass < npass; ++pass) {
        const i64 ch = first + pass * cpp;
        const bool live = ch < a.nchain; // (a group without a chain in the last pass goes through the motions on chain 0 with zero steps)
        const u64 g = live ? (u64)ch : 0ull;
        Chain<Cfg> c; // the trip's base configuration: the same in every lane of the group
        if (a.carry_x) load_carried<Cfg>(a, t, wi.lb, carried_from(a, wi.lb, (i64)g), c);
        else {   // initialize!  (montecarlo.jl:151-153): create! on every live slot
            Sample<Cfg> s;
            draw_sample<Cfg>(t, a.seed, st_init, g, s);
            static_for<0, Cfg::NDRAW>([&](auto K) {
                constexpr int k = decltype(K)::value;
                c.x[k] = s.x[k];
                c.bin[k] = s.bin[k];
                c.prob[k] = 1.0 / s.pj[k]; // sampler.jl:303 / :20
            });
        }
        double w[Cfg::NW], pad[NI + 1];
        Cfg::integrand(c.x, w, a.ud, -1); // :155-159
        static_for<0, NI + 1>([&](auto I) { pad[decltype(I)::value] = pad_prob<Cfg, decltype(I)::value>(c); }); // :161
        double probability = rw[NORMI] * pad[NORMI]; // :162
        static_for<0, NI>([&](auto I) { constexpr int i = decltype(I)::value; probability += absw<Cfg, i>(w) * rw[i] * pad[i]; }); // :163-166

        const i64 total = live ? steps : 0;
        i64 ne0 = 1; // first step of the trip (montecarlo.jl:184 counts from 1)
        u32 trips = 0u;
        while (__ballot(ne0 <= total) != 0ull) {
            const i64 ne = ne0 + sp.nd.depth; // the step whose proposal this lane evaluates
            const bool valid = ne <= total;
            const u64 sidx = (g << 32) | (u64)(ne - 1);
            const u32x4 r0 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 0u, st_step, k0, k1);
            const u32x4 r1 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 1u, st_step, k0, k1);
            // ---- changeVariable  updates.jl:45-106: what the step draws does not depend on the configuration it starts from ----
            double upool = u01(r0.x, r0.y); // :50
            if (Cfg::NPOOL > 1 && a.nchain > 1) { // (the pool-pick sequence chains (ch & ~63) .. (ch | 63) of a block share: vegasmc_chains)
                const u64 gidx = ((g & ~63ull) << 32) | (u64)(ne - 1);
                const u32x4 rg = philox4x32_10((u32)gidx, (u32)(gidx >> 32), 0u, a.iteration * 8u + STREAM_MC_GROUP + bs, k0, k1);
                upool = u01(rg.x, rg.y);
            }
            int vi = (int)(upool * (double)Cfg::NPOOL);
            if (vi >= Cfg::NPOOL) vi = Cfg::NPOOL - 1;
            const double uslot = u01(r0.z, r0.w);
            const double uacc = u01(r1.x, r1.y);
            double dxn[MAXNL], dpn[MAXNL];
            int dbn[MAXNL], slot = 0;
            bool active = false;
            static_for<0, MAXNL>([&](auto J) { dxn[decltype(J)::value] = 0.0; dpn[decltype(J)::value] = 1.0; dbn[decltype(J)::value] = 0; });
            static_for<0, Cfg::NPOOL>([&](auto V) {
                constexpr int v = decltype(V)::value;
                constexpr int md = Cfg::pool_maxdof(v), nl = Cfg::pool_nleaf(v), k00 = Cfg::pool_first_draw(v);
                constexpr bool skip = (md <= 0) || (nl == 1 && Cfg::leaf_kind(Cfg::draw_leaf(md > 0 ? k00 : 0)) == 1 &&
                                                    Cfg::leaf_nbin(Cfg::draw_leaf(md > 0 ? k00 : 0)) == 1); // :52-57
                if constexpr (!skip) {
                    if (vi == v) {
                        active = true;
                        slot = (int)(uslot * (double)md); // :58
                        if (slot >= md) slot = md - 1;
                        static_for<0, nl>([&](auto Lf) {
                            constexpr int l = decltype(Lf)::value;
                            constexpr int kk = 3 + l; // RNG draw index within the step
                            double y;
                            if constexpr (kk == 3) y = u01(r1.z, r1.w);
                            else {
                                const u32x4 rr = philox4x32_10((u32)sidx, (u32)(sidx >> 32), (u32)(kk >> 1), st_step, k0, k1);
                                y = (kk & 1) ? u01(rr.z, rr.w) : u01(rr.x, rr.y);
                            }
                            draw_pool_leaf<Cfg, v, l>(t, y, dxn[l], dpn[l], dbn[l]); // shift!  sampler.jl:336-386, :57-71
                        });
                    }
                }
            });
            // ---- the configuration the lane's step starts from (cp) and its proposal (n).  What a step draws does not depend on where it
            // starts, so cp is the trip's base with the draws of the ancestors the way leaves by an accept edge applied in step order.
            // Two ways to get them there (the same cp either way; wave-uniform choice by the shape of the trees the wave's groups are on):
            //   depth by depth -- every lane reads the draw of ONE lane of that depth (they all hold the same) and applies it if its way
            //     accepted there: the exchanges do not depend on each other, one burst of ds_bpermute however many accept levels the tree
            //     has (trees built for an acceptance >= 1/2: as many depths with accept edges as levels);
            //   level by level -- lanes behind an accept edge read the whole configuration their ancestor proposed, one level after the
            //     ancestor built its own (trees with one or two accept levels hanging off a long reject chain: many depths, few levels)
            Chain<Cfg> cp = c;
            if (2 * __popcll(sp.anydepth) <= 3 * sp.maxacc) {
                for (u64 todo = sp.anydepth; todo != 0ull; todo &= todo - 1ull) {
                    const int dd = __builtin_ctzll(todo);
                    const u64 at = (__ballot(sp.nd.depth == dd) >> sp.gbase) & sp.gmask; // lanes of the group that propose step ne0 + dd
                    const int src = sp.gbase + (at ? __builtin_ctzll(at) : sp.m);
                    const int pvi = lane_read(active ? vi : -1, src), pslot = lane_read(slot, src);
                    const bool take = ((sp.nd.accdepth >> dd) & 1ull) != 0ull;
                    static_for<0, MAXNL>([&](auto Lf) {
                        constexpr int l = decltype(Lf)::value;
                        const double px = lane_read(dxn[l], src), pp = lane_read(dpn[l], src);
                        const int pb = lane_read(dbn[l], src);
                        static_for<0, Cfg::NPOOL>([&](auto V) {
                            constexpr int v = decltype(V)::value;
                            if constexpr (Cfg::pool_maxdof(v) > 0 && l < Cfg::pool_nleaf(v)) {
                                if (take && pvi == v) put_slot<Cfg, v, l>(cp, pslot, px, pp, pb);
                            }
                        });
                    });
                }
            } else {
                Chain<Cfg> built = c; // a lane's own proposal once its level has been through (what the lanes behind its accept edge read)
                for (int lvl = 0; lvl <= sp.maxacc; ++lvl) {
                    if (lvl > 0) { // (every lane takes part in the exchange; the lanes of this level keep what they read)
                        const Chain<Cfg> f = lane_read<Cfg>(built, sp.gbase + (sp.nd.anc >= 0 ? sp.nd.anc : sp.m));
                        chain_select<Cfg>(cp, sp.nd.nacc == lvl, f);
                    }
                    if (lvl < sp.maxacc && sp.nd.nacc == lvl) {
                        built = cp;
                        static_for<0, Cfg::NPOOL>([&](auto V) {
                            constexpr int v = decltype(V)::value;
                            if constexpr (Cfg::pool_maxdof(v) > 0) {
                                if (active && vi == v)
                                    static_for<0, Cfg::pool_nleaf(v)>([&](auto Lf) {
                                        constexpr int l = decltype(Lf)::value;
                                        put_slot<Cfg, v, l>(built, slot, dxn[l], dpn[l], dbn[l]);
                                    });
                            }
                        });
                    }
                }
            }
            Chain<Cfg> n = cp;
            double prop = 1.0;
            static_for<0, Cfg::NPOOL>([&](auto V) {
                constexpr int v = decltype(V)::value;
                constexpr int md = Cfg::pool_maxdof(v), nl = Cfg::pool_nleaf(v), k00 = Cfg::pool_first_draw(v);
                constexpr bool skip = (md <= 0) || (nl == 1 && Cfg::leaf_kind(Cfg::draw_leaf(md > 0 ? k00 : 0)) == 1 &&
                                                    Cfg::leaf_nbin(Cfg::draw_leaf(md > 0 ? k00 : 0)) == 1);
                if constexpr (!skip) {
                    if (vi == v) {
                        static_for<0, nl>([&](auto Lf) {
                            constexpr int l = decltype(Lf)::value;
                            double xo, po;
                            int bo;
                            get_slot<Cfg, v, l>(cp, slot, xo, po, bo);
                            put_slot<Cfg, v, l>(n, slot, dxn[l], dpn[l], dbn[l]);
                            prop *= po / dpn[l]; // 1/prob_ratio  sampler.jl:385, :70
                        });
                    }
                }
            });
            const bool go = valid && active && prop > 4.9406564584124654e-324; // :63-65
            double wn[Cfg::NW], padn[NI + 1], newp = 0.0;
            static_for<0, Cfg::NW>([&](auto I) { wn[decltype(I)::value] = 0.0; });
            static_for<0, NI + 1>([&](auto I) { padn[decltype(I)::value] = 0.0; });
            if (go) {
                Cfg::integrand(n.x, wn, a.ud, -1);             // :67-75
                static_for<0, NI + 1>([&](auto I) { padn[decltype(I)::value] = pad_prob<Cfg, decltype(I)::value>(n); }); // :79-81
                newp = rw[NORMI] * padn[NORMI];                // :84
                static_for<0, NI>([&](auto I) { constexpr int i = decltype(I)::value; newp += absw<Cfg, i>(wn) * rw[i] * padn[i]; }); // :85-87
            }
            // weights, paddings and probability of the configuration the step starts from: the base's, or what the ancestor evaluated
            double wp[Cfg::NW], padp[NI + 1], Pp = probability;
            static_for<0, Cfg::NW>([&](auto I) { wp[decltype(I)::value] = w[decltype(I)::value]; });
            static_for<0, NI + 1>([&](auto I) { padp[decltype(I)::value] = pad[decltype(I)::value]; });
            if (sp.maxacc > 0) {
                const int src = sp.gbase + (sp.nd.anc >= 0 ? sp.nd.anc : sp.m);
                const bool behind = sp.nd.anc >= 0;
                const double fp = lane_read(newp, src);
                Pp = behind ? fp : Pp;
                static_for<0, Cfg::NW>([&](auto I) { const double f = lane_read(wn[decltype(I)::value], src); wp[decltype(I)::value] = behind ? f : wp[decltype(I)::value]; });
                static_for<0, NI + 1>([&](auto I) { const double f = lane_read(padn[decltype(I)::value], src); padp[decltype(I)::value] = behind ? f : padp[decltype(I)::value]; });
            }
            const double R = prop * newp / Pp;                 // :88
            const bool ok = go && uacc < R;                    // :91
            const SpecPath path = spec_path(sp, valid, ok);
            // ---- the configuration the lane's step ENDS in: its proposal if accepted (:93-100), else the one it started from (:102) ----
            chain_select<Cfg>(cp, ok, n);
            static_for<0, Cfg::NW>([&](auto I) { wp[decltype(I)::value] = ok ? wn[decltype(I)::value] : wp[decltype(I)::value]; });
            static_for<0, NI + 1>([&](auto I) { padp[decltype(I)::value] = ok ? padn[decltype(I)::value] : padp[decltype(I)::value]; });
            Pp = ok ? newp : Pp;
            if (path.onpath) { // this step is one the chain takes: its bookkeeping
                if (go) {
                    extra[XE] += 1.0;                          // config.neval += 1   :77
                    static_for<0, Cfg::NPOOL>([&](auto V) {
                        constexpr int v = decltype(V)::value;
                        if (vi == v) {
                            npr[v] += 1u;                      // :90
                            nac[v] += ok ? 1u : 0u;            // :92
                        }
                    });
                }
                // ---- histogram  montecarlo.jl:198-211 ----
                {
                    double wh[NI];
                    static_for<0, NI>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        const double aw = absw<Cfg, i>(wp);
                        const double f2 = aw * aw / own_prob<Cfg, i>(cp);                  // :203
                        wh[i] = f2 * padp[i] / Pp;                                         // :204
                    });
                    Sample<Cfg> sb;
                    static_for<0, Cfg::NDRAW>([&](auto K) { sb.bin[decltype(K)::value] = cp.bin[decltype(K)::value]; });
                    hist_update<Cfg>(sb, wh, sH, a.ghist, tile);
                }
                // ---- measurement  montecarlo.jl:213-232 ----
                bool mf = true;
                i64 mj = ne;
                if (a.measurefreq != 1) {
                    mj = ne / a.measurefreq;
                    mf = mj * a.measurefreq == ne;
                }
                if (mf && (double)ne >= a.burnin) { // :213
                    double relw[Cfg::NW];
                    static_for<0, NI>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        extra[XV + i] += absw<Cfg, i>(wp) * fabs(padp[i] * rw[i]) / Pp; // :216
                        static_for<0, Cfg::NCOMP>([&](auto Q) {
                            constexpr int q = i * Cfg::NCOMP + decltype(Q)::value;
                            relw[q] = wp[q] * padp[i] / Pp;                                // :218/:220
                        });
                    });
                    if constexpr (Cfg::HOST_MEASURE != 0) { // :224-227 on the host, after the launch
                        if (tile == 0) host_measure_record<Cfg, Cfg::NW>(a, wi.lb, (i64)g, mj, cp.x, relw, -1);
                    } else
                    measure<Cfg>(cp.x, cp.bin, relw, a.ud, acc, obs_wave<Cfg>(sO));
                    extra[XN] += padp[NORMI] / Pp;                    // :229
                    extra[XV + NORMI] += rw[NORMI] * padp[NORMI] / Pp; // :230
                }
            }
            // ---- the deepest lane on the path hands its end configuration to the group ----
            {
                const int src = sp.gbase + path.last;
                const int adv = path.pathm ? lane_read(sp.nd.depth, src) + 1 : 0;
                if (__ballot((path.okm & path.pathm) != 0ull) != 0ull) { // (a trip of rejections only leaves every group's base as it is)
                    c = lane_read<Cfg>(cp, src);
                    static_for<0, Cfg::NW>([&](auto I) { w[decltype(I)::value] = lane_read(wp[decltype(I)::value], src); });
                    static_for<0, NI + 1>([&](auto I) { pad[decltype(I)::value] = lane_read(padp[decltype(I)::value], src); });
                    probability = lane_read(Pp, src);
                }
                ne0 += adv;
                spec_adapt(a, sp, adv, path.okm & path.pathm);
            }
            if ((++trips & 0x3FFFFFFFu) == 0u) flush_pa(); // (32-bit lane counters: hand over long before they wrap)
        }
        flush_pa();
        if (a.store_x && tile == 0 && live && sp.m == 0) {
            store_carried<Cfg>(a, wi.lb, ch, c);
            if (a.store_P) a.store_P[wi.lb * a.nchain + ch] = probability; // (vegasmc_carry_weights)
        }
    }
    __syncthreads();
    flush_workgroup<Cfg, Lds<Cfg>, true, true>(a, smem, acc, extra, wi.rowid, tile);
}

// =============================================================================================
// MCMC, G lanes per chain  (mcmc/montecarlo.jl:72-184, mcmc/updates.jl:1-147; lane-per-chain form: mci_device.h mcmc_chains -- same
// streams, same arithmetic per step).  A proposal here depends on the configuration it starts from (the integrand index picks the
// neighbor, the slot counts, which uniform a shifted slot consumes), so lanes behind an accept edge build theirs one accept level
// after their ancestor built its own (mcmc_propose runs once per level of the tree, the integrand once per trip).
// =============================================================================================
template <class Cfg> __device__ __forceinline__ void mcmc_chains_spec(const BatchArgs &a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NI = Cfg::NI, NORMI = Cfg::NI, ND = Cfg::NI + 1, NPOOL = Cfg::NPOOL;
    constexpr int NUPD = 2 * NPOOL + 2; // [changeIntegrand, swapVariable, changeVariable x 2*Nv]  montecarlo.jl:127-130
    const int tid = threadIdx.x, T = blockDim.x;
    double *sE = smem + Lds<Cfg>::E, *sDA = smem + Lds<Cfg>::DA, *sDD = smem + Lds<Cfg>::DD;
    double *sH = smem + Lds<Cfg>::H, *sO = smem + Lds<Cfg>::O;
    stage_tables<Cfg>(a.edges, a.dacc, a.ddist, sE, sDA, sDD);
    if constexpr (Mode<Cfg>::HIST_LDS)
        for (int i = tid; i < Cfg::HTILE * Cfg::HCOPY; i += T) sH[i] = 0.0;
    for (int i = tid; i < Cfg::NOBS * ocopy<Cfg>(); i += T) sO[i] = 0.0;
    u64 *sPA = reinterpret_cast<u64 *>(smem + Lds<Cfg>::PA);
    for (int i = tid; i < 2 * PaTable<Cfg>::N; i += T) sPA[i] = 0ull;
    __syncthreads();
    Tables<Cfg> t;
    if constexpr (Mode<Cfg>::EDGE_LDS) t.E = sE;
    else t.E = a.edges;
    t.DA = sDA;
    t.DD = sDD;

    const WorkItem wi = work_item<Cfg>(a);
    const int slice = wi.slice, tile = wi.tile;
    const i64 B = a.block_lo + wi.lb;
    const i64 steps = a.neval_per_block / a.nchain, nburn = a.nburn;
    const u32 bs = (u32)B << 20;
    const u32 st_init = a.iteration * 8u + STREAM_MCMC_INIT + bs, st_step = a.iteration * 8u + STREAM_MCMC_STEP + bs;
    const u32 k0 = (u32)a.seed, k1 = (u32)(a.seed >> 32);
    double rw[ND];
    static_for<0, ND>([&](auto I) { rw[decltype(I)::value] = a.reweight[decltype(I)::value]; });
    auto rw_sel = [&](int i) {
        double r = rw[NORMI];
        static_for<0, NI>([&](auto I) { if (i == decltype(I)::value) r = rw[decltype(I)::value]; });
        return r;
    };
    SpecLane sp = spec_lane(a);

    double acc[Cfg::NW];
    static_for<0, Cfg::NW>([&](auto I) { acc[decltype(I)::value] = 0.0; });
    double extra[Cfg::NCOLS - Cfg::NOBS];
    static_for<0, Cfg::NCOLS - Cfg::NOBS>([&](auto I) { extra[decltype(I)::value] = 0.0; });
    constexpr int XN = Cols<Cfg>::NORM - Cfg::NOBS, XE = Cols<Cfg>::NEVAL - Cfg::NOBS, XV = Cols<Cfg>::VISITED - Cfg::NOBS;

    const i64 cpp = (i64)a.wg_per_block * T / sp.G, first = ((i64)slice * T + tid) / sp.G;
    const i64 npass = (a.nchain + cpp - 1) / cpp;
    for (i64 pass = 0; pass < npass; ++pass) {
        const i64 ch = first + pass * cpp;
        const bool live = ch < a.nchain;
        const u64 g = live ? (u64)ch : 0ull;
        int curr = a.nchain == 1 ? 0 : (int)(g % (u64)ND); // montecarlo.jl:76 idx = 1; many chains start stratified
        Chain<Cfg> c;
        Weight<Cfg> weight; // :116 _State(curr, zero(T), 1.0)
        static_for<0, Cfg::NCOMP>([&](auto Q) { weight.v[decltype(Q)::value] = 0.0; });
        weight.abs = 0.0;
        double probability = 1.0;
        bool fresh = a.carry_x == nullptr;
        if (!fresh) { // continues the previous iteration's chain (BatchArgs::carry_x)
            const i64 from = carried_from(a, wi.lb, (i64)g);
            load_carried<Cfg>(a, t, wi.lb, from, c);
            curr = a.carry_curr[wi.lb * a.carry_nchain + from];
            if (curr != NORMI) {
                weight = eval_sel<Cfg>(curr, c.x, a.ud);        // :197 on the carried configuration
                probability = weight.abs * rw_sel(curr);        // :199
                if (!(probability > 4.940656458412465e-274)) {
                    fresh = true;
                    curr = (int)(g % (u64)ND);
                }
            } else probability = rw[NORMI];                     // :201-202
        }
        for (int tr = 0; fresh && tr < 10000; ++tr) {    // :118-124 (every lane of the group draws the same start)
            Sample<Cfg> s;
            draw_sample<Cfg>(t, a.seed, st_init, g * 16384ull + (u64)tr, s); // initialize!  :190-193
            static_for<0, Cfg::NDRAW>([&](auto K) {
                constexpr int k = decltype(K)::value;
                c.x[k] = s.x[k];
                c.bin[k] = s.bin[k];
                c.prob[k] = 1.0 / s.pj[k];
            });
            static_for<0, NPOOL>([&](auto V) { // FermiK slots are created jointly from their D uniforms (same stream, k = flat draw)
                constexpr int v = decltype(V)::value;
                if constexpr (pool_is_fermik<Cfg>(v)) {
                    constexpr int D = Cfg::pool_nleaf(v), k00 = Cfg::pool_first_draw(v);
                    constexpr double kF = Cfg::leaf_lower(Cfg::draw_leaf(k00));
                    static_for<0, Cfg::pool_maxdof(v)>([&](auto S) {
                        constexpr int kb = k00 + decltype(S)::value * D;
                        double u[D], kk[D];
                        const u64 iidx = g * 16384ull + (u64)tr;
                        static_for<0, D>([&](auto J) {
                            constexpr int kq = kb + decltype(J)::value;
                            const u32x4 rr = philox4x32_10((u32)iidx, (u32)(iidx >> 32), (u32)(kq >> 1), st_init, k0, k1);
                            u[decltype(J)::value] = (kq & 1) ? u01(rr.z, rr.w) : u01(rr.x, rr.y);
                            kk[decltype(J)::value] = kF / sqrt((double)D); // variable.jl:13: the pool's initial content
                        });
                        (void)fermik_create<Cfg, v>(u, kk);
                        static_for<0, D>([&](auto J) { c.x[kb + decltype(J)::value] = kk[decltype(J)::value]; });
                    });
                }
            });
            if (curr != NORMI) {
                weight = eval_sel<Cfg>(curr, c.x, a.ud);        // :197
                probability = weight.abs * rw_sel(curr);        // :199
            } else {
                static_for<0, Cfg::NCOMP>([&](auto Q) { weight.v[decltype(Q)::value] = 0.0; });
                weight.abs = 0.0;
                probability = rw[NORMI];                        // :201-202
            }
            if (curr == NORMI || probability > 4.940656458412465e-274) break; // :120-122 (TINY)
        }
        if (live && sp.m == 0 && curr != NORMI && probability == 0.0) atomicOr(a.status, ST_MCMC_INIT); // :125-126 error(...)

        // holding times (mcmc_chains): the group's lanes keep identical records
        int last[Cfg::NDRAW > 0 ? Cfg::NDRAW : 1], lastc = 0, hmax = 0;
        static_for<0, Cfg::NDRAW>([&](auto K) { last[decltype(K)::value] = 0; });
        const i64 total = live ? steps + nburn : 0;
        i64 it0 = 1; // first step of the trip (:134)
        while (__ballot(it0 <= total) != 0ull) {
            const i64 it = it0 + sp.nd.depth; // the step whose proposal this lane evaluates
            const bool valid = it <= total;
            const u64 sidx = (g << 32) | (u64)(it - 1);
            const u32x4 r0 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 0u, st_step, k0, k1);
            const u32x4 r1 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 1u, st_step, k0, k1);
            const u32x4 r2 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 2u, st_step, k0, k1);
            double uupd = u01(r0.x, r0.y); // :137 rand(rng, updates)
            if (a.nchain > 1) { // (the update-type sequence chains (ch & ~63) .. (ch | 63) of a block share: mcmc_chains)
                const u64 gidx = ((g & ~63ull) << 32) | (u64)(it - 1);
                const u32x4 rg = philox4x32_10((u32)gidx, (u32)(gidx >> 32), 0u, a.iteration * 8u + STREAM_MCMC_GROUP + bs, k0, k1);
                uupd = u01(rg.x, rg.y);
            }
            int upd = (int)(uupd * (double)NUPD);
            if (upd >= NUPD) upd = NUPD - 1;
            const double upick = u01(r0.z, r0.w), us1 = u01(r1.x, r1.y), us2 = u01(r1.z, r1.w), uacc = u01(r2.x, r2.y);
            // ---- the configuration the lane's step starts from (cp, currp) and its proposal, accept level by accept level ----
            Chain<Cfg> cp = c;
            int currp = curr;
            McmcProposal<Cfg> pr;
            pr.n = c;
            pr.prop = 1.0;
            pr.active = false;
            pr.newcurr = curr;
            pr.ut = 0;
            pr.pvi = 0;
            pr.touched = 0ull;
            for (int lvl = 0; lvl <= sp.maxacc; ++lvl) {
                if (lvl > 0) { // (every lane takes part in the exchange; the lanes of this level keep what they read)
                    const int src = sp.gbase + (sp.nd.anc >= 0 ? sp.nd.anc : sp.m);
                    const Chain<Cfg> f = lane_read<Cfg>(pr.n, src);
                    const int fc = lane_read(pr.newcurr, src);
                    chain_select<Cfg>(cp, sp.nd.nacc == lvl, f);
                    currp = sp.nd.nacc == lvl ? fc : currp;
                }
                if (sp.nd.nacc == lvl) pr = mcmc_propose<Cfg>(t, cp, currp, upd, upick, us1, us2, sidx, st_step, k0, k1, r2);
            }
            const double prop = pr.prop;
            const int newcurr = pr.newcurr, ut = pr.ut, pvi = pr.pvi;
            const u64 touched = pr.touched;
            const bool go = valid && pr.active && prop > 4.9406564584124654e-324; // updates.jl:29-31, :88-90, :129-131
            Weight<Cfg> wn;
            static_for<0, Cfg::NCOMP>([&](auto Q) { wn.v[decltype(Q)::value] = 0.0; });
            wn.abs = 0.0;
            double newp = 0.0;
            if (go) {
                if (newcurr != NORMI) wn = eval_sel<Cfg>(newcurr, pr.n.x, a.ud);               // :35-38, :92, :133
                newp = newcurr == NORMI ? rw[NORMI] : wn.abs * rw_sel(newcurr);                 // :42-44, :96, :137
            }
            // weight and probability of the configuration the step starts from: the base's, or what the ancestor evaluated
            Weight<Cfg> wp = weight;
            double Pp = probability;
            if (sp.maxacc > 0) {
                const int src = sp.gbase + (sp.nd.anc >= 0 ? sp.nd.anc : sp.m);
                const bool behind = sp.nd.anc >= 0;
                const double fp = lane_read(newp, src), fa = lane_read(wn.abs, src);
                Pp = behind ? fp : Pp;
                wp.abs = behind ? fa : wp.abs;
                static_for<0, Cfg::NCOMP>([&](auto Q) { const double f = lane_read(wn.v[decltype(Q)::value], src); wp.v[decltype(Q)::value] = behind ? f : wp.v[decltype(Q)::value]; });
            }
            const double R = prop * newp / Pp;                                                  // :46, :97, :138
            const bool ok = go && uacc < R;                                                     // :49, :100, :141
            const SpecPath path = spec_path(sp, valid, ok);
            if (path.onpath) {
                static_for<0, ND>([&](auto I) { extra[XV + decltype(I)::value] += currp == decltype(I)::value ? 1.0 : 0.0; }); // :136
                if (go) {
                    extra[XE] += 1.0;                                                           // :40, :94, :135
                    // propose[1, curr, new] :48,:50 | propose[2, curr, vi] :99,:101 | propose[3, curr, vi] :140,:142
                    pa_count<Cfg, false>(sPA, PaTable<Cfg>::idx(ut, currp, ut == 0 ? newcurr : pvi), true, ok);
                }
            }
            if (a.hold_hist) {
                // the accepted steps on the path, in chain order: every lane of the group applies them to its copy of the records
                u64 mo = 0ull, mn = 0ull; // live draws of the old and of the proposed integrand
                static_for<0, NI>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    mo = currp == i ? Cfg::own_mask(i) : mo;
                    mn = newcurr == i ? Cfg::own_mask(i) : mn;
                });
                const u64 mychg = ut == 0 ? (mn & ~mo) : touched;
                const int myflags = (ut != 0 ? 1 : 0) | (newcurr != currp ? 2 : 0);
                u64 accm = path.pathm & path.okm;
                while (__ballot(accm != 0ull) != 0ull) {
                    const bool has = accm != 0ull;
                    const int s = has ? __builtin_ctzll(accm) : 0, src = sp.gbase + s;
                    const int now = (int)it0 + lane_read(sp.nd.depth, src);
                    const u64 chgm = lane_read(mychg, src);
                    const int fl = lane_read(myflags, src);
                    if (has) {
                        static_for<0, Cfg::NDRAW>([&](auto K) {
                            constexpr int k = decltype(K)::value;
                            const bool chg = ((chgm >> k) & 1ull) != 0ull;
                            const int hold = now - last[k];
                            hmax = (chg && (fl & 1) && hold > hmax) ? hold : hmax;
                            last[k] = chg ? now : last[k];
                        });
                        const bool chgc = (fl & 2) != 0;
                        const int hold = now - lastc;
                        hmax = (chgc && hold > hmax) ? hold : hmax;
                        lastc = chgc ? now : lastc;
                    }
                    accm &= accm - 1ull;
                }
            }
            // ---- the configuration the lane's step ENDS in (:51-53 | the rollbacks) ----
            chain_select<Cfg>(cp, ok, pr.n);
            currp = ok ? newcurr : currp;
            wp.abs = ok ? wn.abs : wp.abs;
            static_for<0, Cfg::NCOMP>([&](auto Q) { wp.v[decltype(Q)::value] = ok ? wn.v[decltype(Q)::value] : wp.v[decltype(Q)::value]; });
            Pp = ok ? newp : Pp;
            // ---- measurement  montecarlo.jl:144-172 ----
            if (path.onpath) {
                bool mf = true;
                i64 mj = it;
                if (a.measurefreq != 1) {
                    mj = it / a.measurefreq;
                    mf = mj * a.measurefreq == it;
                }
                if (mf && it >= nburn) {
                    if (currp != NORMI) {
                        double relw[Cfg::NCOMP]; // :162
                        static_for<0, Cfg::NCOMP>([&](auto Q) { relw[decltype(Q)::value] = wp.v[decltype(Q)::value] / Pp; });
                        if constexpr (Cfg::HOST_MEASURE != 0) { // :166-169 on the host, after the launch
                            if (tile == 0) host_measure_record<Cfg, Cfg::NCOMP>(a, wi.lb, (i64)g, mj, cp.x, relw, currp);
                        }
                        static_for<0, NI>([&](auto I) {
                            constexpr int i = decltype(I)::value;
                            if (currp == i) {
                                static_for<0, Cfg::NDRAW>([&](auto K) { // :147-154  accumulate!(var, pos + offset, 1.0)
                                    constexpr int k = decltype(K)::value;
                                    if constexpr ((Cfg::own_mask(i) >> k) & 1ull) hist_add<Cfg, k>(cp.bin[k], 1.0, sH, a.ghist, tile);
                                });
                                if constexpr (Cfg::HOST_MEASURE != 0) {
                                } else if constexpr (Cfg::CUSTOM_MEASURE != 0) { // measure(idx, var, obs, relative_weight, config)  :166-169
                                    double rwv[Cfg::NW];
                                    static_for<0, Cfg::NW>([&](auto Q) { rwv[decltype(Q)::value] = 0.0; });
                                    static_for<0, Cfg::NCOMP>([&](auto Q) { rwv[i * Cfg::NCOMP + decltype(Q)::value] = relw[decltype(Q)::value]; });
                                    Cfg::measure(cp.x, rwv, a.ud, i, obs_wave<Cfg>(sO));
                                } else if constexpr (Cfg::obs_bin_draw(i) >= 0) {
                                    const int b = cp.bin[Cfg::obs_bin_draw(i)];
                                    if (b >= 0 && b < Cfg::obs_nbin(i)) lds_add(&obs_wave<Cfg>(sO)[Cfg::obs_off(i) + b], relw[0]);
                                }
                            }
                            if constexpr (Cfg::CUSTOM_MEASURE == 0 && Cfg::obs_bin_draw(i) < 0) // :164
                                static_for<0, Cfg::NCOMP>([&](auto Q) { acc[i * Cfg::NCOMP + decltype(Q)::value] += currp == i ? relw[decltype(Q)::value] : 0.0; });
                        });
                    } else {
                        extra[XN] += 1.0 / rw[NORMI]; // :158
                    }
                }
            }
            // ---- the deepest lane on the path hands its end configuration to the group ----
            {
                const int src = sp.gbase + path.last;
                const int adv = path.pathm ? lane_read(sp.nd.depth, src) + 1 : 0;
                if (__ballot((path.okm & path.pathm) != 0ull) != 0ull) { // (a trip of rejections only leaves every group's base as it is)
                    c = lane_read<Cfg>(cp, src);
                    curr = lane_read(currp, src);
                    weight.abs = lane_read(wp.abs, src);
                    static_for<0, Cfg::NCOMP>([&](auto Q) { weight.v[decltype(Q)::value] = lane_read(wp.v[decltype(Q)::value], src); });
                    probability = lane_read(Pp, src);
                }
                it0 += adv;
                spec_adapt(a, sp, adv, path.okm & path.pathm);
            }
        }
        if (a.hold_hist && live && sp.m == 0) { // holds still running when the chain ends count with their length so far
            const int tot = (int)(steps + nburn);
            hmax = (tot - lastc > hmax) ? tot - lastc : hmax;
            static_for<0, NI>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if (curr == i) static_for<0, Cfg::NDRAW>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    constexpr int pv = Cfg::draw_pool(k);
                    constexpr bool fixed = Cfg::pool_nleaf(pv) == 1 && Cfg::leaf_kind(Cfg::draw_leaf(k)) == 1 && Cfg::leaf_nbin(Cfg::draw_leaf(k)) == 1;
                    if constexpr (((Cfg::own_mask(i) >> k) & 1ull) && !fixed) hmax = (tot - last[k] > hmax) ? tot - last[k] : hmax;
                });
            });
            atomicAdd(&a.hold_hist[hmax <= 0 ? 0 : 32 - __clz(hmax)], 1ull);
        }
        if (a.store_x && tile == 0 && live && sp.m == 0) {
            store_carried<Cfg>(a, wi.lb, ch, c);
            a.store_curr[wi.lb * a.nchain + ch] = curr;
        }
    }
    __syncthreads();
    flush_workgroup<Cfg, Lds<Cfg>, true, true>(a, smem, acc, extra, wi.rowid, tile);
}

} // namespace mci
