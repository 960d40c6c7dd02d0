// mci_device.h -- hand-written gfx950 (CDNA4, wave64) kernels of the VEGAS / VegasMC sample batch.
//
// This header is compiled twice: ahead of time by hipcc (static kernels in mci_static_kernels.hip)
// and at run time by hiprtc, where a tiny generated translation unit supplies a `Cfg` traits struct
// (the analogue of Julia specialising Vegas.montecarlo on Configuration{N,V,P,O,T}) plus the user's
// integrand body and instantiates the templates below.  It must stay free of host/std headers.
//
// Reference semantics (file:line under /root/reference/src):
//   draw            distribution/sampler.jl:293-305 (Continuous create!), :13-22 (Discrete create!)
//   sample batch    vegas/montecarlo.jl:117-187
//   padding         distribution/variable.jl:628-641
//   accumulate!     distribution/variable.jl:196-200, :362-367, :474-478
//   chains          vegas_mc/montecarlo.jl:151-232, vegas_mc/updates.jl:45-106
//   mcmc chains     mcmc/montecarlo.jl:72-184, mcmc/updates.jl:1-147
//
// MI355X mapping: one workgroup = one slice of ONE statistical block; the adaptive-grid tables and
// the per-bin weight histograms live in LDS (ds_read_b64 / ds_add_f64), Philox4x32-10 supplies the
// uniforms on chip, observables are reduced with wave64 shuffles.  No MFMA: the path is elementwise
// + reduction.  HBM sees only the table load and the per-workgroup partial flush.
#pragma once

namespace mci {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef long long i64;

template <int I> struct IC { static constexpr int value = I; };

template <int B, int E, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

// ---------------------------------------------------------------------------------------------
// RNG streams (DESIGN.md "RNG streams"; identical to oracle/mci_oracle.c:mcio_uniform)
//   key = (seed lo, seed hi); ctr = (index lo, index hi, k>>1, stream); draw k -> words 2(k&1), 2(k&1)+1
//   52 mantissa bits, [1,2) - 1 : the same resolution as Julia's MersenneTwister rand(Float64).
// ---------------------------------------------------------------------------------------------
struct u32x4 { u32 x, y, z, w; };

// Philox4x32 rounds of every stream: 10, or 7 with the opt-in cheaper generator (mci_set_rng_rounds; set by the generated translation unit)
#ifndef MCI_PHILOX_ROUNDS
#define MCI_PHILOX_ROUNDS 10
#endif

// The ten round keys (k + r * Weyl constant).  They are wave-uniform and would naturally sit in SGPRs -- but v_bitop3_b32 with an
// SGPR source issues at the 3-source rate (measured 1.76 ns per wave-instruction and SIMD, tools/issue_microbench.hip) while its
// all-VGPR form issues at the VOP2 rate (1.2 ns).  IN_VGPR copies them to VGPRs once (a pure, hoistable v_mov); worth 20 registers
// only where the kernel has them to spare (the pipelined :vegas loop on the histogram-copy plan, MCI_PIPE_VGPR_KEYS).
template <bool IN_VGPR> struct RoundKeys {
    u32 a[MCI_PHILOX_ROUNDS], b[MCI_PHILOX_ROUNDS];
};
template <bool IN_VGPR> __device__ __forceinline__ RoundKeys<IN_VGPR> make_round_keys(u32 k0, u32 k1) {
    RoundKeys<IN_VGPR> K;
#pragma unroll
    for (int r = 0; r < MCI_PHILOX_ROUNDS; ++r) {
        const u32 a = k0 + (u32)r * 0x9E3779B9u, b = k1 + (u32)r * 0xBB67AE85u; // wave-uniform: scalar ALU
        if (IN_VGPR) {
            asm("v_mov_b32 %0, %1" : "=v"(K.a[r]) : "s"(a));
            asm("v_mov_b32 %0, %1" : "=v"(K.b[r]) : "s"(b));
        } else {
            K.a[r] = a;
            K.b[r] = b;
        }
    }
    return K;
}

// 32 x 32 -> 64-bit product: one v_mad_u64_u32 (hi and lo in one issue)
__device__ __forceinline__ u64 mul_wide(u32 m, u32 c) { return (u64)m * c; }

template <bool IN_VGPR> __device__ __forceinline__ u32x4 philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, const RoundKeys<IN_VGPR> &K) {
#pragma unroll
    for (int r = 0; r < MCI_PHILOX_ROUNDS; ++r) {
        const u64 p0 = mul_wide(0xD2511F53u, c0); // v_mad_u64_u32: hi and lo in one issue
        const u64 p1 = mul_wide(0xCD9E8D57u, c2);
        // three-input xor in ONE issue: gfx950 has no v_xor3_b32, but v_bitop3_b32 with truth table 0x96 is exactly that;
        // the compiler does not form it by itself from a ^ b ^ c
        const u32 n0 = __builtin_amdgcn_bitop3_b32((u32)(p1 >> 32), c1, K.a[r], 0x96);
        const u32 n2 = __builtin_amdgcn_bitop3_b32((u32)(p0 >> 32), c3, K.b[r], 0x96);
        c1 = (u32)p1;
        c3 = (u32)p0;
        c0 = n0;
        c2 = n2;
    }
    return {c0, c1, c2, c3};
}
__device__ __forceinline__ u32x4 philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1) {
    return philox4x32_10<false>(c0, c1, c2, c3, make_round_keys<false>(k0, k1));
}

// 52 random mantissa bits as a double in [1, 2)
__device__ __forceinline__ double u12(u32 lo, u32 hi) {
    // ((hi:lo) >> 12) | 0x3FF0...0 as two v_alignbit_b32: the low word is (hi:lo) >> 12, the high word (0x3FF:hi) >> 12 =
    // 0x3FF00000 | hi >> 12.  Per instruction a 3-source form is no cheaper than the 64-bit shift + or it replaces
    // (tools/issue_microbench.hip), but with it the compiler keeps every Philox product a single v_mad_u64_u32; with the
    // 64-bit shift in the loop it splits 16 of them into v_mul_lo_u32 + v_mul_hi_u32 pairs (C2: 1.67 vs 1.72 ms per 1e8)
    const u32 wlo = __builtin_amdgcn_alignbit(hi, lo, 12u);
    const u32 whi = __builtin_amdgcn_alignbit(0x3FFu, hi, 12u);
    return __longlong_as_double((i64)(((u64)whi << 32) | wlo));
}
__device__ __forceinline__ double u01(u32 lo, u32 hi) { return u12(lo, hi) - 1.0; }
// the opt-in 32-bit stream of the :vegas solver (Cfg::RNG_BITS == 32, mci_set_rng_bits): ONE Philox word per draw, its 32 bits are the
// top 32 mantissa bits of a double in [1, 2) -- four draws per Philox call instead of two; y has a resolution of 2^-32
__device__ __forceinline__ double u12_word(u32 w) {
    const u32 whi = __builtin_amdgcn_alignbit(0x3FFu, w, 12u); // 0x3FF00000 | w >> 12
    return __longlong_as_double((i64)(((u64)whi << 32) | (w << 20)));
}
// uniform-plus-one of draw j (0 .. DPC-1) of a Philox block: DPC = 2 -> 52-bit draws from word pairs, DPC = 4 -> 32-bit draws
template <int DPC, int J> __device__ __forceinline__ double block_u12(const u32x4 &r) {
    if constexpr (DPC == 2) return J == 0 ? u12(r.x, r.y) : u12(r.z, r.w);
    else return u12_word(J == 0 ? r.x : J == 1 ? r.y : J == 2 ? r.z : r.w);
}

enum { STREAM_VEGAS = 0, STREAM_POOLINIT = 1, STREAM_MC_INIT = 2, STREAM_MC_STEP = 3, STREAM_MCMC_INIT = 4, STREAM_MCMC_STEP = 5, STREAM_MCMC_GROUP = 6, STREAM_MC_GROUP = 7 };
enum { ST_NORMALIZATION = 1, ST_HIST_NONFINITE = 2, ST_HIST_NONPOSITIVE = 4, ST_RESCALE_NONFINITE = 8, ST_MCMC_INIT = 16, ST_PERSIST_STALL = 32 };

// ---------------------------------------------------------------------------------------------
// kernel arguments (plain struct passed by value)
// ---------------------------------------------------------------------------------------------
// One node of a chain's speculation tree = one lane of the group that steps the chain (mci_spec.h; filled by the host, mci_api.hip
// spec_build).  Lanes are numbered ancestors-first.
struct SpecNode {
    int depth;   // the lane evaluates the proposal of step (first step of the trip) + depth
    int anc;     // nearest ancestor the way from the root leaves by its ACCEPT edge: the lane's step starts from that lane's proposal (-1: from the trip's base)
    int nacc;    // accept edges on the way from the root
    int levels;  // low byte: the most accept edges on any way through this node's tree; next byte: its deepest node (the same in all its nodes)
    u64 needacc; // lanes (bit = lane within the group) that must have accepted / rejected for this node to be on the chain's path
    u64 needrej;
    u64 accdepth; // the DEPTHS of the ancestors the way leaves by an accept edge (bit = depth): under :vegasmc a step's draw does not depend
                  // on the configuration it starts from, so a lane builds its starting configuration from the draws of those depths
    u64 anydepth; // ... of any node of this tree (the same in all its nodes): the depths whose draw somebody needs
};

struct BatchArgs {
    const double *edges;    // [NEDGE]  all Continuous grids, concatenated          (variable.jl:94)
    const double *dacc;     // [NDACC]  all Discrete accumulation tables            (variable.jl:280)
    const double *ddist;    // [NDDIST] all Discrete distribution tables            (variable.jl:281)
    const double *reweight; // [NI+1]   vegasmc only                                (configuration.jl:50)
    const double *ud;       // userdata                                             (configuration.jl:42)
    double *part_cols;      // [nWG][NCOLS]   per-workgroup partial statistics
    double *part_hist;      // [nWG][NBIN]    per-workgroup partial histograms (TABLE_MODE 0)
    double *ghist;          // [NBIN]         device histogram for global atomics (TABLE_MODE 1,2)
    double *part_pa;        // [nWG][2*NPA]   per-workgroup propose | accept tables (chain solvers; configuration.jl:185-186)
    u64 seed;
    u32 iteration;
    i64 neval_per_block;    // samples (vegas) or chain steps (vegasmc) per statistical block
    i64 block_lo;           // first global block index handled by this launch
    int wg_per_block;
    i64 measurefreq;
    i64 nchain;             // vegasmc: chains per block
    double burnin;          // vegasmc: a chain measures from step `burnin` on (montecarlo.jl:213; DESIGN.md "chains")
    i64 nburn;              // mcmc: burn-in steps run before the neval/nchain measured ones (mcmc/montecarlo.jl:133)
    int *status;            // error bits (ST_*)
    // vegas with NTILE > 1 histogram tiles: the sample pass keeps tile 0 and parks, per sample, the histogram
    // weights and the 16-bit bins of the other tiles' draws in HBM; mci_vegas_tiles replays them per tile
    double *tile_w;         // [NI][tile_stride]
    u32 *tile_bins;         // [tdraw_words][tile_stride]  32 / ceil(log2(nbin)) bins per word
    i64 tile_stride;        // samples of this launch
    // Many-grid (NTILE > 1) :vegas launches run in CHUNKS of a block's samples so that the parked stream stays bounded whatever neval is
    // (the reference's loop allocates nothing per sample, vegas/montecarlo.jl:117-187): this launch draws the samples chunk_lo <= n <
    // chunk_hi of every block -- same Philox indices as ever -- and parks sample n of local block lb at lb * chunk_len + (n - chunk_lo)
    // (tile_stride = blocks * chunk_len); accum != 0: a later chunk ADDS its partial rows to those of the chunks before it
    i64 chunk_lo, chunk_hi, chunk_len;
    int accum;
    i64 nrows;              // partial rows (block, slice) of this launch
    int tiles_wpb;          // split-all :vegas: replay workgroups per block and tile (mci_vegas_tiles has its own, coarser, partition of a
                            // block's samples: every one of its workgroups flushes a whole LDS tile), 0 = wg_per_block
    i64 tiles_rows;         // ... and the partial-histogram rows they write (= blocks * tiles_wpb), 0 = nrows
    // host integrand ("batch callback", Cfg::HOST_INTEGRAND): weights evaluated on the host for exactly the draws
    // this launch regenerates, host_w[q * tile_stride + sample]
    const double *host_w;
    // host measure ("batch callback", Cfg::HOST_MEASURE): the kernel accumulates no observable; it leaves, per sample of the
    // launch, the draws and the relative weights w * jac_i of the measured samples (0 for the others) for the host closure:
    // host_mx[k * tile_stride + sample], host_relw[q * tile_stride + sample]
    double *host_mx, *host_relw;
    // ... under a chain solver: a chain measures inside its step loop, so every chain leaves its j-th measured configuration (measured
    // steps are j * measurefreq, j = hm_first .. hm_first + hm_count - 1) in the record of its block,
    //     slot = (local block * nchain + chain) * hm_count + (j - hm_first);   host_mx[k * hm_stride + slot],
    //     :vegasmc  host_relw[q * hm_stride + slot], q < NW (vegas_mc/montecarlo.jl:218-227);  host_midx[slot] = -1
    //     :mcmc     host_relw[q * hm_stride + slot], q < NCOMP: the relative weight of the integrand the chain sits on,
    //               host_midx[slot] = that integrand (mcmc/montecarlo.jl:162-169); a chain on the normalization integrand
    //               writes nothing (the host presets host_midx = -1, host_relw = 0)
    // and the host closure runs over a block's records after the launch.
    int *host_midx;
    i64 hm_first, hm_count, hm_stride;
    // mcmc: [64] histogram over the chains of this launch of bit_width(longest holding time), the longest run of steps
    // during which a live slot (or the integrand index) of the chain did not change; feeds the automatic chain length
    unsigned long long *hold_hist;
    // A chain solver with a HOST integrand (Cfg::HOST_INTEGRAND under :vegasmc): the closure sits inside the Markov step
    // (vegas_mc/updates.jl:67-75), so a block's chains advance in lock step, ONE LAUNCH PER STEP, and every launch hands the host
    // the configurations it has to evaluate (hx[k * nc + chain]) and takes back their weights (host_w[q * nc + chain]):
    //   ne = 0            initialise the chains (montecarlo.jl:151-153), hand over their configurations
    //   ne = 1 .. steps   finish step ne-1 with the weights the host returned (ne = 1: the initial weights, :155-166), propose step ne
    //   ne = steps + 1    finish the last step
    // Chain state lives in global memory between launches (SoA, stride nc = chains of the launch).
    // Launch-bound :vegas iterations (a few workgroups, a few thousand samples each): the workgroup adds the non-zero bins of its LDS
    // histogram straight into `ghist` (global f64 atomics) instead of leaving a partial row for a merge launch of its own.  The value is
    // the number of `ghist` buffers the workgroups spread over (row r adds to buffer r % hist_atomic; k_finish sums them): 256 rows on one
    // buffer are 256 serialized atomics per bin
    int hist_atomic;
    // Carried chains (chain solvers, nchain > 1, an iteration that continues the previous one; DESIGN.md "Chains"): chain (block, ch)
    // does not draw a fresh start but continues from the configuration chain (block, ch % carry_nchain) ended the previous iteration
    // with -- bins and probabilities are looked up again on the refined grid -- and leaves its own end configuration for the next
    // iteration: x[k * cap + local block * nchain + ch], curr[...] (:mcmc: the integrand index).  Two buffers, read one, write the other.
    const double *carry_x;  // NULL: every chain starts afresh (montecarlo.jl:151-153, mcmc/montecarlo.jl:118-124)
    const int *carry_curr;
    i64 carry_nchain, carry_cap;
    // :mcmc: chain (block, ch) continues the stored chain carry_src[local block * nchain + ch] of its block -- the stored chains resampled
    // with probability ~ reweight_new[curr] / reweight_old[curr], i.e. to the target doReweight! has just moved (k_resample_chains,
    // mci_static_kernels.h); NULL (:vegasmc): chain ch continues stored chain ch % carry_nchain
    const int *carry_src;
    double *store_x;        // NULL: nothing kept
    int *store_curr;
    i64 store_cap;
    // :vegasmc: the chain's target density  pi(x) = sum_i reweight[i] |f_i(x)| pad_i(x) + reweight[N] pad_N(x)  (config.probability,
    // vegas_mc/montecarlo.jl:162-166) depends on the MAP (the paddings are map densities) and on the reweight factors, and train! /
    // doReweight! move both between iterations: the stored chains are a sample of the OLD target.  Every chain leaves the value of its
    // target at its end configuration (store_P); before the next launch mci_vegasmc_carry_weights evaluates the NEW target at every stored
    // configuration, carry_w = pi_new / pi_old, and k_resample_chains resamples the block's stored chains with probability ~ carry_w
    // (systematic, deterministic) into carry_src -- a start population distributed like the new target (DESIGN.md "Chains")
    const double *carry_P;  // [carry_cap] pi_old of the stored chains
    double *carry_w;        // [carry_total] out: pi_new / pi_old, indexed local block * carry_nchain + stored chain
    i64 carry_total;        // stored chains of this launch's blocks
    double *store_P;
    struct HostStep {
        i64 ne, steps, nc;
        double *cx, *cprob, *cw, *cprobability; // current configuration [NDRAW][nc], weights [NW][nc], config.probability [nc]
        int *cbin;
        double *hx, *pprob, *pprop, *puacc;     // proposal: x (= what the host evaluates) [NDRAW][nc], prob [NDRAW][nc], prop / accept draw [nc]
        int *pbin, *pvi;                        // proposal bins [NDRAW][nc]; pool the step changes, -1: nothing proposed
        // :mcmc (mcmc_host_step) -- every chain carries its own step counter: a chain whose start has to be redrawn
        // (mcmc/montecarlo.jl:118-124) lags behind the others, so the host launches until `done` says every chain finished
        double *cwabs;                          // |weight| of the current configuration [nc]
        int *ccurr, *cit, *ctr;                 // integrand index, steps done (-1: the start is still being evaluated), start tries
        int *pnew, *put;                        // proposal: integrand it lands on, update type (first index of propose[., ., .])
        int *hidx;                              // the integrand the host evaluates for this chain's hx, -1: nothing to evaluate
        int *done;                              // [1] chains that have run all their steps
    } hs;
    // Several lanes per chain (mci_spec.h): a chain is stepped by a group of spec_lanes lanes (a power of two, 2..64) along the
    // speculation tree spec_tab[spec_lanes]; spec_maxacc = the most accept edges on any way through it.  0 / NULL: one lane per chain
    // spec_ntree > 1: spec_tab holds that many trees of spec_lanes nodes each, built for the acceptances spec_accept[]; every group starts
    // on tree spec_first and moves, every few trips, to the tree built for the acceptance its own chain has shown since (the chain is the
    // same chain on any tree: only the number of steps a trip advances depends on it)
    const SpecNode *spec_tab;
    int spec_lanes, spec_maxacc;
    int spec_ntree, spec_first;
    float spec_accept[8];
    // :vegas, timed launches (mci_kernel_clocks): the first wave of workgroup 0 leaves the shader-clock ticks (s_memtime) and the
    // constant-rate reference ticks (s_memrealtime) its sample loop took -- their ratio is the clock the kernel actually ran at
    u64 *clock_out; // [2] or NULL
};

struct DumpArgs {
    const double *edges, *dacc, *ddist, *ud;
    double *x, *jac, *w;
    int soa; // 1: x[k*n + i] (draw-major, what a vectorised host integrand wants), jac/w not written
    u64 seed;
    u32 iteration;
    i64 first_index; // global sample index of the first dumped sample
    i64 n;
};

// ---------------------------------------------------------------------------------------------
// wave64 / workgroup reductions
// ---------------------------------------------------------------------------------------------
// Cross-lane moves through the VALU's data-parallel primitives (DPP: gfx9's row_shr / row_shl within a row of 16 lanes, row_bcast:15 and
// row_bcast:31 across rows): a dependent VALU instruction each, where __shfl_* is a ds_bpermute_b32 round trip through the LDS crossbar per
// 32-bit half (an epilogue's reductions and train!'s scans are chains of them, alone on their CU).  Lanes without a source read 0.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_read(double v) {
    const u64 u = (u64)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u32)u, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u32)(u >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double((long long)(((u64)(u32)hi << 32) | (u64)(u32)lo));
}
// inclusive prefix sums over the 64 lanes of a wave, in a fixed order
__device__ __forceinline__ double wave_scan_incl(double v) {
    v += dpp_read<0x111, 0xf>(v); // row_shr:1
    v += dpp_read<0x112, 0xf>(v); // row_shr:2
    v += dpp_read<0x114, 0xf>(v); // row_shr:4
    v += dpp_read<0x118, 0xf>(v); // row_shr:8
    v += dpp_read<0x142, 0xa>(v); // row_bcast:15 -> rows 1, 3
    v += dpp_read<0x143, 0xc>(v); // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    const u64 u = (u64)__double_as_longlong(wave_scan_incl(v)); // lane 63 holds the total
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)u, 63), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(u >> 32), 63);
    return __longlong_as_double((long long)(((u64)hi << 32) | (u64)lo)); // valid in every lane
}

// LDS f64 add -> ds_add_f64 (no return).  Contention is benign: iy is uniform in y-space by
// construction of the map (sampler.jl:378).
__device__ __forceinline__ void lds_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void global_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// table placement.  TABLE_MODE 0: grids + histograms in LDS; 1: grids in LDS, histograms via
// global f64 atomics; 2: everything from L2/HBM (grids too large for 160 KiB).
// ---------------------------------------------------------------------------------------------
//   TABLE_MODE 3: histograms in LDS, grids gathered from L2 (more than ~9 independent grids: the
//   ds_add_f64 is the part that must not go to global memory); when all histograms do not fit either they
//   are split into NTILE tiles and each (block, slice) is run by NTILE workgroups, workgroup `tile`
//   keeping only its tile's histograms (draws + integrand are recomputed: compute is cheaper than atomics).
template <class Cfg> struct Mode {
    static constexpr bool EDGE_LDS = Cfg::TABLE_MODE <= 1;
    static constexpr bool HIST_LDS = Cfg::TABLE_MODE == 0 || Cfg::TABLE_MODE == 3;
};

template <class Cfg> struct Tables {
    const double *EC; // split-all sample pass: LDS cache of the leading leaves' edges (Cfg::leaf_ecoff)
    const double *E;  // grid edges (LDS or global)
    const double *DA; // discrete accumulation (LDS)
    const double *DD; // discrete distribution (LDS)
};

// one leaf draw: create! in its Jacobian form.  Returns x, the bin index (0-based) and `raw` with
// 1/prob = raw * jac_scale(K): raw = dx for a Continuous leaf (scale N), 1/distribution for a Discrete one.
// With Cfg::PAIR_TABLE the LDS table holds (g[i], g[i+1]-g[i]) pairs: ONE aligned ds_read_b128 per draw and
// no subtraction on the critical path (the pair is formed with the same rounding when the table is staged).
// U12 = true: `y` is the uniform PLUS ONE (in [1, 2), see u12()).
// y * N of a Continuous draw (FMA: y is the uniform plus one and (y - 1) * N is formed as fma(y, N, -N), see draw_leaf)
template <int N, bool FMA, bool U12> __device__ __forceinline__ double cont_yn(double y) {
    if constexpr (FMA) {
        double Nd = (double)N;
        asm("" : "+s"(Nd)); // an SGPR pair the compiler cannot fold into a literal
        return __builtin_fma(y, Nd, -Nd);
    } else return (U12 ? y - 1.0 : y) * (double)N;
}
// every draw a Continuous leaf served from the LDS pair table (the batched reads of draw_sample)
template <class Cfg> constexpr bool all_draws_pair_table() {
    if (Cfg::PAIR_TABLE == 0 || Cfg::TABLE_MODE > 1) return false;
    for (int k = 0; k < Cfg::NDRAW; ++k)
        if (Cfg::leaf_kind(Cfg::draw_leaf(k)) != 0) return false;
    return true;
}

// A grid gathered from global memory (table modes 2 / 3): edges g[iy], g[iy + 1] of the increment a draw falls into.  MCI_GATHER_X4 (an
// experiment, off: profiles/r05_ablation.txt): ONE 16-byte load -- dword-aligned global loads of any width are legal on gfx950 -- instead
// of two 8-byte loads of neighbouring addresses.
#ifndef MCI_GATHER_X4
#define MCI_GATHER_X4 0
#endif
__device__ __forceinline__ void gather_edges(const double *e, double &g0, double &g1) {
#if MCI_GATHER_X4
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    const d2u v = *reinterpret_cast<const d2u *>(e);
    g0 = v.x;
    g1 = v.y;
#else
    g0 = e[0];
    g1 = e[1];
#endif
}

template <class Cfg, int K, bool U12 = false, bool ECACHE = false> __device__ __forceinline__ void draw_leaf(const Tables<Cfg> &t, double y, double &x, double &raw, int &bin) {
    constexpr int leaf = Cfg::draw_leaf(K);
    if constexpr (Cfg::leaf_kind(leaf) != 0 && U12) y -= 1.0;
    if constexpr (Cfg::leaf_kind(leaf) == 0) {
        // sampler.jl:295-303:  iy = floor(y*N)+1; dy = y*N-(iy-1); x = g[iy] + dy*(g[iy+1]-g[iy]); prob = 1/(N*dx)
        constexpr int N = Cfg::leaf_nbin(leaf);
        // (y1 - 1) * N in one instruction (C2 -0.9 %, C4 -0.5 %, C5 :vegas -2 %: 515 instead of 531 VALU instructions per C2 sample): y1 - 1 is exact for y1 in [1, 2), so fma(y1, N, -N) rounds the same real number once -- the same bits
        // (N through an opaque SGPR pair: with the literal the compiler picks v_fmac_f64 and rebuilds the -N accumulator with two
        // v_mov_b32 per draw; v_fma_f64 v, v, s, -s uses one SGPR pair twice, which the constant bus allows.  The whole v_fma_f64 as
        // inline asm made the C3 :vegas kernel 17 % slower: different inlining, 143 -> 157 VGPRs)
        const double yn = cont_yn<N, (U12 && Cfg::NDRAW >= 8), U12>(y); // (few draws: nothing to gain, and the C3 :vegas kernel came out 17 % slower with either fma form)
        const int iy = (int)yn;                           // y*N >= 0: trunc == floor
        const double dy = __builtin_amdgcn_fract(yn);     // v_fract_f64 == yn - floor(yn), exact
        double g0, dx;
        if constexpr (Cfg::PAIR_TABLE != 0 && Cfg::TABLE_MODE <= 1) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            // The pair table starts at LDS address 0: the JIT kernels keep no static LDS, so the dynamic segment -- whose first entry the
            // table is (Lds::E == 0) -- begins there; the host checks .group_segment_fixed_size == 0 on every code object it loads.
            // With the address formed from that constant the byte offset is ONE VOP2 shift and the leaf's offset rides in the
            // instruction's immediate; through the `smem` symbol the compiler emits v_lshl_add_u32 v, iy, 4, <base = 0, known only
            // after instruction selection>, a three-source form that issues at 1.75 ns instead of 1.0 (tools/issue_microbench.hip)
            typedef const d2 __attribute__((address_space(3))) lds_d2;
            const u32 boff = ((u32)iy << 4) + (u32)(Cfg::leaf_poff(leaf) * 8);
            const d2 e = *(lds_d2 *)boff;
            g0 = e.x;
            dx = e.y;
        } else if constexpr (ECACHE && Cfg::leaf_ecoff(leaf) >= 0) {
            constexpr int eoff = Cfg::leaf_ecoff(leaf); // this leaf's edges sit in the LDS cache of the split-all sample pass
            g0 = t.EC[eoff + iy];
            dx = t.EC[eoff + iy + 1] - g0;
        } else {
            constexpr int eoff = Cfg::leaf_eoff(leaf);
            double g1;
            gather_edges(t.E + eoff + iy, g0, g1); // (L2 gathers in table modes 2/3: non-temporal loads measured 30 % slower)
            dx = g1 - g0;
        }
        x = g0 + dy * dx;
        raw = dx;
        bin = iy;
    } else if constexpr (Cfg::leaf_kind(leaf) == 2) {
        // a component of a FermiK slot: the D components are created jointly (fermik_create), not per draw
        (void)t;
        (void)y;
        x = 0.0;
        raw = 1.0;
        bin = 0;
    } else {
        // sampler.jl:17-20 + common.jl:16-25 bisection on accumulation[1..K+1]
        constexpr int Kn = Cfg::leaf_nbin(leaf);
        constexpr int aoff = Cfg::leaf_eoff(leaf);
        constexpr int doff = Cfg::leaf_doff(leaf);
        int jl = 1, ju = Kn + 2;
        while (ju - jl > 1) {
            const int jm = (jl + ju) >> 1;
            if (y < t.DA[aoff + jm - 1]) ju = jm;
            else jl = jm;
        }
        if (jl > Kn) jl = Kn; // accumulation[end] <= y < 1 through rounding: reference raises (common.jl:10-12)
        x = Cfg::leaf_lower(leaf) + (double)(jl - 1);
        raw = 1.0 / t.DD[doff + jl - 1];
        bin = jl - 1;
    }
}

// 1/prob = raw * jac_scale(k)
template <class Cfg> constexpr double jac_scale(int k) {
    return Cfg::leaf_kind(Cfg::draw_leaf(k)) == 0 ? (double)Cfg::leaf_nbin(Cfg::draw_leaf(k)) : 1.0;
}
// product of jac_scale over the draws lo <= k < hi of `mask` (compile-time).  The N factors are applied once per group of
// 8 draws, not once per sample: the bare product of 8 increments cannot underflow, the bare product of 48 narrow ones does
// (D = 48 sharply peaked dimensions with increments ~1e-7 gave weights of exactly 0 when the scale was applied at the end)
template <class Cfg> constexpr double jac_scale_product(unsigned long long mask, int lo = 0, int hi = 64) {
    double p = 1.0;
    for (int k = lo; k < Cfg::NDRAW && k < hi; ++k)
        if ((mask >> k) & 1ull) p *= jac_scale<Cfg>(k);
    return p;
}
constexpr int kJacGroup = 8;

// draws whose histogram is replayed by mci_vegas_tiles (adaptive and covered by some integrand; tile >= 1, or every tile in
// split-all mode), in draw order
template <class Cfg> constexpr bool is_tdraw(int k) {
    return Cfg::NTILE > 1 && Cfg::leaf_adapt(Cfg::draw_leaf(k)) != 0 && Cfg::cover_mask(k) != 0ull && Cfg::leaf_tile(Cfg::draw_leaf(k)) >= (Cfg::SPLIT_ALL != 0 ? 0 : 1);
}
template <class Cfg> constexpr int tdraw_count() {
    int n = 0;
    for (int k = 0; k < Cfg::NDRAW; ++k) n += is_tdraw<Cfg>(k) ? 1 : 0;
    return n;
}
template <class Cfg> constexpr int tdraw_pos(int k) { // position of draw k in that list
    int n = 0;
    for (int j = 0; j < k; ++j) n += is_tdraw<Cfg>(j) ? 1 : 0;
    return n;
}
// parked bins are packed tdraw_bits() bits each (10 bits for the default 999-bin grids)
template <class Cfg> constexpr int tdraw_bits() {
    int mx = 2;
    for (int k = 0; k < Cfg::NDRAW; ++k)
        if (is_tdraw<Cfg>(k) && Cfg::leaf_nbin(Cfg::draw_leaf(k)) > mx) mx = Cfg::leaf_nbin(Cfg::draw_leaf(k));
    int b = 1;
    while ((1 << b) < mx) ++b;
    return b;
}
// ... contiguously: tdraw m occupies bits [m * BITS, (m + 1) * BITS) of the word array, a field may straddle two words.  32 grids of
// 999 bins: 10 words per sample instead of 11 (3 fields per word, 2 bits idle), and a tile of 16 grids is exactly 5 words -- 12 % less
// for the replay to read, which is bound by exactly that.
template <class Cfg> constexpr int tdraw_bitpos(int m) { return m * tdraw_bits<Cfg>(); }
template <class Cfg> constexpr int tdraw_words() { return (tdraw_count<Cfg>() * tdraw_bits<Cfg>() + 31) / 32; }
// does tdraw m touch word j?
template <class Cfg> constexpr bool tdraw_in_word(int m, int j) {
    const int lo = tdraw_bitpos<Cfg>(m), hi = lo + tdraw_bits<Cfg>() - 1;
    return lo / 32 == j || hi / 32 == j;
}
// the field of tdraw M out of a sample's words
template <class Cfg, int M> __device__ __forceinline__ int tdraw_extract(const u32 *word) {
    constexpr int BITS = tdraw_bits<Cfg>(), off = tdraw_bitpos<Cfg>(M), w = off / 32, sh = off % 32;
    if constexpr (sh + BITS <= 32) return (int)((word[w] >> sh) & ((1u << BITS) - 1u));
    else return (int)(((word[w] >> sh) | (word[w + 1] << (32 - sh))) & ((1u << BITS) - 1u));
}

// blockIdx -> (statistical block, slice of the block, histogram tile)
// all NDRAW draws of one sample + Jacobians.  jaci[i] = product of 1/prob over integrand i's own draws
// ( = weights*padding_probability*jac of vegas/montecarlo.jl:152 up to rounding ).
template <class Cfg> struct Sample {
    double x[Cfg::NDRAW > 0 ? Cfg::NDRAW : 1];
    int bin[Cfg::NDRAW > 0 ? Cfg::NDRAW : 1];
    double pj[Cfg::NDRAW > 0 ? Cfg::NDRAW : 1]; // 1/prob per draw (dead-code eliminated where unused)
    // several histogram tiles: the bins of the replayed draws, packed as they are drawn (tdraw_bits() bits each) -- 32 separate
    // bin registers kept until the sample is parked cost the 32-grid sample pass its third and fourth wave per SIMD
    u32 word[tdraw_words<Cfg>() > 0 ? tdraw_words<Cfg>() : 1];
    double jac;
    double jaci[Cfg::NI];
};

template <class Cfg, int K> __device__ __forceinline__ void pack_bin(Sample<Cfg> &s) {
    if constexpr (is_tdraw<Cfg>(K)) {
        constexpr int BITS = tdraw_bits<Cfg>(), off = tdraw_bitpos<Cfg>(tdraw_pos<Cfg>(K)), w = off / 32, sh = off % 32;
        s.word[w] |= (u32)s.bin[K] << sh; // (the words start at zero: the gather phase draws out of order)
        if constexpr (sh + BITS > 32) s.word[w + 1] |= (u32)s.bin[K] >> (32 - sh);
    }
}
template <class Cfg, bool ECACHE = false, bool KV = false, int DPC = 2> __device__ __forceinline__ void draw_sample(const Tables<Cfg> &t, const RoundKeys<KV> &keys, u32 stream, u64 index, Sample<Cfg> &s) {
    const u32 ilo = (u32)index, ihi = (u32)(index >> 32);
    constexpr unsigned long long ALL = Cfg::NDRAW >= 64 ? ~0ull : ((1ull << Cfg::NDRAW) - 1ull);
    s.jac = 1.0;
    static_for<0, Cfg::NI>([&](auto I) { s.jaci[decltype(I)::value] = 1.0; });
    static_for<0, tdraw_words<Cfg>()>([&](auto J) { s.word[decltype(J)::value] = 0u; });
    // LDS pair tables, 8..16 draws: every Philox block of the sample first, fenced, then the draws.  That is the schedule the compiler
    // used to pick by itself for the 16-D headline loop (102 VGPRs, two ds_read_b128 in flight); with the loop specialised on
    // measurefreq == 1 it interleaved blocks and draws instead (92 VGPRs, one read in flight, each waited for at once) and the shorter
    // loop ran 3 % SLOWER.  The fence pins the order.
    constexpr int NCH = (Cfg::NDRAW + DPC - 1) / DPC;
    constexpr bool PHILOX_FIRST = DPC == 2 && Cfg::NDRAW >= 8 && Cfg::NDRAW <= 16 && !ECACHE && Cfg::PAIR_TABLE != 0 &&
                                  Cfg::TABLE_MODE <= 1;
    u32x4 rr[PHILOX_FIRST ? NCH : 1];
    if constexpr (PHILOX_FIRST) {
        static_for<0, NCH>([&](auto C) { rr[decltype(C)::value] = philox4x32_10<KV>(ilo, ihi, (u32)decltype(C)::value, stream, keys); });
        __builtin_amdgcn_sched_barrier(0);
    }
    // ... and the table reads of RB draws are issued back to back before the first pair is used: a wave then waits for the LDS
    // twice per 16 draws instead of eight times.  The loop's VALU work alone takes 1.20 ms per 1e8 samples, its LDS work alone 1.17 ms,
    // and with two reads in flight per wave (what the compiler schedules) the two only overlap to 1.43 ms (profiles/r02_ablation.txt)
    constexpr int RB = (PHILOX_FIRST && all_draws_pair_table<Cfg>()) ? 8 : 0;
    static_assert(RB % DPC == 0, "a batch of reads covers whole Philox blocks");
    typedef double pair_d2 __attribute__((ext_vector_type(2)));
    pair_d2 pe[RB > 0 ? RB : 1];
    double pdy[RB > 0 ? RB : 1];
    static_for<0, NCH>([&](auto C) {
        constexpr int c = decltype(C)::value;
        u32x4 r;
        if constexpr (PHILOX_FIRST) r = rr[c];
        else r = philox4x32_10<KV>(ilo, ihi, (u32)c, stream, keys);
        if constexpr (RB > 0) {
            if constexpr ((DPC * c) % RB == 0) { // open a batch: bins, fractions and reads of draws DPC*c .. DPC*c + RB - 1
                static_for<0, RB>([&](auto J) {
                    constexpr int j = decltype(J)::value, k = DPC * c + j;
                    if constexpr (k < Cfg::NDRAW) {
                        constexpr int leaf = Cfg::draw_leaf(k), N = Cfg::leaf_nbin(leaf);
                        const double yn = cont_yn<N, true, true>(block_u12<DPC, k % DPC>(rr[k / DPC])); // as draw_leaf forms it
                        s.bin[k] = (int)yn;
                        pdy[j] = __builtin_amdgcn_fract(yn);
                        typedef const pair_d2 __attribute__((address_space(3))) lds_pair;
                        pe[j] = *(lds_pair *)(((u32)s.bin[k] << 4) + (u32)(Cfg::leaf_poff(leaf) * 8));
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        static_for<0, DPC>([&](auto H) {
            constexpr int k = DPC * c + decltype(H)::value;
            if constexpr (k < Cfg::NDRAW) {
                const double y1 = block_u12<DPC, decltype(H)::value>(r);
                double raw;
                if constexpr (RB > 0) { // x = g[iy] + dy * (g[iy+1] - g[iy])   sampler.jl:299
                    s.x[k] = pe[k % RB].x + pdy[k % RB] * pe[k % RB].y;
                    raw = pe[k % RB].y;
                    (void)y1;
                } else
                draw_leaf<Cfg, k, true, ECACHE>(t, y1, s.x[k], raw, s.bin[k]);
                pack_bin<Cfg, k>(s);
                s.pj[k] = raw * jac_scale<Cfg>(k);
                s.jac *= raw; // jac /= prob   vegas/montecarlo.jl:126 (scale applied below)
                static_for<0, Cfg::NI>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (((Cfg::own_mask(i) >> k) & 1ull) && Cfg::own_mask(i) != ALL) s.jaci[i] *= raw;
                });
            }
        });
        if constexpr (((DPC * c + DPC) % kJacGroup == 0 || DPC * c + DPC >= Cfg::NDRAW)) { // close a group of draws: apply its N factors
            constexpr int hi = DPC * c + DPC, lo = ((hi - 1) / kJacGroup) * kJacGroup;
            constexpr double sc = jac_scale_product<Cfg>(ALL, lo, hi);
            if constexpr (sc != 1.0) s.jac *= sc;
            static_for<0, Cfg::NI>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr double si = jac_scale_product<Cfg>(Cfg::own_mask(i), lo, hi);
                if constexpr (Cfg::own_mask(i) != ALL && si != 1.0) s.jaci[i] *= si;
            });
        }
    });
    static_for<0, Cfg::NI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (Cfg::own_mask(i) == ALL) s.jaci[i] = s.jac; // dof[i] == maxdof: no padding (vegas/montecarlo.jl:82)
    });
}

template <class Cfg, bool ECACHE = false> __device__ __forceinline__ void draw_sample(const Tables<Cfg> &t, u64 seed, u32 stream, u64 index, Sample<Cfg> &s) {
    draw_sample<Cfg, ECACHE, false>(t, make_round_keys<false>((u32)seed, (u32)(seed >> 32)), stream, index, s);
}

// draws whose grid edges are gathered from global memory (table modes 2/3: not staged in LDS, not in the LDS edge cache)
template <class Cfg, bool ECACHE> constexpr bool is_gather_draw(int k) {
    return Cfg::TABLE_MODE >= 2 && Cfg::leaf_kind(Cfg::draw_leaf(k)) == 0 && !(ECACHE && Cfg::leaf_ecoff(Cfg::draw_leaf(k)) >= 0);
}
template <class Cfg, bool ECACHE> constexpr int gather_draw_count() {
    int n = 0;
    for (int k = 0; k < Cfg::NDRAW; ++k) n += is_gather_draw<Cfg, ECACHE>(k) ? 1 : 0;
    return n;
}

// The draws of a sample, GATHER DRAWS FIRST AND DIMENSION-MAJOR (draw_gather_phase_pipe), then the rest (draw_rest_phase): every wave
// of the workgroup walks the gathered grids in the same order, with a workgroup barrier after every Philox chunk -- so at any moment
// the whole CU reads ONE or two 8 KB edge tables, which then live in its 32 KB L1 (measured: 58 ns per wave-gather and SIMD from an
// L1-resident table against 148 ns when 13..32 tables compete for the L1 and every access is an L2 line fill,
// tools/issue_microbench.hip).  The Philox streams are counter-based, so the order of evaluation is free; a chunk that holds one
// gathered and one cached draw is simply computed in both phases.  Jacobians are products of the per-draw 1/prob (no bare
// increment products, so no scaling groups are needed).  Must be called by every thread of the workgroup (barriers inside).
template <class Cfg, bool ECACHE, int DPC, int K, class SampleT> __device__ __forceinline__ void phased_one_draw(const Tables<Cfg> &t, const u32x4 &r, SampleT &s) {
    constexpr unsigned long long ALL = Cfg::NDRAW >= 64 ? ~0ull : ((1ull << Cfg::NDRAW) - 1ull);
    const double y1 = block_u12<DPC, K % DPC>(r);
    double raw;
    draw_leaf<Cfg, K, true, ECACHE>(t, y1, s.x[K], raw, s.bin[K]);
    pack_bin<Cfg, K>(s);
    const double pj = raw * jac_scale<Cfg>(K);
    s.pj[K] = pj;
    s.jac *= pj; // jac /= prob   vegas/montecarlo.jl:126
    static_for<0, Cfg::NI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (((Cfg::own_mask(i) >> K) & 1ull) && Cfg::own_mask(i) != ALL) s.jaci[i] *= pj;
    });
}
// does chunk c (draws DPC*c .. DPC*c + DPC - 1) hold a gathered draw / a draw of the other kind
template <class Cfg, bool ECACHE, int DPC> constexpr bool chunk_has(int c, bool gather) {
    for (int j = 0; j < DPC; ++j) {
        const int k = DPC * c + j;
        if (k < Cfg::NDRAW && is_gather_draw<Cfg, ECACHE>(k) == gather) return true;
    }
    return false;
}
// The gather phase of ONE sample per lane, software-pipelined by hand one chunk deep: chunk c's table reads are issued, the
// workgroup barrier passed, and only then chunk c-1's reads are consumed (x, bin, Jacobian) -- so at most two chunks' loads
// (edges + fractions) are live at any point instead of the whole phase's, which is what the scheduler does when left alone
// (13 gathers x 6 registers on C4: 6.76 -> 6.46 ms per 1e8 samples at 768 threads).  sched_barrier keeps the order.  Must be
// called by every thread of the workgroup (barriers inside).
struct PendingGather {
    double g0, g1, dy;
    int iy;
};
template <class Cfg, bool ECACHE, int DPC> constexpr int prev_gather_chunk(int c) {
    for (int j = c - 1; j >= 0; --j)
        if (chunk_has<Cfg, ECACHE, DPC>(j, true)) return j;
    return -1;
}
template <class Cfg, bool ECACHE, int DPC, int C> __device__ __forceinline__ void finish_gather_chunk(const PendingGather (&pend)[DPC], Sample<Cfg> &s) {
    constexpr unsigned long long ALL = Cfg::NDRAW >= 64 ? ~0ull : ((1ull << Cfg::NDRAW) - 1ull);
    static_for<0, DPC>([&](auto J) {
        constexpr int j = decltype(J)::value, k = DPC * C + j;
        if constexpr (k < Cfg::NDRAW) {
            if constexpr (is_gather_draw<Cfg, ECACHE>(k)) {
                const double dx = pend[j].g1 - pend[j].g0;
                s.x[k] = pend[j].g0 + pend[j].dy * dx;
                s.bin[k] = pend[j].iy;
                pack_bin<Cfg, k>(s);
                const double pj = dx * jac_scale<Cfg>(k);
                s.pj[k] = pj;
                s.jac *= pj;
                static_for<0, Cfg::NI>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (((Cfg::own_mask(i) >> k) & 1ull) && Cfg::own_mask(i) != ALL) s.jaci[i] *= pj;
                });
            }
        }
    });
}
template <class Cfg, bool ECACHE, bool KV, int DPC> __device__ __forceinline__ void draw_gather_phase_pipe(const Tables<Cfg> &t, const RoundKeys<KV> &keys, u32 stream,
                                                                                                         const u64 index, Sample<Cfg> &s) {
    constexpr int NCHUNK = (Cfg::NDRAW + DPC - 1) / DPC;
    s.jac = 1.0;
    static_for<0, Cfg::NI>([&](auto I) { s.jaci[decltype(I)::value] = 1.0; });
    static_for<0, tdraw_words<Cfg>()>([&](auto J) { s.word[decltype(J)::value] = 0u; });
    PendingGather pend[2][DPC];
    static_for<0, NCHUNK>([&](auto C) {
        constexpr int c = decltype(C)::value;
        if constexpr (chunk_has<Cfg, ECACHE, DPC>(c, true)) {
            constexpr int slot = [] { int n = 0; for (int j = 0; j < c; ++j) n += chunk_has<Cfg, ECACHE, DPC>(j, true) ? 1 : 0; return n & 1; }();
            const u32x4 r = philox4x32_10<KV>((u32)index, (u32)(index >> 32), (u32)c, stream, keys);
            __builtin_amdgcn_s_setprio(1); // (the gathers of this chunk go out ahead of the other waves' Philox blocks)
            static_for<0, DPC>([&](auto J) {
                constexpr int j = decltype(J)::value, k = DPC * c + j;
                if constexpr (k < Cfg::NDRAW) {
                    if constexpr (is_gather_draw<Cfg, ECACHE>(k)) {
                        constexpr int leaf = Cfg::draw_leaf(k), N = Cfg::leaf_nbin(leaf), eoff = Cfg::leaf_eoff(leaf);
                        const double yn = (block_u12<DPC, j>(r) - 1.0) * (double)N;
                        const int iy = (int)yn;
                        pend[slot][j].iy = iy;
                        pend[slot][j].dy = __builtin_amdgcn_fract(yn);
                        gather_edges(t.E + eoff + iy, pend[slot][j].g0, pend[slot][j].g1);
                    }
                }
            });
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier(); // keeps the waves on the same tables; no data is exchanged
            constexpr int pc = prev_gather_chunk<Cfg, ECACHE, DPC>(c);
            if constexpr (pc >= 0) finish_gather_chunk<Cfg, ECACHE, DPC, pc>(pend[slot ^ 1], s);
            __builtin_amdgcn_sched_barrier(0);
        }
    });
    constexpr int last = prev_gather_chunk<Cfg, ECACHE, DPC>(NCHUNK);
    if constexpr (last >= 0) {
        constexpr int lslot = [] { int n = 0; for (int j = 0; j < last; ++j) n += chunk_has<Cfg, ECACHE, DPC>(j, true) ? 1 : 0; return n & 1; }();
        finish_gather_chunk<Cfg, ECACHE, DPC, last>(pend[lslot], s);
    }
}
// the remaining draws of ONE sample (LDS-resident grids, Discrete tables), right before its integrand is evaluated
template <class Cfg, bool ECACHE, bool KV, int DPC> __device__ __forceinline__ void draw_rest_phase(const Tables<Cfg> &t, const RoundKeys<KV> &keys, u32 stream, u64 index,
                                                                                                  Sample<Cfg> &s) {
    constexpr unsigned long long ALL = Cfg::NDRAW >= 64 ? ~0ull : ((1ull << Cfg::NDRAW) - 1ull);
    constexpr int NCHUNK = (Cfg::NDRAW + DPC - 1) / DPC;
    static_for<0, NCHUNK>([&](auto C) {
        constexpr int c = decltype(C)::value;
        if constexpr (chunk_has<Cfg, ECACHE, DPC>(c, false)) {
            const u32x4 r = philox4x32_10<KV>((u32)index, (u32)(index >> 32), (u32)c, stream, keys);
            __builtin_amdgcn_s_setprio(1); // (issue priority for the table reads behind a Philox block, as in draw_sample_pipe: C4 4.89 -> 4.81 ms with both phases)
            static_for<0, DPC>([&](auto J) {
                constexpr int k = DPC * c + decltype(J)::value;
                if constexpr (k < Cfg::NDRAW) {
                    if constexpr (!is_gather_draw<Cfg, ECACHE>(k)) phased_one_draw<Cfg, ECACHE, DPC, k>(t, r, s);
                }
            });
            __builtin_amdgcn_s_setprio(0);
        }
    });
    static_for<0, Cfg::NI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (Cfg::own_mask(i) == ALL) s.jaci[i] = s.jac; // dof[i] == maxdof: no padding (vegas/montecarlo.jl:82)
    });
}

// stage the tables into LDS (coalesced 8-byte loads, once per workgroup)
template <class Cfg> __device__ __forceinline__ void stage_tables(const double *gE, const double *gDA, const double *gDD, double *sE, double *sDA, double *sDD) {
    const int tid = threadIdx.x, T = blockDim.x;
    if constexpr (Cfg::TABLE_MODE <= 1) {
        if constexpr (Cfg::PAIR_TABLE != 0) {
            static_for<0, Cfg::NLEAF>([&](auto Lf) {
                constexpr int l = decltype(Lf)::value;
                if constexpr (Cfg::leaf_kind(l) == 0) {
                    constexpr int eoff = Cfg::leaf_eoff(l), poff = Cfg::leaf_poff(l);
                    for (int i = tid; i < Cfg::leaf_nbin(l); i += T) {
                        const double g0 = gE[eoff + i], g1 = gE[eoff + i + 1];
                        sE[poff + 2 * i] = g0;
                        sE[poff + 2 * i + 1] = g1 - g0;
                    }
                }
            });
        } else {
            for (int i = tid; i < Cfg::NEDGE; i += T) sE[i] = gE[i];
        }
    }
    for (int i = tid; i < Cfg::NDACC; i += T) sDA[i] = gDA[i];
    for (int i = tid; i < Cfg::NDDIST; i += T) sDD[i] = gDD[i];
}

// config.propose / config.accept (configuration.jl:185-186): [3 updates][Nd integrands][max(Nd, Nv) targets], row-major, 0-based:
//   changeIntegrand [0][curr][new]  (mcmc/updates.jl:48,50)      changeVariable [1][curr][vi]  (mcmc/updates.jl:100,102;
//   swapVariable    [2][curr][vi]   (mcmc/updates.jl:138,140)    vegasmc: [1][0][vi], vegas_mc/updates.jl:90,92)
// A workgroup counts in LDS (64-bit integers: exact, order-free) and flushes one row of part_pa.
template <class Cfg> struct PaTable {
    static constexpr int ND = Cfg::NI + 1;
    static constexpr int M = ND > Cfg::NPOOL ? ND : Cfg::NPOOL;
    static constexpr int N = 3 * ND * M;
    static __device__ __forceinline__ int idx(int ut, int curr, int target) { return (ut * ND + curr) * M + target; }
};
__device__ __forceinline__ void lds_count(u64 *p, u64 n) { __hip_atomic_fetch_add(p, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// propose[idx] += 1 on every lane of the wave for which `pred` holds, accept[idx] likewise under `acc`.  UNIFORM: idx is the same
// on all active lanes (one ballot + one LDS add per wave instead of up to 64 colliding atomics)
template <class Cfg, bool UNIFORM> __device__ __forceinline__ void pa_count(u64 *sPA, int idx, bool pred, bool acc) {
    if constexpr (UNIFORM) {
        const u64 mp = __ballot(pred), ma = __ballot(pred && acc);
        if (mp != 0ull && (int)(threadIdx.x & 63) == __builtin_ctzll(mp)) {
            lds_count(&sPA[idx], (u64)__popcll(mp));
            if (ma != 0ull) lds_count(&sPA[PaTable<Cfg>::N + idx], (u64)__popcll(ma));
        }
    } else if (pred) {
        lds_count(&sPA[idx], 1ull);
        if (acc) lds_count(&sPA[PaTable<Cfg>::N + idx], 1ull);
    }
}

// LDS carve (doubles).  Order: grid table | dacc | ddist | hist | obs | reduction scratch | propose/accept counters
template <class Cfg> struct Lds {
    static constexpr int E = 0;
    static constexpr int DA = E + (Cfg::TABLE_MODE <= 1 ? (Cfg::PAIR_TABLE != 0 ? Cfg::NPAIR : Cfg::NEDGE) : 0);
    static constexpr int DD = DA + Cfg::NDACC;
    static constexpr int H = DD + Cfg::NDDIST;
    static constexpr int O = H + (Mode<Cfg>::HIST_LDS ? Cfg::HTILE * Cfg::HCOPY : 0);
    static constexpr int R = O + Cfg::NOBS * (Cfg::DET != 0 ? Cfg::HCOPY : 1); // (deterministic mode: one observable array per wave, obs_wave)
    static constexpr int PA = R + 16 /*waves*/ * Cfg::NCOLS; // propose | accept counters (u64), chain solvers
    static constexpr int END = PA + 2 * PaTable<Cfg>::N;
};

template <bool B, class X, class Y> struct SelectType { using type = X; };
template <class X, class Y> struct SelectType<false, X, Y> { using type = Y; };
// the split-all sample pass has no histogram: observables and the reduction scratch move up, the edge cache takes the rest
template <class Cfg> struct LdsEC {
    static constexpr int E = 0, DA = Lds<Cfg>::DA, DD = Lds<Cfg>::DD, H = Lds<Cfg>::H;
    static constexpr int O = H;
    static constexpr int R = O + Cfg::NOBS * (Cfg::DET != 0 ? Cfg::HCOPY : 1);
    static constexpr int EC = R + 16 * Cfg::NCOLS;
    static constexpr int END = EC + Cfg::EC_DOUBLES;
};

// (draw_leaf addresses the pair table from LDS address 0)
template <class Cfg> struct LdsTableAtZero { static_assert(Lds<Cfg>::E == 0 && LdsEC<Cfg>::E == 0, "the edge / pair table opens the dynamic LDS segment"); };

// partial-statistics columns written per workgroup:
//   [0, NOBS) observables | NOBS normalization | NOBS+1 neval | NOBS+2 .. +NI+1 visited(N+1)
template <class Cfg> struct Cols {
    static constexpr int NORM = Cfg::NOBS;
    static constexpr int NEVAL = Cfg::NOBS + 1;
    static constexpr int VISITED = Cfg::NOBS + 2;
    static_assert(VISITED + Cfg::NI + 1 == Cfg::NCOLS, "column layout");
};

// Histogram copies (Cfg::HCOPY, a power of two; one histogram tile only): the workgroup keeps HCOPY interleaved copies of its LDS
// histograms, sH[bin * HCOPY + copy], and lane l adds to copy l % HCOPY.  The 64 / HCOPY lanes that share a copy then share
// 32 / HCOPY bank pairs instead of all 64 lanes colliding at random over the 32 (a random-address ds_add_f64 costs 41 ns per
// wave-instruction and SIMD, a conflict-free one 13.4; tools/issue_microbench.hip); the copies are summed, in a fixed order,
// when the workgroup writes its partial histogram.
// Deterministic mode (Cfg::DET, mci_set_deterministic): HCOPY = the workgroup's waves and every WAVE owns a copy.  A wave's adds to
// its copy happen in program order, and the lanes of one ds_add_f64 that hit the same bin are served in the hardware's fixed lane
// order, so the contents of every copy -- and, the copies being summed in a fixed order, the workgroup's partial histogram -- do
// not depend on how the waves of the workgroup were scheduled: a fixed seed gives bit-identical histograms, hence grids, run to run
// (the reference's sequential loop is reproducible the same way, configuration.jl:190).  Same for the LDS observables (obs_wave).
template <class Cfg> __device__ __forceinline__ int hslot(int flat) {
    if constexpr (Cfg::HCOPY == 1) return flat;
    else if constexpr (Cfg::DET != 0) return (int)(threadIdx.x >> 6) * Cfg::HTILE + flat; // copy-major: a wave's random bins spread over all banks
                                                                                         // (bin-major, all 64 lanes would share 32 / HCOPY bank pairs)
    else return flat * Cfg::HCOPY + (int)(threadIdx.x & (unsigned)(Cfg::HCOPY - 1));
}
// copies of the LDS observable array: one per wave in deterministic mode (HCOPY is the wave count there)
template <class Cfg> constexpr int ocopy() { return Cfg::DET != 0 ? Cfg::HCOPY : 1; }
template <class Cfg> __device__ __forceinline__ double *obs_wave(double *sO) {
    if constexpr (ocopy<Cfg>() == 1) return sO;
    else return sO + (int)(threadIdx.x >> 6) * Cfg::NOBS;
}

// histogram update of one sample: accumulate!(var, pos+offset, weight) for every (integrand i, draw k in own(i))
// (vegas/montecarlo.jl:170-185).  The per-integrand weights covering the same draw are summed first,
// so each draw costs one ds_add_f64.
// TILE >= 0: compile-time histogram tile of this workgroup (the bins of the other tiles' draws are then dead
// values and leave the register file); TILE < 0: run-time `tile`.
template <class Cfg, int TILE = -1> __device__ __forceinline__ void hist_update(const Sample<Cfg> &s, const double *wh /*[NI]*/, double *sH, double *gH, int tile) {
    static_for<0, Cfg::NDRAW>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int leaf = Cfg::draw_leaf(k);
        if constexpr (Cfg::leaf_adapt(leaf) != 0 && Cfg::cover_mask(k) != 0ull) { // T.adapt  variable.jl:197,:363
            double wk = 0.0;
            static_for<0, Cfg::NI>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr ((Cfg::own_mask(i) >> k) & 1ull) wk += wh[i];
            });
            if constexpr (Mode<Cfg>::HIST_LDS) {
                constexpr int lt = Cfg::leaf_tile(leaf);
                if constexpr (TILE >= 0) {
                    if constexpr (Cfg::NTILE == 1 || TILE == lt) lds_add(&sH[hslot<Cfg>(Cfg::leaf_boff(leaf) - Cfg::tile_boff(lt) + s.bin[k])], wk);
                } else {
                    if (Cfg::NTILE == 1 || tile == lt) lds_add(&sH[hslot<Cfg>(Cfg::leaf_boff(leaf) - Cfg::tile_boff(lt) + s.bin[k])], wk);
                }
            } else {
                global_add(&gH[Cfg::leaf_boff(leaf) + s.bin[k]], wk);
            }
        }
    });
}

// the histogram add of ONE draw (hist_update's body for draw K; one LDS tile)
template <class Cfg, int K> __device__ __forceinline__ void hist_add_draw(int bin, const double *wh /*[NI]*/, double *sH) {
    constexpr int leaf = Cfg::draw_leaf(K);
    if constexpr (Cfg::leaf_adapt(leaf) != 0 && Cfg::cover_mask(K) != 0ull) {
        double wk = 0.0;
        static_for<0, Cfg::NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr ((Cfg::own_mask(i) >> K) & 1ull) wk += wh[i];
        });
        lds_add(&sH[hslot<Cfg>(Cfg::leaf_boff(leaf) - Cfg::tile_boff(Cfg::leaf_tile(leaf)) + bin)], wk);
    }
}

// Software-pipelined :vegas sample (the kernels of pipe_eligible()): the histogram adds of the PREVIOUS sample and the table reads of
// this one are spread between the Philox blocks, and a pair is consumed one block after its read was issued (at once on the 32-bit stream).  Left to the
// compiler a trip is a long VALU-only stretch (eight Philox blocks, ~280 instructions) followed by an LDS-heavy one (16 reads, each
// waited for, then 16 atomics back to back), and since all waves of a CU run the same code at the same pace they tend to queue for the
// same pipe: measured on the 16-D headline loop, VALU work alone 1.20 ms per 1e8 samples, LDS work alone 1.17 ms, both together 1.43 ms.
// With every stretch of the instruction stream carrying the same VALU : LDS mix the two pipes overlap whatever the phase of the waves.
// Same draws, same arithmetic in the same order as draw_sample + hist_update (the atomics of a sample land one trip later).
template <class Cfg> struct PendingHist {
    int bin[Cfg::NDRAW > 0 ? Cfg::NDRAW : 1];
    double wh[Cfg::NI];
};
template <class Cfg> constexpr bool pipe_eligible() {
    // (the opt-in 32-bit stream too: four draws per Philox block, four reads and four atomics per stage)
    return Cfg::NDRAW >= 8 && Cfg::NDRAW <= 16 && all_draws_pair_table<Cfg>() && Cfg::NTILE == 1 &&
           Mode<Cfg>::HIST_LDS && Cfg::HOST_INTEGRAND == 0 && Cfg::HOST_MEASURE == 0 && Cfg::EC_DOUBLES == 0;
}
template <class Cfg, bool KV, int DPC> __device__ __forceinline__ void draw_sample_pipe(const RoundKeys<KV> &keys, u32 stream, u64 index, Sample<Cfg> &s,
                                                                                        const PendingHist<Cfg> &pend, PendingHist<Cfg> &next, double *sH) {
    constexpr int NCH = (Cfg::NDRAW + DPC - 1) / DPC;
    constexpr int LAG = DPC == 4 ? 0 : 1; // blocks between a read and its use: with four reads per block they cover each other (and LAG 1 spills at 1024 threads)
    constexpr unsigned long long ALL = Cfg::NDRAW >= 64 ? ~0ull : ((1ull << Cfg::NDRAW) - 1ull);
    const u32 ilo = (u32)index, ihi = (u32)(index >> 32);
    s.jac = 1.0;
    static_for<0, Cfg::NI>([&](auto I) { s.jaci[decltype(I)::value] = 1.0; });
    static_for<0, tdraw_words<Cfg>()>([&](auto J) { s.word[decltype(J)::value] = 0u; });
    typedef double pair_d2 __attribute__((ext_vector_type(2)));
    pair_d2 pe[Cfg::NDRAW];
    double pdy[Cfg::NDRAW];
    static_for<0, NCH + LAG>([&](auto C) {
        constexpr int c = decltype(C)::value;
        if constexpr (c < NCH) {
            const u32x4 r = philox4x32_10<KV>(ilo, ihi, (u32)c, stream, keys);
            // The wave asks for issue priority while it hands its LDS work over (s_setprio 1 ... 0): its two reads and two atomics then
            // go out ahead of the other waves' Philox stretches instead of queueing behind them, and the LDS pipe has them while this
            // wave computes its next block.  Measured on the headline loop, kernel ms per 1e8 samples on two boxes: 1.357 / 1.371 ->
            // 1.308 / 1.318 and 1.367 / 1.368 -> 1.340 / 1.341; priority 2 or 3 the same; around the reads alone or the atomics
            // alone: nothing or worse; held through the consumption of the pairs: half of it (profiles/r04_ablation.txt); a third box,
            // three interleaved pairs of runs: 1.392 / 1.360 / 1.370 -> 1.317 / 1.320 / 1.325.
            __builtin_amdgcn_s_setprio(1);
            static_for<0, DPC>([&](auto H) { // bins, fractions and table reads of this block's draws
                constexpr int k = DPC * c + decltype(H)::value;
                if constexpr (k < Cfg::NDRAW) {
                    constexpr int leaf = Cfg::draw_leaf(k), N = Cfg::leaf_nbin(leaf);
                    const double yn = cont_yn<N, true, true>(block_u12<DPC, decltype(H)::value>(r)); // as draw_leaf forms it
                    s.bin[k] = next.bin[k] = (int)yn; // (the bins go straight into the record the NEXT trip adds from)
                    pdy[k] = __builtin_amdgcn_fract(yn);
                    typedef const pair_d2 __attribute__((address_space(3))) lds_pair;
                    pe[k] = *(lds_pair *)(((u32)s.bin[k] << 4) + (u32)(Cfg::leaf_poff(leaf) * 8));
                }
            });
            static_for<0, DPC>([&](auto H) { // two of the previous sample's histogram adds
                constexpr int k = DPC * c + decltype(H)::value;
                if constexpr (k < Cfg::NDRAW) hist_add_draw<Cfg, k>(pend.bin[k], pend.wh, sH);
            });
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (c >= LAG) { // the draws whose pairs were asked for LAG blocks ago: x = g[iy] + dy * (g[iy+1] - g[iy])  sampler.jl:299
            constexpr int cc = c - LAG;
            static_for<0, DPC>([&](auto H) {
                constexpr int k = DPC * cc + decltype(H)::value;
                if constexpr (k < Cfg::NDRAW) {
                    s.x[k] = pe[k].x + pdy[k] * pe[k].y;
                    const double raw = pe[k].y;
                    pack_bin<Cfg, k>(s);
                    s.pj[k] = raw * jac_scale<Cfg>(k);
                    s.jac *= raw; // jac /= prob   vegas/montecarlo.jl:126 (scale applied below)
                    static_for<0, Cfg::NI>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        if constexpr (((Cfg::own_mask(i) >> k) & 1ull) && Cfg::own_mask(i) != ALL) s.jaci[i] *= raw;
                    });
                }
            });
            if constexpr (((DPC * cc + DPC) % kJacGroup == 0 || DPC * cc + DPC >= Cfg::NDRAW)) { // close a group of draws: apply its N factors
                constexpr int hi = DPC * cc + DPC, lo = ((hi - 1) / kJacGroup) * kJacGroup;
                constexpr double sc = jac_scale_product<Cfg>(ALL, lo, hi);
                if constexpr (sc != 1.0) s.jac *= sc;
                static_for<0, Cfg::NI>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    constexpr double si = jac_scale_product<Cfg>(Cfg::own_mask(i), lo, hi);
                    if constexpr (Cfg::own_mask(i) != ALL && si != 1.0) s.jaci[i] *= si;
                });
            }
            if constexpr (c < NCH) __builtin_amdgcn_sched_barrier(0);
        }
    });
    static_for<0, Cfg::NI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (Cfg::own_mask(i) == ALL) s.jaci[i] = s.jac; // dof[i] == maxdof: no padding (vegas/montecarlo.jl:82)
    });
}

// abs(weights[i]) (vegas/montecarlo.jl:173): |w| for Float64, the modulus for ComplexF64 weights stored (re, im)
template <class Cfg, int I> __device__ __forceinline__ double absw(const double *w) {
    if constexpr (Cfg::NCOMP == 1) return fabs(w[I]);
    else return hypot(w[2 * I], w[2 * I + 1]);
}

// default measure (vegas/montecarlo.jl:151-153), "bin by a Discrete draw" (example/bubble.jl:81-84), or the user's
// measure body (vegas/montecarlo.jl:156-161), which accumulates into the LDS observable array through obs_add(k, v).
// relw: NI*NCOMP relative weights; acc: NI*NCOMP register accumulators of the default measure.
template <class Cfg> __device__ __forceinline__ void measure(const double *x, const int *bin, const double *relw, const double *ud,
                                                             double *acc, double *sO) {
    if constexpr (Cfg::CUSTOM_MEASURE != 0) {
        Cfg::measure(x, relw, ud, -1, sO);
    } else {
        static_for<0, Cfg::NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (Cfg::obs_bin_draw(i) < 0) {
                static_for<0, Cfg::NCOMP>([&](auto Q) { acc[i * Cfg::NCOMP + decltype(Q)::value] += relw[i * Cfg::NCOMP + decltype(Q)::value]; });
            } else {
                static_assert(Cfg::NCOMP == 1 || Cfg::obs_bin_draw(i) < 0, "binned observables are real");
                constexpr int kd = Cfg::obs_bin_draw(i);
                const int b = bin[kd];
                if (b >= 0 && b < Cfg::obs_nbin(i)) lds_add(&sO[Cfg::obs_off(i) + b], relw[i]);
            }
        });
    }
}

// host measure under a chain solver (Cfg::HOST_MEASURE, BatchArgs::host_mx): record of the chain's measured step j * measurefreq
template <class Cfg, int NR> __device__ __forceinline__ void host_measure_record(const BatchArgs &a, i64 lb, i64 ch, i64 j, const double *x, const double *relw /*[NR]*/, int idx) {
    const i64 ord = j - a.hm_first;
    if (ord < 0 || ord >= a.hm_count) return;
    const i64 slot = (lb * a.nchain + ch) * a.hm_count + ord;
    static_for<0, Cfg::NDRAW>([&](auto K) { a.host_mx[decltype(K)::value * a.hm_stride + slot] = x[decltype(K)::value]; });
    static_for<0, NR>([&](auto Q) { a.host_relw[decltype(Q)::value * a.hm_stride + slot] = relw[decltype(Q)::value]; });
    a.host_midx[slot] = idx;
}

// workgroup epilogue: registers -> wave shuffle -> LDS -> one row of part_cols; LDS histogram -> part_hist
template <class Cfg, class L = Lds<Cfg>, bool WRITE_HIST = true, bool WRITE_PA = false, bool ACCUM = false> __device__ __forceinline__ void flush_workgroup(const BatchArgs &a, double *smem, const double *acc, const double *extra /*[NCOLS-NOBS]*/, i64 rowid, int tile) {
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = T >> 6;
    double *sO = smem + L::O, *sR = smem + L::R, *sH = smem + L::H;
    const bool accum = ACCUM || (Cfg::NTILE > 1 && a.accum != 0); // (a later chunk of a many-grid :vegas launch, BatchArgs::accum)
    // scalar observables of the default measure (register accumulators)
    if constexpr (Cfg::CUSTOM_MEASURE == 0) {
        static_for<0, Cfg::NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (Cfg::obs_bin_draw(i) < 0) {
                static_for<0, Cfg::NCOMP>([&](auto Q) {
                    const double v = wave_sum(acc[i * Cfg::NCOMP + decltype(Q)::value]);
                    if (lane == 0) sR[wave * Cfg::NCOLS + Cfg::obs_off(i) + decltype(Q)::value] = v;
                });
            }
        });
    }
    static_for<Cfg::NOBS, Cfg::NCOLS>([&](auto Cc) {
        constexpr int c = decltype(Cc)::value;
        const double v = wave_sum(extra[c - Cfg::NOBS]);
        if (lane == 0) sR[wave * Cfg::NCOLS + c] = v;
    });
    __syncthreads();
    double *row = a.part_cols + rowid * Cfg::NCOLS;
    for (int c = tid; c < Cfg::NCOLS && tile == 0; c += T) { // the NTILE workgroups of a slice hold identical statistics
        bool binned = Cfg::CUSTOM_MEASURE != 0; // in LDS (sO): binned observables, everything under a user measure
        static_for<0, Cfg::NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (c >= Cfg::obs_off(i) && c < Cfg::obs_off(i) + Cfg::obs_nbin(i)) binned = binned || Cfg::obs_bin_draw(i) >= 0;
        });
        double v = 0.0;
        if (c < Cfg::NOBS && Cfg::HOST_MEASURE != 0) v = 0.0; // the host closure's observables join the row later (k_add_host_obs)
        else if (c < Cfg::NOBS && binned) {
            v = sO[c];
            static_for<1, ocopy<Cfg>()>([&](auto Cc) { v += sO[decltype(Cc)::value * Cfg::NOBS + c]; }); // (deterministic mode: the waves' copies, fixed order)
        }
        else
            for (int w = 0; w < nwave; ++w) v += sR[w * Cfg::NCOLS + c]; // fixed order: deterministic
        row[c] = accum ? row[c] + v : v; // (ACCUM: one launch per Markov step adds to the row the host zeroed)
    }
    if constexpr (WRITE_PA) { // the workgroup's propose | accept counters -> its row of part_pa (exact integers below 2^53)
        const u64 *sPA = reinterpret_cast<const u64 *>(smem + L::PA);
        double *prow = a.part_pa + rowid * (2 * PaTable<Cfg>::N);
        for (int i = tid; i < 2 * PaTable<Cfg>::N && tile == 0; i += T) prow[i] = ACCUM ? prow[i] + (double)sPA[i] : (double)sPA[i];
    }
    if constexpr (Mode<Cfg>::HIST_LDS && WRITE_HIST) {
        // ABLATION SWITCH, off (MCI_JIT_FLAGS=-DMCI_COPY_SUM_DPP=1; tools/run_batch.sh midsize): the interleaved copies summed IN PLACE
        // first -- lane j reads slot j (a wave reads 64 consecutive doubles), the HCOPY neighbouring lanes that hold one bin's copies add
        // them up with row_shr moves, the last lane of the group stores the sum at sH[bin] (slots are read a round before anything is
        // stored over them, behind the round's barrier).  Measured 1.7 us SLOWER per launch at every size than the loop below, whose
        // HCOPY consecutive doubles per lane the compiler already reads as 16-byte loads (profiles/r06_latency.txt).
#ifndef MCI_COPY_SUM_DPP
#define MCI_COPY_SUM_DPP 0
#endif
        constexpr bool SUMMED = MCI_COPY_SUM_DPP != 0 && Cfg::DET == 0 && Cfg::HCOPY > 1 && Cfg::HCOPY <= 16 && Cfg::NTILE == 1;
        if constexpr (SUMMED) {
            constexpr int N = Cfg::HTILE * Cfg::HCOPY, K = 4;
            for (int base = 0; base < N; base += T * K) {
                double v[K];
                static_for<0, K>([&](auto Kk) {
                    constexpr int k = decltype(Kk)::value;
                    const int j = base + k * T + tid;
                    v[k] = j < N ? sH[j] : 0.0;
                });
                static_for<0, K>([&](auto Kk) {
                    constexpr int k = decltype(Kk)::value;
                    if constexpr (Cfg::HCOPY > 1) v[k] += dpp_read<0x111, 0xf>(v[k]); // row_shr:1 (groups of HCOPY lanes never straddle a row of 16)
                    if constexpr (Cfg::HCOPY > 2) v[k] += dpp_read<0x112, 0xf>(v[k]);
                    if constexpr (Cfg::HCOPY > 4) v[k] += dpp_read<0x114, 0xf>(v[k]);
                    if constexpr (Cfg::HCOPY > 8) v[k] += dpp_read<0x118, 0xf>(v[k]);
                });
                __syncthreads();
                static_for<0, K>([&](auto Kk) {
                    constexpr int k = decltype(Kk)::value;
                    const int j = base + k * T + tid;
                    if (j < N && (tid & (Cfg::HCOPY - 1)) == Cfg::HCOPY - 1) sH[j / Cfg::HCOPY] = v[k];
                });
            }
            __syncthreads();
        }
        static_for<0, Cfg::NTILE>([&](auto Tt) {
            constexpr int tt = decltype(Tt)::value;
            if (tile == tt) {
                double *hrow = a.part_hist + rowid * Cfg::NBIN + Cfg::tile_boff(tt);
                for (int i = tid; i < Cfg::tile_nbin(tt); i += T) {
                    constexpr int SB = (Cfg::DET != 0 || SUMMED) ? 1 : Cfg::HCOPY, SC = Cfg::DET != 0 ? Cfg::HTILE : 1; // strides of bin and copy (hslot)
                    double v = sH[i * SB];
                    if constexpr (!SUMMED)
                    static_for<1, Cfg::HCOPY>([&](auto Cc) { v += sH[i * SB + decltype(Cc)::value * SC]; }); // fixed order
                    if (!accum && a.hist_atomic) { // (BatchArgs::hist_atomic: no merge launch behind this one)
                        if (v != 0.0) global_add(&a.ghist[(rowid % a.hist_atomic) * Cfg::NBIN + Cfg::tile_boff(tt) + i], v);
                    } else
                    hrow[i] = accum ? hrow[i] + v : v;
                }
            }
        });
    }
}

struct WorkItem {
    i64 rowid, lb;
    int slice, tile;
};
template <class Cfg> __device__ __forceinline__ WorkItem work_item(const BatchArgs &a) {
    WorkItem w;
    w.tile = Cfg::NTILE == 1 ? 0 : (int)(blockIdx.x % Cfg::NTILE);
    w.rowid = Cfg::NTILE == 1 ? (i64)blockIdx.x : (i64)(blockIdx.x / Cfg::NTILE);
    w.lb = w.rowid / a.wg_per_block;
    w.slice = (int)(w.rowid % a.wg_per_block);
    return w;
}

// =============================================================================================
// VEGAS sample batch  (vegas/montecarlo.jl:117-187)
// =============================================================================================
// SPLIT (NTILE > 1): this pass owns histogram tile 0 only and parks (weights, bins of the other tiles' draws)
// per sample for mci_vegas_tiles; one workgroup per (block, slice).
template <class Cfg, bool SPLIT = false> __device__ __forceinline__ void vegas_batch(const BatchArgs &a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, T = blockDim.x;
    // split-all (SPLIT and Cfg::SPLIT_ALL): no histogram in this pass; the LDS it would take caches the leading leaves' edges.
    // One tile with L2-gathered grids (table mode 3): the LDS left over next to the histogram caches as many grids as fit.
    constexpr bool NOHIST = SPLIT && Cfg::SPLIT_ALL != 0;
    constexpr bool EC = Cfg::EC_DOUBLES > 0 && (NOHIST || !SPLIT);
    using L = typename SelectType<NOHIST, LdsEC<Cfg>, Lds<Cfg>>::type;
    double *sE = smem + L::E, *sDA = smem + L::DA, *sDD = smem + L::DD;
    double *sH = smem + L::H, *sO = smem + L::O;
    stage_tables<Cfg>(a.edges, a.dacc, a.ddist, sE, sDA, sDD);
#ifndef MCI_ZERO_B128
#define MCI_ZERO_B128 0 // (ablation switch, off: 16-byte zeroing stores made no difference, profiles/r06_latency.txt)
#endif
    if constexpr (Mode<Cfg>::HIST_LDS && !NOHIST) {
        if constexpr (MCI_ZERO_B128 != 0 && L::H % 2 == 0 && (Cfg::HTILE * Cfg::HCOPY) % 2 == 0) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 *z = reinterpret_cast<d2 *>(sH);
            for (int i = tid; i < Cfg::HTILE * Cfg::HCOPY / 2; i += T) z[i] = d2{0.0, 0.0};
        } else
        for (int i = tid; i < Cfg::HTILE * Cfg::HCOPY; i += T) sH[i] = 0.0;
    }
    for (int i = tid; i < Cfg::NOBS * ocopy<Cfg>(); i += T) sO[i] = 0.0;
    Tables<Cfg> t;
    t.EC = nullptr;
    if constexpr (EC) {
        double *sEC = smem + (NOHIST ? LdsEC<Cfg>::EC : Lds<Cfg>::END);
        static_for<0, Cfg::NLEAF>([&](auto Lf) {
            constexpr int l = decltype(Lf)::value;
            if constexpr (Cfg::leaf_kind(l) == 0 && Cfg::leaf_ecoff(l) >= 0)
                for (int i = tid; i <= Cfg::leaf_nbin(l); i += T) sEC[Cfg::leaf_ecoff(l) + i] = a.edges[Cfg::leaf_eoff(l) + i];
        });
        t.EC = sEC;
    }
    __syncthreads();
    if constexpr (Mode<Cfg>::EDGE_LDS) t.E = sE;
    else t.E = a.edges;
    t.DA = sDA;
    t.DD = sDD;

    WorkItem wi = work_item<Cfg>(a);
    if constexpr (SPLIT) {
        wi.tile = 0;
        wi.rowid = (i64)blockIdx.x;
        wi.lb = wi.rowid / a.wg_per_block;
        wi.slice = (int)(wi.rowid % a.wg_per_block);
    }
    const int slice = wi.slice, tile = wi.tile;
    const i64 B = a.block_lo + wi.lb; // global statistical block
    const u32 stream = a.iteration * 8u + STREAM_VEGAS;
    const i64 stride = (i64)a.wg_per_block * T;

    double acc[Cfg::NW];
    static_for<0, Cfg::NW>([&](auto I) { acc[decltype(I)::value] = 0.0; });
    double extra[Cfg::NCOLS - Cfg::NOBS];
    static_for<0, Cfg::NCOLS - Cfg::NOBS>([&](auto I) { extra[decltype(I)::value] = 0.0; });

    // measurement cadence (n + 1) % measurefreq == 0 (:148) without a 64-bit division in the sample loop: the remainder is
    // carried along, n advances by `stride` per trip
#ifndef MCI_PIPE_VGPR_KEYS
#define MCI_PIPE_VGPR_KEYS 0 // (set by the host for the first compile of a copy-plan kernel: mci_api.hip compile_solver)
#endif
    // round keys in VGPRs: 20 registers, for the pipelined loop when the host found them free
    constexpr bool KV = MCI_PIPE_VGPR_KEYS != 0 && pipe_eligible<Cfg>() && !SPLIT && Cfg::EC_DOUBLES == 0;
    constexpr int DPC = Cfg::RNG_BITS == 32 ? 4 : 2;           // draws per Philox block of the :vegas sample stream
    const RoundKeys<KV> keys = make_round_keys<KV>((u32)a.seed, (u32)(a.seed >> 32));
    const i64 mfreq = a.measurefreq;
    // the block's samples this launch draws: all of them, or one chunk of a many-grid launch (BatchArgs::chunk_lo)
    const i64 n_lo = SPLIT ? a.chunk_lo : 0, n_hi = SPLIT ? a.chunk_hi : a.neval_per_block;
    i64 mrem = mfreq == 1 ? 0 : (n_lo + (i64)slice * T + tid + 1) % mfreq;
    const i64 mstep = mfreq == 1 ? 0 : stride % mfreq;
    // gathered grids (table mode 3) are walked dimension-major so that they are served from L1
    constexpr bool PHASED = Cfg::L1_PHASE > 0 && Cfg::HOST_INTEGRAND == 0 && gather_draw_count<Cfg, EC>() > 0;
    // the sample loop, specialised on the workgroup's histogram tile and on measurefreq == 1 (the reference's default, main.jl:84: every
    // sample is measured and the carried remainder with its 64-bit compare / select -- a dozen VALU instructions per sample -- is gone)
    auto run = [&](auto TT, auto MF1c) {
    constexpr bool MF1 = decltype(MF1c)::value != 0;
    auto process = [&](const i64 n, const Sample<Cfg> &s, double *defer_wh = nullptr) { // everything after the draws of sample n
        double w[Cfg::NW];
        if constexpr (Cfg::HOST_INTEGRAND != 0) { // the closure ran on the host over the dumped draws
            const i64 hidx = wi.lb * a.neval_per_block + n;
            static_for<0, Cfg::NW>([&](auto Q) { w[decltype(Q)::value] = a.host_w[decltype(Q)::value * a.tile_stride + hidx]; });
        } else {
            Cfg::integrand(s.x, w, a.ud, -1); // vegas/montecarlo.jl:140-144
        }
        extra[Cols<Cfg>::NEVAL - Cfg::NOBS] += 1.0; // config.neval += 1   :118
        bool domeasure = true; // :148
        if constexpr (!MF1) {
            domeasure = mrem == 0;
            mrem += mstep;
            mrem = mrem >= mfreq ? mrem - mfreq : mrem;
        }
        if constexpr (Cfg::HOST_MEASURE != 0) { // measure(vars, obs, relative_weights, config) runs on the host over this launch's samples (:156-161)
            const i64 hidx = wi.lb * a.neval_per_block + n;
            static_for<0, Cfg::NDRAW>([&](auto K) { a.host_mx[decltype(K)::value * a.tile_stride + hidx] = s.x[decltype(K)::value]; });
            static_for<0, Cfg::NW>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                a.host_relw[q * a.tile_stride + hidx] = domeasure ? w[q] * s.jaci[q / Cfg::NCOMP] : 0.0; // :152
            });
            if (domeasure) extra[Cols<Cfg>::NORM - Cfg::NOBS] += 1.0; // :164
        } else if (domeasure) {
            double relw[Cfg::NW];
            static_for<0, Cfg::NW>([&](auto Q) { constexpr int q = decltype(Q)::value; relw[q] = w[q] * s.jaci[q / Cfg::NCOMP]; }); // :152
            measure<Cfg>(s.x, s.bin, relw, a.ud, acc, obs_wave<Cfg>(sO));
            extra[Cols<Cfg>::NORM - Cfg::NOBS] += 1.0; // :164
        }
        double wh[Cfg::NI];
        static_for<0, Cfg::NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const double wj = absw<Cfg, i>(w) * s.jac; // :173-174 (full jac, not the integrand's own: author's warning :175)
            wh[i] = wj * wj;                      // :180
        });
        if (defer_wh) { // pipelined loop: the adds of this sample are issued during the next sample's draws (draw_sample_pipe)
            static_for<0, Cfg::NI>([&](auto I) { defer_wh[decltype(I)::value] = wh[decltype(I)::value]; });
        } else if constexpr (!NOHIST) hist_update<Cfg, decltype(TT)::value>(s, wh, sH, a.ghist, tile);
        if constexpr (SPLIT) { // park what the other tiles need: coalesced (lane == consecutive sample) 8- and 4-byte stores
            const i64 idx = wi.lb * a.chunk_len + (n - n_lo);
            static_for<0, Cfg::NI>([&](auto I) { a.tile_w[decltype(I)::value * a.tile_stride + idx] = wh[decltype(I)::value]; });
            constexpr int NWORD = tdraw_words<Cfg>();
            static_for<0, NWORD>([&](auto J) { a.tile_bins[decltype(J)::value * a.tile_stride + idx] = s.word[decltype(J)::value]; });
        }
    };
    if constexpr (PHASED) {
        // The phased trips are those in which every thread of the workgroup holds a valid sample (barriers inside): their body is
        // unconditional -- the integrand may consume the draws as they come instead of keeping all of them for a guarded call -- and what
        // is left at the end of the block, at most one sample per lane, goes through the plain loop.
        const i64 n0 = n_lo + (i64)slice * T + tid, first = n_lo + (i64)slice * T;
        const i64 jfull = first + T <= n_hi ? (n_hi - first - T) / stride + 1 : 0;
        i64 n = n0;
        for (i64 j = 0; j < jfull; ++j, n += stride) {
            Sample<Cfg> sm;
            const u64 index = (u64)(B * a.neval_per_block + n);
            draw_gather_phase_pipe<Cfg, EC, KV, DPC>(t, keys, stream, index, sm);
            __builtin_amdgcn_sched_barrier(0);
            draw_rest_phase<Cfg, EC, KV, DPC>(t, keys, stream, index, sm);
            process(n, sm);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (; n < n_hi; n += stride) {
            Sample<Cfg> s;
            draw_sample<Cfg, EC, KV, DPC>(t, keys, stream, (u64)(B * a.neval_per_block + n), s);
            process(n, s);
        }
    } else if constexpr (pipe_eligible<Cfg>() && !SPLIT && !EC) {
        // two samples per trip, the two pending records swapping roles: with one record the bins of the sample just drawn would be
        // copied into it on every trip (16 v_mov_b32 on the headline loop)
        PendingHist<Cfg> pa, pb; // nothing pending yet: a zero weight on bin 0
        static_for<0, Cfg::NDRAW>([&](auto K) { pa.bin[decltype(K)::value] = 0; });
        static_for<0, Cfg::NI>([&](auto I) { pa.wh[decltype(I)::value] = 0.0; });
        auto flush = [&](const PendingHist<Cfg> &q) { // a lane's last sample
            static_for<0, Cfg::NDRAW>([&](auto K) { hist_add_draw<Cfg, decltype(K)::value>(q.bin[decltype(K)::value], q.wh, sH); });
        };
        i64 n = (i64)slice * T + tid;
        for (; n + stride < a.neval_per_block; n += 2 * stride) {
            {
                Sample<Cfg> s;
                draw_sample_pipe<Cfg, KV, DPC>(keys, stream, (u64)(B * a.neval_per_block + n), s, pa, pb, sH);
                process(n, s, pb.wh);
            }
            {
                Sample<Cfg> s;
                draw_sample_pipe<Cfg, KV, DPC>(keys, stream, (u64)(B * a.neval_per_block + n + stride), s, pb, pa, sH);
                process(n + stride, s, pa.wh);
            }
        }
        if (n < a.neval_per_block) {
            Sample<Cfg> s;
            draw_sample_pipe<Cfg, KV, DPC>(keys, stream, (u64)(B * a.neval_per_block + n), s, pa, pb, sH);
            process(n, s, pb.wh);
            flush(pb);
        } else flush(pa);
    } else {
        for (i64 n = n_lo + (i64)slice * T + tid; n < n_hi; n += stride) {
            Sample<Cfg> s;
            draw_sample<Cfg, EC, KV, DPC>(t, keys, stream, (u64)(B * a.neval_per_block + n), s);
            process(n, s);
        }
    }
    };
    // MCI_MF_ONLY (set by the generated translation unit): 1 = this code object is the loop for measurefreq == 1 and nothing else (the
    // host launches it for that cadence only), 0 = the loop for any cadence; each is compiled the first time a launch needs it
#ifndef MCI_MF_ONLY
#define MCI_MF_ONLY 0
#endif
    auto run_mf = [&](auto TT) { run(TT, IC<(MCI_MF_ONLY != 0 ? 1 : 0)>{}); };
    // (one-tile kernels only: the four scalar pairs cost the split-all pass of BASELINE configs[3] seven VGPRs -- 124 -> 131, one rung of
    // its workgroup-size ladder)
    const bool stamp = !SPLIT && Cfg::NTILE == 1 && a.clock_out != nullptr && blockIdx.x == 0 && tid < 64; // (wave-uniform: scalar reads of the two counters)
    u64 ck0 = 0ull, rt0 = 0ull;
    if (stamp) {
        ck0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    if constexpr (Cfg::NTILE == 1 || SPLIT) run_mf(IC<0>{});
    else static_for<0, Cfg::NTILE>([&](auto TT) { if (tile == decltype(TT)::value) run_mf(TT); });
    if (stamp) {
        const u64 ck1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            a.clock_out[0] = ck1 - ck0;
            a.clock_out[1] = rt1 - rt0;
        }
    }
    __syncthreads();
    flush_workgroup<Cfg, L, !NOHIST>(a, smem, acc, extra, wi.rowid, tile);
}

// histogram tiles 1 .. NTILE-1 (split-all: 0 .. NTILE-1) of a SPLIT vegas pass: workgroup = (block, slice, tile); replays the parked
// (weights, bins) of the same samples its sample-pass workgroup drew -- no RNG, no gathers, no integrand:
// 8*NI + 2 bytes per tiled draw of coalesced HBM reads and one ds_add_f64 per draw.
#ifndef MCI_THREADS
#define MCI_THREADS 256 // (the JIT translation units define it: the workgroup size their kernels are compiled for)
#endif
constexpr int kTilesU = MCI_THREADS >= 768 ? 8 : 4; // replay: samples per lane and trip (measured: 8 at 768 threads 6.46 -> 6.38 ms, 16 slower; 4 at 512; with the 16-byte loads of four consecutive samples per lane: 4 | 8 | 16 at 1024 threads 4.89 | 4.90 | 5.5 ms for sample pass + replay)
// Replay with the tile's histograms BIN-MAJOR in LDS, sH[bin * G + grid], and the lanes of a wave walking the tile's G grids in
// skewed order (lane l adds to grid (d + l % G) % G at step d).  A wave's ds_add_f64 then lands on G different grids at once and
// the lanes that share a grid (64 / G of them) share only the 2 bank pairs {grid, grid + 16} of a 16-grid tile, instead of 64
// random addresses colliding all over the 32 bank pairs: the random-address atomic costs 41 ns per wave-instruction and SIMD,
// a conflict-free one 13.4 (tools/issue_microbench.hip).  The skew is a rotation of the lane's G bins by a per-lane constant:
// a 4-stage log shifter of v_cndmask_b32 (the replay has the VALU slots to spare: it is bound by the atomics and by HBM).
// Taken when the tile is G <= 16 one-draw Continuous leaves of equal size covered by the same integrands (C4: 2 tiles of 16).
template <int... D> struct ISeq {};
template <int N, int... D> struct MakeISeq : MakeISeq<N - 1, N - 1, D...> {};
template <int... D> struct MakeISeq<0, D...> { typedef ISeq<D...> type; };
// b[d] <- b[(d + SH) % G] of a G-element register vector (SSA value: no private array the compiler could index dynamically)
template <int G, int SH, class V, int... D> __device__ __forceinline__ V rotate_lanes(const V b, ISeq<D...>) {
    return __builtin_shufflevector(b, b, ((D + SH) % G)...);
}
template <class Cfg> constexpr int tile_draw_count(int tt) {
    int n = 0;
    for (int k = 0; k < Cfg::NDRAW; ++k) n += (is_tdraw<Cfg>(k) && Cfg::leaf_tile(Cfg::draw_leaf(k)) == tt) ? 1 : 0;
    return n;
}
template <class Cfg> constexpr int tile_draw(int tt, int gi) { // the gi-th replayed draw of tile tt, in draw order
    int n = 0;
    for (int k = 0; k < Cfg::NDRAW; ++k)
        if (is_tdraw<Cfg>(k) && Cfg::leaf_tile(Cfg::draw_leaf(k)) == tt) {
            if (n == gi) return k;
            ++n;
        }
    return 0;
}
template <class Cfg> constexpr bool tile_banked(int tt) {
    const int G = tile_draw_count<Cfg>(tt);
    if (G < 2 || G > 16) return false;
    const int k0 = tile_draw<Cfg>(tt, 0), NB = Cfg::leaf_nbin(Cfg::draw_leaf(k0));
    if (Cfg::tile_nbin(tt) != G * NB) return false;
    for (int gi = 0; gi < G; ++gi) {
        const int k = tile_draw<Cfg>(tt, gi), leaf = Cfg::draw_leaf(k);
        if (Cfg::leaf_kind(leaf) != 0 || Cfg::leaf_nbin(leaf) != NB || Cfg::cover_mask(k) != Cfg::cover_mask(k0)) return false;
        if (Cfg::leaf_boff(leaf) - Cfg::tile_boff(tt) != gi * NB) return false;
    }
    return true;
}
template <class Cfg> __device__ __forceinline__ void vegas_tiles(const BatchArgs &a) {
    static_assert(Cfg::HCOPY == 1, "histogram copies go with one tile");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, T = blockDim.x;
    double *sH = smem + Lds<Cfg>::H;
    constexpr int T0 = Cfg::SPLIT_ALL != 0 ? 0 : 1; // first replayed tile
    constexpr int NTM = Cfg::NTILE - T0 > 0 ? Cfg::NTILE - T0 : 1;
    // XCD-aware mapping: workgroups go round-robin to the 8 XCDs, so blockIdx % 8 is the XCD.  The NTM tile-workgroups of one
    // row sit on ONE XCD, back to back (the second finds the row's weights in that XCD's L2), and every tile is spread over all
    // XCDs (tile = blockIdx % NTM would pin each tile to a subset of the XCDs: measured 2.7 ms for tile 0 alone on 4 XCDs).
    const i64 q = (i64)blockIdx.x / 8;
    const int tile = T0 + (int)(q % NTM);
    const i64 rowid = (q / NTM) * 8 + (i64)(blockIdx.x % 8);
    const int wpb = a.tiles_wpb > 0 ? a.tiles_wpb : a.wg_per_block;
    if (rowid >= (a.tiles_rows > 0 ? a.tiles_rows : a.nrows)) return; // the grid is rounded up to a multiple of 8 * NTM
    const i64 lb = rowid / wpb;
    const int slice = (int)(rowid % wpb);
    for (int i = tid; i < Cfg::HTILE * Cfg::HCOPY; i += T) sH[i] = 0.0;
    __syncthreads();
    const i64 stride = (i64)wpb * T;
    static_for<T0, Cfg::NTILE>([&](auto TT) {
        constexpr int tt = decltype(TT)::value;
        if (tile == tt) {
            // U samples per lane and trip: all their loads are issued before the first ds_add_f64 (the kernel has one
            // 512-thread workgroup per CU, so memory-level parallelism has to come from the loop body)
            constexpr int U = kTilesU;
            constexpr int NWORD = tdraw_words<Cfg>();
            // A lane takes FOUR consecutive samples at a time: their weights and each of their bin words are then 16-byte loads (a wave
            // reads 1 KB per instruction instead of 256 B: 7 wide loads per four samples instead of 24 narrow ones), which is what a pass
            // that streams 5 GB wants.  Needs the block's first sample 16-byte aligned in both parked arrays; otherwise sample by sample.
            static_assert(U % 4 == 0, "the replay takes its samples in groups of four");
            const i64 cnt = a.chunk_hi - a.chunk_lo; // the block's parked samples (this chunk of them, BatchArgs::chunk_lo)
            const bool wide = (cnt & 3) == 0 && (a.chunk_len & 3) == 0 && (a.tile_stride & 3) == 0;
            const i64 per = wide ? 4 : 1; // consecutive samples per lane and group
            for (i64 n0 = ((i64)slice * T + tid) * per; n0 < cnt; n0 += stride * U) {
                double wh[U][Cfg::NI];
                u32 word[U][NWORD > 0 ? NWORD : 1];
                bool live[U];
                static_for<0, NWORD>([&](auto J) { // (words no draw of this tile lives in are never loaded)
                    static_for<0, U>([&](auto Uu) { word[decltype(Uu)::value][decltype(J)::value] = 0u; });
                });
                if (wide) {
                    static_for<0, U / 4>([&](auto Q) {
                        constexpr int q = decltype(Q)::value;
                        const i64 n = n0 + (i64)q * stride * 4; // this group's first sample (a multiple of four, like the block's length)
                        const bool lv = n < cnt;
                        const i64 idx = lb * a.chunk_len + (lv ? n : n0);
                        static_for<0, 4>([&](auto Vv) { live[4 * q + decltype(Vv)::value] = lv; });
                        typedef double d2 __attribute__((ext_vector_type(2)));
                        typedef u32 u4 __attribute__((ext_vector_type(4)));
                        static_for<0, Cfg::NI>([&](auto I) {
                            constexpr int i = decltype(I)::value;
                            const d2 *pw = reinterpret_cast<const d2 *>(a.tile_w + (i64)i * a.tile_stride + idx);
                            const d2 w01 = pw[0], w23 = pw[1];
                            wh[4 * q + 0][i] = w01.x;
                            wh[4 * q + 1][i] = w01.y;
                            wh[4 * q + 2][i] = w23.x;
                            wh[4 * q + 3][i] = w23.y;
                        });
                        static_for<0, NWORD>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            constexpr bool need = [] { // only the words that hold a draw of this tile
                                for (int k = 0; k < Cfg::NDRAW; ++k)
                                    if (is_tdraw<Cfg>(k) && Cfg::leaf_tile(Cfg::draw_leaf(k)) == tt && tdraw_in_word<Cfg>(tdraw_pos<Cfg>(k), j)) return true;
                                return false;
                            }();
                            if constexpr (need) {
                                const u4 wv = *reinterpret_cast<const u4 *>(a.tile_bins + (i64)j * a.tile_stride + idx);
                                word[4 * q + 0][j] = wv.x;
                                word[4 * q + 1][j] = wv.y;
                                word[4 * q + 2][j] = wv.z;
                                word[4 * q + 3][j] = wv.w;
                            }
                        });
                    });
                } else
                static_for<0, U>([&](auto Uu) {
                    constexpr int u = decltype(Uu)::value;
                    const i64 n = n0 + (i64)u * stride;
                    live[u] = n < cnt;
                    const i64 idx = lb * a.chunk_len + (live[u] ? n : n0);
                    static_for<0, Cfg::NI>([&](auto I) { wh[u][decltype(I)::value] = a.tile_w[decltype(I)::value * a.tile_stride + idx]; });
                    static_for<0, NWORD>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        // only the words that hold a draw of this tile
                        constexpr bool need = [] {
                            for (int k = 0; k < Cfg::NDRAW; ++k)
                                if (is_tdraw<Cfg>(k) && Cfg::leaf_tile(Cfg::draw_leaf(k)) == tt && tdraw_in_word<Cfg>(tdraw_pos<Cfg>(k), j)) return true;
                            return false;
                        }();
                        if constexpr (need) word[u][j] = a.tile_bins[j * a.tile_stride + idx];
                    });
                });
                if constexpr (tile_banked<Cfg>(tt)) {
                    constexpr int G = tile_draw_count<Cfg>(tt), K0 = tile_draw<Cfg>(tt, 0);
                    const int r = (tid & 63) % G; // this lane's skew
                    static_for<0, U>([&](auto Uu) {
                        constexpr int u = decltype(Uu)::value;
                        if (live[u]) {
                            double wk = 0.0;
                            static_for<0, Cfg::NI>([&](auto I) {
                                constexpr int i = decltype(I)::value;
                                if constexpr ((Cfg::own_mask(i) >> K0) & 1ull) wk += wh[u][i];
                            });
                            typedef int bvec __attribute__((ext_vector_type(G)));
                            bvec b;
                            static_for<0, G>([&](auto Gi) {
                                b[decltype(Gi)::value] = tdraw_extract<Cfg, tdraw_pos<Cfg>(tile_draw<Cfg>(tt, decltype(Gi)::value))>(word[u]);
                            });
                            static_for<0, 4>([&](auto Ss) { // b[d] <- b[(d + r) % G], one conditional rotation per bit of r
                                constexpr int sh = 1 << decltype(Ss)::value;
                                if constexpr (sh < G) {
                                    const bvec c = rotate_lanes<G, sh>(b, typename MakeISeq<G>::type{});
                                    b = (r & sh) != 0 ? c : b;
                                }
                            });
                            int g = r;
                            static_for<0, G>([&](auto D) {
                                lds_add(&sH[b[decltype(D)::value] * G + g], wk);
                                g = g + 1 == G ? 0 : g + 1;
                            });
                        }
                    });
                } else
                static_for<0, U>([&](auto Uu) {
                    constexpr int u = decltype(Uu)::value;
                    if (live[u]) {
                        static_for<0, Cfg::NDRAW>([&](auto K) {
                            constexpr int k = decltype(K)::value;
                            if constexpr (is_tdraw<Cfg>(k)) {
                                constexpr int leaf = Cfg::draw_leaf(k);
                                if constexpr (Cfg::leaf_tile(leaf) == tt) {
                                    const int bin = tdraw_extract<Cfg, tdraw_pos<Cfg>(k)>(word[u]);
                                    double wk = 0.0;
                                    static_for<0, Cfg::NI>([&](auto I) {
                                        constexpr int i = decltype(I)::value;
                                        if constexpr ((Cfg::own_mask(i) >> k) & 1ull) wk += wh[u][i];
                                    });
                                    lds_add(&sH[Cfg::leaf_boff(leaf) - Cfg::tile_boff(tt) + bin], wk);
                                }
                            }
                        });
                    }
                });
            }
            __syncthreads();
            double *hrow = a.part_hist + rowid * Cfg::NBIN + Cfg::tile_boff(tt);
            if constexpr (tile_banked<Cfg>(tt)) { // [bin][grid] in LDS -> [grid][bin] rows (coalesced stores; the strided LDS reads are a few us per tile)
                constexpr int G = tile_draw_count<Cfg>(tt), NB = Cfg::tile_nbin(tt) / G;
                for (int i = tid; i < Cfg::tile_nbin(tt); i += T) hrow[i] = (a.accum ? hrow[i] : 0.0) + sH[(i % NB) * G + i / NB];
            } else
            for (int i = tid; i < Cfg::tile_nbin(tt); i += T) hrow[i] = (a.accum ? hrow[i] : 0.0) + sH[i];
        }
    });
}

// =============================================================================================
// VegasMC: independent Metropolis chains, one per lane  (vegas_mc/montecarlo.jl:112-241,
// vegas_mc/updates.jl:45-106).  The reference runs ONE chain of neval steps per block; a block here
// is `nchain` chains of neval/nchain steps (nchain = 1 reproduces the reference's chain).  Chain state
// (x, prob, bin per draw; weights; probability) stays in registers; the proposal touches one
// (pool, slot), selected by a compile-time switch so that every table access keeps static offsets.
//   chain g = ch, the chain's index within its block; the block index is added to the stream word as block << 20
//   init  : stream MC_INIT, index g,            k = flat draw
//   step s: stream MC_STEP, index (g<<32 | s),  k = 0 pool pick, 1 slot pick, 2 accept, 3+l leaf l
// =============================================================================================
template <class Cfg> struct Chain {
    double x[Cfg::NDRAW > 0 ? Cfg::NDRAW : 1];
    double prob[Cfg::NDRAW > 0 ? Cfg::NDRAW : 1]; // leaf prob[idx]  (variable.jl:90)
    int bin[Cfg::NDRAW > 0 ? Cfg::NDRAW : 1];
};

// padding_probability(config, i) (variable.jl:628-641) and probability(config, i) (:606-619) as products
// over the draws outside / inside integrand i's dof; i == NI is the normalisation integrand (dof = 0).
template <class Cfg, int I> __device__ __forceinline__ double pad_prob(const Chain<Cfg> &c) {
    double p = 1.0;
    static_for<0, Cfg::NDRAW>([&](auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr (!((Cfg::own_mask(I) >> k) & 1ull)) p *= c.prob[k];
    });
    return p;
}
template <class Cfg, int I> __device__ __forceinline__ double own_prob(const Chain<Cfg> &c) {
    double p = 1.0;
    static_for<0, Cfg::NDRAW>([&](auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr ((Cfg::own_mask(I) >> k) & 1ull) p *= c.prob[k];
    });
    return p;
}


// Chain-state access with a RUN-TIME slot and compile-time (pool, leaf).  Written as value selects on purpose:
// stores under data-dependent branches get sunk by LLVM into a single store through a phi of addresses, which
// would move the register-resident chain state into scratch memory.
template <class Cfg, int V, int L> __device__ __forceinline__ void get_slot(const Chain<Cfg> &c, int slot, double &x, double &p, int &b) {
    constexpr int md = Cfg::pool_maxdof(V), nl = Cfg::pool_nleaf(V), k00 = Cfg::pool_first_draw(V);
    x = 0.0;
    p = 1.0;
    b = 0;
    static_for<0, md>([&](auto S) {
        constexpr int k = k00 + decltype(S)::value * nl + L;
        const bool hit = slot == decltype(S)::value;
        x = hit ? c.x[k] : x;
        p = hit ? c.prob[k] : p;
        b = hit ? c.bin[k] : b;
    });
}
template <class Cfg, int V, int L> __device__ __forceinline__ void put_slot(Chain<Cfg> &c, int slot, double x, double p, int b) {
    constexpr int md = Cfg::pool_maxdof(V), nl = Cfg::pool_nleaf(V), k00 = Cfg::pool_first_draw(V);
    static_for<0, md>([&](auto S) {
        constexpr int k = k00 + decltype(S)::value * nl + L;
        const bool hit = slot == decltype(S)::value;
        c.x[k] = hit ? x : c.x[k];
        c.prob[k] = hit ? p : c.prob[k];
        c.bin[k] = hit ? b : c.bin[k];
    });
}
// one leaf draw for pool V, leaf L (every slot of a pool shares the leaf's table): x, prob = 1/(raw*scale), bin
template <class Cfg, int V, int L> __device__ __forceinline__ void draw_pool_leaf(const Tables<Cfg> &t, double y, double &x, double &p, int &b) {
    constexpr int k = Cfg::pool_first_draw(V) + L; // slot 0 of the pool: same leaf as every other slot
    double raw;
    draw_leaf<Cfg, k>(t, y, x, raw, b);
    p = 1.0 / (raw * jac_scale<Cfg>(k));
}

// A carried chain's draw K on the CURRENT map: the bin that holds x and prob = 1/(N dx) (sampler.jl:303) | distribution[bin] (:20).
// Continuous: the largest increment whose lower edge is <= x (bisection over the leaf's table, wherever it lives); the grid's end
// points never move (variable.jl:217-218), so x stays inside.  A FermiK component has neither.
template <class Cfg, int K> __device__ __forceinline__ void relocate_draw(const Tables<Cfg> &t, const double x, double &prob, int &bin) {
    constexpr int leaf = Cfg::draw_leaf(K);
    if constexpr (Cfg::leaf_kind(leaf) == 0) {
        constexpr int N = Cfg::leaf_nbin(leaf);
        constexpr bool PAIR = Cfg::PAIR_TABLE != 0 && Cfg::TABLE_MODE <= 1;
        constexpr int off = PAIR ? Cfg::leaf_poff(leaf) : Cfg::leaf_eoff(leaf);
        int lo = 0, hi = N - 1; // invariant: edge(lo) <= x (or lo == 0), edge(hi + 1) > x (or hi == N - 1)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            const double e = PAIR ? t.E[off + 2 * mid] : t.E[off + mid];
            if (e <= x) lo = mid;
            else hi = mid - 1;
        }
        const double dx = PAIR ? t.E[off + 2 * lo + 1] : t.E[off + lo + 1] - t.E[off + lo];
        bin = lo;
        prob = 1.0 / (dx * jac_scale<Cfg>(K));
    } else if constexpr (Cfg::leaf_kind(leaf) == 1) {
        constexpr int Kn = Cfg::leaf_nbin(leaf);
        int b = (int)(x - Cfg::leaf_lower(leaf));
        b = b < 0 ? 0 : (b >= Kn ? Kn - 1 : b);
        bin = b;
        prob = 1.0 / ((1.0 / t.DD[Cfg::leaf_doff(leaf) + b]) * jac_scale<Cfg>(K)); // as create! forms it here (draw_leaf + draw_pool_leaf)
    } else {
        bin = 0;
        prob = 1.0;
    }
}
// the whole configuration of a carried chain
// (block-local index of) the stored chain that chain `ch` of local block `lb` continues
__device__ __forceinline__ i64 carried_from(const BatchArgs &a, const i64 lb, const i64 ch) {
    return a.carry_src ? (i64)a.carry_src[lb * a.nchain + ch] : ch % a.carry_nchain;
}
template <class Cfg> __device__ __forceinline__ void load_carried(const BatchArgs &a, const Tables<Cfg> &t, const i64 lb, const i64 from, Chain<Cfg> &c) {
    const i64 slot = lb * a.carry_nchain + from;
    static_for<0, Cfg::NDRAW>([&](auto K) {
        constexpr int k = decltype(K)::value;
        c.x[k] = a.carry_x[(i64)k * a.carry_cap + slot];
        relocate_draw<Cfg, k>(t, c.x[k], c.prob[k], c.bin[k]);
    });
}
template <class Cfg> __device__ __forceinline__ void store_carried(const BatchArgs &a, const i64 lb, const i64 ch, const Chain<Cfg> &c) {
    const i64 slot = lb * a.nchain + ch;
    static_for<0, Cfg::NDRAW>([&](auto K) { constexpr int k = decltype(K)::value; a.store_x[(i64)k * a.store_cap + slot] = c.x[k]; });
}

template <class Cfg> __device__ __forceinline__ void vegasmc_chains(const BatchArgs &a) {
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
    // chain identity = (block, chain within the block): the block index rides in the top 12 bits of the stream word, so the
    // streams of a block do not depend on how many chains any other block (or rank) runs
    const u32 bs = (u32)B << 20;
    const u32 st_init = a.iteration * 8u + STREAM_MC_INIT + bs, st_step = a.iteration * 8u + STREAM_MC_STEP + bs;
    const u32 k0 = (u32)a.seed, k1 = (u32)(a.seed >> 32);
    double rw[NI + 1];
    static_for<0, NI + 1>([&](auto I) { rw[decltype(I)::value] = a.reweight[decltype(I)::value]; });

    double acc[Cfg::NW];
    static_for<0, Cfg::NW>([&](auto I) { acc[decltype(I)::value] = 0.0; });
    double extra[Cfg::NCOLS - Cfg::NOBS];
    static_for<0, Cfg::NCOLS - Cfg::NOBS>([&](auto I) { extra[decltype(I)::value] = 0.0; });
    constexpr int XN = Cols<Cfg>::NORM - Cfg::NOBS, XE = Cols<Cfg>::NEVAL - Cfg::NOBS, XV = Cols<Cfg>::VISITED - Cfg::NOBS;
    u32 npr[Cfg::NPOOL], nac[Cfg::NPOOL]; // propose[2, 1, vi], accept[2, 1, vi] of the lane's current chain (vegas_mc/updates.jl:90-92)
    static_for<0, Cfg::NPOOL>([&](auto V) { npr[decltype(V)::value] = 0u; nac[decltype(V)::value] = 0u; });
    auto flush_pa = [&]() { // lane counters -> the workgroup's 64-bit table in LDS (once per chain: off the step loop)
        static_for<0, Cfg::NPOOL>([&](auto V) {
            constexpr int v = decltype(V)::value;
            if (npr[v]) lds_count(&sPA[PaTable<Cfg>::idx(1, 0, v)], (u64)npr[v]);
            if (nac[v]) lds_count(&sPA[PaTable<Cfg>::N + PaTable<Cfg>::idx(1, 0, v)], (u64)nac[v]);
            npr[v] = 0u;
            nac[v] = 0u;
        });
    };

    for (i64 ch = (i64)slice * T + tid; ch < a.nchain; ch += (i64)a.wg_per_block * T) {
        const u64 g = (u64)ch;
        Chain<Cfg> c;
        if (a.carry_x) load_carried<Cfg>(a, t, wi.lb, carried_from(a, wi.lb, ch), c); // continues the previous iteration's chain (BatchArgs::carry_x)
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

        i64 mcnt = 0, mj = 0; // ne % measurefreq and ne / measurefreq, carried (no 64-bit division per step)
        for (i64 ne = 1; ne <= steps; ++ne) { // :184
            const u64 sidx = (g << 32) | (u64)(ne - 1);
            const u32x4 r0 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 0u, st_step, k0, k1);
            const u32x4 r1 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 1u, st_step, k0, k1);
            // ---- changeVariable  updates.jl:45-106 ----
            // :50 rand(1:Nv).  With many chains per block the 64 chains of a wave share the pool-pick sequence (it does
            // not depend on the chain states): the pool dispatch below becomes a scalar branch
            double upool = u01(r0.x, r0.y);
            if (Cfg::NPOOL > 1 && a.nchain > 1) {
                const u64 gidx = ((u64)(ch & ~(i64)63) << 32) | (u64)(ne - 1);
                const u32x4 rg = philox4x32_10((u32)gidx, (u32)(gidx >> 32), 0u, a.iteration * 8u + STREAM_MC_GROUP + bs, k0, k1);
                upool = u01(rg.x, rg.y);
            }
            int vi = (int)(upool * (double)Cfg::NPOOL);
            if (vi >= Cfg::NPOOL) vi = Cfg::NPOOL - 1;
            if (Cfg::NPOOL > 1 && a.nchain > 1) vi = __builtin_amdgcn_readfirstlane(vi);
            const double uslot = u01(r0.z, r0.w);
            const double uacc = u01(r1.x, r1.y);
            Chain<Cfg> n = c; // proposal; unchanged draws are copy-propagated
            double prop = 1.0;
            bool active = false;
            static_for<0, Cfg::NPOOL>([&](auto V) {
                constexpr int v = decltype(V)::value;
                constexpr int md = Cfg::pool_maxdof(v), nl = Cfg::pool_nleaf(v), k00 = Cfg::pool_first_draw(v);
                // :52-57  a lone single-valued Discrete, or a pool nobody uses, has nothing to sample
                constexpr bool skip = (md <= 0) || (nl == 1 && Cfg::leaf_kind(Cfg::draw_leaf(md > 0 ? k00 : 0)) == 1 &&
                                                    Cfg::leaf_nbin(Cfg::draw_leaf(md > 0 ? k00 : 0)) == 1);
                if constexpr (!skip) {
                    if (vi == v) {
                        active = true;
                        int slot = (int)(uslot * (double)md); // :58
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
                            double xo, po, xn, pn;
                            int bo, bn;
                            get_slot<Cfg, v, l>(c, slot, xo, po, bo);
                            draw_pool_leaf<Cfg, v, l>(t, y, xn, pn, bn); // shift!  sampler.jl:336-386, :57-71
                            put_slot<Cfg, v, l>(n, slot, xn, pn, bn);
                            prop *= po / pn;                              // 1/prob_ratio  sampler.jl:385, :70
                        });
                    }
                }
            });
            if (active && prop > 4.9406564584124654e-324) { // :63-65
                double wn[Cfg::NW], padn[NI + 1];
                Cfg::integrand(n.x, wn, a.ud, -1);             // :67-75
                extra[XE] += 1.0;                              // config.neval += 1   :77
                static_for<0, NI + 1>([&](auto I) { padn[decltype(I)::value] = pad_prob<Cfg, decltype(I)::value>(n); }); // :79-81
                double newp = rw[NORMI] * padn[NORMI];         // :84
                static_for<0, NI>([&](auto I) { constexpr int i = decltype(I)::value; newp += absw<Cfg, i>(wn) * rw[i] * padn[i]; }); // :85-87
                const double R = prop * newp / probability;    // :88
                const bool ok = uacc < R;                      // :91
                static_for<0, Cfg::NPOOL>([&](auto V) { // (vi is wave-uniform with many chains per block: one scalar branch is taken)
                    constexpr int v = decltype(V)::value;
                    if (vi == v) {
                        npr[v] += 1u;                          // :90
                        nac[v] += ok ? 1u : 0u;                // :92
                    }
                });
                if (ok) {
                    c = n;
                    static_for<0, Cfg::NW>([&](auto I) { w[decltype(I)::value] = wn[decltype(I)::value]; }); // :93-95
                    static_for<0, NI + 1>([&](auto I) { pad[decltype(I)::value] = padn[decltype(I)::value]; }); // :96-98
                    probability = newp;                        // :100
                } // else shiftRollback!  :102  (the proposal copy is dropped)
            }
            // ---- histogram  montecarlo.jl:198-211 ----
            {
                double wh[NI];
                static_for<0, NI>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    const double aw = absw<Cfg, i>(w);
                    const double f2 = aw * aw / own_prob<Cfg, i>(c);                   // :203
                    wh[i] = f2 * pad[i] / probability;                                 // :204
                });
                Sample<Cfg> sb;
                static_for<0, Cfg::NDRAW>([&](auto K) { sb.bin[decltype(K)::value] = c.bin[decltype(K)::value]; });
                hist_update<Cfg>(sb, wh, sH, a.ghist, tile);
            }
            if ((ne & 0x3FFFFFFF) == 0) flush_pa(); // (32-bit lane counters: hand over long before they wrap)
            // ---- measurement  montecarlo.jl:213-232 ----
            mcnt = mcnt + 1 == a.measurefreq ? 0 : mcnt + 1;
            const bool mf = mcnt == 0;
            mj += mf ? 1 : 0;
            if (mf && (double)ne >= a.burnin) { // :213
                double relw[Cfg::NW];
                static_for<0, NI>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    extra[XV + i] += absw<Cfg, i>(w) * fabs(pad[i] * rw[i]) / probability; // :216
                    static_for<0, Cfg::NCOMP>([&](auto Q) {
                        constexpr int q = i * Cfg::NCOMP + decltype(Q)::value;
                        relw[q] = w[q] * pad[i] / probability;                             // :218/:220
                    });
                });
                if constexpr (Cfg::HOST_MEASURE != 0) { // :224-227 on the host, after the launch
                    if (tile == 0) host_measure_record<Cfg, Cfg::NW>(a, wi.lb, ch, mj, c.x, relw, -1);
                } else
                measure<Cfg>(c.x, c.bin, relw, a.ud, acc, obs_wave<Cfg>(sO));
                extra[XN] += pad[NORMI] / probability;                // :229
                extra[XV + NORMI] += rw[NORMI] * pad[NORMI] / probability; // :230
            }
        }
        flush_pa();
        if (a.store_x && tile == 0) {
            store_carried<Cfg>(a, wi.lb, ch, c);
            if (a.store_P) a.store_P[wi.lb * a.nchain + ch] = probability; // the target at the configuration the chain stopped at
        }
    }
    __syncthreads();
    flush_workgroup<Cfg, Lds<Cfg>, true, true>(a, smem, acc, extra, wi.rowid, tile);
}

// The NEW target at every stored configuration over the old one (BatchArgs::carry_w): one lane per stored chain; the same arithmetic in
// the same order as a chain's start (montecarlo.jl:155-166) on the refined map and the moved reweight factors.
template <class Cfg> __device__ __forceinline__ void vegasmc_carry_weights(const BatchArgs &a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NI = Cfg::NI, NORMI = Cfg::NI;
    const int tid = threadIdx.x, T = blockDim.x;
    double *sE = smem + Lds<Cfg>::E, *sDA = smem + Lds<Cfg>::DA, *sDD = smem + Lds<Cfg>::DD;
    stage_tables<Cfg>(a.edges, a.dacc, a.ddist, sE, sDA, sDD);
    __syncthreads();
    Tables<Cfg> t;
    if constexpr (Mode<Cfg>::EDGE_LDS) t.E = sE;
    else t.E = a.edges;
    t.DA = sDA;
    t.DD = sDD;
    double rw[NI + 1];
    static_for<0, NI + 1>([&](auto I) { rw[decltype(I)::value] = a.reweight[decltype(I)::value]; });
    for (i64 j = (i64)blockIdx.x * T + tid; j < a.carry_total; j += (i64)gridDim.x * T) {
        const i64 lb = j / a.carry_nchain, from = j % a.carry_nchain;
        Chain<Cfg> c;
        load_carried<Cfg>(a, t, lb, from, c);
        double w[Cfg::NW], pad[NI + 1];
        if constexpr (Cfg::HOST_INTEGRAND != 0) // (a host closure: evaluated at the stored configurations before this launch, host_w[q * carry_total + j])
            static_for<0, Cfg::NW>([&](auto Q) { w[decltype(Q)::value] = a.host_w[decltype(Q)::value * a.carry_total + j]; });
        else
        Cfg::integrand(c.x, w, a.ud, -1);
        static_for<0, NI + 1>([&](auto I) { pad[decltype(I)::value] = pad_prob<Cfg, decltype(I)::value>(c); });
        double probability = rw[NORMI] * pad[NORMI];
        static_for<0, NI>([&](auto I) { constexpr int i = decltype(I)::value; probability += absw<Cfg, i>(w) * rw[i] * pad[i]; });
        const double r = probability / a.carry_P[lb * a.carry_nchain + from];
        a.carry_w[j] = (r == r && r < 1.7976931348623157e308 && r > 0.0) ? r : 0.0; // (a configuration neither target can have produced continues nothing)
    }
}

// ---------------------------------------------------------------------------------------------
// VegasMC with the integrand on the HOST (BatchArgs::HostStep): the step of vegasmc_chains cut at the integrand call.  Every
// launch finishes the step whose weights just came back and proposes the next one; the same Philox streams, the same
// arithmetic in the same order as vegasmc_chains, so a closure and the same function as device source give the same chains.
// Partial rows are ADDED to (the host zeroes them before the first launch).  One histogram tile only.
// ---------------------------------------------------------------------------------------------
template <class Cfg> __device__ __forceinline__ void vegasmc_host_step(const BatchArgs &a) {
    static_assert(Cfg::NTILE == 1, "host-closure chains keep one histogram tile");
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
    const BatchArgs::HostStep &h = a.hs;
    const i64 B = a.block_lo + wi.lb;
    const u32 bs = (u32)B << 20;
    const u32 st_init = a.iteration * 8u + STREAM_MC_INIT + bs, st_step = a.iteration * 8u + STREAM_MC_STEP + bs;
    const u32 k0 = (u32)a.seed, k1 = (u32)(a.seed >> 32);
    double rw[NI + 1];
    static_for<0, NI + 1>([&](auto I) { rw[decltype(I)::value] = a.reweight[decltype(I)::value]; });
    double acc[Cfg::NW];
    static_for<0, Cfg::NW>([&](auto I) { acc[decltype(I)::value] = 0.0; });
    double extra[Cfg::NCOLS - Cfg::NOBS];
    static_for<0, Cfg::NCOLS - Cfg::NOBS>([&](auto I) { extra[decltype(I)::value] = 0.0; });
    constexpr int XN = Cols<Cfg>::NORM - Cfg::NOBS, XE = Cols<Cfg>::NEVAL - Cfg::NOBS, XV = Cols<Cfg>::VISITED - Cfg::NOBS;
    const i64 nc = h.nc;

    for (i64 ch = (i64)wi.slice * T + tid; ch < a.nchain; ch += (i64)a.wg_per_block * T) {
        const u64 g = (u64)ch;
        const i64 cid = wi.lb * a.nchain + ch; // the chain's column in the state arrays
        Chain<Cfg> c;
        if (h.ne == 0) { // initialize!  (montecarlo.jl:151-153): create! on every live slot; the host evaluates it (:155-159)
            if (a.carry_x) load_carried<Cfg>(a, t, wi.lb, carried_from(a, wi.lb, ch), c); // ... or the previous iteration's chain goes on (BatchArgs::carry_x)
            else {
                Sample<Cfg> s;
                draw_sample<Cfg>(t, a.seed, st_init, g, s);
                static_for<0, Cfg::NDRAW>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    c.x[k] = s.x[k];
                    c.bin[k] = s.bin[k];
                    c.prob[k] = 1.0 / s.pj[k]; // sampler.jl:303 / :20
                });
            }
            static_for<0, Cfg::NDRAW>([&](auto K) {
                constexpr int k = decltype(K)::value;
                h.cx[k * nc + cid] = c.x[k];
                h.hx[k * nc + cid] = c.x[k];
                h.cbin[k * nc + cid] = c.bin[k];
                h.cprob[k * nc + cid] = c.prob[k];
            });
            continue;
        }
        static_for<0, Cfg::NDRAW>([&](auto K) {
            constexpr int k = decltype(K)::value;
            c.x[k] = h.cx[k * nc + cid];
            c.prob[k] = h.cprob[k * nc + cid];
            c.bin[k] = h.cbin[k * nc + cid];
        });
        double w[Cfg::NW], pad[NI + 1], probability;
        static_for<0, NI + 1>([&](auto I) { pad[decltype(I)::value] = pad_prob<Cfg, decltype(I)::value>(c); }); // :161
        if (h.ne == 1) { // the weights of the initial configuration: config.probability  (:162-166)
            static_for<0, Cfg::NW>([&](auto Q) { w[decltype(Q)::value] = a.host_w[decltype(Q)::value * nc + cid]; });
            probability = rw[NORMI] * pad[NORMI];
            static_for<0, NI>([&](auto I) { constexpr int i = decltype(I)::value; probability += absw<Cfg, i>(w) * rw[i] * pad[i]; });
        } else {
            static_for<0, Cfg::NW>([&](auto Q) { w[decltype(Q)::value] = h.cw[decltype(Q)::value * nc + cid]; });
            probability = h.cprobability[cid];
            const i64 ne = h.ne - 1; // the step being finished
            const int vi = h.pvi[cid];
            const double prop = h.pprop[cid];
            if (vi >= 0 && prop > 4.9406564584124654e-324) { // :63-65
                Chain<Cfg> n;
                static_for<0, Cfg::NDRAW>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    n.x[k] = h.hx[k * nc + cid];
                    n.prob[k] = h.pprob[k * nc + cid];
                    n.bin[k] = h.pbin[k * nc + cid];
                });
                double wn[Cfg::NW], padn[NI + 1];
                static_for<0, Cfg::NW>([&](auto Q) { wn[decltype(Q)::value] = a.host_w[decltype(Q)::value * nc + cid]; }); // :67-75, on the host
                extra[XE] += 1.0;                              // config.neval += 1   :77
                static_for<0, NI + 1>([&](auto I) { padn[decltype(I)::value] = pad_prob<Cfg, decltype(I)::value>(n); }); // :79-81
                double newp = rw[NORMI] * padn[NORMI];         // :84
                static_for<0, NI>([&](auto I) { constexpr int i = decltype(I)::value; newp += absw<Cfg, i>(wn) * rw[i] * padn[i]; }); // :85-87
                const double R = prop * newp / probability;    // :88
                const bool ok = h.puacc[cid] < R;              // :91
                lds_count(&sPA[PaTable<Cfg>::idx(1, 0, vi)], 1ull);                              // :90
                if (ok) lds_count(&sPA[PaTable<Cfg>::N + PaTable<Cfg>::idx(1, 0, vi)], 1ull);    // :92
                if (ok) {
                    c = n;
                    static_for<0, Cfg::NW>([&](auto I) { w[decltype(I)::value] = wn[decltype(I)::value]; }); // :93-95
                    static_for<0, NI + 1>([&](auto I) { pad[decltype(I)::value] = padn[decltype(I)::value]; }); // :96-98
                    probability = newp;                        // :100
                } // else shiftRollback!  :102
            }
            {   // histogram  montecarlo.jl:198-211
                double wh[NI];
                static_for<0, NI>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    const double aw = absw<Cfg, i>(w);
                    const double f2 = aw * aw / own_prob<Cfg, i>(c);                   // :203
                    wh[i] = f2 * pad[i] / probability;                                 // :204
                });
                Sample<Cfg> sb;
                static_for<0, Cfg::NDRAW>([&](auto K) { sb.bin[decltype(K)::value] = c.bin[decltype(K)::value]; });
                hist_update<Cfg>(sb, wh, sH, a.ghist, 0);
            }
            if (ne % a.measurefreq == 0 && (double)ne >= a.burnin) { // measurement  :213-232
                double relw[Cfg::NW];
                static_for<0, NI>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    extra[XV + i] += absw<Cfg, i>(w) * fabs(pad[i] * rw[i]) / probability; // :216
                    static_for<0, Cfg::NCOMP>([&](auto Q) {
                        constexpr int q = i * Cfg::NCOMP + decltype(Q)::value;
                        relw[q] = w[q] * pad[i] / probability;                             // :218/:220
                    });
                });
                if constexpr (Cfg::HOST_MEASURE != 0) host_measure_record<Cfg, Cfg::NW>(a, wi.lb, ch, ne / a.measurefreq, c.x, relw, -1);
                else measure<Cfg>(c.x, c.bin, relw, a.ud, acc, obs_wave<Cfg>(sO));
                extra[XN] += pad[NORMI] / probability;                // :229
                extra[XV + NORMI] += rw[NORMI] * pad[NORMI] / probability; // :230
            }
        }
        if (h.ne <= h.steps) { // ---- changeVariable, up to the integrand call  updates.jl:45-66 ----
            const i64 ne = h.ne;
            const u64 sidx = (g << 32) | (u64)(ne - 1);
            const u32x4 r0 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 0u, st_step, k0, k1);
            const u32x4 r1 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 1u, st_step, k0, k1);
            double upool = u01(r0.x, r0.y);
            if (Cfg::NPOOL > 1 && a.nchain > 1) { // (the 64 chains of a wave share the pool pick, as in vegasmc_chains)
                const u64 gidx = ((u64)(ch & ~(i64)63) << 32) | (u64)(ne - 1);
                const u32x4 rg = philox4x32_10((u32)gidx, (u32)(gidx >> 32), 0u, a.iteration * 8u + STREAM_MC_GROUP + bs, k0, k1);
                upool = u01(rg.x, rg.y);
            }
            int vi = (int)(upool * (double)Cfg::NPOOL);
            if (vi >= Cfg::NPOOL) vi = Cfg::NPOOL - 1;
            const double uslot = u01(r0.z, r0.w);
            Chain<Cfg> n = c;
            double prop = 1.0;
            bool active = false;
            static_for<0, Cfg::NPOOL>([&](auto V) {
                constexpr int v = decltype(V)::value;
                constexpr int md = Cfg::pool_maxdof(v), nl = Cfg::pool_nleaf(v), k00 = Cfg::pool_first_draw(v);
                constexpr bool skip = (md <= 0) || (nl == 1 && Cfg::leaf_kind(Cfg::draw_leaf(md > 0 ? k00 : 0)) == 1 &&
                                                    Cfg::leaf_nbin(Cfg::draw_leaf(md > 0 ? k00 : 0)) == 1);
                if constexpr (!skip) {
                    if (vi == v) {
                        active = true;
                        int slot = (int)(uslot * (double)md); // :58
                        if (slot >= md) slot = md - 1;
                        static_for<0, nl>([&](auto Lf) {
                            constexpr int l = decltype(Lf)::value;
                            constexpr int kk = 3 + l;
                            double y;
                            if constexpr (kk == 3) y = u01(r1.z, r1.w);
                            else {
                                const u32x4 rr = philox4x32_10((u32)sidx, (u32)(sidx >> 32), (u32)(kk >> 1), st_step, k0, k1);
                                y = (kk & 1) ? u01(rr.z, rr.w) : u01(rr.x, rr.y);
                            }
                            double xo, po, xn, pn;
                            int bo, bn;
                            get_slot<Cfg, v, l>(c, slot, xo, po, bo);
                            draw_pool_leaf<Cfg, v, l>(t, y, xn, pn, bn); // shift!  sampler.jl:336-386, :57-71
                            put_slot<Cfg, v, l>(n, slot, xn, pn, bn);
                            prop *= po / pn;                              // 1/prob_ratio  sampler.jl:385, :70
                        });
                    }
                }
            });
            static_for<0, Cfg::NDRAW>([&](auto K) {
                constexpr int k = decltype(K)::value;
                h.hx[k * nc + cid] = n.x[k];
                h.pprob[k * nc + cid] = n.prob[k];
                h.pbin[k * nc + cid] = n.bin[k];
            });
            h.pprop[cid] = prop;
            h.puacc[cid] = u01(r1.x, r1.y);
            h.pvi[cid] = active ? vi : -1;
        }
        static_for<0, Cfg::NDRAW>([&](auto K) {
            constexpr int k = decltype(K)::value;
            h.cx[k * nc + cid] = c.x[k];
            h.cprob[k * nc + cid] = c.prob[k];
            h.cbin[k * nc + cid] = c.bin[k];
        });
        static_for<0, Cfg::NW>([&](auto Q) { h.cw[decltype(Q)::value * nc + cid] = w[decltype(Q)::value]; });
        h.cprobability[cid] = probability;
        if (a.store_x && h.ne == h.steps + 1) { // the chain's last step is through
            store_carried<Cfg>(a, wi.lb, ch, c);
            if (a.store_P) a.store_P[wi.lb * a.nchain + ch] = probability;
        }
    }
    __syncthreads();
    flush_workgroup<Cfg, Lds<Cfg>, true, true, true>(a, smem, acc, extra, wi.rowid, 0);
}

// =============================================================================================
// MCMC: Metropolis chains over (integrand index, live variables), one chain per lane
// (mcmc/montecarlo.jl:72-184, mcmc/updates.jl:1-147).  The reference runs ONE chain of neval (+ burn-in)
// steps per block; a block here is `nchain` chains of neval/nchain measured steps, each after its own
// burn-in (nchain = 1 reproduces the reference's chain).  Only the integrand the chain sits on is
// evaluated per step (the integrand body sees `idx`); the neighbor graph (configuration.jl:201-227) and
// the dof table are compile-time, so every register-array access keeps a static index.
//   chain g = ch, the chain's index within its block; the block index is added to the stream word as block << 20
//   init try t: stream MCMC_INIT, index g*16384 + t,  k = flat draw
//   step s    : stream MCMC_STEP, index (g<<32 | s),  k = 0 update pick, 1 neighbor/pool pick, 2 slot pick,
//               3 second slot pick (swap), 4 accept, 5 + flat draw index of a created/shifted draw
// =============================================================================================
// ---------------------------------------------------------------------------------------------
// FermiK{D} (variable.jl:1-20, sampler.jl:109-281): a momentum on a shell |k| in (kF - dk, kF + dk), D = 2 | 3
// components per slot, no adaptive map, :mcmc only.  kF = leaf_lower, dk = leaf_upper, D = pool_nleaf(V).
// ---------------------------------------------------------------------------------------------
#define MCI_PI 3.14159265358979323846
// create!  sampler.jl:109-148.  u = D uniforms; returns the proposal weight (0: rejected, k untouched)
template <class Cfg, int V> __device__ __forceinline__ double fermik_create(const double *u, double *k) {
    constexpr int D = Cfg::pool_nleaf(V), leaf = Cfg::draw_leaf(Cfg::pool_first_draw(V));
    constexpr double kF = Cfg::leaf_lower(leaf), dk = Cfg::leaf_upper(leaf);
    const double Kamp = kF + (u[0] - 0.5) * 2.0 * dk; // :121
    if (Kamp <= 0.0) return 0.0;                       // :122
    const double phi = 2.0 * MCI_PI * u[1];            // :124
    if constexpr (D == 3) {
        const double theta = MCI_PI * u[2];            // :126
        k[0] = Kamp * cos(phi) * sin(theta);           // :129-131
        k[1] = Kamp * sin(phi) * sin(theta);
        k[2] = Kamp * cos(theta);
        return 2 * dk * 2 * MCI_PI * MCI_PI * (sin(theta) * Kamp * Kamp); // :132
    } else {
        k[0] = Kamp * cos(phi);                        // :139-140
        k[1] = Kamp * sin(phi);
        return 2 * dk * 2 * MCI_PI * Kamp;             // :141
    }
}
// remove!  sampler.jl:158-188
template <class Cfg, int V> __device__ __forceinline__ double fermik_remove(const double *k) {
    constexpr int D = Cfg::pool_nleaf(V), leaf = Cfg::draw_leaf(Cfg::pool_first_draw(V));
    constexpr double kF = Cfg::leaf_lower(leaf), dk = Cfg::leaf_upper(leaf);
    double k2 = 0.0;
    static_for<0, D>([&](auto J) { k2 += k[decltype(J)::value] * k[decltype(J)::value]; });
    const double Kamp = sqrt(k2);                                  // :171
    if (!(kF - dk < Kamp && Kamp < kF + dk)) return 0.0;           // :172-174
    if constexpr (D == 3) {
        const double sint = sqrt(k[0] * k[0] + k[1] * k[1]) / Kamp; // :177
        if (sint < 1.0e-15) return 0.0;                             // :178
        return 1.0 / (2 * dk * 2 * MCI_PI * MCI_PI * sint * Kamp * Kamp); // :179
    } else {
        return 1.0 / (2 * dk * 2 * MCI_PI * Kamp);                  // :183
    }
}
// shift!  sampler.jl:198-246: scale | rotate | shift, picked by upick; u = up to D more uniforms; k is updated in place
template <class Cfg, int V> __device__ __forceinline__ double fermik_shift(double upick, const double *u, double *k) {
    constexpr int D = Cfg::pool_nleaf(V), leaf = Cfg::draw_leaf(Cfg::pool_first_draw(V));
    constexpr double dk = Cfg::leaf_upper(leaf);
    if (upick < 1.0 / 3) { // :206-212
        const double lambda = 1.5;
        const double ratio = 1.0 / lambda + u[0] * (lambda - 1.0 / lambda);
        static_for<0, D>([&](auto J) { k[decltype(J)::value] *= ratio; });
        return D == 2 ? 1.0 : ratio;
    } else if (upick < 2.0 / 3) { // :213-229
        const double phi = u[0] * 2.0 * MCI_PI;
        if constexpr (D == 3) {
            const double theta = acos(1.0 - 2.0 * u[1]);
            const double Kamp = sqrt(k[0] * k[0] + k[1] * k[1] + k[2] * k[2]);
            k[0] = Kamp * cos(phi) * sin(theta);
            k[1] = Kamp * sin(phi) * sin(theta);
            k[2] = Kamp * cos(theta);
        } else {
            const double Kamp = sqrt(k[0] * k[0] + k[1] * k[1]);
            k[0] = Kamp * cos(phi);
            k[1] = Kamp * sin(phi);
        }
        return 1.0;
    }
    static_for<0, D>([&](auto J) { k[decltype(J)::value] += (u[decltype(J)::value] - 0.5) * dk; }); // :231-243
    return 1.0;
}
template <class Cfg> constexpr bool pool_is_fermik(int v) {
    return Cfg::pool_maxdof(v) > 0 && Cfg::leaf_kind(Cfg::draw_leaf(Cfg::pool_first_draw(v))) == 2;
}

// weight of ONE integrand: value (re [, im]) and modulus
template <class Cfg> struct Weight {
    double v[Cfg::NCOMP];
    double abs;
};
template <class Cfg, int I> __device__ __forceinline__ Weight<Cfg> eval_one(const double *x, const double *ud) {
    double w[Cfg::NW];
    Cfg::integrand(x, w, ud, I); // the other outputs are dead code after inlining
    Weight<Cfg> r;
    static_for<0, Cfg::NCOMP>([&](auto Q) { r.v[decltype(Q)::value] = w[I * Cfg::NCOMP + decltype(Q)::value]; });
    r.abs = absw<Cfg, I>(w);
    return r;
}
template <class Cfg> __device__ __forceinline__ Weight<Cfg> eval_sel(int curr, const double *x, const double *ud) {
    Weight<Cfg> r;
    static_for<0, Cfg::NCOMP>([&](auto Q) { r.v[decltype(Q)::value] = 0.0; });
    r.abs = 0.0;
    static_for<0, Cfg::NI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if (curr == i) r = eval_one<Cfg, i>(x, ud);
    });
    return r;
}
// uniform k of a chain step; chunk 2 (k = 4, 5) is shared with the accept draw
template <int K> __device__ __forceinline__ double step_uniform(u64 sidx, u32 stream, u32 k0, u32 k1, const u32x4 &r2) {
    if constexpr ((K >> 1) == 2) return (K & 1) ? u01(r2.z, r2.w) : u01(r2.x, r2.y);
    else {
        const u32x4 r = philox4x32_10((u32)sidx, (u32)(sidx >> 32), (u32)(K >> 1), stream, k0, k1);
        return (K & 1) ? u01(r.z, r.w) : u01(r.x, r.y);
    }
}
// the same with a run-time k (k >= 5: the shifted slot of changeVariable is a run-time value)
__device__ __forceinline__ double step_uniform_dyn(int k, u64 sidx, u32 stream, u32 k0, u32 k1) {
    const u32x4 r = philox4x32_10((u32)sidx, (u32)(sidx >> 32), (u32)(k >> 1), stream, k0, k1);
    return (k & 1) ? u01(r.z, r.w) : u01(r.x, r.y);
}
// histogram add of one draw with the table-mode dispatch of hist_update
template <class Cfg, int K> __device__ __forceinline__ void hist_add(int bin, double wk, double *sH, double *gH, int tile) {
    constexpr int leaf = Cfg::draw_leaf(K);
    if constexpr (Cfg::leaf_adapt(leaf) != 0) {
        if constexpr (Mode<Cfg>::HIST_LDS) {
            constexpr int lt = Cfg::leaf_tile(leaf);
            if (Cfg::NTILE == 1 || tile == lt) lds_add(&sH[hslot<Cfg>(Cfg::leaf_boff(leaf) - Cfg::tile_boff(lt) + bin)], wk);
        } else {
            global_add(&gH[Cfg::leaf_boff(leaf) + bin], wk);
        }
    }
}

// One MCMC proposal (mcmc/updates.jl:1-147): from the chain's configuration c on integrand curr and the step's update type and
// uniforms, the proposed configuration n, the proposal ratio, the integrand it lands on, and what the bookkeeping needs
// (ut: first index of propose[., ., .] -- 0 changeIntegrand, 1 changeVariable, 2 swapVariable; pvi: the pool picked; touched: the
// draws of the slot(s) it moves).  Shared by mcmc_chains and its host-closure form mcmc_host_step.
template <class Cfg> struct McmcProposal {
    Chain<Cfg> n;
    double prop;
    bool active;
    int newcurr, ut, pvi;
    u64 touched;
};
template <class Cfg> __device__ __forceinline__ McmcProposal<Cfg> mcmc_propose(const Tables<Cfg> &t, const Chain<Cfg> &c, const int curr, const int upd, const double upick,
                                                                              const double us1, const double us2, const u64 sidx, const u32 st_step,
                                                                              const u32 k0, const u32 k1, const u32x4 &r2) {
    constexpr int NI = Cfg::NI, NORMI = Cfg::NI, ND = Cfg::NI + 1, NPOOL = Cfg::NPOOL;
    (void)NI;
    // ---- build the proposal (n, prop, newcurr); ONE evaluate-and-accept tail serves all three updates, so
    // lanes that diverged on the update type reconverge before the expensive part ----
    Chain<Cfg> n = c;
    double prop = 1.0;
    bool active = false;
    int newcurr = curr, ut = 0; // ut: first index of propose[., ., .]: 0 changeIntegrand, 1 changeVariable, 2 swapVariable
    int pvi = 0;                // the variable pool changeVariable / swapVariable picked (last index of propose)
    u64 touched = 0ull; // draws of the slot(s) this proposal moves (changeVariable, swapVariable)
    if (upd == 0) {
        // ---- changeIntegrand  updates.jl:1-69 ----
        static_for<0, ND>([&](auto C0) {
            constexpr int c0 = decltype(C0)::value;
            constexpr int nn = Cfg::nneighbor(c0);
            if (curr == c0) {
                int j = (int)(upick * (double)nn); // :6
                if (j >= nn) j = nn - 1;
                static_for<0, nn>([&](auto J) {
                    constexpr int nw = Cfg::neighbor(c0 * Cfg::NBMAX + decltype(J)::value);
                    if constexpr (nw != c0) { // :7
                        if (j == decltype(J)::value) {
                            active = true;
                            newcurr = nw;
                            prop = (double)nn / (double)Cfg::nneighbor(nw); // :12
                            static_for<0, NPOOL>([&](auto V) { // :15-26
                                constexpr int v = decltype(V)::value;
                                constexpr int cd = Cfg::dof(c0 * NPOOL + v), nd = Cfg::dof(nw * NPOOL + v);
                                constexpr int nl = Cfg::pool_nleaf(v), k00 = Cfg::pool_first_draw(v);
                                if constexpr (pool_is_fermik<Cfg>(v) && cd != nd) {
                                    static_for<(cd < nd ? cd : nd), (cd < nd ? nd : cd)>([&](auto S) {
                                        constexpr int kb = k00 + decltype(S)::value * nl;
                                        double kk[nl];
                                        static_for<0, nl>([&](auto J) { kk[decltype(J)::value] = c.x[kb + decltype(J)::value]; });
                                        if constexpr (cd < nd) { // create!  sampler.jl:109-148
                                            double u[nl];
                                            static_for<0, nl>([&](auto J) { u[decltype(J)::value] = step_uniform<5 + kb + decltype(J)::value>(sidx, st_step, k0, k1, r2); });
                                            prop *= fermik_create<Cfg, v>(u, kk);
                                            static_for<0, nl>([&](auto J) { n.x[kb + decltype(J)::value] = kk[decltype(J)::value]; });
                                        } else {                 // remove!  sampler.jl:158-188
                                            prop *= fermik_remove<Cfg, v>(kk);
                                        }
                                    });
                                } else if constexpr (cd < nd) {
                                    static_for<cd * nl, nd * nl>([&](auto Q) { // create!  sampler.jl:293-305, :13-22
                                        constexpr int k = k00 + decltype(Q)::value;
                                        const double y = step_uniform<5 + k>(sidx, st_step, k0, k1, r2);
                                        double raw;
                                        draw_leaf<Cfg, k>(t, y, n.x[k], raw, n.bin[k]);
                                        const double ip = raw * jac_scale<Cfg>(k);
                                        n.prob[k] = 1.0 / ip;
                                        prop *= ip;
                                    });
                                } else if constexpr (cd > nd) {
                                    static_for<nd * nl, cd * nl>([&](auto Q) { // remove!  sampler.jl:318-323, :36-40
                                        prop *= c.prob[k00 + decltype(Q)::value];
                                    });
                                }
                            });
                        }
                    }
                });
            }
        });
    } else if (curr != NORMI) { // updates.jl:73, :115
        int vi = (int)(upick * (double)NPOOL); // :77, :119
        if (vi >= NPOOL) vi = NPOOL - 1;
        pvi = vi;
        int cdv = 0; // currdof[vi]
        static_for<0, NI>([&](auto I) {
            static_for<0, NPOOL>([&](auto V) {
                if (curr == decltype(I)::value && vi == decltype(V)::value) cdv = Cfg::dof(decltype(I)::value * NPOOL + decltype(V)::value);
            });
        });
        if (upd == 1) {
            // ---- swapVariable  updates.jl:113-147 ----
            ut = 2;
            if (cdv > 0) { // :121
                int s1 = (int)(us1 * (double)cdv), s2 = (int)(us2 * (double)cdv); // :122-123
                if (s1 >= cdv) s1 = cdv - 1;
                if (s2 >= cdv) s2 = cdv - 1;
                if (s1 != s2) { // :124
                    active = true;
                    static_for<0, NPOOL>([&](auto V) {
                        constexpr int v = decltype(V)::value;
                        if (vi == v) {
                            constexpr u64 slotbits = (1ull << Cfg::pool_nleaf(v)) - 1ull;
                            touched = (slotbits << (Cfg::pool_first_draw(v) + s1 * Cfg::pool_nleaf(v))) |
                                      (slotbits << (Cfg::pool_first_draw(v) + s2 * Cfg::pool_nleaf(v)));
                            static_for<0, Cfg::pool_nleaf(v)>([&](auto Lf) { // swap!  sampler.jl:395-408, :86-97, :448-455
                                constexpr int l = decltype(Lf)::value;
                                double xa, xb, pa, pb;
                                int ba, bb;
                                get_slot<Cfg, v, l>(c, s1, xa, pa, ba);
                                get_slot<Cfg, v, l>(c, s2, xb, pb, bb);
                                put_slot<Cfg, v, l>(n, s1, xb, pb, bb);
                                put_slot<Cfg, v, l>(n, s2, xa, pa, ba);
                            });
                        }
                    });
                }
            }
        } else {
            // ---- changeVariable  updates.jl:71-111 ----
            ut = 1;
            static_for<0, NPOOL>([&](auto V) {
                constexpr int v = decltype(V)::value;
                constexpr int md = Cfg::pool_maxdof(v), nl = Cfg::pool_nleaf(v), k00 = Cfg::pool_first_draw(v);
                constexpr bool skip = (md <= 0) || (nl == 1 && Cfg::leaf_kind(Cfg::draw_leaf(md > 0 ? k00 : 0)) == 1 &&
                                                    Cfg::leaf_nbin(Cfg::draw_leaf(md > 0 ? k00 : 0)) == 1); // :79-81
                if constexpr (!skip) {
                    if (vi == v && cdv > 0) { // :82
                        active = true;
                        int slot = (int)(us1 * (double)cdv); // :83
                        if (slot >= cdv) slot = cdv - 1;
                        touched = ((1ull << nl) - 1ull) << (k00 + slot * nl);
                        if constexpr (pool_is_fermik<Cfg>(v)) { // shift!  sampler.jl:198-246; the move is picked by uniform 3
                            double u[nl], kk[nl], po;
                            int bo;
                            static_for<0, nl>([&](auto J) {
                                constexpr int j = decltype(J)::value;
                                u[j] = step_uniform_dyn(5 + k00 + slot * nl + j, sidx, st_step, k0, k1);
                                get_slot<Cfg, v, j>(c, slot, kk[j], po, bo);
                            });
                            prop *= fermik_shift<Cfg, v>(us2, u, kk);
                            static_for<0, nl>([&](auto J) { put_slot<Cfg, v, decltype(J)::value>(n, slot, kk[decltype(J)::value], 1.0, 0); });
                        } else
                        static_for<0, nl>([&](auto Lf) { // shift!  sampler.jl:336-386, :57-71, :431-440
                            constexpr int l = decltype(Lf)::value;
                            const double y = step_uniform_dyn(5 + k00 + slot * nl + l, sidx, st_step, k0, k1);
                            double xo, po, xn, pn;
                            int bo, bn;
                            get_slot<Cfg, v, l>(c, slot, xo, po, bo);
                            draw_pool_leaf<Cfg, v, l>(t, y, xn, pn, bn);
                            put_slot<Cfg, v, l>(n, slot, xn, pn, bn);
                            prop *= po / pn; // 1/prob_ratio  sampler.jl:385, :70
                        });
                    }
                }
            });
        }
    }
    McmcProposal<Cfg> out;
    out.n = n;
    out.prop = prop;
    out.active = active;
    out.newcurr = newcurr;
    out.ut = ut;
    out.pvi = pvi;
    out.touched = touched;
    return out;
}

template <class Cfg> __device__ __forceinline__ void mcmc_chains(const BatchArgs &a) {
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
    const u32 bs = (u32)B << 20; // (block, chain within the block) identify a chain: see vegasmc_chains
    const u32 st_init = a.iteration * 8u + STREAM_MCMC_INIT + bs, st_step = a.iteration * 8u + STREAM_MCMC_STEP + bs;
    const u32 k0 = (u32)a.seed, k1 = (u32)(a.seed >> 32);
    double rw[ND];
    static_for<0, ND>([&](auto I) { rw[decltype(I)::value] = a.reweight[decltype(I)::value]; });
    auto rw_sel = [&](int i) {
        double r = rw[NORMI];
        static_for<0, NI>([&](auto I) { if (i == decltype(I)::value) r = rw[decltype(I)::value]; });
        return r;
    };

    double acc[Cfg::NW];
    static_for<0, Cfg::NW>([&](auto I) { acc[decltype(I)::value] = 0.0; });
    double extra[Cfg::NCOLS - Cfg::NOBS];
    static_for<0, Cfg::NCOLS - Cfg::NOBS>([&](auto I) { extra[decltype(I)::value] = 0.0; });
    constexpr int XN = Cols<Cfg>::NORM - Cfg::NOBS, XE = Cols<Cfg>::NEVAL - Cfg::NOBS, XV = Cols<Cfg>::VISITED - Cfg::NOBS;

    for (i64 ch = (i64)slice * T + tid; ch < a.nchain; ch += (i64)a.wg_per_block * T) {
        const u64 g = (u64)ch;
        int curr = a.nchain == 1 ? 0 : (int)(g % (u64)ND); // montecarlo.jl:76 idx = 1; many chains start stratified
        Chain<Cfg> c;
        Weight<Cfg> weight; // :116 _State(curr, zero(T), 1.0)
        static_for<0, Cfg::NCOMP>([&](auto Q) { weight.v[decltype(Q)::value] = 0.0; });
        weight.abs = 0.0;
        double probability = 1.0;
        bool fresh = a.carry_x == nullptr;
        if (!fresh) { // continues the previous iteration's chain: its configuration and the integrand it sat on (BatchArgs::carry_x)
            const i64 from = carried_from(a, wi.lb, ch);
            load_carried<Cfg>(a, t, wi.lb, from, c);
            curr = a.carry_curr[wi.lb * a.carry_nchain + from];
            if (curr != NORMI) {
                weight = eval_sel<Cfg>(curr, c.x, a.ud);        // :197 on the carried configuration
                probability = weight.abs * rw_sel(curr);        // :199
                if (!(probability > 4.940656458412465e-274)) {  // (cannot happen while the integrand is the one that left it there)
                    fresh = true;
                    curr = (int)(g % (u64)ND);
                }
            } else probability = rw[NORMI];                     // :201-202
        }
        for (int tr = 0; fresh && tr < 10000; ++tr) {    // :118-124
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
        if (curr != NORMI && probability == 0.0) atomicOr(a.status, ST_MCMC_INIT); // :125-126 error(...)

        // holding times (this engine's own diagnostic, DESIGN.md "chains"): step of the last change of every slot and of
        // the integrand index, and the longest completed or still running hold
        int last[Cfg::NDRAW > 0 ? Cfg::NDRAW : 1], lastc = 0, hmax = 0;
        static_for<0, Cfg::NDRAW>([&](auto K) { last[decltype(K)::value] = 0; });
        i64 mcnt = 0, mj = 0; // it % measurefreq and it / measurefreq, carried
        for (i64 it = 1; it <= steps + nburn; ++it) { // :134
            const u64 sidx = (g << 32) | (u64)(it - 1);
            static_for<0, ND>([&](auto I) { extra[XV + decltype(I)::value] += curr == decltype(I)::value ? 1.0 : 0.0; }); // :136
            const u32x4 r0 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 0u, st_step, k0, k1);
            const u32x4 r1 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 1u, st_step, k0, k1);
            const u32x4 r2 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 2u, st_step, k0, k1);
            // :137 rand(rng, updates).  With many chains per block the 64 chains of a wave (chains ch & ~63 .. | 63 of ONE
            // block) share the update-type sequence: it is independent of the chain states, so every chain is still a
            // valid Markov chain, blocks stay independent, and the wave no longer walks through all three update bodies
            // at every step.  nchain = 1 (the reference's chain) draws its own.
            double uupd = u01(r0.x, r0.y);
            if (a.nchain > 1) {
                const u64 gidx = ((u64)(ch & ~(i64)63) << 32) | (u64)(it - 1);
                const u32x4 rg = philox4x32_10((u32)gidx, (u32)(gidx >> 32), 0u, a.iteration * 8u + STREAM_MCMC_GROUP + bs, k0, k1);
                uupd = u01(rg.x, rg.y);
            }
            int upd = (int)(uupd * (double)NUPD);
            if (upd >= NUPD) upd = NUPD - 1;
            if (a.nchain > 1) upd = __builtin_amdgcn_readfirstlane(upd); // wave-uniform by construction: a scalar branch
            const double upick = u01(r0.z, r0.w), us1 = u01(r1.x, r1.y), us2 = u01(r1.z, r1.w), uacc = u01(r2.x, r2.y);
            // ---- build the proposal (n, prop, newcurr); ONE evaluate-and-accept tail serves all three updates, so
            // lanes that diverged on the update type reconverge before the expensive part ----
            const McmcProposal<Cfg> pr = mcmc_propose<Cfg>(t, c, curr, upd, upick, us1, us2, sidx, st_step, k0, k1, r2);
            const Chain<Cfg> &n = pr.n;
            const double prop = pr.prop;
            const bool active = pr.active;
            const int newcurr = pr.newcurr, ut = pr.ut, pvi = pr.pvi;
            const u64 touched = pr.touched;
            if (active && prop > 4.9406564584124654e-324) { // updates.jl:29-31, :88-90, :129-131
                Weight<Cfg> wn;
                static_for<0, Cfg::NCOMP>([&](auto Q) { wn.v[decltype(Q)::value] = 0.0; });
                wn.abs = 0.0;
                if (newcurr != NORMI) wn = eval_sel<Cfg>(newcurr, n.x, a.ud);                  // :35-38, :92, :133
                extra[XE] += 1.0;                                                               // :40, :94, :135
                const double newp = newcurr == NORMI ? rw[NORMI] : wn.abs * rw_sel(newcurr);    // :42-44, :96, :137
                const double R = prop * newp / probability;                                     // :46, :97, :138
                const bool ok = uacc < R;                                                       // :49, :100, :141
                // propose[1, curr, new] :48,:50 | propose[2, curr, vi] :99,:101 | propose[3, curr, vi] :140,:142
                pa_count<Cfg, false>(sPA, PaTable<Cfg>::idx(ut, curr, ut == 0 ? newcurr : pvi), true, ok);
                if (a.hold_hist) {
                    const int now = (int)it;
                    u64 mo = 0ull, mn = 0ull; // live draws of the old and of the proposed integrand
                    static_for<0, NI>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        mo = curr == i ? Cfg::own_mask(i) : mo;
                        mn = newcurr == i ? Cfg::own_mask(i) : mn;
                    });
                    static_for<0, Cfg::NDRAW>([&](auto K) {
                        constexpr int k = decltype(K)::value;
                        // changeIntegrand: the slots it creates start their first hold (they held nothing before);
                        // changeVariable / swapVariable: an accepted move of the slot ends its hold (also when a Discrete
                        // redraw lands on the same value: the chain was free to move)
                        const bool chg = ok && (((ut == 0 ? (mn & ~mo) : touched) >> k) & 1ull) != 0ull;
                        const int hold = now - last[k];
                        hmax = (chg && ut != 0 && hold > hmax) ? hold : hmax;
                        last[k] = chg ? now : last[k];
                    });
                    const bool chg = ok && newcurr != curr;
                    const int hold = now - lastc;
                    hmax = (chg && hold > hmax) ? hold : hmax;
                    lastc = chg ? now : lastc;
                }
                if (ok) {
                    c = n;
                    curr = newcurr;                                                             // :51-53
                    weight = wn;
                    probability = newp;
                } // else the proposal copy is dropped: createRollback!/removeRollback! are no-ops (sampler.jl:306, :324),
                  // shiftRollback!/swapRollback! restore the slot (:105, :145)
            }
            // ---- measurement  montecarlo.jl:144-172 ----
            mcnt = mcnt + 1 == a.measurefreq ? 0 : mcnt + 1;
            const bool mf = mcnt == 0;
            mj += mf ? 1 : 0;
            if (mf && it >= nburn) {
                if (curr != NORMI) {
                    double relw[Cfg::NCOMP]; // :162
                    static_for<0, Cfg::NCOMP>([&](auto Q) { relw[decltype(Q)::value] = weight.v[decltype(Q)::value] / probability; });
                    if constexpr (Cfg::HOST_MEASURE != 0) { // measure(idx, var, obs, relative_weight, config) on the host, after the launch  :166-169
                        if (tile == 0) host_measure_record<Cfg, Cfg::NCOMP>(a, wi.lb, ch, mj, c.x, relw, curr);
                    }
                    static_for<0, NI>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        if (curr == i) {
                            static_for<0, Cfg::NDRAW>([&](auto K) { // :147-154  accumulate!(var, pos + offset, 1.0)
                                constexpr int k = decltype(K)::value;
                                if constexpr ((Cfg::own_mask(i) >> k) & 1ull) hist_add<Cfg, k>(c.bin[k], 1.0, sH, a.ghist, tile);
                            });
                            if constexpr (Cfg::HOST_MEASURE != 0) {
                            } else if constexpr (Cfg::CUSTOM_MEASURE != 0) { // measure(idx, var, obs, relative_weight, config)  :166-169
                                double rwv[Cfg::NW];
                                static_for<0, Cfg::NW>([&](auto Q) { rwv[decltype(Q)::value] = 0.0; });
                                static_for<0, Cfg::NCOMP>([&](auto Q) { rwv[i * Cfg::NCOMP + decltype(Q)::value] = relw[decltype(Q)::value]; });
                                Cfg::measure(c.x, rwv, a.ud, i, obs_wave<Cfg>(sO));
                            } else if constexpr (Cfg::obs_bin_draw(i) >= 0) {
                                const int b = c.bin[Cfg::obs_bin_draw(i)];
                                if (b >= 0 && b < Cfg::obs_nbin(i)) lds_add(&obs_wave<Cfg>(sO)[Cfg::obs_off(i) + b], relw[0]);
                            }
                        }
                        if constexpr (Cfg::CUSTOM_MEASURE == 0 && Cfg::obs_bin_draw(i) < 0) // :164
                            static_for<0, Cfg::NCOMP>([&](auto Q) { acc[i * Cfg::NCOMP + decltype(Q)::value] += curr == i ? relw[decltype(Q)::value] : 0.0; });
                    });
                } else {
                    extra[XN] += 1.0 / rw[NORMI]; // :158
                }
            }
        }
        if (a.hold_hist) { // holds still running when the chain ends count with their length so far
            const int tot = (int)(steps + nburn);
            hmax = (tot - lastc > hmax) ? tot - lastc : hmax;
            static_for<0, NI>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if (curr == i) static_for<0, Cfg::NDRAW>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    // (a lone single-valued Discrete has nothing to sample, updates.jl:79-81: it never moves and holds nothing)
                    constexpr int pv = Cfg::draw_pool(k);
                    constexpr bool fixed = Cfg::pool_nleaf(pv) == 1 && Cfg::leaf_kind(Cfg::draw_leaf(k)) == 1 && Cfg::leaf_nbin(Cfg::draw_leaf(k)) == 1;
                    if constexpr (((Cfg::own_mask(i) >> k) & 1ull) && !fixed) hmax = (tot - last[k] > hmax) ? tot - last[k] : hmax;
                });
            });
            atomicAdd(&a.hold_hist[hmax <= 0 ? 0 : 32 - __clz(hmax)], 1ull);
        }
        if (a.store_x && tile == 0) {
            store_carried<Cfg>(a, wi.lb, ch, c);
            a.store_curr[wi.lb * a.nchain + ch] = curr;
        }
    }
    __syncthreads();
    flush_workgroup<Cfg, Lds<Cfg>, true, true>(a, smem, acc, extra, wi.rowid, tile);
}

// ---------------------------------------------------------------------------------------------
// MCMC with the integrand on the HOST (BatchArgs::HostStep): the step of mcmc_chains cut at the integrand call
// (mcmc/updates.jl:35-38, :92, :133).  Every launch finishes, per chain, the evaluation that just came back -- the start
// configuration (retried like montecarlo.jl:118-124) or the proposal of step it -- and proposes the next step; the host evaluates
// integrand hidx[chain] at hx[.][chain].  Same streams, same arithmetic as mcmc_chains (the holding-time diagnostic is left out).
// ---------------------------------------------------------------------------------------------
template <class Cfg> __device__ __forceinline__ void mcmc_host_step(const BatchArgs &a) {
    static_assert(Cfg::NTILE == 1, "host-closure chains keep one histogram tile");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NI = Cfg::NI, NORMI = Cfg::NI, ND = Cfg::NI + 1, NPOOL = Cfg::NPOOL;
    constexpr int NUPD = 2 * NPOOL + 2;
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
    const BatchArgs::HostStep &h = a.hs;
    const i64 B = a.block_lo + wi.lb;
    const i64 total = a.neval_per_block / a.nchain + a.nburn, nburn = a.nburn, nc = h.nc;
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
    double acc[Cfg::NW];
    static_for<0, Cfg::NW>([&](auto I) { acc[decltype(I)::value] = 0.0; });
    double extra[Cfg::NCOLS - Cfg::NOBS];
    static_for<0, Cfg::NCOLS - Cfg::NOBS>([&](auto I) { extra[decltype(I)::value] = 0.0; });
    constexpr int XN = Cols<Cfg>::NORM - Cfg::NOBS, XE = Cols<Cfg>::NEVAL - Cfg::NOBS, XV = Cols<Cfg>::VISITED - Cfg::NOBS;

    for (i64 ch = (i64)wi.slice * T + tid; ch < a.nchain; ch += (i64)a.wg_per_block * T) {
        const u64 g = (u64)ch;
        const i64 cid = wi.lb * a.nchain + ch;
        Chain<Cfg> c;
        auto draw_start = [&](const int tr) { // initialize!  :190-193: try `tr` of the start configuration -> state + what the host evaluates
            Sample<Cfg> s;
            draw_sample<Cfg>(t, a.seed, st_init, g * 16384ull + (u64)tr, s);
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
                            kk[decltype(J)::value] = kF / sqrt((double)D);
                        });
                        (void)fermik_create<Cfg, v>(u, kk);
                        static_for<0, D>([&](auto J) { c.x[kb + decltype(J)::value] = kk[decltype(J)::value]; });
                    });
                }
            });
            static_for<0, Cfg::NDRAW>([&](auto K) {
                constexpr int k = decltype(K)::value;
                h.cx[k * nc + cid] = c.x[k];
                h.hx[k * nc + cid] = c.x[k];
                h.cbin[k * nc + cid] = c.bin[k];
                h.cprob[k * nc + cid] = c.prob[k];
            });
        };
        if (h.ne == 0) {
            int curr0 = a.nchain == 1 ? 0 : (int)(g % (u64)ND); // montecarlo.jl:76 idx = 1; many chains start stratified
            if (a.carry_x) { // the previous iteration's chain goes on: its configuration and the integrand it sat on (BatchArgs::carry_x)
                const i64 from = carried_from(a, wi.lb, ch);
                load_carried<Cfg>(a, t, wi.lb, from, c);
                curr0 = a.carry_curr[wi.lb * a.carry_nchain + from];
                static_for<0, Cfg::NDRAW>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    h.cx[k * nc + cid] = c.x[k];
                    h.hx[k * nc + cid] = c.x[k];
                    h.cbin[k * nc + cid] = c.bin[k];
                    h.cprob[k * nc + cid] = c.prob[k];
                });
            } else
            draw_start(0);
            h.ccurr[cid] = curr0;
            h.cit[cid] = -1;
            h.ctr[cid] = 0;
            h.hidx[cid] = curr0 != NORMI ? curr0 : -1;
            continue;
        }
        int it = h.cit[cid];
        if (it >= total) continue; // this chain is through
        int curr = h.ccurr[cid];
        static_for<0, Cfg::NDRAW>([&](auto K) {
            constexpr int k = decltype(K)::value;
            c.x[k] = h.cx[k * nc + cid];
            c.prob[k] = h.cprob[k * nc + cid];
            c.bin[k] = h.cbin[k * nc + cid];
        });
        Weight<Cfg> weight;
        double probability;
        if (it < 0) { // ---- the start configuration was evaluated  :118-126, :195-203 ----
            static_for<0, Cfg::NCOMP>([&](auto Q) { weight.v[decltype(Q)::value] = 0.0; });
            weight.abs = 0.0;
            if (curr != NORMI) {
                static_for<0, Cfg::NCOMP>([&](auto Q) { weight.v[decltype(Q)::value] = a.host_w[decltype(Q)::value * nc + cid]; });
                if constexpr (Cfg::NCOMP == 1) weight.abs = fabs(weight.v[0]);
                else weight.abs = hypot(weight.v[0], weight.v[Cfg::NCOMP - 1]);
                probability = weight.abs * rw_sel(curr);        // :199
            } else {
                probability = rw[NORMI];                        // :201-202
            }
            if (!(curr == NORMI || probability > 4.940656458412465e-274)) { // :120-122 (TINY): draw the start again
                const int tr = h.ctr[cid] + 1;
                if (tr >= 10000) {
                    if (probability == 0.0) atomicOr(a.status, ST_MCMC_INIT); // :125-126 error(...)
                } else {
                    h.ctr[cid] = tr;
                    draw_start(tr);
                    h.hidx[cid] = curr;
                    continue;
                }
            }
            it = 0;
        } else { // ---- the proposal of step it + 1 was evaluated: accept or drop it, then measure ----
            static_for<0, Cfg::NCOMP>([&](auto Q) { weight.v[decltype(Q)::value] = h.cw[decltype(Q)::value * nc + cid]; });
            weight.abs = h.cwabs[cid];
            probability = h.cprobability[cid];
            const int step = it + 1;
            const double prop = h.pprop[cid];
            const int newcurr = h.pnew[cid], ut = h.put[cid], pvi = h.pvi[cid];
            if (ut >= 0 && prop > 4.9406564584124654e-324) { // updates.jl:29-31, :88-90, :129-131  (ut < 0: nothing was proposed)
                Chain<Cfg> n;
                static_for<0, Cfg::NDRAW>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    n.x[k] = h.hx[k * nc + cid];
                    n.prob[k] = h.pprob[k * nc + cid];
                    n.bin[k] = h.pbin[k * nc + cid];
                });
                Weight<Cfg> wn;
                static_for<0, Cfg::NCOMP>([&](auto Q) { wn.v[decltype(Q)::value] = 0.0; });
                wn.abs = 0.0;
                if (newcurr != NORMI) {                                                        // :35-38, :92, :133 -- on the host
                    static_for<0, Cfg::NCOMP>([&](auto Q) { wn.v[decltype(Q)::value] = a.host_w[decltype(Q)::value * nc + cid]; });
                    if constexpr (Cfg::NCOMP == 1) wn.abs = fabs(wn.v[0]);
                    else wn.abs = hypot(wn.v[0], wn.v[Cfg::NCOMP - 1]);
                }
                extra[XE] += 1.0;                                                               // :40, :94, :135
                const double newp = newcurr == NORMI ? rw[NORMI] : wn.abs * rw_sel(newcurr);    // :42-44, :96, :137
                const double R = prop * newp / probability;                                     // :46, :97, :138
                const bool ok = h.puacc[cid] < R;                                               // :49, :100, :141
                pa_count<Cfg, false>(sPA, PaTable<Cfg>::idx(ut, curr, ut == 0 ? newcurr : pvi), true, ok);
                if (ok) {
                    c = n;
                    curr = newcurr;                                                             // :51-53
                    weight = wn;
                    probability = newp;
                }
            }
            if (step % a.measurefreq == 0 && step >= nburn) { // ---- measurement  montecarlo.jl:144-172 ----
                if (curr != NORMI) {
                    double relw[Cfg::NCOMP]; // :162
                    static_for<0, Cfg::NCOMP>([&](auto Q) { relw[decltype(Q)::value] = weight.v[decltype(Q)::value] / probability; });
                    if constexpr (Cfg::HOST_MEASURE != 0) host_measure_record<Cfg, Cfg::NCOMP>(a, wi.lb, ch, step / a.measurefreq, c.x, relw, curr);
                    static_for<0, NI>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        if (curr == i) {
                            static_for<0, Cfg::NDRAW>([&](auto K) { // :147-154  accumulate!(var, pos + offset, 1.0)
                                constexpr int k = decltype(K)::value;
                                if constexpr ((Cfg::own_mask(i) >> k) & 1ull) hist_add<Cfg, k>(c.bin[k], 1.0, sH, a.ghist, 0);
                            });
                            if constexpr (Cfg::HOST_MEASURE != 0) {
                            } else if constexpr (Cfg::CUSTOM_MEASURE != 0) { // measure(idx, var, obs, relative_weight, config)  :166-169
                                double rwv[Cfg::NW];
                                static_for<0, Cfg::NW>([&](auto Q) { rwv[decltype(Q)::value] = 0.0; });
                                static_for<0, Cfg::NCOMP>([&](auto Q) { rwv[i * Cfg::NCOMP + decltype(Q)::value] = relw[decltype(Q)::value]; });
                                Cfg::measure(c.x, rwv, a.ud, i, obs_wave<Cfg>(sO));
                            } else if constexpr (Cfg::obs_bin_draw(i) >= 0) {
                                const int b = c.bin[Cfg::obs_bin_draw(i)];
                                if (b >= 0 && b < Cfg::obs_nbin(i)) lds_add(&obs_wave<Cfg>(sO)[Cfg::obs_off(i) + b], relw[0]);
                            }
                        }
                        if constexpr (Cfg::CUSTOM_MEASURE == 0 && Cfg::obs_bin_draw(i) < 0) // :164
                            static_for<0, Cfg::NCOMP>([&](auto Q) { acc[i * Cfg::NCOMP + decltype(Q)::value] += curr == i ? relw[decltype(Q)::value] : 0.0; });
                    });
                } else {
                    extra[XN] += 1.0 / rw[NORMI]; // :158
                }
            }
            it = step;
        }
        if (it < total) { // ---- propose step it + 1 ----
            const u64 sidx = (g << 32) | (u64)it;
            static_for<0, ND>([&](auto I) { extra[XV + decltype(I)::value] += curr == decltype(I)::value ? 1.0 : 0.0; }); // :136
            const u32x4 r0 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 0u, st_step, k0, k1);
            const u32x4 r1 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 1u, st_step, k0, k1);
            const u32x4 r2 = philox4x32_10((u32)sidx, (u32)(sidx >> 32), 2u, st_step, k0, k1);
            double uupd = u01(r0.x, r0.y);
            if (a.nchain > 1) { // (the 64 chains of a wave share the update-type sequence, as in mcmc_chains)
                const u64 gidx = ((u64)(ch & ~(i64)63) << 32) | (u64)it;
                const u32x4 rg = philox4x32_10((u32)gidx, (u32)(gidx >> 32), 0u, a.iteration * 8u + STREAM_MCMC_GROUP + bs, k0, k1);
                uupd = u01(rg.x, rg.y);
            }
            int upd = (int)(uupd * (double)NUPD);
            if (upd >= NUPD) upd = NUPD - 1;
            const double upick = u01(r0.z, r0.w), us1 = u01(r1.x, r1.y), us2 = u01(r1.z, r1.w);
            const McmcProposal<Cfg> pr = mcmc_propose<Cfg>(t, c, curr, upd, upick, us1, us2, sidx, st_step, k0, k1, r2);
            static_for<0, Cfg::NDRAW>([&](auto K) {
                constexpr int k = decltype(K)::value;
                h.hx[k * nc + cid] = pr.n.x[k];
                h.pprob[k * nc + cid] = pr.n.prob[k];
                h.pbin[k * nc + cid] = pr.n.bin[k];
            });
            h.pprop[cid] = pr.prop;
            h.puacc[cid] = u01(r2.x, r2.y);
            h.pnew[cid] = pr.newcurr;
            h.put[cid] = pr.active ? pr.ut : -1;
            h.pvi[cid] = pr.pvi;
            h.hidx[cid] = (pr.active && pr.prop > 4.9406564584124654e-324 && pr.newcurr != NORMI) ? pr.newcurr : -1;
        } else {
            h.hidx[cid] = -1;
            atomicAdd(h.done, 1);
            if (a.store_x) { // the chain's last step is through
                store_carried<Cfg>(a, wi.lb, ch, c);
                a.store_curr[wi.lb * a.nchain + ch] = curr;
            }
        }
        static_for<0, Cfg::NDRAW>([&](auto K) {
            constexpr int k = decltype(K)::value;
            h.cx[k * nc + cid] = c.x[k];
            h.cprob[k * nc + cid] = c.prob[k];
            h.cbin[k * nc + cid] = c.bin[k];
        });
        static_for<0, Cfg::NCOMP>([&](auto Q) { h.cw[decltype(Q)::value * nc + cid] = weight.v[decltype(Q)::value]; });
        h.cwabs[cid] = weight.abs;
        h.cprobability[cid] = probability;
        h.ccurr[cid] = curr;
        h.cit[cid] = it;
    }
    __syncthreads();
    flush_workgroup<Cfg, Lds<Cfg>, true, true, true>(a, smem, acc, extra, wi.rowid, 0);
}

// the map + integrand alone, for parity tests of a2/a3 and for host-side consumers
template <class Cfg> __device__ __forceinline__ void sample_dump(const DumpArgs &a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sE = smem + Lds<Cfg>::E, *sDA = smem + Lds<Cfg>::DA, *sDD = smem + Lds<Cfg>::DD;
    stage_tables<Cfg>(a.edges, a.dacc, a.ddist, sE, sDA, sDD);
    __syncthreads();
    Tables<Cfg> t;
    if constexpr (Cfg::TABLE_MODE <= 1) t.E = sE;
    else t.E = a.edges;
    t.DA = sDA;
    t.DD = sDD;
    const u32 stream = a.iteration * 8u + STREAM_VEGAS;
    for (i64 n = (i64)blockIdx.x * blockDim.x + threadIdx.x; n < a.n; n += (i64)gridDim.x * blockDim.x) {
        Sample<Cfg> s;
        draw_sample<Cfg, false, false, (Cfg::RNG_BITS == 32 ? 4 : 2)>(t, make_round_keys<false>((u32)a.seed, (u32)(a.seed >> 32)), stream, (u64)(a.first_index + n), s);
        if (a.soa) {
            static_for<0, Cfg::NDRAW>([&](auto K) { constexpr int k = decltype(K)::value; a.x[(i64)k * a.n + n] = s.x[k]; });
            continue;
        }
        double w[Cfg::NW];
        if constexpr (Cfg::HOST_INTEGRAND != 0) static_for<0, Cfg::NW>([&](auto I) { w[decltype(I)::value] = 0.0; });
        else Cfg::integrand(s.x, w, a.ud, -1);
        static_for<0, Cfg::NDRAW>([&](auto K) { constexpr int k = decltype(K)::value; a.x[n * Cfg::NDRAW + k] = s.x[k]; });
        a.jac[n] = s.jac;
        static_for<0, Cfg::NW>([&](auto I) { constexpr int i = decltype(I)::value; a.w[n * Cfg::NW + i] = w[i]; });
    }
}

} // namespace mci
