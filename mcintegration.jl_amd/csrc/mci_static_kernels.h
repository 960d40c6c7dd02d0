// mci_static_kernels.h -- configuration-independent gfx950 kernels, compiled ahead of time by hipcc:
// partial reduction (block merge, reference src/main.jl:273-287, src/configuration.jl:252-262),
// reweighting (src/main.jl:322-346) and grid refinement (src/distribution/variable.jl:206-239,
// :369-382 with src/distribution/common.jl:43-82).  These are O(bins) per iteration; they stay on the
// device so that an iteration is one asynchronous chain  sample -> merge -> all-reduce -> train  with
// no host round trip.
#pragma once
#include <hip/hip_runtime.h>

#include "mci_train.h" // merge / train device functions, ST_* status bits

namespace mci {

// stage 1 of the histogram merge: out[g][bin] = sum over this group's workgroups (fixed order)
__global__ void __launch_bounds__(256) k_hist_stage1(const double *__restrict__ part_hist, int nwg, int nbin, int ngroup,
                                                     double *__restrict__ out) {
    const int bin = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (bin >= nbin) return;
    const int per = (nwg + ngroup - 1) / ngroup;
    const int w0 = g * per, w1 = min(nwg, w0 + per);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int w = w0;
    for (; w + 3 < w1; w += 4) {
        s0 += part_hist[(size_t)(w + 0) * nbin + bin];
        s1 += part_hist[(size_t)(w + 1) * nbin + bin];
        s2 += part_hist[(size_t)(w + 2) * nbin + bin];
        s3 += part_hist[(size_t)(w + 3) * nbin + bin];
    }
    for (; w < w1; ++w) s0 += part_hist[(size_t)w * nbin + bin];
    out[(size_t)g * nbin + bin] = (s0 + s1) + (s2 + s3);
}

// host measure: obs[b][o] (accumulated by the host closure over block b's samples) goes into the observable columns of the
// block's first partial row, which the kernel left at zero
__global__ void __launch_bounds__(256) k_add_host_obs(const double *__restrict__ obs, int nblocks, int nobs, int ncols, int wg_per_block,
                                                      double *__restrict__ part_cols) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nblocks * nobs) return;
    const int b = i / nobs, o = i % nobs;
    part_cols[(size_t)b * wg_per_block * ncols + o] += obs[i];
}

__global__ void __launch_bounds__(256) k_finalize(MergeArgs m) {
    const int nhb = (m.nbin + 255) / 256;
    const int hoff = 2 * m.nobs + 2 + m.ni + 1;
    if ((int)blockIdx.x < nhb) {
        const int bin = blockIdx.x * 256 + threadIdx.x;
        if (bin < m.nbin) m.packed[hoff + bin] = merge_hist_bin(m, bin);
        return;
    }
    if ((int)blockIdx.x > nhb) { // grid = nhb + 1 + merge_pa_blocks
        merge_pa(m, (int)blockIdx.x - nhb - 1);
        return;
    }
    merge_stats(m);
}

// Carried :mcmc chains (this engine's many-chain decomposition; DESIGN.md "Chains"): which stored chain every chain of the next launch
// continues.  The chains a block stored are a sample of the finished iteration's target ~ reweight_old[idx] |f_idx(x)|; doReweight! has
// moved the factors since, so the next target differs from it by the known ratio w[idx] = reweight_new[idx] / reweight_old[idx]:
// systematic resampling of the block's stored chains (chain order, one fixed offset u -- deterministic) with probability ~ w[curr].
//   W[j] = sum_i w[i] * #(stored chains j' <= j that ended on integrand i)        (added over i = 0 .. nd-1 in that order)
//   new chain c continues the first stored chain j with W[j] > (c + u) * (W[n_old-1] / n_new),  u = (sqrt(5) - 1) / 2  (with u = 1/2
//   many chain counts make (c + u) n_old / n_new an integer: exact ties, decided by the last bit of w, whenever a block's stored chains
//   all sit on one integrand)
// One workgroup per block; mirrored by mcio_resample_chains (same operations in the same order: same picks).
struct ResampleArgs {
    const int *curr_old;   // [nblocks][n_old]
    long long n_old, n_new;
    int nd;
    const double *rw_now, *rw_used; // [nd]
    int *src;              // [nblocks][n_new]
    double *W;             // [nblocks][n_old] scratch
    const double *w_chain; // [nblocks][n_old] or NULL: a weight per stored chain (:vegasmc: new target / old target, vegasmc_carry_weights)
                           // instead of a weight per integrand index
};
__global__ void __launch_bounds__(256) k_resample_chains(ResampleArgs a) {
    constexpr int kLdsW = 8192; // stored chains per block whose running weights also sit in LDS: the bisections below then read LDS
    __shared__ double sW[kLdsW];
    __shared__ long long part[256];
    const int tid = threadIdx.x, T = 256;
    const int *co = a.curr_old + (size_t)blockIdx.x * a.n_old;
    double *W = a.W + (size_t)blockIdx.x * a.n_old;
    int *src = a.src + (size_t)blockIdx.x * a.n_new;
    const long long per = (a.n_old + T - 1) / T, j0 = tid * per < a.n_old ? tid * per : a.n_old, j1 = j0 + per < a.n_old ? j0 + per : a.n_old;
    constexpr int kNd = 16; // integrands (+ the normalisation) whose running counts a thread keeps in registers
    if (a.w_chain) {
        // W[j] = (sum of the stretches before this thread's, added in thread order) + (running sum along the thread's own stretch): a
        // fixed association the oracle repeats (mcio_resample_weighted)
        __shared__ double dpart[256];
        const double *wc = a.w_chain + (size_t)blockIdx.x * a.n_old;
        double mine = 0.0;
        for (long long j = j0; j < j1; ++j) mine += wc[j];
        dpart[tid] = mine;
        __syncthreads();
        double below = 0.0;
        for (int t = 0; t < tid; ++t) below += dpart[t];
        double run = 0.0;
        for (long long j = j0; j < j1; ++j) {
            run += wc[j];
            const double v = below + run;
            W[j] = v;
            if (a.n_old <= kLdsW) sW[j] = v;
        }
    } else
    if (a.nd <= kNd) {
        // Two passes over the thread's stretch of the stored chains instead of one read-modify-write pass over W per integrand: the
        // per-integrand counts below the stretch first, then W[j] = sum_i w[i] * cnt_i(j), the terms added over i = 0 .. nd-1 in that
        // order -- the same numbers in the same order as the pass per integrand (the counts are integers, every product is formed once).
        long long cnt[kNd];
        double w[kNd];
#pragma unroll
        for (int i = 0; i < kNd; ++i) {
            cnt[i] = 0;
            w[i] = i < a.nd ? a.rw_now[i] / a.rw_used[i] : 0.0;
        }
        for (long long j = j0; j < j1; ++j) {
            const int c = co[j];
#pragma unroll
            for (int i = 0; i < kNd; ++i) cnt[i] += c == i ? 1 : 0;
        }
        for (int i = 0; i < a.nd; ++i) { // this thread's counts -> the counts of everything before its stretch
            long long mine = 0;
#pragma unroll
            for (int q = 0; q < kNd; ++q) mine = q == i ? cnt[q] : mine;
            __syncthreads();
            part[tid] = mine;
            __syncthreads();
            long long below = 0;
            for (int t = 0; t < tid; ++t) below += part[t];
#pragma unroll
            for (int q = 0; q < kNd; ++q) cnt[q] = q == i ? below : cnt[q];
        }
        for (long long j = j0; j < j1; ++j) {
            const int c = co[j];
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < kNd; ++i) {
                cnt[i] += c == i ? 1 : 0;
                if (i < a.nd) {
                    const double term = w[i] * (double)cnt[i];
                    acc = i == 0 ? term : acc + term;
                }
            }
            W[j] = acc;
            if (a.n_old <= kLdsW) sW[j] = acc;
        }
    } else
    for (int i = 0; i < a.nd; ++i) { // one pass per integrand: its running count along the stored chains, times its ratio, onto W
        const double w = a.rw_now[i] / a.rw_used[i];
        long long mine = 0;
        for (long long j = j0; j < j1; ++j) mine += co[j] == i ? 1 : 0;
        __syncthreads();
        part[tid] = mine;
        __syncthreads();
        long long cnt = 0;
        for (int t = 0; t < tid; ++t) cnt += part[t];
        for (long long j = j0; j < j1; ++j) {
            cnt += co[j] == i ? 1 : 0;
            const double term = w * (double)cnt;
            W[j] = i == 0 ? term : W[j] + term;
        }
    }
    __threadfence_block();
    __syncthreads();
    const bool in_lds = (a.w_chain != nullptr || a.nd <= kNd) && a.n_old <= kLdsW;
    const double step = W[a.n_old - 1] / (double)a.n_new;
    if (!(W[a.n_old - 1] > 0.0)) { // every stored chain has weight zero (or the total is not a number): nothing to resample BY -- spread the
        // new chains over the stored ones instead of continuing all of them from the last (they burn in from there like fresh starts)
        for (long long c = tid; c < a.n_new; c += T) src[c] = (int)(c % a.n_old);
        return;
    }
    for (long long c = tid; c < a.n_new; c += T) {
        const double target = ((double)c + 0.6180339887498949) * step;
        long long lo = 0, hi = a.n_old - 1; // smallest j with W[j] > target
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if ((in_lds ? sW[mid] : W[mid]) > target) hi = mid;
            else lo = mid + 1;
        }
        src[c] = (int)lo;
    }
}

// One workgroup per leaf: Dist.train! then clearStatistics!; workgroup `nleaf`: bookkeeping.
__global__ void __launch_bounds__(512) k_train(TrainArgs a) { // (launched with 256 or 512 threads; the serial walk's lane wants ~200 registers)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double ps[256]; // block_prefix scratch
    __shared__ int bad;
    __shared__ double ssum;
    if ((int)blockIdx.x == a.nleaf) {
        iteration_bookkeeping(a);
        return;
    }
    if (!a.do_train) return;
    const LeafDev L = a.leaves[blockIdx.x];
    if (!L.adapt) return; // variable.jl:208, :370
    double *h = a.packed + a.nstat + L.boff;
    train_leaf(L, h, h, sm, ps, bad, ssum, a.edges, a.dacc, a.ddist, a.serial_walk, a.status, false, nullptr, false, a.spare ? sm + train_lds_doubles(a.maxn) : nullptr);
}

// Single-rank iterations need no all-reduce between the merge and the refinement: k_finalize and k_train as ONE
// launch.  Workgroup l < nleaf merges its leaf's histogram (second stage) into LDS and `packed`, then trains from
// the LDS copy; workgroup nleaf merges the statistics head, then does the bookkeeping.
__global__ void __launch_bounds__(512) k_finish(MergeArgs m, TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[]; // [train_lds_doubles(maxn)] train scratch | [maxn] merged histogram, then ([train_spare_doubles(maxn)], a.spare) the serial walk's slots
    __shared__ double ps[256];
    __shared__ int bad;
    __shared__ double ssum;
    const int tid = threadIdx.x, T = blockDim.x;
    if ((int)blockIdx.x == a.nleaf) {
        merge_stats(m);
        __syncthreads(); // the head of `packed` was written by this workgroup
        iteration_bookkeeping(a);
        return;
    }
    if ((int)blockIdx.x > a.nleaf) { // grid = nleaf + 1 + merge_pa_blocks
        merge_pa(m, (int)blockIdx.x - a.nleaf - 1);
        return;
    }
    const LeafDev L = a.leaves[blockIdx.x];
    double *hl = sm + train_lds_doubles(a.maxn);
    double *hp = a.packed + a.nstat + L.boff;
    const bool train = a.do_train && L.adapt;
    if (train) train_stage_grid(L, sm, a.edges); // (the old grid's loads fly together with the histogram's)
    for (int i = tid; i < L.nbin; i += T) {
        const double v = merge_hist_bin(m, L.boff + i);
        hl[i] = v;
        hp[i] = v;
    }
    __syncthreads();
    if (!train) return;
    train_leaf(L, hl, hp, sm, ps, bad, ssum, a.edges, a.dacc, a.ddist, a.serial_walk, a.status, true, nullptr, false, a.spare ? hl : nullptr); // (hl: free once smoothed)
}

} // namespace mci
