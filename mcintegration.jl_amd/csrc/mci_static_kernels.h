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

// One workgroup per leaf: Dist.train! then clearStatistics!; workgroup `nleaf`: bookkeeping.
__global__ void __launch_bounds__(1024) k_train(TrainArgs a) {
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
    train_leaf(L, h, h, sm, ps, bad, ssum, a.edges, a.dacc, a.ddist, a.serial_walk, a.status);
}

// Single-rank iterations need no all-reduce between the merge and the refinement: k_finalize and k_train as ONE
// launch.  Workgroup l < nleaf merges its leaf's histogram (second stage) into LDS and `packed`, then trains from
// the LDS copy; workgroup nleaf merges the statistics head, then does the bookkeeping.
__global__ void __launch_bounds__(1024) k_finish(MergeArgs m, TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[]; // [train_lds_doubles(maxn)] train scratch | [maxn] merged histogram
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
    train_leaf(L, hl, hp, sm, ps, bad, ssum, a.edges, a.dacc, a.ddist, a.serial_walk, a.status, true);
}

} // namespace mci
