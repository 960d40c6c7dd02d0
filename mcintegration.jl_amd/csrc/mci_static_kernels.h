// mci_static_kernels.h -- configuration-independent gfx950 kernels, compiled ahead of time by hipcc:
// partial reduction (block merge, reference src/main.jl:273-287, src/configuration.jl:252-262),
// reweighting (src/main.jl:322-346) and grid refinement (src/distribution/variable.jl:206-239,
// :369-382 with src/distribution/common.jl:43-82).  These are O(bins) per iteration; they stay on the
// device so that an iteration is one asynchronous chain  sample -> merge -> all-reduce -> train  with
// no host round trip.
#pragma once
#include <hip/hip_runtime.h>

#include "mci_device.h" // ST_* status bits

namespace mci {

struct LeafDev {
    int kind;   // 0 continuous, 1 discrete
    int nbin;   // continuous: npts-1 ; discrete: K
    int eoff;   // continuous: offset into edges ; discrete: offset into dacc
    int doff;   // discrete: offset into ddist
    int boff;   // offset into the histogram section
    int adapt;
    double alpha;
};


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

// packed = [obsSum(nobs) | obsSqSum(nobs) | normalization | neval | visited(ni+1) | hist(nbin)]
// Workgroups [0, nhb) merge the histogram section; the last workgroup merges the statistics columns
// block by block:  m = observable/normalization; obsSum += m; obsSquaredSum += m*m   (main.jl:275-287)
// with every block and the merged config starting from clearStatistics! values (configuration.jl:238-250):
// normalization 1e-10, visited 1e-8, histogram 1e-10.
enum { kMergeGroups = 32 }; // first-stage histogram groups (k_hist_stage1 launches exactly this many)

struct MergeArgs {
    const double *part_cols; // [rows][ncols]
    int ncols, nobs, ni, nblocks, wg_per_block;
    const double *stage1;    // [ngroup][nbin]
    int ngroup;
    double *ghist;           // global-atomics histogram (table modes 1, 2)
    int use_ghist, nbin;
    double *packed;
    int *status;
    double *scratch;         // [nblocks*ncols]
    const double *part_pa;   // [nrows][2*npa] per-workgroup propose | accept tables of a chain solver; NULL after a :vegas pass
    int npa, nrows;          // npa = 3 * (ni+1) * max(ni+1, npool)   (configuration.jl:185-186)
};
// packed = [ ... | hist(nbin) | propose(npa) | accept(npa) ]: the tables ride in the all-reduce like MPIreduceConfig! reduces them
// (configuration.jl:297-298).  One wave per entry, lanes stride over the workgroup rows.
__device__ inline int merge_pa_blocks(const MergeArgs &m) { return (2 * m.npa + 3) / 4; }
__device__ inline void merge_pa(const MergeArgs &m, int blk) {
    const int lane = threadIdx.x & 63, e = blk * 4 + (int)(threadIdx.x >> 6);
    if (e >= 2 * m.npa || (threadIdx.x >> 6) >= 4) return; // (four entries per workgroup whatever its size)
    double s = 0.0;
    if (m.part_pa)
        for (int r = lane; r < m.nrows; r += 64) s += m.part_pa[(size_t)r * (2 * m.npa) + e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    // clearStatistics! of every block's config and of the summed one: propose 1e-8, accept 1e-10 (configuration.jl:247-248)
    const double off0 = e < m.npa ? 1.0e-8 : 1.0e-10;
    if (lane == 0) m.packed[2 * m.nobs + 2 + m.ni + 1 + m.nbin + e] = s + (double)(m.nblocks + 1) * off0;
}

// one histogram bin of the merged config: clearStatistics! offsets + the second merge stage
__device__ inline double merge_hist_bin(const MergeArgs &m, int bin) {
    double s = (double)(m.nblocks + 1) * 1.0e-10;
    if (m.use_ghist) {
        s += m.ghist[bin];
        m.ghist[bin] = 0.0; // ready for the next iteration
    } else {
        // all group partials in flight at once (a rolled loop would serialise ngroup L2 round trips), summed in group order
        double v[kMergeGroups];
#pragma unroll
        for (int g = 0; g < kMergeGroups; ++g) v[g] = m.stage1[(size_t)g * m.nbin + bin];
#pragma unroll
        for (int g = 0; g < kMergeGroups; ++g) s += v[g];
    }
    return s;
}

// the statistics head of `packed`, by one workgroup
__device__ inline void merge_stats(const MergeArgs &m) {
    const double *__restrict__ part_cols = m.part_cols;
    const int ncols = m.ncols, nobs = m.nobs, ni = m.ni, nblocks = m.nblocks, wg_per_block = m.wg_per_block;
    double *__restrict__ packed = m.packed, *__restrict__ scratch = m.scratch;
    int *status = m.status;
    // --- statistics columns ---
    // scratch[b][c] = sum over the block's workgroup rows, in a fixed order: 8 lanes per (block, column) stride
    // over the rows (the loads of different rows are independent, so they pipeline), then a 3-step butterfly
    for (int base = 0; base < nblocks * ncols; base += blockDim.x / 8) {
        const int idx = base + (int)threadIdx.x / 8, part = threadIdx.x & 7;
        double s = 0.0;
        if (idx < nblocks * ncols) {
            const int b = idx / ncols, c = idx % ncols;
#pragma unroll 8
            for (int w = part; w < wg_per_block; w += 8) s += part_cols[(size_t)(b * wg_per_block + w) * ncols + c];
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (idx < nblocks * ncols && part == 0) scratch[idx] = s;
    }
    __syncthreads();
    const int cnorm = nobs, cneval = nobs + 1, cvis = nobs + 2;
    for (int o = threadIdx.x; o < nobs; o += blockDim.x) {
        double sum = 0.0, sq = 0.0;
#pragma unroll 8
        for (int b = 0; b < nblocks; ++b) {
            const double norm = scratch[b * ncols + cnorm] + 1.0e-10;
            const double m = scratch[b * ncols + o] / norm;
            sum += m;
            sq += m * m;
        }
        packed[o] = sum;
        packed[nobs + o] = sq;
    }
    if (threadIdx.x == 0) {
        double norm = 1.0e-10, neval = 0.0;
        int bad = 0;
#pragma unroll 8
        for (int b = 0; b < nblocks; ++b) {
            const double nb = scratch[b * ncols + cnorm] + 1.0e-10;
            if (!(nb > 0.0)) bad = 1; // main.jl:269-271
            norm += nb;
            neval += scratch[b * ncols + cneval];
        }
        packed[2 * nobs] = norm;
        packed[2 * nobs + 1] = neval;
        if (bad) atomicOr(status, ST_NORMALIZATION);
    }
    for (int i = threadIdx.x; i < ni + 1; i += blockDim.x) {
        double v = 1.0e-8;
#pragma unroll 8
        for (int b = 0; b < nblocks; ++b) v += scratch[b * ncols + cvis + i] + 1.0e-8;
        packed[2 * nobs + 2 + i] = v;
    }
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

// doReweight!  main.jl:322-346 (goal = nullptr: no reweight_goal)
__device__ inline void do_reweight_dev(double *reweight, const double *visited, int nd, double gamma, const double *goal) {
    double avgstep = 0.0;
    for (int i = 0; i < nd; ++i) avgstep += visited[i];
    for (int i = 0; i < nd; ++i) {
        if (visited[i] <= 1) reweight[i] *= pow(avgstep, gamma);
        else reweight[i] *= pow(avgstep / visited[i], gamma);
    }
    if (goal) { // main.jl:334-337
        double gs = 0.0;
        for (int i = 0; i < nd; ++i) gs += goal[i];
        for (int i = 0; i < nd; ++i) reweight[i] *= goal[i] / gs;
    }
    double s = 0.0;
    for (int i = 0; i < nd; ++i) s += reweight[i];
    for (int i = 0; i < nd; ++i) reweight[i] /= s; // main.jl:339
}

// Inclusive prefix sum of v[0..n) into out[0..n) with a fixed summation order: a contiguous chunk per thread,
// a shuffle scan of the chunk totals inside each wave64, then the (<= 16) wave totals; returns the total.
// ps: LDS scratch [>= blockDim.x / 64].
__device__ inline double block_prefix(const double *v, double *out, int n, double *ps) {
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = T >> 6;
    const int per = (n + T - 1) / T, b = tid * per, e = min(n, b + per);
    double loc = 0.0;
    for (int k = b; k < e; ++k) loc += v[k];
    double x = loc; // inclusive scan of loc over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    __syncthreads();
    if (lane == 63) ps[wave] = x;
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < wave; ++w) base += ps[w];
    (void)nwave;
    double run = base + (x - loc); // exclusive prefix of this thread's chunk
    for (int k = b; k < e; ++k) {
        run += v[k];
        out[k] = run;
    }
    __syncthreads();
    return out[n - 1];
}

// LDS doubles train_leaf needs for a leaf of n bins: d[n+kWalkPad] | sg[n+1] | wa[n+kWalkPad] (+ alignment)
enum { kWalkPad = 64 }; // zeros behind d[]: the serial loops read 16 bins at a time, two trips ahead
__host__ __device__ inline int train_lds_doubles(int n) { return (n + kWalkPad) + (n + 2) + (n + kWalkPad) + 2; }

// Julia's sum() over a histogram-length vector (common.jl:72, variable.jl:226) is mapreduce_impl's `@simd` loop below its pairwise
// block size of 1024: a vectorised reduction whose association is the CPU's (lanes x interleave), not left to right.  Oracle and
// device fix the AVX2 shape: 16 interleaved partial sums (element i -> partial i mod 16, each left to right), folded
// p[l] += p[l + h] for h = 8, 4, 2, 1.  Called by every thread of the workgroup; every 16-lane group computes the total for itself.
// Limit of the claim: from 1025 elements on Julia's mapreduce_impl splits the range pairwise at its midpoint before it reaches the
// @simd loop; grids of more than 1025 increments (the default is 999) are summed here -- and in the oracle, mcio_sum16 -- with the
// same 16-lane shape over the whole range, so for them the last bits of f_ninc and of the rescale sum need not be Julia's.
__device__ inline double sum16(const double *v, int n) {
    double s = 0.0;
    int i = threadIdx.x & 15;
    for (; i + 112 < n; i += 128) { // eight loads in flight, then their adds in order (one load per add costs an LDS round trip each)
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[i + 16 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += t[k];
    }
    for (; i < n; i += 16) s += v[i];
#pragma unroll
    for (int h = 8; h >= 1; h >>= 1) s += __shfl_down(s, h, 16);
    return __shfl(s, 0, 16);
}

// b ^ alpha of rescale (common.jl:75).  The learning rates the reference's constructors hand out are small integers (alpha = 2
// by default, variable.jl:137; the bubble example uses 3): for those Julia's `^(::Float64, ::Float64)` takes its
// power-by-squaring path, whose result is the correctly rounded product -- b * b here, one rounding, instead of exp(alpha * log(b)),
// which also was a third of a launch-bound iteration's refinement time.  Any other exponent goes through pow().
__device__ inline double rescale_pow(double b, double alpha) {
    if (alpha == 2.0) return b * b;
    if (alpha == 1.0) return b;
    if (alpha == 3.0) return b * b * b;
    return pow(b, alpha);
}

struct TrainArgs {
    const LeafDev *leaves;
    int nleaf;
    double *packed;
    int nstat;
    double *edges, *dacc, *ddist;
    double *iter_log_row;
    double *reweight;
    const double *goal;
    int nd, do_reweight;
    double gamma;
    int do_train, serial_walk;
    int *status;
};

// Sixteen bins of the refinement walk on ONE lane (variable.jl:228-232, bin-major).  On entry acc = acc_f AFTER bin 0 of the trip was
// consumed; per bin:  rec[j] = acc_f;  while acc_f >= f_ninc: acc_f -= f_ninc;  acc_f += avg_f[j + 1]  (vnext = the first bin of
// the next trip).  Written in ISA because the point is the instruction count and the branch round trips of a chain that one wave
// issues alone: both decisions of a bin (`one new point?`, `a second one?`) are computed before the first branch, and the sums
// each outcome needs (acc_f + next, acc_f - f_ninc + next) are formed while the compare is in flight -- the same floating-point
// operations on the same operands as the reference's loop, only issued early and the unused one dropped.  More than one new
// point per bin (narrow peaks, early iterations) takes the out-of-line loop.   %0 acc_f  %1 a1  %2 second decision  %3 f_ninc
// %4 LDS address of rec[j0]  %5..%20 the trip's bins  %21 the next trip's first bin
#define MCI_WALK_BIN(K, OFF, DN)                                                                                                       \
    "ds_write_b64 %4, %0 offset:" OFF "\n\tv_cmp_ge_f64 vcc, %0, %3\n\tv_add_f64 %1, %0, -%3\n\tv_cmp_ge_f64_e64 %2, %1, %3\n\t"          \
    "v_add_f64 %0, %0, " DN "\n\ts_cbranch_vccz .Lwd" K "_%=\n\tv_add_f64 %0, %1, " DN "\n\ts_cmp_lg_u64 %2, 0\n\t"                      \
    "s_cbranch_scc1 .Lwr" K "_%=\n.Lwd" K "_%=:\n\t"
#define MCI_WALK_MORE(K, DN)                                                                                                           \
    ".Lwr" K "_%=:\n\tv_add_f64 %1, %1, -%3\n\tv_cmp_ge_f64 vcc, %1, %3\n\ts_cbranch_vccnz .Lwr" K "_%=\n\tv_add_f64 %0, %1, " DN "\n\t"   \
    "s_branch .Lwd" K "_%=\n\t"
__device__ __forceinline__ void walk_bins16(double &acc, const double (&v)[16], const double vnext, const double f, const unsigned rec_addr) {
    double a1;
    unsigned long long c2;
    asm volatile(
                 MCI_WALK_BIN("0", "0", "%6")
                 MCI_WALK_BIN("1", "8", "%7")
                 MCI_WALK_BIN("2", "16", "%8")
                 MCI_WALK_BIN("3", "24", "%9")
                 MCI_WALK_BIN("4", "32", "%10")
                 MCI_WALK_BIN("5", "40", "%11")
                 MCI_WALK_BIN("6", "48", "%12")
                 MCI_WALK_BIN("7", "56", "%13")
                 MCI_WALK_BIN("8", "64", "%14")
                 MCI_WALK_BIN("9", "72", "%15")
                 MCI_WALK_BIN("10", "80", "%16")
                 MCI_WALK_BIN("11", "88", "%17")
                 MCI_WALK_BIN("12", "96", "%18")
                 MCI_WALK_BIN("13", "104", "%19")
                 MCI_WALK_BIN("14", "112", "%20")
                 MCI_WALK_BIN("15", "120", "%21")
                 "s_branch .Lwend_%=\n\t"
                 MCI_WALK_MORE("0", "%6")
                 MCI_WALK_MORE("1", "%7")
                 MCI_WALK_MORE("2", "%8")
                 MCI_WALK_MORE("3", "%9")
                 MCI_WALK_MORE("4", "%10")
                 MCI_WALK_MORE("5", "%11")
                 MCI_WALK_MORE("6", "%12")
                 MCI_WALK_MORE("7", "%13")
                 MCI_WALK_MORE("8", "%14")
                 MCI_WALK_MORE("9", "%15")
                 MCI_WALK_MORE("10", "%16")
                 MCI_WALK_MORE("11", "%17")
                 MCI_WALK_MORE("12", "%18")
                 MCI_WALK_MORE("13", "%19")
                 MCI_WALK_MORE("14", "%20")
                 MCI_WALK_MORE("15", "%21")
                 ".Lwend_%=:"
                 : "+v"(acc), "=&v"(a1), "=&s"(c2)
                 : "v"(f), "v"(rec_addr), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
                   "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(vnext)
                 : "vcc", "scc", "memory");
}

// The same sixteen bins without a branch, for trips in which no bin yields more than one new point (the rule once the grid has
// adapted): `acc_f -= f_ninc` runs under the compare's own lane mask (v_cmpx writes EXEC), six instructions per bin.  One wave
// alone issues an instruction every ~4-5 ns whatever it is, so the instruction count is the cost (a variant that forms both outcomes
// ahead of the compare has a shorter chain, one instruction more, and measured the same).  m = the largest acc_f left after a
// subtraction: m >= f_ninc means some bin needed a second one -- the caller then redoes the trip with walk_bins16 from the saved
// acc_f (identical records where both are valid).   %0 acc_f  %1 m  %2 saved EXEC  %3 f_ninc  %4 rec  %5..%21 bins
#define MCI_WALK_BIN1(OFF, DN)                                                                                                         \
    "ds_write_b64 %4, %0 offset:" OFF "\n\tv_cmpx_ge_f64 vcc, %0, %3\n\tv_add_f64 %0, %0, -%3\n\ts_mov_b64 exec, %2\n\t"                  \
    "v_max_f64 %1, %1, %0\n\tv_add_f64 %0, %0, " DN "\n\t"
__device__ __forceinline__ void walk_bins16_single(double &acc, double &m, const double (&v)[16], const double vnext, const double f, const unsigned rec_addr) {
    unsigned long long sv;
    asm volatile("s_mov_b64 %2, exec\n\tv_mov_b64 %1, 0\n\t"
                 MCI_WALK_BIN1("0", "%6")
                 MCI_WALK_BIN1("8", "%7")
                 MCI_WALK_BIN1("16", "%8")
                 MCI_WALK_BIN1("24", "%9")
                 MCI_WALK_BIN1("32", "%10")
                 MCI_WALK_BIN1("40", "%11")
                 MCI_WALK_BIN1("48", "%12")
                 MCI_WALK_BIN1("56", "%13")
                 MCI_WALK_BIN1("64", "%14")
                 MCI_WALK_BIN1("72", "%15")
                 MCI_WALK_BIN1("80", "%16")
                 MCI_WALK_BIN1("88", "%17")
                 MCI_WALK_BIN1("96", "%18")
                 MCI_WALK_BIN1("104", "%19")
                 MCI_WALK_BIN1("112", "%20")
                 MCI_WALK_BIN1("120", "%21")
                 : "+v"(acc), "=&v"(m), "=&s"(sv)
                 : "v"(f), "v"(rec_addr), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
                   "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(vnext)
                 : "vcc", "memory");
}
#undef MCI_WALK_BIN
#undef MCI_WALK_MORE
#undef MCI_WALK_BIN1

__device__ __forceinline__ void walk_trip(double &acc, const double (&v)[16], const double vnext, const double f, const unsigned rec_addr) {
    const double acc0 = acc;
    double m;
    walk_bins16_single(acc, m, v, vnext, f, rec_addr);
    if (__builtin_amdgcn_ballot_w64(!(m < f)) != 0ull) { // some bin of the trip yields two or more points: the general form, from the start of the trip
        acc = acc0;
        walk_bins16(acc, v, vnext, f, rec_addr);
    }
}

// Dist.train! for one leaf by one workgroup, then clearStatistics!.  h: the merged histogram (global or LDS);
// hclear: its home in `packed`, reset for the next iteration.  sm: train_lds_doubles(N) doubles of LDS.
__device__ inline void train_leaf(const LeafDev &L, const double *h, double *hclear, double *sm, double *ps, int &bad, double &ssum,
                                  double *__restrict__ edges, double *__restrict__ dacc, double *__restrict__ ddist, int serial_walk,
                                  int *__restrict__ status) {
    const int tid = threadIdx.x, T = blockDim.x;
    const int N = L.nbin;
    double *d = sm;                     // [N+kWalkPad] smoothed / rescaled distribution, zeros behind it
    double *sg = sm + N + kWalkPad;     // [N+1] old grid staged in LDS
    double *wa = sg + N + 2;            // [N+kWalkPad] scan form: prefix sums; serial form: acc_f after each bin
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int i = tid; i < N; i += T) {
        const double v = h[i];
        if (!isfinite(v)) atomicOr(&bad, ST_HIST_NONFINITE);      // variable.jl:212
        else if (!(v > 0.0)) atomicOr(&bad, ST_HIST_NONPOSITIVE); // variable.jl:213 / common.jl:71
    }
    __syncthreads();
    if (bad) {
        if (tid == 0) atomicOr(status, bad);
        return;
    }
    if (L.kind == 0) {
        double *g = edges + L.eoff;
        for (int i = tid; i <= N; i += T) sg[i] = g[i];
        if (tid < kWalkPad) d[N + tid] = 0.0;
        // smooth(hist, 6)  common.jl:43-54
        for (int i = tid; i < N; i += T) {
            double v;
            if (N <= 1) v = h[i];
            else if (i == 0) v = (h[0] * 7.0 + h[1]) / 8.0;
            else if (i == N - 1) v = (h[N - 1] * 7.0 + h[N - 2]) / 8.0;
            else v = (h[i - 1] + h[i] * 6.0 + h[i + 1]) / 8.0;
            d[i] = v;
        }
        __syncthreads();
        // rescale  common.jl:67-82
        if (N > 1) {
            const double s = sum16(d, N); // :72
            __syncthreads(); // every 16-lane group reads ALL of d[] for its total: nobody overwrites d[] before the last group is through
            for (int i = tid; i < N; i += T) {
                double v = d[i] / s;
                if (v > 0 && v <= 0.99999999) v = rescale_pow(-(1 - v) / log(v), L.alpha);
                if (!isfinite(v)) atomicOr(&bad, ST_RESCALE_NONFINITE); // common.jl:79
                d[i] = v;
            }
            __syncthreads();
            if (bad) {
                if (tid == 0) atomicOr(status, bad);
                return;
            }
        }
        // refinement walk  variable.jl:216-235.  The recurrence on (j, acc_f) is inherently serial and is
        // kept in the reference's order (bit-for-bit the oracle's); lane 0 runs it with a 4-deep register
        // window over d[] so that no LDS latency sits on the dependency chain, and only records (j, acc_f)
        // per new grid point.  The divisions/interpolations (:233) are then done by all lanes.
        if (!serial_walk) {
            // Parallel form of the same walk (default).  With C[j] = sum_{k<=j} avg_f[k] the loop :228-232 leaves,
            // at new grid point i,  j = min{ j : C[j] >= (i-1)*f_ninc }  and  acc_f = C[j] - (i-1)*f_ninc :
            // one fixed-order prefix scan + a bisection per point instead of a 2N-step serial recurrence.
            // Rounding differs from the serial order by O(eps*C[j]/avg_f[j]) of a bin width -- the serial
            // recurrence has the same forward error; device pow/log differ from libm by as much.
            const double total = block_prefix(d, wa, N, ps); // wa[j] = C[j] (inclusive)
            const double f_ninc = total / (double)N;         // :226
            for (int i = tid; i <= N; i += T) {
                double v;
                if (i == 0 || i == N) v = sg[i]; // :217-218, :235
                else {
                    const double target = (double)i * f_ninc;
                    int lo = 0, hi = N - 1; // smallest j0 with C[j0] >= target
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (wa[mid] >= target) hi = mid;
                        else lo = mid + 1;
                    }
                    const double acc_f = wa[lo] - target;
                    v = sg[lo + 1] - (acc_f / d[lo]) * (sg[lo + 1] - sg[lo]); // :233 with j = lo+1
                }
                g[i] = v;
            }
            __syncthreads();
            for (int i = tid; i < N; i += T) hclear[i] = 1.0e-10; // clearStatistics!  variable.jl:238 -> :565
            return;
        }
        // Serial form: the reference's recurrence, floating-point operation for operation (bit-for-bit the oracle's).  Bin-major:
        // consuming avg_f[j] and then emitting new points while acc_f >= f_ninc is the same sequence of operations and decisions
        // as `for i: while acc_f < f_ninc: j += 1; acc_f += avg_f[j]; end; acc_f -= f_ninc` (:227-232).  Lane 0 runs only the
        // chain -- add, compare, subtract -- and records acc_f after each bin (walk_bins16); how many points a bin yields, their
        // acc_f (the same subtractions again), the division and the interpolation (:233) are recomputed from that record by all
        // lanes.  acc_f <= (N + 1) f_ninc, so a subtraction always makes progress.
        const double f_ninc = sum16(d, N) / (double)N; // :226
        if (tid == 0) {
            if (f_ninc > 0.0 && isfinite(f_ninc)) {
                const unsigned rec = (unsigned)(size_t)wa;
                double va[16], vb[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) va[k] = d[k];
                double acc_f = 0.0 + va[0]; // :222, and the first `j += 1; acc_f += avg_f[j]` (:229-230)
                for (int jb = 0; jb < N; jb += 32) { // two trips per turn, the next trip's bins are loaded before this trip's chain
#pragma unroll
                    for (int k = 0; k < 16; ++k) vb[k] = d[jb + 16 + k];
                    walk_trip(acc_f, va, vb[0], f_ninc, rec + 8u * (unsigned)jb);
#pragma unroll
                    for (int k = 0; k < 16; ++k) va[k] = d[jb + 32 + k];
                    walk_trip(acc_f, vb, va[0], f_ninc, rec + 8u * (unsigned)(jb + 16));
                }
            } else {
                atomicOr(status, ST_RESCALE_NONFINITE);
            }
        }
        __syncthreads();
        {
            if (!(f_ninc > 0.0 && isfinite(f_ninc))) return;
            // lane t owns the bins [t*per, (t+1)*per): count their new points, exclusive scan over the lanes, then write them
            const int lane = tid & 63, wave = tid >> 6;
            const int per = (N + T - 1) / T, b = min(N, tid * per), e = min(N, b + per);
            int cnt = 0;
            for (int j = b; j < e; ++j)
                for (double a = wa[j]; a >= f_ninc; a -= f_ninc) cnt += 1;
            int x = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(x, off, 64);
                if (lane >= off) x += y;
            }
            int *pw = (int *)ps;
            if (lane == 63) pw[wave] = x;
            __syncthreads();
            int base = 0, total = 0;
            for (int w = 0; w < (T >> 6); ++w) {
                if (w < wave) base += pw[w];
                total += pw[w];
            }
            int i = 1 + base + x - cnt; // first new grid point of this lane's bins (0-based index into the new grid)
            for (int j = b; j < e; ++j)
                for (double a = wa[j]; a >= f_ninc;) {
                    a -= f_ninc; // :232
                    if (i < N) g[i] = sg[j + 1] - (a / d[j]) * (sg[j + 1] - sg[j]); // :233 (1-based j of the reference = j + 1)
                    i += 1;
                }
            for (int k = 1 + total + tid; k < N; k += T) g[k] = sg[N]; // (points the walk did not reach: rounding at the very end)
            if (tid == 0) {
                g[0] = sg[0]; // :217
                g[N] = sg[N]; // :218, :235
            }
        }
    } else {
        // train!(Discrete)  variable.jl:369-382 : rescale (no smoothing), normalise, prefix sum
        double *acc = dacc + L.eoff, *dist = ddist + L.doff;
        const double s = N > 1 ? sum16(h, N) : 1.0; // rescale's sum(dist), common.jl:72
        if (tid == 0) {
            int lbad = 0;
            if (N > 1) {
                for (int i = 0; i < N; ++i) {
                    double v = h[i] / s;
                    if (v > 0 && v <= 0.99999999) v = rescale_pow(-(1 - v) / log(v), L.alpha);
                    if (!isfinite(v)) lbad = ST_RESCALE_NONFINITE;
                    d[i] = v;
                }
            } else {
                d[0] = h[0];
            }
            if (lbad) {
                atomicOr(status, lbad);
            } else {
                double s = 0.0;
                for (int i = 0; i < N; ++i) s += d[i];
                double run = 0.0;
                acc[0] = 0.0;
                for (int i = 0; i < N; ++i) {
                    const double v = d[i] / s;
                    dist[i] = v;
                    run += v;
                    acc[i + 1] = run;
                }
            }
        }
    }
    // clearStatistics!(T)  variable.jl:238/:381 -> :565 (the next iteration's merge starts from its own fill)
    __syncthreads();
    for (int i = tid; i < N; i += T) hclear[i] = 1.0e-10;
}

// per-iteration bookkeeping by one workgroup: statistics head -> iteration log; doReweight! for the chain solvers
__device__ inline void iteration_bookkeeping(const TrainArgs &a) {
    const int tid = threadIdx.x, T = blockDim.x;
    double *row = a.iter_log_row;
    if (row)
        for (int i = tid; i < a.nstat; i += T) row[i] = a.packed[i];
    if (a.do_reweight && tid == 0) do_reweight_dev(a.reweight, a.packed + (a.nstat - a.nd), a.nd, a.gamma, a.goal);
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
    int maxn = 1;
    for (int l = 0; l < a.nleaf; ++l) maxn = a.leaves[l].nbin > maxn ? a.leaves[l].nbin : maxn;
    double *hl = sm + train_lds_doubles(maxn);
    double *hp = a.packed + a.nstat + L.boff;
    for (int i = tid; i < L.nbin; i += T) {
        const double v = merge_hist_bin(m, L.boff + i);
        hl[i] = v;
        hp[i] = v;
    }
    __syncthreads();
    if (!a.do_train || !L.adapt) return;
    train_leaf(L, hl, hp, sm, ps, bad, ssum, a.edges, a.dacc, a.ddist, a.serial_walk, a.status);
}

} // namespace mci
