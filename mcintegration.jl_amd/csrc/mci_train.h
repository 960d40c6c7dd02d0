// mci_train.h -- the O(bins) tail of an iteration as device functions: block merge (reference src/main.jl:273-287,
// src/configuration.jl:252-262), reweighting (src/main.jl:322-346) and grid refinement (src/distribution/variable.jl:206-239,
// :369-382 with src/distribution/common.jl:43-82).  Compiled twice, like mci_device.h: ahead of time by hipcc into the
// configuration-independent kernels of mci_static_kernels.h (k_finalize / k_train / k_finish), and at run time by hiprtc into the
// persistent :vegas kernel (mci_vegas_persist, below), where the whole iteration loop of a launch-bound integrate() call is ONE launch.
// Free of host / std headers.
#pragma once
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
    double *block_means;     // [nblocks][nobs] or NULL: every block's m = observable / normalization of this iteration (main.jl:275-280),
                             // kept per iteration for the block-lineage error of carried chains (mci_lineage_sums)
    const unsigned long long *hold; // [64] or NULL: the :mcmc holding-time histogram of the launch; its counts follow the tables in `packed`
                                    // (exact doubles), so that they ride in the iteration's one all-reduce
};
// packed = [ ... | hist(nbin) | propose(npa) | accept(npa) ]: the tables ride in the all-reduce like MPIreduceConfig! reduces them
// (configuration.jl:297-298).  One wave per entry, lanes stride over the workgroup rows.
__device__ inline int merge_pa_blocks(const MergeArgs &m) { return (2 * m.npa + 3) / 4; }
__device__ inline void merge_pa(const MergeArgs &m, int blk) {
    const int lane = threadIdx.x & 63, e = blk * 4 + (int)(threadIdx.x >> 6);
    if (blk == 0 && threadIdx.x < 64) // packed = [ ... | accept(npa) | holding-time histogram(64) ]
        m.packed[2 * m.nobs + 2 + m.ni + 1 + m.nbin + 2 * m.npa + (int)threadIdx.x] = m.hold ? (double)m.hold[threadIdx.x] : 0.0;
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
    if (m.use_ghist) { // (use_ghist = the number of buffers the launch spread its atomics over)
        for (int g = 0; g < m.use_ghist; ++g) {
            s += m.ghist[(size_t)g * m.nbin + bin];
            m.ghist[(size_t)g * m.nbin + bin] = 0.0; // ready for the next iteration
        }
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
    double *__restrict__ packed = m.packed, *__restrict__ scratch = m.scratch, *__restrict__ block_means = m.block_means;
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
            if (block_means) block_means[(size_t)b * nobs + o] = m;
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
    const double x = wave_scan_incl(loc); // inclusive scan of loc over the wave
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
__host__ __device__ inline int train_spare_doubles(int n) { return 2 * (2 * n + kWalkPad) + (2 * n + kWalkPad) / 16 + 2; } // the serial walk's slots, their record, the record's heads (train_leaf)

// Julia's sum() over a histogram-length vector (common.jl:72, variable.jl:226) is mapreduce_impl's `@simd` loop below its pairwise
// block size of 1024: a vectorised reduction whose association is the CPU's (lanes x interleave), not left to right.  Oracle and
// device fix ONE association of that family (the lanes x interleave of an AVX2 build; not any particular Julia binary's order to the
// last bit, see oracle/mci_oracle.c mcio_sum16): 16 interleaved partial sums (element i -> partial i mod 16, each left to right), folded
// p[l] += p[l + h] for h = 8, 4, 2, 1.  Called by every thread of the workgroup; every 16-lane group computes the total for itself.
// From 1025 elements on mapreduce_impl splits the range at its midpoint and adds the sums of the two halves: sum_julia below.
__device__ inline double sum16(const double *v, int n) {
    double s = 0.0;
    int i = threadIdx.x & 15;
    for (; i + 496 < n; i += 512) { // 32 loads in flight, then their adds in order (one load per add costs an LDS round trip each)
        double t[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) t[k] = v[i + 16 * k];
#pragma unroll
        for (int k = 0; k < 32; ++k) s += t[k];
    }
    for (; i + 112 < n; i += 128) {
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[i + 16 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += t[k];
    }
    for (; i < n; i += 16) s += v[i];
    s += dpp_read<0x108, 0xf>(s); // p[l] += p[l + 8]   (row_shl:8 -- the rows of the DPP are the 16-lane groups)
    s += dpp_read<0x104, 0xf>(s); // p[l] += p[l + 4]
    s += dpp_read<0x102, 0xf>(s); // p[l] += p[l + 2]
    s += dpp_read<0x101, 0xf>(s); // p[l] += p[l + 1]
    return __shfl(s, 0, 16);
}

// Julia's sum() of a Vector{Float64} of any length (base/reduce.jl mapreduce_impl, pairwise_blocksize = 1024): the @simd block up to
// 1024 elements, above that the halves [ifirst, imid], [imid + 1, ilast] with imid = ifirst + (ilast - ifirst) >> 1, summed the same way
// and added.  The recursion is three deep at the largest grid a workgroup refines (kMaxLeafBins = 4400) and unrolled at compile time.
// (MCI_TRAIN_SHORT_SUMS: a translation unit whose vectors are known to be no longer than 1024 elements -- the persistent kernel of a
// grid of up to 1024 increments -- takes the @simd block alone: the unrolled recursion is fifteen inlined copies of it for the JIT)
#ifdef MCI_TRAIN_SHORT_SUMS
template <int DEPTH = 3> __device__ inline double sum_julia(const double *v, int n) { return sum16(v, n); }
#else
template <int DEPTH = 3> __device__ inline double sum_julia(const double *v, int n) {
    if constexpr (DEPTH == 0) return sum16(v, n);
    else {
        if (n <= 1024) return sum16(v, n);
        const int h = ((n - 1) >> 1) + 1;
        return sum_julia<DEPTH - 1>(v, h) + sum_julia<DEPTH - 1>(v + h, n - h);
    }
}
#endif

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
    int maxn; // bins of the largest leaf (k_finish: where the merged histogram sits behind the refinement's scratch)
    int spare;  // the launch carries train_spare_doubles(maxn) doubles of LDS behind train_lds_doubles(maxn) (else maxn: k_finish's merged histogram)
};

#ifndef MCI_TRAIN_SCAN_ONLY // (the persistent kernel refines with the prefix-scan walk only: the hand-written recurrence stays out of its translation unit)
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

#undef MCI_WALK_BIN
#undef MCI_WALK_MORE

// Sixteen SLOTS of the walk with its decisions given (train_leaf, serial form): acc_f += e[k], one dependent addition per slot.  Lane 0
// records acc_f at the head of every sixteen slots only (`lone`); the values in between are the same additions again, done for all
// trips side by side by as many threads.
template <bool RECORD> __device__ __forceinline__ void walk_slots16(double &acc, const double (&e)[16], double *__restrict__ rec) {
    if (!RECORD) rec[0] = acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (RECORD) rec[k] = acc;
        acc = acc + e[k];
    }
}

// one trip of the general form (the fall-back of the slots below: a decision that did not hold, grids too long for the slots' LDS,
// mci_set_train_walk(prob, 2))
__device__ __forceinline__ void walk_trip(double &acc, const double (&v)[16], const double vnext, const double f, const unsigned rec_addr) {
    walk_bins16(acc, v, vnext, f, rec_addr);
}
#endif // MCI_TRAIN_SCAN_ONLY

enum { kTrainQ = 4 }; // bins a thread works on side by side (999-bin grids on 256 threads: all of a thread's bins)

// the old grid of a Continuous leaf -> its LDS home inside `sm` (and the zeros behind d[]).  A caller that has something else to
// wait for first (the merged histogram) issues this ahead of it and passes staged = true to train_leaf: one memory round trip, not two.
__device__ inline void train_stage_grid(const LeafDev &L, double *sm, const double *__restrict__ edges) {
    if (L.kind != 0) return;
    const int tid = threadIdx.x, T = blockDim.x, N = L.nbin;
    double *d = sm, *sg = sm + N + kWalkPad;
    const double *g = edges + L.eoff;
    for (int base = 0; base <= N; base += kTrainQ * T) {
        double v[kTrainQ];
#pragma unroll
        for (int q = 0; q < kTrainQ; ++q) {
            const int i = base + q * T + tid;
            v[q] = i <= N ? g[i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kTrainQ; ++q) {
            const int i = base + q * T + tid;
            if (i <= N) sg[i] = v[q];
        }
    }
    if (tid < kWalkPad) d[N + tid] = 0.0;
}

// Dist.train! for one leaf by one workgroup, then clearStatistics!.  h: the merged histogram (global or LDS);
// hclear: its home in `packed`, reset for the next iteration (NULL: somebody else's business).  sm: train_lds_doubles(N) doubles of LDS.
__device__ inline void train_leaf(const LeafDev &L, const double *h, double *hclear, double *sm, double *ps, int &bad, double &ssum,
                                  double *__restrict__ edges, double *__restrict__ dacc, double *__restrict__ ddist, int serial_walk,
                                  int *__restrict__ status, bool staged = false, unsigned long long *tt = nullptr, bool checked = false,
                                  double *spare = nullptr) { // spare: train_spare_doubles(N) more doubles of LDS (may start at h: h is not read after the smoothing pass) -- the serial walk's slots
#ifdef MCI_PERSIST_TRACE // development aid (tools/persist_trace.py): wall-clock stamps of the phases, into LDS
#define MCI_TT(k) if (tt && threadIdx.x == 0) tt[k] = wall_clock64();
#else
#define MCI_TT(k)
    (void)tt;
#endif
    const int tid = threadIdx.x, T = blockDim.x;
    const int N = L.nbin;
    double *d = sm;                     // [N+kWalkPad] smoothed / rescaled distribution, zeros behind it
    double *sg = sm + N + kWalkPad;     // [N+1] old grid staged in LDS
    double *wa = sg + N + 2;            // [N+kWalkPad] scan form: prefix sums; serial form: acc_f after each bin
    MCI_TT(0)
    if (!checked) { // (checked: the caller looked at every bin while it merged them, `bad` holds the verdict and a barrier has passed)
        if (tid == 0) bad = 0;
        __syncthreads();
        for (int i = tid; i < N; i += T) {
            const double v = h[i];
            if (!isfinite(v)) atomicOr(&bad, ST_HIST_NONFINITE);      // variable.jl:212
            else if (!(v > 0.0)) atomicOr(&bad, ST_HIST_NONPOSITIVE); // variable.jl:213 / common.jl:71
        }
        __syncthreads();
    }
    if (bad) {
        if (tid == 0) atomicOr(status, bad);
        return;
    }
    // A lone workgroup is latency-bound here: whatever a thread does for its (up to kTrainQ) bins is written as kTrainQ independent
    // chains side by side -- the same operations per bin, issued interleaved -- instead of one bin after the other.
    if (L.kind == 0) {
        double *g = edges + L.eoff;
        MCI_TT(1)
        if (!staged) train_stage_grid(L, sm, edges);
        // smooth(hist, 6)  common.jl:43-54
        double mine = 0.0; // (prefix-scan form: the sum of the smoothed bins rides along, thread by thread, instead of a pass of its own)
        for (int i = tid; i < N; i += T) {
            double v;
            if (N <= 1) v = h[i];
            else if (i == 0) v = (h[0] * 7.0 + h[1]) / 8.0;
            else if (i == N - 1) v = (h[N - 1] * 7.0 + h[N - 2]) / 8.0;
            else v = (h[i - 1] + h[i] * 6.0 + h[i + 1]) / 8.0;
            d[i] = v;
            mine += v;
        }
        if (!serial_walk) {
            const double w = wave_sum(mine);
            if ((tid & 63) == 0) ps[32 + (tid >> 6)] = w;
        }
        __syncthreads();
        MCI_TT(2)
        // rescale  common.jl:67-82
        if (N > 1) {
            // sum(dist) :72 -- the serial form sums in sum_julia's fixed association (one of the family Julia's @simd sum() belongs to: what the walk, bit-for-bit the oracle's, starts from); the prefix-scan form,
            // which rounds differently from the reference's recurrence anyway, takes the waves' partial sums in a fixed order
            double s;
            if (serial_walk) {
                s = sum_julia(d, N);
                __syncthreads(); // every 16-lane group reads ALL of d[] for its total: nobody overwrites d[] before the last group is through
            } else {
                s = 0.0;
                for (int w = 0; w < (T >> 6); ++w) s += ps[32 + w];
            }
            MCI_TT(3)
            int anybad = 0;
            auto rescale_q = [&](auto power, auto QC) { // power(b) = b ^ alpha (rescale_pow, its exponent decided once for the leaf)
                constexpr int Q = decltype(QC)::value;  // chains side by side: as many as a thread has bins (a dummy chain costs a real logarithm)
                for (int base = 0; base < N; base += Q * T) {
                    double v[Q];
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const int i = base + q * T + tid;
                        v[q] = i < N ? d[i] / s : 1.0; // (1.0: left alone by the rescale, finite)
                    }
#pragma unroll
                    for (int q = 0; q < Q; ++q) { // (evaluated for every bin and selected: straight-line code, so the Q logarithms interleave)
                        const double r = power(-(1 - v[q]) / log(v[q]));
                        v[q] = (v[q] > 0 && v[q] <= 0.99999999) ? r : v[q];
                    }
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const int i = base + q * T + tid;
                        if (!isfinite(v[q])) anybad = 1; // common.jl:79
                        if (i < N) d[i] = v[q];
                    }
                }
            };
            auto rescale_bins = [&](auto power) {
#ifdef MCI_TRAIN_SCAN_ONLY // (the persistent kernel: its workgroup size is this translation unit's)
                rescale_q(power, IC<(MCI_THREADS >= 512 ? 2 : kTrainQ)>{});
#else
                if (2 * T >= N) rescale_q(power, IC<2>{}); // (k_finish runs 512 threads on grids of more than 256 increments)
                else rescale_q(power, IC<kTrainQ>{});
#endif
            };
            const double alpha = L.alpha;
            // MCI_TRAIN_POWER (the persistent kernel's translation unit knows its one leaf): 2, 3, 1 = that exponent and nothing else,
            // 4 = pow() alone; undefined = decided here (one instance of the loop per form costs the JIT half a second)
#if defined(MCI_TRAIN_POWER) && MCI_TRAIN_POWER == 2
            rescale_bins([](double b) { return b * b; });
#elif defined(MCI_TRAIN_POWER) && MCI_TRAIN_POWER == 3
            rescale_bins([](double b) { return b * b * b; });
#elif defined(MCI_TRAIN_POWER) && MCI_TRAIN_POWER == 1
            rescale_bins([](double b) { return b; });
#elif defined(MCI_TRAIN_POWER) && MCI_TRAIN_POWER == 4
            rescale_bins([alpha](double b) { return pow(b, alpha); });
#else
            if (alpha == 2.0) rescale_bins([](double b) { return b * b; });
            else if (alpha == 3.0) rescale_bins([](double b) { return b * b * b; });
            else if (alpha == 1.0) rescale_bins([](double b) { return b; });
            else rescale_bins([alpha](double b) { return pow(b, alpha); });
#endif
            if (anybad) atomicOr(&bad, ST_RESCALE_NONFINITE);
            __syncthreads();
            MCI_TT(4)
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
            MCI_TT(5)
            auto place_points = [&](auto QC) {
                constexpr int Q = decltype(QC)::value;
                for (int base = 0; base <= N; base += Q * T) {
                    // smallest j0 with C[j0] >= target, for Q new points at once: a branch-free lower bound whose interval length is the
                    // same on every lane (scalar loop control; per halving one LDS read, one compare, one conditional add per point)
                    int lo[Q];
                    double target[Q];
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        lo[q] = 0;
                        target[q] = (double)(base + q * T + tid) * f_ninc;
                    }
                    for (int len = N; len > 1;) {
                        const int half = len >> 1;
#pragma unroll
                        for (int q = 0; q < Q; ++q) lo[q] += wa[lo[q] + half - 1] < target[q] ? half : 0;
                        len -= half;
                    }
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        lo[q] += wa[lo[q]] < target[q] ? 1 : 0;
                        lo[q] = lo[q] > N - 1 ? N - 1 : lo[q];
                    }
                    double vnew[Q];
#pragma unroll
                    for (int q = 0; q < Q; ++q) { // (straight-line: the Q interpolations interleave; the end points are selected afterwards)
                        const int i = base + q * T + tid, j = lo[q];
                        const double acc_f = wa[j] - target[q];
                        const double v = sg[j + 1] - (acc_f / d[j]) * (sg[j + 1] - sg[j]); // :233 with j = lo+1
                        vnew[q] = (i == 0 || i >= N) ? sg[i < N ? 0 : N] : v;              // :217-218, :235
                    }
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const int i = base + q * T + tid;
                        if (i <= N) g[i] = vnew[q];
                    }
                }
            };
#ifdef MCI_TRAIN_SCAN_ONLY
            place_points(IC<(MCI_THREADS >= 512 ? 2 : kTrainQ)>{});
#else
            if (2 * T > N) place_points(IC<2>{});
            else place_points(IC<kTrainQ>{});
#endif
            __syncthreads();
            MCI_TT(6)
            if (hclear)
                for (int i = tid; i < N; i += T) hclear[i] = 1.0e-10; // clearStatistics!  variable.jl:238 -> :565
            MCI_TT(7)
            return;
        }
#ifndef MCI_TRAIN_SCAN_ONLY
        // Serial form: the reference's recurrence -- its additions and subtractions on its operands in its order (bit-for-bit the oracle's;
        // the two sums it starts from are in ONE fixed association, sum16 above, where Julia's own depends on the CPU).  Bin-major:
        // consuming avg_f[j] and then emitting new points while acc_f >= f_ninc is the same sequence of operations and decisions
        // as `for i: while acc_f < f_ninc: j += 1; acc_f += avg_f[j]; end; acc_f -= f_ninc` (:227-232).  Lane 0 runs only the
        // chain -- add, compare, subtract -- and records acc_f after each bin (walk_bins16); how many points a bin yields, their
        // acc_f (the same subtractions again), the division and the interpolation (:233) are recomputed from that record by all
        // lanes.  acc_f <= (N + 1) f_ninc, so a subtraction always makes progress.
        const double f_ninc = sum_julia(d, N) / (double)N; // :226
        if (!(f_ninc > 0.0 && isfinite(f_ninc))) {
            if (tid == 0) atomicOr(status, ST_RESCALE_NONFINITE);
            return;
        }
        // The chain costs one lone wave ~4.6 ns per instruction it issues, ~8 instructions per bin in the general form (walk_trip): most
        // of them DECIDE (compare, masked subtract, the exec round trip).  The decisions can be had ahead of time: with C[j] the
        // prefix-scan form's partial sums, bin j yields c[j] = floor(C[j] / f_ninc) - floor(C[j-1] / f_ninc) new points -- right wherever
        // acc_f is not within rounding of f_ninc.  With them the recurrence is a list of SLOTS, acc_f += e[s]:
        //     bin j:  c[j] times  `acc_f -= f_ninc`  (e = -f_ninc),  then  `acc_f += avg_f[j + 1]`  (e = avg_f[j + 1])
        // -- the reference's operations on the reference's operands, nothing else.  The last subtraction of a bin is EXACT (the loop
        // :228 stops at acc_f < f_ninc, so it starts from f_ninc <= acc_f < 2 f_ninc: Sterbenz), and for f_ninc / 2 <= avg_f[j + 1] <=
        // 2 f_ninc so is avg_f[j + 1] - f_ninc: then fl(fl(acc_f - f_ninc) + avg_f[j + 1]) = fl(acc_f + (avg_f[j + 1] - f_ninc)) and the two
        // slots are one.  On an adapted grid that is nearly every bin: ~N slots, one dependent addition each (walk_slots16), 1.5
        // instructions per slot with the loads.  Every thread builds its bins' slots, lane 0 walks them recording acc_f at the head of
        // every sixteen, one thread per sixteen slots redoes the same additions for the record in between, every thread then counts its bins' points from the exact record the way the general form does and compares with c[j]: where
        // all agree the record IS the recurrence's (induction over the bins); one disagreement sends the walk through the general
        // form.  serial_walk == 2: the general form at once; 3: slots with one wrong decision (test hook).  spare: [2N + kWalkPad] slots | [2N + kWalkPad] their record | its heads.
        int *flag = (int *)(ps + 64);
        int *pw = (int *)ps;
        const int lane = tid & 63, wave = tid >> 6;
        const int per = (N + T - 1) / T, b = min(N, tid * per), e = min(N, b + per); // thread t owns the bins [t*per, (t+1)*per)
        bool given = spare != nullptr && serial_walk != 2 && N > 1;
        double *es = spare, *rs = spare + 2 * N + kWalkPad, *heads = rs + 2 * N + kWalkPad;
        const double f_inv = 1.0 / f_ninc, f_lo = 0.5 * f_ninc, f_hi = 2.0 * f_ninc;
        auto points = [&](int j) { // c[j] from C[] (in wa[] while `given`); the last bin's decision changes no record: "none", never checked
            if (j >= N - 1) return 0;
            const double c = floor(wa[j] * f_inv) - (j > 0 ? floor(wa[j - 1] * f_inv) : 0.0);
            if (serial_walk == 3 && j == N / 2) return c > 0.0 ? (int)c + 1 : 1; // test hook: one decision deliberately wrong -- the check must catch it
            return c > 0.0 ? (int)c : 0;
        };
        auto slots = [&](int j, int c) { return c + ((c >= 1 && d[j + 1] >= f_lo && d[j + 1] <= f_hi) ? 0 : 1); };
        int sbase = 0, nslot = 0;
        if (given) {
            if (tid == 0) *flag = 0;
            block_prefix(d, wa, N, ps); // wa[j] = C[j]; the barriers inside order the flag's reset
            int mine = 0;
            for (int j = b; j < e; ++j) mine += slots(j, points(j));
            int x = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(x, off, 64);
                if (lane >= off) x += y;
            }
            if (lane == 63) pw[wave] = x;
            __syncthreads();
            for (int w = 0; w < (T >> 6); ++w) {
                if (w < wave) sbase += pw[w];
                nslot += pw[w];
            }
            sbase += x - mine; // (nslot <= N + the points of bins 0 .. N-2 <= 2N - 1)
            int sl = sbase;
            for (int j = b; j < e; ++j) {
                const int c = points(j);
                for (int k = 0; k < c - 1; ++k) es[sl++] = -f_ninc;
                if (c >= 1 && slots(j, c) == c) es[sl++] = d[j + 1] - f_ninc;
                else {
                    if (c >= 1) es[sl++] = -f_ninc;
                    es[sl++] = d[j + 1];
                }
            }
            for (int k = nslot + tid; k < nslot + kWalkPad; k += T) es[k] = 0.0;
            __syncthreads();
        }
        int cnt;
        for (;;) {
            if (tid == 0) {
                double va[16], vb[16];
                double acc_f = 0.0 + d[0]; // :222, and the first `j += 1; acc_f += avg_f[j]` (:229-230)
                if (given) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) va[k] = es[k];
                    for (int sb = 0; sb < nslot; sb += 32) { // two trips per turn, the next trip's slots are loaded before this trip's chain
#pragma unroll
                        for (int k = 0; k < 16; ++k) vb[k] = es[sb + 16 + k];
                        walk_slots16<false>(acc_f, va, heads + (sb >> 4));
#pragma unroll
                        for (int k = 0; k < 16; ++k) va[k] = es[sb + 32 + k];
                        walk_slots16<false>(acc_f, vb, heads + (sb >> 4) + 1);
                    }
                } else {
                    const unsigned rec = (unsigned)(size_t)wa;
#pragma unroll
                    for (int k = 0; k < 16; ++k) va[k] = d[k];
                    for (int jb = 0; jb < N; jb += 32) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) vb[k] = d[jb + 16 + k];
                        walk_trip(acc_f, va, vb[0], f_ninc, rec + 8u * (unsigned)jb);
#pragma unroll
                        for (int k = 0; k < 16; ++k) va[k] = d[jb + 32 + k];
                        walk_trip(acc_f, vb, va[0], f_ninc, rec + 8u * (unsigned)(jb + 16));
                    }
                }
            }
            __syncthreads();
            if (given) { // the record between the heads: sixteen slots per thread
                for (int t = tid; t * 16 < nslot; t += T) {
                    double ev[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) ev[k] = es[t * 16 + k];
                    double acc_f = heads[t];
                    walk_slots16<true>(acc_f, ev, rs + t * 16);
                }
                __syncthreads();
            }
            // count the new points of this thread's bins (the same subtractions again)
            cnt = 0;
            int wrong = 0, sl = sbase;
            for (int j = b; j < e; ++j) {
                int c = 0;
                for (double a = given ? rs[sl] : wa[j]; a >= f_ninc; a -= f_ninc) c += 1;
                cnt += c;
                if (given) {
                    const int cp = points(j);
                    wrong |= (j < N - 1 && c != cp) ? 1 : 0;
                    sl += slots(j, cp);
                }
            }
            if (!given) break;
            if (wrong) atomicOr(flag, 1);
            __syncthreads();
            if (*flag == 0) break;
            given = false; // (nobody reads wa[] or the flag between this barrier and lane 0's second walk)
        }
        if (tid == 0) atomicAdd(status + (given ? 1 : 2), 1); // (mci_debug_walk_counts)
        {
            // exclusive scan of the counts over the lanes, then every lane writes its bins' points
            int x = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(x, off, 64);
                if (lane >= off) x += y;
            }
            if (lane == 63) pw[wave] = x;
            __syncthreads();
            int base = 0, total = 0;
            for (int w = 0; w < (T >> 6); ++w) {
                if (w < wave) base += pw[w];
                total += pw[w];
            }
            int i = 1 + base + x - cnt; // first new grid point of this lane's bins (0-based index into the new grid)
            int sl = sbase;
            for (int j = b; j < e; ++j) {
                for (double a = given ? rs[sl] : wa[j]; a >= f_ninc;) {
                    a -= f_ninc; // :232
                    if (i < N) g[i] = sg[j + 1] - (a / d[j]) * (sg[j + 1] - sg[j]); // :233 (1-based j of the reference = j + 1)
                    i += 1;
                }
                if (given) sl += slots(j, points(j));
            }
            for (int k = 1 + total + tid; k < N; k += T) g[k] = sg[N]; // (points the walk did not reach: rounding at the very end)
            if (tid == 0) {
                g[0] = sg[0]; // :217
                g[N] = sg[N]; // :218, :235
            }
        }
#endif // MCI_TRAIN_SCAN_ONLY
    }
#ifndef MCI_TRAIN_CONTINUOUS_ONLY // (the persistent kernel refines one Continuous grid: the Discrete form and its pow() stay out of its translation unit)
    else {
        // train!(Discrete)  variable.jl:369-382 : rescale (no smoothing), normalise, prefix sum
        double *acc = dacc + L.eoff, *dist = ddist + L.doff;
        const double s = N > 1 ? sum_julia(h, N) : 1.0; // rescale's sum(dist), common.jl:72
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
#endif
    // clearStatistics!(T)  variable.jl:238/:381 -> :565 (the next iteration's merge starts from its own fill)
    __syncthreads();
    if (hclear)
        for (int i = tid; i < N; i += T) hclear[i] = 1.0e-10;
#undef MCI_TT
}

// per-iteration bookkeeping by one workgroup: statistics head -> iteration log; doReweight! for the chain solvers
__device__ inline void iteration_bookkeeping(const TrainArgs &a) {
    const int tid = threadIdx.x, T = blockDim.x;
    double *row = a.iter_log_row;
    if (row)
        for (int i = tid; i < a.nstat; i += T) row[i] = a.packed[i];
#ifndef MCI_TRAIN_CONTINUOUS_ONLY // (:vegas never reweights, main.jl:183)
    if (a.do_reweight && tid == 0) do_reweight_dev(a.reweight, a.packed + (a.nstat - a.nd), a.nd, a.gamma, a.goal);
#endif
}

// =============================================================================================
// Persistent :vegas iterations: the whole loop of integrate() (main.jl:142-207) as ONE launch
// =============================================================================================
// At the reference's own sizes (neval = 1e4 .. 1e5 per iteration, main.jl:76) an iteration is a microsecond of sampling and ~8 us of
// train! on one CU; as a chain of launches it also pays two dependent dispatches with their ramps and the idle queue between them.
// For the reference's most common shape -- ONE Continuous variable type, i.e. one adaptive grid shared by all dimensions -- a grid of
// G <= 256 co-resident sampling workgroups plus one statistics workgroup stays on the chip for all `niter` iterations:
//
//     sampling workgroup   build the pair table from ITS OWN copy of the map (LDS), sample its slice (vegas_batch: the same code, the
//                          same Philox indices, the same sums), flush its partial row (double-buffered by the turn's parity) and -- with
//                          global f64 atomics -- its histogram into buffer `turn % 3`                                    ->  arrive
//                          once all G have arrived: read the merged histogram and run train! on it ITSELF (prefix-scan walk) -- every
//                          workgroup refines its own copy of the map with the same arithmetic on the same numbers, so the copies stay
//                          bit-identical and nobody waits for a refined map to travel through HBM: one grid-wide wait per turn
//     workgroup 0          writes its copy of the map back to HBM when the last turn is through (what the host reads, what the next launch starts from)
//     statistics workgroup once all G have arrived: block merge -> statistics head -> iteration log; zeroes the histogram buffer of the
//                          turn before; in the last turn leaves the merged histogram / clearStatistics! state in `packed`         ->  done
//
// The waits are counters in HBM that only grow (targets are computed from the launch's starting values; nothing is reset between
// launches); a workgroup publishes with  every wave's s_waitcnt vmcnt(0) -> barrier -> agent-scope release fence + relaxed atomic add by
// its first thread  and consumes with  relaxed atomic load -> agent-scope acquire fence -> barrier  (those fences write back / invalidate the XCD's L2).
// Three histogram buffers: buffer t % 3 is added to in turn t, read by everybody after the arrive of turn t, zeroed by the statistics
// workgroup after the arrive of turn t + 1 (all its readers have arrived there) and next added to in turn t + 3.  Two partial-row buffers:
// the rows of turn t are read by the statistics workgroup, which every sampling workgroup checks has finished turn t - 1 -- rows merged,
// buffer zeroed -- before it leaves the arrive of turn t.  (One poller per workgroup: the first lane of every wave polling out of step
// was tried and was slower, 15.2 against 13.6 us per iteration -- the polls queue at the memory side with the arrivals' own atomics.)  Residency: the host launches no more workgroups than fit the chip at once next to another such grid; a wait is
// bounded in wall-clock time and a stall (ST_PERSIST_STALL) ends the launch with an error, not a hang.
struct PersistArgs {
    MergeArgs m;              // (use_ghist = 1; part_pa = NULL; part_cols: [2][G][ncols]; ghist: [3][nbin])
    TrainArgs t;              // t.iter_log_row: the log row of this launch's FIRST iteration
    int niter;
    int map_off;              // doubles: LDS behind BOTH the sample loop's carve and the refinement's (train scratch [train_lds_doubles(N)] |
                              // merged histogram [N] | scan scratch [256], from address 0: each is dead while the other runs):
                              // this workgroup's map [N + 2] | flags [4]
    u64 *ctr;                 // [0] low 40 bits: sampling workgroups that have flushed / finished reading; high 24 bits: turns the statistics
                              // workgroup has done (one word: one load per poll); [2] != 0: a wait gave up
    u64 arrive0, done0;       // the two counts when this launch starts
    u64 spin_ticks;           // bound of one wait, in wall_clock64() ticks (100 MHz)
};

enum : u64 { kPersistDoneShift = 40, kPersistArriveMask = (1ull << 40) - 1 };
// thread 0 polls until `arrived` >= t0 and `done` >= t1 (mod 2^24); false (for every thread of the workgroup) when the launch is to be abandoned
__device__ __forceinline__ bool persist_wait(const PersistArgs &f, u64 t0, u64 t1, int *verdict) {
    if (threadIdx.x == 0) {
        int ok = 1;
        const u64 start = wall_clock64();
        unsigned n = 0;
        for (;;) {
            const u64 w = __hip_atomic_load(&f.ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (the 24-bit turn count wraps: compared as a signed distance)
            if ((w & kPersistArriveMask) >= t0 && (long long)(((w >> kPersistDoneShift) - t1) << kPersistDoneShift) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
            // (the clock is looked at every 256th poll; a limit of a few ticks -- the test hook of csrc/mci_debug.h -- at every poll)
            if (((++n & 255u) == 0u || f.spin_ticks < 256ull) &&
                (__hip_atomic_load(&f.ctr[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull || wall_clock64() - start > f.spin_ticks)) {
                ok = 0;
                break;
            }
        }
        if (!ok) {
            __hip_atomic_store(&f.ctr[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(f.m.status, ST_PERSIST_STALL);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *verdict = ok;
    }
    __syncthreads();
    return *verdict != 0;
}
// Publish: EVERY wave first waits until its own stores and atomics have been performed (s_waitcnt vmcnt(0): a workgroup barrier alone
// orders global memory only through the CU's in-order path to each L2 channel, and the counter lives on another channel than the data;
// gfx9 counts stores and atomics without return in vmcnt), then the barrier, then the first thread writes the XCD's L2 back
// (agent-scope release fence: every wave's stores have reached it) and counts the workgroup in.  (A release fence by every wave instead
// -- four buffer_wbl2 per workgroup -- costs 1.3 us more per turn.)
__device__ __forceinline__ void persist_signal(u64 *ctr, u64 inc) {
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <class Cfg> __device__ __forceinline__ void vegas_persist(const BatchArgs &a0, const PersistArgs &f) {
    static_assert(Cfg::NLEAF == 1 && Cfg::leaf_kind(0) == 0, "the persistent kernel refines ONE Continuous grid (the host checks)");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, T = blockDim.x;
    constexpr int N = Cfg::leaf_nbin(0);
    double *sm = smem, *hl = sm + train_lds_doubles(N), *ps = hl + N, *gcur = smem + f.map_off, *flags = gcur + N + 2;
    int *verdict = reinterpret_cast<int *>(flags + 2);
    const u64 G = gridDim.x - 1; // sampling workgroups; workgroup G merges the statistics
    const int wg = (int)blockIdx.x;
    const size_t rows = (size_t)G * f.m.ncols;
#ifdef MCI_PERSIST_TRACE // development aid (tools/persist_trace.py): wall-clock stamps (10 ns) of workgroups 0, G - 1 and G at every phase of the first 8 turns
#define MCI_PT(k)                                                                                                                      \
    if (tid == 0 && it < 8 && (wg == 0 || wg == (int)G - 1 || wg == (int)G)) {                                                          \
        f.ctr[8 + ((wg == 0 ? 0 : wg == (int)G ? 1 : 2) * 8 + it) * 8 + (k)] = wall_clock64();                                          \
        if (wg == 0 && (k) == 0 && (it == 0 || it == 7)) f.ctr[8 + 3 * 8 * 8 + 8 + (it == 0 ? 0 : 1)] = clock64(); /* shader clock */  \
    }
#else
#define MCI_PT(k)
#endif
    const LeafDev L = f.t.leaves[0];
    const bool train = f.t.do_train && L.adapt; // variable.jl:208
    if (wg == (int)G) { // ---- the statistics workgroup (and everything else that is nobody's critical path)
        for (int it = 0; it < f.niter; ++it) {
            MCI_PT(0)
            if (!persist_wait(f, f.arrive0 + (u64)(it + 1) * G, f.done0 + (u64)it, verdict)) return;
            MCI_PT(3)
            MergeArgs m = f.m;
            m.part_cols = f.m.part_cols + (size_t)(it & 1) * rows;
            merge_stats(m); // main.jl:273-287
            // config.propose / config.accept of a :vegas iteration: the clearStatistics! offsets alone (configuration.jl:247-248)
            for (int e = tid; e < 2 * m.npa; e += T)
                m.packed[2 * m.nobs + 2 + m.ni + 1 + m.nbin + e] = (double)(m.nblocks + 1) * (e < m.npa ? 1.0e-8 : 1.0e-10);
            if (it > 0) { // the histogram buffer of the turn before: everybody who read it has arrived again
                double *gz = a0.ghist + (size_t)((it + 2) % 3) * Cfg::NBIN + L.boff;
                for (int i = tid; i < N; i += T) gz[i] = 0.0;
            }
            if (it == f.niter - 1) { // what the launch leaves in `packed`: the merged histogram, or train!'s clearStatistics! (variable.jl:238 -> :565)
                const double *gh = a0.ghist + (size_t)(it % 3) * Cfg::NBIN + L.boff;
                double *hp = f.t.packed + f.t.nstat + L.boff;
                for (int i = tid; i < N; i += T)
                    hp[i] = train ? 1.0e-10 : (double)(m.nblocks + 1) * 1.0e-10 + __hip_atomic_load(&gh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads(); // the head of `packed` was written by this workgroup
            TrainArgs t = f.t;
            if (t.iter_log_row) t.iter_log_row += (size_t)it * t.nstat;
            iteration_bookkeeping(t);
            __syncthreads();
            MCI_PT(4)
            persist_signal(&f.ctr[0], 1ull << kPersistDoneShift);
            MCI_PT(5)
        }
        // the last turn's histogram buffer: zeroed once everybody has read it
        if (!persist_wait(f, f.arrive0 + (u64)(f.niter + 1) * G, f.done0, verdict)) return;
        double *gz = a0.ghist + (size_t)((f.niter - 1) % 3) * Cfg::NBIN + L.boff;
        for (int i = tid; i < N; i += T) gz[i] = 0.0;
        return;
    }
    // ---- a sampling workgroup
    for (int i = tid; i <= N; i += T) gcur[i] = f.t.edges[L.eoff + i]; // this workgroup's copy of the map
    for (int it = 0; it < f.niter; ++it) {
        BatchArgs a = a0;
        a.iteration = a0.iteration + (u32)it;
        a.edges = gcur - L.eoff;                              // (LDS through the generic address space: stage_tables reads it once)
        a.part_cols = a0.part_cols + (size_t)(it & 1) * rows;
        a.ghist = a0.ghist + (size_t)(it % 3) * Cfg::NBIN;
        MCI_PT(0)
        if (tid == 0) *reinterpret_cast<int *>(flags) = 0; // train_leaf's `bad`
        __syncthreads(); // (gcur complete; the previous turn's scratch is free)
        vegas_batch<Cfg, false>(a); // pair table <- gcur, samples, partial row, histogram atomics
        __syncthreads();
        MCI_PT(1)
        persist_signal(&f.ctr[0], 1ull);
        MCI_PT(2)
        // all G rows and histograms of this turn are out; the statistics workgroup is through with the turn before (its rows, and the
        // histogram buffer it zeroed)
        if (!persist_wait(f, f.arrive0 + (u64)(it + 1) * G, f.done0 + (u64)it, verdict)) return;
        MCI_PT(3)
        const double *gh = a.ghist + L.boff;
        int hbad = 0; // train!'s checks of the histogram ride along (train_leaf(..., checked = true)); `bad` was cleared before the arrive
        for (int base = 0; base < N; base += kTrainQ * T) { // merge_hist_bin: clearStatistics! offsets + what the workgroups added
            double v[kTrainQ];
#pragma unroll
            for (int q = 0; q < kTrainQ; ++q) {
                const int i = base + q * T + tid;
                v[q] = i < N ? __hip_atomic_load(&gh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < kTrainQ; ++q) {
                const int i = base + q * T + tid;
                if (i < N) {
                    const double h = (double)(f.m.nblocks + 1) * 1.0e-10 + v[q];
                    hl[i] = h;
                    if (!isfinite(h)) hbad |= ST_HIST_NONFINITE;      // variable.jl:212
                    else if (!(h > 0.0)) hbad |= ST_HIST_NONPOSITIVE; // variable.jl:213 / common.jl:71
                }
            }
        }
        if (hbad) atomicOr(reinterpret_cast<int *>(flags), hbad);
        __syncthreads();
#ifdef MCI_PERSIST_TRACE
        u64 *tt = (wg == 0 && it == 7) ? reinterpret_cast<u64 *>(ps + 200) : nullptr; // (stamps go to LDS: a global store would be waited for at the next barrier)
#else
        u64 *tt = nullptr;
#endif
        if (train) train_leaf(L, hl, nullptr, sm, ps, *reinterpret_cast<int *>(flags), flags[1], gcur - L.eoff, f.t.dacc, f.t.ddist, 0, f.t.status, false, tt, true);
        __syncthreads();
#ifdef MCI_PERSIST_TRACE
        if (tt && tid < 8) f.ctr[8 + 3 * 8 * 8 + tid] = tt[tid];
#endif
        MCI_PT(4)
    }
    __syncthreads();
    persist_signal(&f.ctr[0], 1ull); // "through with the last turn's histogram": the statistics workgroup zeroes it
    if (wg == 0 && train)
        for (int i = tid; i <= N; i += T) f.t.edges[L.eoff + i] = gcur[i]; // the refined map, for the host and the next launch
#undef MCI_PT
}

} // namespace mci
