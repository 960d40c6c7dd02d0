// mci_jit.h -- run-time specialisation of the sample-batch kernels (host side).
//
// The reference gets its per-problem specialisation from Julia's JIT: Vegas.montecarlo is compiled for
// each Configuration{N,V,P,O,T} and the integrand closure is inlined (src/vegas/montecarlo.jl:72-75,
// :140-144).  Here a small translation unit -- a `Cfg` traits struct holding the problem's static
// shape and the user's integrand body -- is generated, compiled with hiprtc for gfx950 against the
// hand-written kernels of mci_device.h, and cached on disk as a code object.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <link.h>
#include <sys/stat.h>
#include <unistd.h>

namespace mcijit {

// text of mci_device.h, embedded at build time (see __graft_entry__.build)
extern const char *const kDeviceHeader;
// text of mci_train.h (merge + train! device functions and the persistent :vegas kernel): the second header of a kUnitVegasPersist unit
extern const char *const kTrainHeader;
// text of mci_spec.h (the chain solvers with several lanes per chain): the second header of a kUnitSpec unit
extern const char *const kSpecHeader;

struct ProblemShape {
    int ndraw = 0, nleaf = 0, ni = 0, npool = 0, nobs = 0, ncols = 0, table_mode = 0;
    int nedge = 0, ndacc = 0, nddist = 0, nbin = 0, pair_table = 0, npair = 0, ntile = 1, htile = 0;
    std::vector<int> leaf_tile, tile_boff, tile_nbin; // histogram tiles (contiguous leaves)
    std::vector<int> draw_leaf, draw_pool, draw_slot;
    std::vector<int> leaf_kind, leaf_nbin, leaf_eoff, leaf_doff, leaf_boff, leaf_adapt, leaf_poff;
    std::vector<double> leaf_lower, leaf_upper;
    std::vector<unsigned long long> own_mask;   // [ni+1] draws covered by integrand i (last = normalisation: 0)
    std::vector<unsigned long long> cover_mask; // [ndraw] integrands covering draw k
    std::vector<int> obs_off, obs_nbin, obs_bin_draw;
    std::vector<int> pool_maxdof, pool_nleaf, pool_first_draw;
    // vegas with NTILE > 1 ("split-all"): the sample pass keeps NO histogram, caches the edges of the leading leaves in the
    // LDS the histogram tile would take (leaf_ecoff >= 0: offset in that cache, doubles), and every tile is replayed
    int split_all = 0, ec_doubles = 0;
    int rng_rounds = 10; // Philox4x32 rounds of every stream: 10 (default) or 7 (mci_set_rng_rounds)
    int rng_bits = 52;  // :vegas sample stream: 52 random mantissa bits per draw (two draws per Philox block) or 32 (four per block)
    int l1_phase = 0; // :vegas, grids gathered from global memory: samples per lane and trip of the dimension-major gather phase (0 = off)
    std::vector<int> leaf_ecoff;
    std::vector<int> dof;                 // [(ni+1)*npool] incl. the normalisation row (zeros)
    int nbmax = 1;
    int ncomp = 1;            // 1: Float64 weights; 2: ComplexF64 stored (re, im)
    std::string measure_body; // user measure (empty = default / bin-by-Discrete)
    int host_integrand = 0;   // weights come from a host callback: over dumped draws (vegas), per Markov step (vegasmc)
    int hcopy = 1;            // :vegas sample kernel: interleaved copies of the LDS histograms (mci_device.h hslot), power of two
    int host_measure = 0;     // observables are accumulated by a host callback over the launch's (measured) configurations and relative weights
    int det = 0;              // deterministic mode: hcopy = waves per workgroup, one histogram / observable copy per wave, for every solver
    std::vector<int> nneighbor, neighbor; // [ni+1], [(ni+1)*nbmax] 0-based, padded with the integrand itself
    std::string body;
};

template <class T> static std::string arr(const std::vector<T> &v, const char *ty, const char *suffix = "") {
    std::ostringstream o;
    o.precision(17);
    o << "{";
    for (size_t i = 0; i < v.size(); ++i) o << (i ? ", " : "") << v[i] << suffix;
    if (v.empty()) o << "0";
    o << "}";
    (void)ty;
    return o.str();
}

static std::string fn_table(const char *ret, const char *name, const std::string &values) {
    std::ostringstream o;
    o << "    static constexpr " << ret << " " << name << "(int i) { constexpr " << ret << " t[] = " << values
      << "; return t[i]; }\n";
    return o.str();
}

static std::string dbl_arr(const std::vector<double> &v) {
    std::ostringstream o;
    o << "{";
    char buf[64];
    for (size_t i = 0; i < v.size(); ++i) {
        snprintf(buf, sizeof buf, "%a", v[i]); // hex float: exact round trip
        o << (i ? ", " : "") << buf;
    }
    if (v.empty()) o << "0.0";
    o << "}";
    return o.str();
}

// solver: 0 vegas, 1 vegasmc, 2 mcmc.  unit: what the translation unit holds -- the solver's kernel (for :vegas: the loop for any
// measurefreq), the :vegas kernel specialised on measurefreq == 1, or the sample-dump kernel alone; each is built on first use
// kUnitVegasPersist: the :vegas loop for measurefreq == 1 inside the persistent kernel of mci_train.h (all iterations of a launch-bound
// integrate() call in one launch): sample loop + block merge + train!
// kUnitSpec: a chain solver's kernel with several lanes per chain (mci_spec.h: vegasmc_chains_spec / mcmc_chains_spec)
enum { kUnitSolver = 0, kUnitVegasMf1 = 1, kUnitDump = 2, kUnitVegasPersist = 3, kUnitSpec = 4 };
// which headers a unit is compiled against next to mci_device.h
enum { kHdrNone = 0, kHdrTrain = 1, kHdrSpec = 2 };
inline std::string generate_source(const ProblemShape &s, int solver, int unit = kUnitSolver, double persist_alpha = 0.0) {
    std::ostringstream o;
    if (unit == kUnitVegasPersist) { // the learning rate and the size of the one leaf the persistent kernel refines (mci_train.h rescale, sum_julia)
        o << "#define MCI_TRAIN_POWER " << (persist_alpha == 2.0 ? 2 : persist_alpha == 3.0 ? 3 : persist_alpha == 1.0 ? 1 : 4) << "\n";
        if (!s.leaf_nbin.empty() && s.leaf_nbin[0] <= 1024) o << "#define MCI_TRAIN_SHORT_SUMS 1\n";
    }
    if (solver == 0 && unit != kUnitDump) o << "#define MCI_MF_ONLY " << (unit == kUnitVegasMf1 || unit == kUnitVegasPersist ? 1 : 0) << "\n";
    if (unit == kUnitVegasPersist) o << "#define MCI_TRAIN_SCAN_ONLY 1\n#define MCI_TRAIN_CONTINUOUS_ONLY 1\n";
    if (s.rng_rounds != 10) o << "#define MCI_PHILOX_ROUNDS " << s.rng_rounds << "\n"; // opt-in cheaper stream (mci_set_rng_rounds)
    o << "#include \"mci_device.h\"\n";
    if (unit == kUnitVegasPersist) o << "#include \"mci_train.h\"\n";
    if (unit == kUnitSpec) o << "#include \"mci_spec.h\"\n";
    o << "#ifndef M_PI\n#define M_PI 3.14159265358979323846\n#endif\n";
    o << "#ifndef MCI_CHAIN_KERNEL_ATTR\n#define MCI_CHAIN_KERNEL_ATTR\n#endif\n"; // (occupancy experiments on the lane-per-chain kernels: MCI_JIT_FLAGS=-DMCI_CHAIN_KERNEL_ATTR=...)
    o << "namespace {\nstruct Cfg {\n";
    o << "    static constexpr int NDRAW = " << s.ndraw << ", NLEAF = " << s.nleaf << ", NI = " << s.ni
      << ", NPOOL = " << s.npool << ", NOBS = " << s.nobs << ", NCOLS = " << s.ncols << ";\n";
    o << "    static constexpr int TABLE_MODE = " << s.table_mode << ", NEDGE = " << s.nedge << ", NDACC = " << s.ndacc
      << ", NDDIST = " << s.nddist << ", NBIN = " << s.nbin << ", PAIR_TABLE = " << s.pair_table << ", NPAIR = " << s.npair << ";\n";
    o << fn_table("int", "draw_leaf", arr(s.draw_leaf, "int"));
    o << fn_table("int", "draw_pool", arr(s.draw_pool, "int"));
    o << fn_table("int", "draw_slot", arr(s.draw_slot, "int"));
    o << fn_table("int", "leaf_kind", arr(s.leaf_kind, "int"));
    o << fn_table("int", "leaf_nbin", arr(s.leaf_nbin, "int"));
    o << fn_table("int", "leaf_eoff", arr(s.leaf_eoff, "int"));
    o << fn_table("int", "leaf_doff", arr(s.leaf_doff, "int"));
    o << fn_table("int", "leaf_boff", arr(s.leaf_boff, "int"));
    o << fn_table("int", "leaf_adapt", arr(s.leaf_adapt, "int"));
    o << fn_table("int", "leaf_poff", arr(s.leaf_poff, "int"));
    o << "    static constexpr int NTILE = " << s.ntile << ", HTILE = " << s.htile << ", HCOPY = " << (((solver == 0 || s.det) && s.hcopy > 0) ? s.hcopy : 1) << ", DET = " << (s.det ? 1 : 0) << ";\n";
    o << "    static constexpr int SPLIT_ALL = " << (solver == 0 ? s.split_all : 0) << ", EC_DOUBLES = " << (solver == 0 ? s.ec_doubles : 0)
      << ", L1_PHASE = " << (solver == 0 ? s.l1_phase : 0) << ", RNG_BITS = " << (solver == 0 ? s.rng_bits : 52) << ";\n";
    {
        std::vector<int> ec = s.leaf_ecoff;
        if (solver != 0 || s.ec_doubles <= 0 || ec.size() != s.leaf_kind.size()) ec.assign(s.leaf_kind.size(), -1);
        o << fn_table("int", "leaf_ecoff", arr(ec, "int"));
    }
    o << fn_table("int", "leaf_tile", arr(s.leaf_tile, "int"));
    o << fn_table("int", "tile_boff", arr(s.tile_boff, "int"));
    o << fn_table("int", "tile_nbin", arr(s.tile_nbin, "int"));
    o << fn_table("double", "leaf_lower", dbl_arr(s.leaf_lower));
    o << fn_table("double", "leaf_upper", dbl_arr(s.leaf_upper));
    o << fn_table("unsigned long long", "own_mask", arr(s.own_mask, "u64", "ull"));
    o << fn_table("unsigned long long", "cover_mask", arr(s.cover_mask, "u64", "ull"));
    o << fn_table("int", "obs_off", arr(s.obs_off, "int"));
    o << fn_table("int", "obs_nbin", arr(s.obs_nbin, "int"));
    o << fn_table("int", "obs_bin_draw", arr(s.obs_bin_draw, "int"));
    o << fn_table("int", "pool_maxdof", arr(s.pool_maxdof, "int"));
    o << fn_table("int", "pool_nleaf", arr(s.pool_nleaf, "int"));
    o << fn_table("int", "pool_first_draw", arr(s.pool_first_draw, "int"));
    o << "    static constexpr int NBMAX = " << s.nbmax << ", NCOMP = " << s.ncomp << ", NW = " << s.ni * s.ncomp
      << ", CUSTOM_MEASURE = " << (s.measure_body.empty() ? 0 : 1) << ", HOST_INTEGRAND = " << s.host_integrand
      << ", HOST_MEASURE = " << s.host_measure << ";\n";
    o << fn_table("int", "dof", arr(s.dof, "int"));
    o << fn_table("int", "nneighbor", arr(s.nneighbor, "int"));
    o << fn_table("int", "neighbor", arr(s.neighbor, "int"));
    o << "    // the user's integrand (reference: the `integrand` closure, vegas/montecarlo.jl:140-144)\n";
    o << "    // idx: the one output that is needed (mcmc: `integrand(idx, var, config)`, mcmc/montecarlo.jl:34), -1 = all\n";
    o << "    static __device__ __forceinline__ void integrand(const double* __restrict__ x, double* __restrict__ w, "
         "const double* __restrict__ ud, const int idx) {\n    (void)idx;\n"
      << s.body << "\n    }\n";
    o << "    // the user's measure (vegas/montecarlo.jl:156-161; mcmc/montecarlo.jl:166-169): rw = relative weights\n"
         "    // [NI*NCOMP], idx = -1 (all integrands) or the integrand an mcmc chain sits on; obs_add(k, v) accumulates\n";
    o << "    static __device__ __forceinline__ void measure(const double* __restrict__ x, const double* __restrict__ rw, "
         "const double* __restrict__ ud, const int idx, double* __restrict__ mci_obs_) {\n"
         "    (void)x; (void)rw; (void)ud; (void)idx; (void)mci_obs_;\n"
         "#define obs_add(k, v) mci::lds_add(&mci_obs_[(k)], (v))\n"
      << s.measure_body << "\n#undef obs_add\n    }\n";
    o << "};\n}\n";
    if (solver == 0 && unit == kUnitDump) {
        o << "extern \"C\" __global__ void __launch_bounds__(256) mci_sample_dump(mci::DumpArgs a) { "
             "mci::sample_dump<Cfg>(a); }\n";
    } else if (solver == 0 && unit == kUnitVegasPersist) {
        o << "extern \"C\" __global__ void __launch_bounds__(MCI_THREADS) mci_vegas_persist(mci::BatchArgs a, mci::PersistArgs f) { "
             "mci::vegas_persist<Cfg>(a, f); }\n";
    } else if (solver == 0) {
        o << "extern \"C\" __global__ void __launch_bounds__(MCI_THREADS) mci_vegas_batch(mci::BatchArgs a) { "
             "mci::vegas_batch<Cfg, (Cfg::NTILE > 1)>(a); }\n";
        if (s.ntile > 1)
            o << "extern \"C\" __global__ void __launch_bounds__(MCI_THREADS) mci_vegas_tiles(mci::BatchArgs a) { "
                 "mci::vegas_tiles<Cfg>(a); }\n";
    } else if (unit == kUnitSpec) {
        o << "extern \"C\" __global__ void __launch_bounds__(MCI_THREADS) " << (solver == 1 ? "mci_vegasmc_spec" : "mci_mcmc_spec") << "(mci::BatchArgs a) { "
          << (solver == 1 ? "mci::vegasmc_chains_spec<Cfg>(a); }\n" : "mci::mcmc_chains_spec<Cfg>(a); }\n");
        if (solver == 1)
            o << "extern \"C\" __global__ void __launch_bounds__(MCI_THREADS) mci_vegasmc_carry_weights(mci::BatchArgs a) { mci::vegasmc_carry_weights<Cfg>(a); }\n";
    } else if (solver == 1) {
        // (a host integrand: the step cut at the integrand call, one launch per Markov step -- same entry point)
        o << "extern \"C\" __global__ void __launch_bounds__(MCI_THREADS) MCI_CHAIN_KERNEL_ATTR mci_vegasmc_chains(mci::BatchArgs a) { "
          << (s.host_integrand ? "mci::vegasmc_host_step<Cfg>(a); }\n" : "mci::vegasmc_chains<Cfg>(a); }\n");
        // (carried chains: the new target over the old one at every stored configuration, before they are resampled)
        o << "extern \"C\" __global__ void __launch_bounds__(MCI_THREADS) mci_vegasmc_carry_weights(mci::BatchArgs a) { mci::vegasmc_carry_weights<Cfg>(a); }\n";
    } else {
        o << "extern \"C\" __global__ void __launch_bounds__(MCI_THREADS) MCI_CHAIN_KERNEL_ATTR mci_mcmc_chains(mci::BatchArgs a) { "
          << (s.host_integrand ? "mci::mcmc_host_step<Cfg>(a); }\n" : "mci::mcmc_chains<Cfg>(a); }\n");
    }
    return o.str();
}

// .private_segment_fixed_size (scratch bytes per work-item: register spills, stack) of one kernel, read from the code object's
// AMDGPU metadata note (msgpack; LLVM writes a kernel's keys in sorted order, so the first such key after `.name <kernel>` is that
// kernel's).  -1 when the note does not have the expected shape.
inline long kernel_note_value(const std::vector<char> &code, const char *kernel, const char *keyname) {
    const std::string blob(code.begin(), code.end());
    std::string name = "\xa5.name";
    const size_t kl = strlen(kernel);
    if (kl < 32) name += (char)(0xa0 | kl);
    else { name += (char)0xd9; name += (char)kl; }
    name += kernel;
    const size_t at = blob.find(name);
    if (at == std::string::npos) return -1;
    std::string key;
    const size_t nl = strlen(keyname);
    if (nl < 32) key += (char)(0xa0 | nl);
    else { key += (char)0xd9; key += (char)nl; }
    key += keyname;
    const size_t k = blob.find(key, at);
    if (k == std::string::npos || k + key.size() >= blob.size()) return -1;
    const unsigned char *v = (const unsigned char *)blob.data() + k + key.size();
    const size_t left = blob.size() - (k + key.size());
    if (v[0] <= 0x7f) return v[0];
    if (v[0] == 0xcc && left > 1) return v[1];
    if (v[0] == 0xcd && left > 2) return ((long)v[1] << 8) | v[2];
    if (v[0] == 0xce && left > 4) return ((long)v[1] << 24) | ((long)v[2] << 16) | ((long)v[3] << 8) | v[4];
    return -1;
}
// largest .group_segment_fixed_size (static LDS bytes) over the kernels of a code object; -1 when the note does not have the expected
// shape.  The sample-batch kernels address their first LDS table from address 0 (mci_device.h draw_leaf, MCI_LDS_ABS): that holds as
// long as no kernel of the translation unit declares static LDS, which is what this reads back.
inline long max_static_lds_bytes(const std::vector<char> &code) {
    const std::string blob(code.begin(), code.end());
    const char *keyname = ".group_segment_fixed_size";
    std::string key;
    key += (char)(0xa0 | strlen(keyname));
    key += keyname;
    long worst = -1;
    for (size_t k = blob.find(key); k != std::string::npos; k = blob.find(key, k + 1)) {
        if (k + key.size() >= blob.size()) return -1;
        const unsigned char *v = (const unsigned char *)blob.data() + k + key.size();
        const size_t left = blob.size() - (k + key.size());
        long val;
        if (v[0] <= 0x7f) val = v[0];
        else if (v[0] == 0xcc && left > 1) val = v[1];
        else if (v[0] == 0xcd && left > 2) val = ((long)v[1] << 8) | v[2];
        else if (v[0] == 0xce && left > 4) val = ((long)v[1] << 24) | ((long)v[2] << 16) | ((long)v[3] << 8) | v[4];
        else return -1;
        if (val > worst) worst = val;
    }
    return worst;
}
inline long kernel_scratch_bytes(const std::vector<char> &code, const char *kernel) { return kernel_note_value(code, kernel, ".private_segment_fixed_size"); }
// .vgpr_count of one kernel (sorts after .name like .private_segment_fixed_size does)
inline long kernel_vgprs(const std::vector<char> &code, const char *kernel) { return kernel_note_value(code, kernel, ".vgpr_count"); }

inline uint64_t fnv1a(const std::string &s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

inline std::string cache_dir() {
    if (const char *e = getenv("MCI_KERNEL_CACHE")) return e;
    Dl_info info;
    std::string dir = ".";
    if (dladdr((void *)&cache_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname; // .../mcintegration.jl_amd/lib/libmci_hip.so
        size_t k = p.rfind('/');
        if (k != std::string::npos) dir = p.substr(0, k) + "/../kernel_cache";
    }
    return dir;
}

// Which compiler made a cached code object: hiprtc's version, the file of the hiprtc library the process resolved (name as resolved,
// which carries its version, and size), the file of the code-object manager -- libamd_comgr holds the clang / LLVM that compiles -- and
// the target.  Part of the cache key: a ROCm upgrade (say, the fix of the pass the units are compiled without) must not keep serving
// the old objects, a downgrade must not serve objects of a compiler nothing here was tested with.  And neither must ANOTHER COPY in the
// same image: a process that has imported PyTorch before this library resolves hiprtc and comgr to the copies PyTorch bundles -- a
// different compiler build (the headline loop: 523.5 instead of 503.5 VALU instructions per wave and sample, 2.2 % slower; campaign case
// 205 compiles correctly there; profiles/r06_ablation.txt E).  hiprtc opens comgr by its soname, so the comgr in use is the FIRST one
// loaded into the process whichever hiprtc asks; before any is loaded, the one next to the hiprtc library (its RUNPATH).
// mci_debug_compiler_id overrides the identity for tests.
inline std::string &compiler_id_override() {
    static std::string s;
    return s;
}
inline int first_comgr_cb(struct dl_phdr_info *info, size_t, void *data) {
    std::string *out = (std::string *)data;
    if (out->empty() && info->dlpi_name && strstr(info->dlpi_name, "amd_comgr")) *out = info->dlpi_name;
    return 0;
}
inline std::string compiler_id() {
    if (!compiler_id_override().empty()) return compiler_id_override();
    int maj = 0, min = 0;
    (void)hiprtcVersion(&maj, &min);
    std::string s = "hiprtc " + std::to_string(maj) + "." + std::to_string(min);
    auto file_id = [](const std::string &path) {
        char real[4096];
        struct stat st;
        if (!realpath(path.c_str(), real) || stat(real, &st) != 0) return std::string("?");
        std::string r = real;
        const size_t k = r.rfind('/');
        return (k == std::string::npos ? r : r.substr(k + 1)) + ":" + std::to_string((long long)st.st_size);
    };
    Dl_info info;
    std::string comgr;
    dl_iterate_phdr(first_comgr_cb, &comgr); // (objects in load order: the first comgr is the one a dlopen by soname returns)
    if (dladdr((void *)&hiprtcVersion, &info) && info.dli_fname) {
        const std::string lib = info.dli_fname;
        s += " | " + file_id(lib);
        if (comgr.empty()) {
            const size_t k = lib.rfind('/');
            comgr = (k == std::string::npos ? std::string(".") : lib.substr(0, k)) + "/libamd_comgr.so";
        }
    }
    return s + " | " + (comgr.empty() ? std::string("?") : file_id(comgr)) + " | gfx950";
}

// hiprtc's first compile of a process loads the compiler (comgr, ~0.3 s on this image): mci_ctx_create starts it on a thread of its
// own, next to the HIP runtime's own device initialisation, so that the first real compile finds it loaded
struct WarmUp {
    std::thread th;
    std::mutex mu;
    bool started = false;
};
inline WarmUp &warm_up_state() {
    static WarmUp w;
    return w;
}
inline void warm_up_async() {
    WarmUp &w = warm_up_state();
    std::lock_guard<std::mutex> g(w.mu);
    if (w.started) return;
    w.started = true;
    w.th = std::thread([] {
        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, "extern \"C\" __global__ void mci_warm_up(double* a) { a[0] = 1.0; }", "mci_warm_up.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return;
        const char *opts[] = {"--offload-arch=gfx950"};
        (void)hiprtcCompileProgram(prog, 1, opts);
        hiprtcDestroyProgram(&prog);
    });
}
inline void warm_up_join() {
    WarmUp &w = warm_up_state();
    std::lock_guard<std::mutex> g(w.mu);
    if (w.th.joinable()) w.th.join();
}

// returns the gfx950 code object for `src`, from the on-disk cache or by compiling with hiprtc
inline int compile(const std::string &src, int threads, std::vector<char> &code, std::string &log, bool &from_cache, std::string *cache_path = nullptr,
                   int extra_hdr = kHdrNone, bool cache_only = false, bool no_exec_mask_flag = false) {
    std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics",
                                     "-ffp-contract=off", "-DMCI_THREADS=" + std::to_string(threads)};
    // EVERY unit is compiled WITHOUT the backend's pre-RA exec-mask optimisation.  One layout of the randomised campaigns -- a composite
    // pool of three leaves next to a Discrete pool nobody uses, ten draws -- came out of ROCm 7.2's compiler with the right chains and
    // statistics and the :vegasmc histogram adds of its several-lanes-per-chain kernel in the wrong bins; -opt-bisect-limit pins the flip
    // on ONE machine pass, si-optimize-exec-masking-pre-ra on that kernel (right with the first 55582 passes, wrong from 55583 on:
    // profiles/r05_fuzz.txt, tools/repro_case.py 205).  The pass rewrites EXEC save / restore sequences, of which the group kernels --
    // nested divergent regions around wave-wide exchanges -- have hundreds; every user integrand is a new translation unit, and nothing
    // says the next victim is a group kernel.  Round 6 measured what the pass is worth here: nothing (headline 1.3393 | 1.3398 ms with |
    // without it, C2 on 16 grids, C3, C4, C5 under all three solvers and the default call within +-0.5 %, profiles/r06_ablation.txt), so it
    // is off for all of them.  MCI_JIT_FLAGS that names the switch itself decides it (the A/B; the guard test that re-enables the pass to
    // see the self-check of a new group code object trip, mci_host_jit.h spec_self_check); no_exec_mask_flag: the retry of a unit whose
    // compilation the switch itself broke (a later compiler that no longer knows it).
    (void)extra_hdr;
    const char *jf = getenv("MCI_JIT_FLAGS");
    if (!no_exec_mask_flag && !(jf && strstr(jf, "amdgpu-opt-exec-mask-pre-ra"))) {
        opts.push_back("-mllvm");
        opts.push_back("-amdgpu-opt-exec-mask-pre-ra=0");
    }
    if (const char *e = jf) {
        std::istringstream is(e);
        std::string t;
        while (is >> t) opts.push_back(t);
    }
    std::string key = src + "\n//HDR\n" + kDeviceHeader;
    if (extra_hdr == kHdrTrain) key += std::string("\n//HDR\n") + kTrainHeader;
    if (extra_hdr == kHdrSpec) key += std::string("\n//HDR\n") + kSpecHeader;
    for (auto &f : opts) key += "\n//" + f;
    key += "\n//COMPILER " + compiler_id();
    char name[64];
    snprintf(name, sizeof name, "mci_%016llx.hsaco", (unsigned long long)fnv1a(key));
    const std::string dir = cache_dir(), path = dir + "/" + name;
    if (cache_path) *cache_path = path;
    from_cache = false;
    {
        std::ifstream f(path, std::ios::binary);
        if (f) {
            code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
            if (!code.empty()) {
                from_cache = true;
                return 0;
            }
        }
    }
    if (cache_only) return -1; // (not in the cache: the caller compiles it elsewhere, e.g. on a thread of its own)
    warm_up_join();
    hiprtcProgram prog;
    const char *hdr[2] = {kDeviceHeader, extra_hdr == kHdrSpec ? kSpecHeader : kTrainHeader}, *hname[2] = {"mci_device.h", extra_hdr == kHdrSpec ? "mci_spec.h" : "mci_train.h"};
    if (hiprtcCreateProgram(&prog, src.c_str(), "mci_problem.hip", extra_hdr != kHdrNone ? 2 : 1, hdr, hname) != HIPRTC_SUCCESS) {
        log = "hiprtcCreateProgram failed";
        return 1;
    }
    std::vector<const char *> copts;
    for (auto &f : opts) copts.push_back(f.c_str());
    hiprtcResult r = hiprtcCompileProgram(prog, (int)copts.size(), copts.data());
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) {
        log.resize(ls);
        hiprtcGetProgramLog(prog, &log[0]);
    }
    if (r != HIPRTC_SUCCESS) {
        hiprtcDestroyProgram(&prog);
        if (log.empty()) log = hiprtcGetErrorString(r);
        return 2;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    mkdir(dir.c_str(), 0755);
    // unique per process AND per call: concurrent host threads compiling the same key must not share a temporary file
    static std::atomic<unsigned long> seq{0};
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid()) + "." + std::to_string(seq.fetch_add(1));
    {
        std::ofstream f(tmp, std::ios::binary);
        if (f) {
            f.write(code.data(), (std::streamsize)code.size());
            f.close();
            if (rename(tmp.c_str(), path.c_str()) != 0) unlink(tmp.c_str());
        }
    }
    return 0;
}

} // namespace mcijit
