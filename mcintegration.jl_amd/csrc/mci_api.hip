// mci_api.hip -- host core of libmci_hip.so: the C ABI of include/mci.h.  One translation unit; its sections live in the
// mci_host_*.h files included at the bottom, in this order: types, ctx, problem, jit, iteration, integrate, access, statistics.
//
// Owns: the Configuration analogue (src/configuration.jl:105-194), the device-resident state (grids,
// distributions, histograms, packed statistics), the per-iteration launch chain
//     sample batch (JIT, mci_device.h) -> merge (k_hist_stage1, k_finalize) -> RCCL all-reduce -> k_train
// and the iteration loop + Result statistics (src/main.jl:142-218, :296-320, src/statistics.jl:186-220).
// There is NO CPU fallback: without a HIP device every compute entry point fails with MCI_ERR_NO_DEVICE.
#include "../../include/mci.h"
#include "mci_debug.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <link.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "mci_device.h" // BatchArgs / DumpArgs (the templates themselves are instantiated by the JIT)
#include "mci_jit.h"
#include "mci_static_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(x)                                                                                           \
    do {                                                                                                    \
        hipError_t e_ = (x);                                                                                \
        if (e_ != hipSuccess) return fail(MCI_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

} // namespace

#include "mci_host_types.h"

extern "C" {

#include "mci_host_ctx.h"
#include "mci_host_problem.h"
#include "mci_host_jit.h"
#include "mci_host_iteration.h"
#include "mci_host_integrate.h"
#include "mci_host_access.h"
#include "mci_host_statistics.h"

} // extern "C"
