// mci_host_ctx.h -- part of the ONE translation unit mci_api.hip (included there, in order; not a stand-alone header):
// errors and versions, the context (device, stream, RCCL communicator).

const char *mci_last_error(void) { return g_err.c_str(); }
// "mci-hip <abi>.<revision>": <abi> changes whenever a struct of include/mci.h changes its layout (mci_result grew `correlated` and
// `warmup` in ABI 4; ABI 5 adds entry points only) -- a caller built against another header compares it before passing structs
const char *mci_version(void) { return "mci-hip 6.0 (gfx950)"; }
int32_t mci_abi_version(void) { return 5; }

int mci_device_count(int32_t *count) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return MCI_OK;
}

int mci_ctx_create(int32_t device, mci_ctx **out) {
    if (!out) return fail(MCI_ERR_INVALID, "out is NULL");
    mci_ctx *c = new mci_ctx();
    if (device < 0) { // offline / compile-only
        c->offline = true;
        *out = c;
        return MCI_OK;
    }
    mcijit::warm_up_async(); // (the compiler loads while the HIP runtime initialises the device below)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        mcijit::warm_up_join();
        delete c;
        return fail(MCI_ERR_NO_DEVICE, "no HIP device visible: the MI355X path has no CPU fallback");
    }
    if (device >= n) {
        delete c;
        return fail(MCI_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    }
    c->device = device;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    *out = c;
    return MCI_OK;
}

int mci_ctx_destroy(mci_ctx *c) {
    if (!c) return MCI_OK;
    mcijit::warm_up_join();
    persist_orphans_join();
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->spare) (void)hipFree(c->spare); // (the parked-stream buffer the last many-grid problem left behind)
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MCI_OK;
}

void *mci_ctx_stream(mci_ctx *c) { return c ? (void *)c->stream : nullptr; }

int mci_comm_unique_id(void *id128) {
    int rc = rccl_load();
    if (rc) return rc;
    int r = g_rccl.GetUniqueId(id128);
    if (r) return fail(MCI_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return MCI_OK;
}

int mci_comm_init(mci_ctx *c, int32_t rank, int32_t nranks, const void *id128) {
    if (!c || c->offline) return fail(MCI_ERR_INVALID, "communicator needs an online context");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(MCI_ERR_INVALID, "bad rank %d / %d", rank, nranks);
    int rc = rccl_load();
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    Id128 id;
    memcpy(id.b, id128, 128);
    int r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r) return fail(MCI_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    c->rank = rank;
    c->nranks = nranks;
    return MCI_OK;
}

int mci_comm_rank(const mci_ctx *c, int32_t *rank, int32_t *nranks) {
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return MCI_OK;
}

