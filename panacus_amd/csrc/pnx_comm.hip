// pnx_comm.hip -- RCCL communicator owned by a context: the multi-GPU exchange of the hot path.
//
// The reference has no collective (it is a single-process rayon program); what the MI355X path
// adds is one all-reduce of small integer counters over xGMI: hist[G+1] when the items are sharded
// by node range, out[R][T][G] when the orders (or the items) of a permuted-growth call are.  A host
// written against this ABI -- the Rust host of INTEGRATION.md -- does not have to bring its own
// collective library: rank 0 asks for an id, the host hands the 128 bytes to the other processes
// (a file, a pipe, MPI, torch.distributed's store -- the library does not care), every rank calls
// pnx_comm_init, and from then on
//   * every coverage pass of the context is followed, ON THE CONTEXT'S STREAM and before its counters
//     are copied to the host, by an all-reduce (sum) of the pass's flag block and histogram: pnx_hist /
//     pnx_hist_fetch return the GLOBAL histogram, and because the verification flags are reduced with
//     it every rank takes the same "run it again" decision -- the collectives of a re-run stay
//     matched across ranks (the failure mode the round-1 advisor found in a host-side reduce);
//   * pnx_comm_allreduce_u64 reduces any device buffer in place, enqueued on the same stream, e.g.
//     the result of pnx_ordered_growth_enqueued.
// librccl.so is opened with dlopen on the first use: a single-GPU process never loads it, and
// libpanacus_hip.so keeps libamdhip64 as its only link-time dependency.
#include <dlfcn.h>
#include <glob.h>
#include <link.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <rccl/rccl.h>  // types and enums only: the entry points are resolved at run time

#include "pnx_context.hpp"

namespace pnx {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

static RcclApi &rccl() {
    static RcclApi api;
    if (api.lib || !api.err.empty()) return api;
    // PNX_RCCL_LIB (a path) first; then a copy that is ALREADY MAPPED into the process, whatever its file is called (PyTorch
    // wheels ship their own librccl.so: two copies of RCCL in one process interpose each other's symbols and the process
    // dies in a double free when it exits); then the loader's own search; then ROCm's; then the copies of PyTorch wheels in
    // <site-packages>/torch/lib -- a host without torch in the process (the Rust host of INTEGRATION.md) still finds the
    // library of the image.  A process that loads torch AFTER this library names torch's copy in PNX_RCCL_LIB
    // (panacus_amd/capi.py does).
    std::vector<std::string> names;
    if (const char *env = getenv("PNX_RCCL_LIB")) names.push_back(env);
    dl_iterate_phdr(
        [](struct dl_phdr_info *info, size_t, void *out) {
            if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) static_cast<std::vector<std::string> *>(out)->push_back(info->dlpi_name);
            return 0;
        },
        &names);
    for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) names.push_back(n);
    for (const char *pat : {"/usr/local/lib/python3*/dist-packages/torch/lib/librccl.so*", "/usr/lib/python3*/site-packages/torch/lib/librccl.so*",
                            "/usr/local/lib/python3*/site-packages/torch/lib/librccl.so*", "/opt/conda/lib/python3*/site-packages/torch/lib/librccl.so*"}) {
        glob_t g;
        if (glob(pat, 0, nullptr, &g) == 0)
            for (size_t k = 0; k < g.gl_pathc; ++k) names.push_back(g.gl_pathv[k]);
        globfree(&g);
    }
    std::string tried;
    for (const std::string &name : names) {
        api.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
        tried += (tried.empty() ? "" : ", ") + name;
    }
    if (!api.lib) {
        api.err = "cannot load librccl.so (set PNX_RCCL_LIB to its path); tried: " + tried;
        return api;
    }
    auto sym = [&](const char *n) {
        void *p = dlsym(api.lib, n);
        if (!p && api.err.empty()) api.err = std::string("librccl.so lacks ") + n;
        return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!api.err.empty()) {
        dlclose(api.lib);
        api.lib = nullptr;
    }
    return api;
}

static int rccl_fail(pnx_ctx *ctx, const char *what, ncclResult_t r) {
    return ctx->fail(PNX_EHIP, "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
}

// the flags of a pass count events; before they are summed over the ranks they are clipped so that
// the sum of a u32 pair carried in one u64 word can never spill into its neighbour
__global__ void k_clip_flags(uint32_t *flags) {
    if (threadIdx.x < 8 && flags[threadIdx.x] > (1u << 20)) flags[threadIdx.x] = 1u << 20;
}

int comm_allreduce_u64(pnx_ctx *ctx, uint64_t *d_buf, size_t n, hipStream_t stream) {
    if (!ctx->comm) return ctx->fail(PNX_EINVAL, "no communicator: call pnx_comm_init first");
    if (n == 0) return PNX_OK;
    ncclResult_t r = rccl().AllReduce(d_buf, d_buf, n, ncclUint64, ncclSum, (ncclComm_t)ctx->comm, stream ? stream : ctx->stream);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclAllReduce", r);
    return PNX_OK;
}

// flags (u32[8] = 4 words) + histogram ((G+1) words) of the pass in `t`, in place, behind the pass
int comm_reduce_pass(pnx_ctx *ctx, Ticket *t) {
    if (!ctx->comm || !ctx->comm_reduce_hist) return PNX_OK;
    hipLaunchKernelGGL(k_clip_flags, dim3(1), dim3(64), 0, ctx->s_post, t->d_flags);
    return comm_allreduce_u64(ctx, (uint64_t *)t->d_block.p, 4 + (size_t)ctx->n_groups + 1, ctx->s_post);
}

}  // namespace pnx

using namespace pnx;

extern "C" {

int pnx_comm_unique_id(uint8_t id[PNX_COMM_ID_BYTES]) {
    if (!id) return PNX_EINVAL;
    RcclApi &api = rccl();
    if (!api.lib) return PNX_ENODEV;
    static_assert(sizeof(ncclUniqueId) == PNX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    if (api.GetUniqueId(&u) != ncclSuccess) return PNX_EHIP;
    std::memcpy(id, &u, sizeof u);
    return PNX_OK;
}

int pnx_comm_init(pnx_ctx *ctx, const uint8_t id[PNX_COMM_ID_BYTES], int rank, int world) {
    if (!ctx) return PNX_EINVAL;
    if (!id || world < 1 || rank < 0 || rank >= world) return ctx->fail(PNX_EINVAL, "pnx_comm_init: bad rank %d of %d", rank, world);
    if (ctx->comm) return ctx->fail(PNX_EINVAL, "pnx_comm_init: the context already owns a communicator");
    RcclApi &api = rccl();
    if (!api.lib) return ctx->fail(PNX_ENODEV, "%s", api.err.c_str());
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    ncclResult_t r = api.CommInitRank(&comm, world, u, rank);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclCommInitRank", r);
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;  // (PNX_CFG_COMM_REDUCE_HIST keeps whatever the caller configured, before or after this call)
    return PNX_OK;
}

int pnx_comm_allreduce_u64(pnx_ctx *ctx, uint64_t *d_buf, size_t n) {
    if (!ctx) return PNX_EINVAL;
    if (!d_buf && n) return ctx->fail(PNX_EINVAL, "pnx_comm_allreduce_u64: NULL buffer");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    return comm_allreduce_u64(ctx, d_buf, n);
}

int pnx_comm_barrier(pnx_ctx *ctx) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->comm) return ctx->fail(PNX_EINVAL, "no communicator: call pnx_comm_init first");
    PNX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->d_comm_word, 8);
    if (rc) return rc;
    PNX_HIP(ctx, hipMemsetAsync(ctx->d_comm_word.p, 0, 8, ctx->stream));
    if ((rc = comm_allreduce_u64(ctx, (uint64_t *)ctx->d_comm_word.p, 1))) return rc;
    PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PNX_OK;
}

int pnx_comm_free(pnx_ctx *ctx) {
    if (!ctx) return PNX_EINVAL;
    if (!ctx->comm) return PNX_OK;
    (void)hipSetDevice(ctx->device);
    (void)drain_streams(ctx);
    ncclResult_t r = rccl().CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclCommDestroy", r);
    return PNX_OK;
}

}  // extern "C"
