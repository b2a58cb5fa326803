// Multi-GPU support of the path: the ONE collective it has -- replicating the feature index from the
// rank that read it to every other GPU of the node -- and the device-side checksum that verifies the replica.
//
// One process per GPU.  The communicator is RCCL's own (ncclCommInitRank over a 128-byte id that the host side hands
// from rank 0 to the others, e.g. through torch.distributed's store); the broadcast is one ncclBroadcast of raw bytes
// on the caller's stream, which RCCL runs as a ring over the xGMI links.  RCCL is bound at run time (dlopen/dlsym), not
// at link time: a single-GPU box never touches it, and inside a PyTorch process the already-loaded librccl is reused
// instead of mapping a second copy.
#include <dlfcn.h>
#include <stdlib.h>

#include <mutex>

#include <rccl/rccl.h>   // types only; no symbol of it is linked

#include "common.h"

struct rvc_comm {
    ncclComm_t comm;
    int n_ranks, rank;
};

namespace rvc {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    char where[512] = "";
    char error[512] = "";
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;

static void rccl_bind() {
    RcclApi &a = g_rccl;
    const char *env = getenv("RVC_RCCL_PATH");
    // already mapped copies first (PyTorch-ROCm ships its own librccl.so), then the ROCm installation
    const char *names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (int pass = 0; pass < 2 && !a.handle; ++pass)
        for (const char *n : names) {
            if (!n) continue;
            a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (a.handle) {
                snprintf(a.where, sizeof(a.where), "%s%s", n, pass == 0 ? " (already loaded)" : "");
                break;
            }
        }
    if (!a.handle) {
        snprintf(a.error, sizeof(a.error), "librccl.so not found (set RVC_RCCL_PATH): %s", dlerror());
        return;
    }
#define RVC_BIND(field, sym)                                                             \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, sym));                 \
    if (!a.field) { snprintf(a.error, sizeof(a.error), "%s lacks %s", a.where, sym); return; }
    RVC_BIND(GetUniqueId, "ncclGetUniqueId")
    RVC_BIND(CommInitRank, "ncclCommInitRank")
    RVC_BIND(CommDestroy, "ncclCommDestroy")
    RVC_BIND(CommCount, "ncclCommCount")
    RVC_BIND(Broadcast, "ncclBroadcast")
    RVC_BIND(GetErrorString, "ncclGetErrorString")
    RVC_BIND(GetVersion, "ncclGetVersion")
#undef RVC_BIND
}

static const RcclApi *rccl() {
    std::call_once(g_rccl_once, rccl_bind);
    if (g_rccl.error[0]) {
        set_error("RCCL unavailable: %s", g_rccl.error);
        return nullptr;
    }
    return &g_rccl;
}

#define RVC_NCCL(api, expr)                                                                          \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess) return fail("%s failed: %s", #expr, (api)->GetErrorString(_r));       \
    } while (0)

// Position-weighted 64-bit sums over the buffer's 32-bit words: s1 = sum w_i, s2 = sum (i + 1) w_i (mod 2^64).  Integer
// addition is associative, so the value does not depend on how the grid splits the buffer; a trailing 1-3 bytes are
// zero-extended into one last word.
__global__ void __launch_bounds__(256)
checksum64_kernel(const uint32_t *__restrict__ words, uint64_t n_words, const unsigned char *__restrict__ tail, int n_tail,
                  unsigned long long *__restrict__ out) {
    uint64_t s1 = 0, s2 = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
        const uint64_t w = words[i];
        s1 += w;
        s2 += (i + 1) * w;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_tail > 0) {
        uint64_t w = 0;
        for (int b = 0; b < n_tail; ++b) w |= (uint64_t)tail[b] << (8 * b);
        s1 += w;
        s2 += (n_words + 1) * w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor((unsigned long long)s1, o);
        s2 += __shfl_xor((unsigned long long)s2, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], (unsigned long long)s1);
        atomicAdd(&out[1], (unsigned long long)s2);
    }
}

}  // namespace rvc

using namespace rvc;

extern "C" int rvc_comm_unique_id(unsigned char *id_host) {
    if (!id_host) return fail("rvc_comm_unique_id: null pointer");
    const RcclApi *a = rccl();
    if (!a) return 1;
    ncclUniqueId id;
    RVC_NCCL(a, a->GetUniqueId(&id));
    static_assert(sizeof(id) == RVC_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_host, &id, sizeof(id));
    return 0;
}

extern "C" int rvc_comm_create(const unsigned char *id_host, int n_ranks, int rank, rvc_comm **out) {
    if (!id_host || !out) return fail("rvc_comm_create: null pointer");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail("rvc_comm_create: rank %d of %d", rank, n_ranks);
    const RcclApi *a = rccl();
    if (!a) return 1;
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    ncclComm_t c = nullptr;
    RVC_NCCL(a, a->CommInitRank(&c, n_ranks, id, rank));   // collective: every rank of the job calls it, on its own device
    *out = new rvc_comm{c, n_ranks, rank};
    return 0;
}

extern "C" int rvc_comm_destroy(rvc_comm *comm) {
    if (!comm) return 0;
    const RcclApi *a = rccl();
    if (!a) return 1;
    ncclResult_t r = a->CommDestroy(comm->comm);
    delete comm;
    if (r != ncclSuccess) return fail("ncclCommDestroy failed: %s", a->GetErrorString(r));
    return 0;
}

extern "C" int rvc_comm_info(const rvc_comm *comm, int *n_ranks, int *rank, int *rccl_version, char *library_path,
                             size_t library_path_bytes) {
    const RcclApi *a = rccl();
    if (!a) return 1;
    if (comm) {
        int count = 0;
        RVC_NCCL(a, a->CommCount(comm->comm, &count));   // what RCCL itself says, not what we were told
        if (n_ranks) *n_ranks = count;
        if (rank) *rank = comm->rank;
    }
    if (rccl_version) RVC_NCCL(a, a->GetVersion(rccl_version));
    if (library_path && library_path_bytes) snprintf(library_path, library_path_bytes, "%s", a->where);
    return 0;
}

extern "C" int rvc_index_broadcast(rvc_comm *comm, void *buf_dev, size_t bytes, int root, void *stream) {
    if (!comm || !buf_dev) return fail("rvc_index_broadcast: null pointer");
    if (root < 0 || root >= comm->n_ranks) return fail("rvc_index_broadcast: root %d of %d ranks", root, comm->n_ranks);
    const RcclApi *a = rccl();
    if (!a) return 1;
    if (bytes == 0) return 0;
    RVC_NCCL(a, a->Broadcast(buf_dev, buf_dev, bytes, ncclUint8, root, comm->comm, (hipStream_t)stream));
    return 0;
}

extern "C" int rvc_checksum64(const void *buf_dev, size_t bytes, uint64_t *out2_dev, void *stream) {
    if (!out2_dev || (!buf_dev && bytes)) return fail("rvc_checksum64: null pointer");
    if ((uintptr_t)buf_dev % 4) return fail("rvc_checksum64: buffer must be 4-byte aligned");
    RVC_HIP(hipMemsetAsync(out2_dev, 0, 2 * sizeof(uint64_t), (hipStream_t)stream));
    const uint64_t n_words = bytes / 4;
    const int n_tail = (int)(bytes % 4);
    unsigned blocks = (unsigned)ceil_div((int64_t)(n_words ? n_words : 1), 256 * 16);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(checksum64_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)buf_dev, n_words,
                       (const unsigned char *)buf_dev + n_words * 4, n_tail, (unsigned long long *)out2_dev);
    RVC_LAUNCH_CHECK();
    return 0;
}
