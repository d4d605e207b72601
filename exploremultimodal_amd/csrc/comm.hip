// Gradient exchange entry points of the C-ABI (vlmo_comm_{init,reduce_scatter,all_gather,destroy}).
//
// The reference's exchange step is torch DDP / DeepSpeed ZeRO-2 over NCCL (train/pretrain/multimodal.py:61-95,
// conf/ds_stage/l2.yaml); here it is RCCL over xGMI, one communicator per process (one process per GPU), every
// collective enqueued on the caller's stream.  RCCL is resolved at RUN time: a process that already carries a copy (the
// one PyTorch-ROCm ships) must not get a second one with the same symbols, and a build box without RCCL still links
// the library.  The pack / unpack kernels are the two passes the data-parallel reducer makes over a gradient arena:
// fp32 -> bf16 with the 1 / world scaling folded in, and back.
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include "common.h"
#include "vlmo_hip.h"

namespace {

// the slice of rccl.h this file needs (rccl.h: ncclDataType_t / ncclRedOp_t values are ABI-stable across NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[VLMO_COMM_ID_BYTES]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclFloat16 = 6, kNcclFloat32 = 7, kNcclBfloat16 = 9 };

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl* rccl() {
    // resolved once, thread-safe: two threads making their first vlmo_comm_* call must not race on the table
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        // a copy that is in the process already (PyTorch's) first, then the system's
        for (int pass = 0; pass < 2 && !r.lib; ++pass)
            for (const char* n : names) {
                r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (r.lib) break;
            }
        if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) return;
#define SYM(field, name) *(void**)(&r.field) = dlsym(r.lib, name)
        SYM(GetUniqueId, "ncclGetUniqueId");
        SYM(CommInitRank, "ncclCommInitRank");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(CommCount, "ncclCommCount");
        SYM(AllReduce, "ncclAllReduce");
        SYM(ReduceScatter, "ncclReduceScatter");
        SYM(AllGather, "ncclAllGather");
        SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather;
    });
    return r.ok ? &r : nullptr;
}

struct Comm {
    uint32_t magic;
    ncclComm_t comm;
    int rank, world;
};
constexpr uint32_t kMagic = 0x564c434du;   // "VLCM"

int nccl_type(int dtype) {
    switch (dtype) {
        case VLMO_BF16: return kNcclBfloat16;
        case VLMO_F16: return kNcclFloat16;
        case VLMO_F32: return kNcclFloat32;
    }
    return -1;
}

int fail(Rccl* r, const char* what, int rc) {
    vlmo_set_error("%s: RCCL error %d (%s)", what, rc, r->GetErrorString ? r->GetErrorString(rc) : "?");
    return 1000 + rc;       // > 0: a runtime error, distinguishable from hipError_t values (< 1000)
}

#define GET_COMM(c, handle, name)                                                                    \
    Rccl* r = rccl();                                                                                \
    VLMO_CHECK_ARG(r, "%s: RCCL (librccl.so) not found", name);                                      \
    Comm* c = (Comm*)(handle);                                                                       \
    VLMO_CHECK_ARG(c && c->magic == kMagic, "%s: not a communicator handle", name)

// ---- pack / unpack ------------------------------------------------------------------------------------------------
// 16 B per lane on the narrow side (8 bf16), two 16-B accesses on the fp32 side; grid-stride over 2 048-element chunks.
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int64_t n,
                                                   float scale) {
    const int64_t nv = n >> 3;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (int64_t)gridDim.x * 256) {
        const f32x4 a = __builtin_nontemporal_load((const f32x4*)src + 2 * v);
        const f32x4 b = __builtin_nontemporal_load((const f32x4*)src + 2 * v + 1);
        bf16x8 o;
        for (int i = 0; i < 4; ++i) {
            o[i] = (bf16)(a[i] * scale);
            o[4 + i] = (bf16)(b[i] * scale);
        }
        ((bf16x8*)dst)[v] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = (nv << 3) + threadIdx.x;
        dst[i] = (bf16)(src[i] * scale);
    }
}

__global__ __launch_bounds__(256) void unpack_kernel(const bf16* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t nv = n >> 3;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (int64_t)gridDim.x * 256) {
        const bf16x8 a = __builtin_nontemporal_load((const bf16x8*)src + v);
        f32x4 lo, hi;
        for (int i = 0; i < 4; ++i) {
            lo[i] = (float)a[i];
            hi[i] = (float)a[4 + i];
        }
        ((f32x4*)dst)[2 * v] = lo;
        ((f32x4*)dst)[2 * v + 1] = hi;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = (nv << 3) + threadIdx.x;
        dst[i] = (float)src[i];
    }
}

int pack_grid(int64_t n) {
    const int64_t blocks = ((n >> 3) + 255) / 256;
    return (int)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));
}

}  // namespace

extern "C" {

int vlmo_comm_available(void) { return rccl() ? 1 : 0; }

int vlmo_comm_unique_id(void* id) {
    Rccl* r = rccl();
    VLMO_CHECK_ARG(r, "vlmo_comm_unique_id: RCCL (librccl.so) not found");
    VLMO_CHECK_ARG(id, "vlmo_comm_unique_id: null output");
    ncclUniqueId u;
    const int rc = r->GetUniqueId(&u);
    if (rc != kNcclSuccess) return fail(r, "vlmo_comm_unique_id", rc);
    memcpy(id, &u, sizeof u);
    return 0;
}

int vlmo_comm_init(void** out, const void* id, int rank, int world) {
    Rccl* r = rccl();
    VLMO_CHECK_ARG(r, "vlmo_comm_init: RCCL (librccl.so) not found");
    VLMO_CHECK_ARG(out && id, "vlmo_comm_init: null argument");
    VLMO_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "vlmo_comm_init: rank %d of %d", rank, world);
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t c = nullptr;
    const int rc = r->CommInitRank(&c, world, u, rank);     // collective over the ranks: blocks until all arrived
    if (rc != kNcclSuccess) return fail(r, "vlmo_comm_init", rc);
    Comm* h = new Comm{kMagic, c, rank, world};
    *out = h;
    return 0;
}

int vlmo_comm_count(void* comm, int* ranks, int* rank) {
    GET_COMM(c, comm, "vlmo_comm_count");
    VLMO_CHECK_ARG(ranks, "vlmo_comm_count: null output");
    int n = c->world;
    if (r->CommCount) {         // what the COMMUNICATOR says, not what the caller passed to vlmo_comm_init
        const int rc = r->CommCount(c->comm, &n);
        if (rc != kNcclSuccess) return fail(r, "vlmo_comm_count", rc);
    }
    *ranks = n;
    if (rank) *rank = c->rank;
    return 0;
}

int vlmo_comm_destroy(void* comm) {
    if (!comm) return 0;
    GET_COMM(c, comm, "vlmo_comm_destroy");
    // every collective enqueued through this communicator must have finished: the device is drained here, the caller
    // does not have to remember the streams it used
    (void)hipDeviceSynchronize();
    const int rc = r->CommDestroy(c->comm);
    c->magic = 0;
    delete c;
    if (rc != kNcclSuccess) return fail(r, "vlmo_comm_destroy", rc);
    return 0;
}

int vlmo_comm_all_reduce(void* comm, const void* send, void* recv, int64_t count, int dtype, hipStream_t stream) {
    GET_COMM(c, comm, "vlmo_comm_all_reduce");
    const int t = nccl_type(dtype);
    VLMO_CHECK_ARG(t >= 0 && send && recv && count >= 0, "vlmo_comm_all_reduce: bad argument");
    if (count == 0) return 0;
    const int rc = r->AllReduce(send, recv, (size_t)count, t, kNcclSum, c->comm, stream);
    return rc == kNcclSuccess ? 0 : fail(r, "vlmo_comm_all_reduce", rc);
}

int vlmo_comm_reduce_scatter(void* comm, const void* send, void* recv, int64_t recv_count, int dtype,
                             hipStream_t stream) {
    GET_COMM(c, comm, "vlmo_comm_reduce_scatter");
    const int t = nccl_type(dtype);
    VLMO_CHECK_ARG(t >= 0 && send && recv && recv_count >= 0, "vlmo_comm_reduce_scatter: bad argument");
    if (recv_count == 0) return 0;
    const int rc = r->ReduceScatter(send, recv, (size_t)recv_count, t, kNcclSum, c->comm, stream);
    return rc == kNcclSuccess ? 0 : fail(r, "vlmo_comm_reduce_scatter", rc);
}

int vlmo_comm_all_gather(void* comm, const void* send, void* recv, int64_t send_count, int dtype,
                         hipStream_t stream) {
    GET_COMM(c, comm, "vlmo_comm_all_gather");
    const int t = nccl_type(dtype);
    VLMO_CHECK_ARG(t >= 0 && send && recv && send_count >= 0, "vlmo_comm_all_gather: bad argument");
    if (send_count == 0) return 0;
    const int rc = r->AllGather(send, recv, (size_t)send_count, t, c->comm, stream);
    return rc == kNcclSuccess ? 0 : fail(r, "vlmo_comm_all_gather", rc);
}

int vlmo_grad_pack(const float* src, void* dst, int64_t n, float scale, hipStream_t stream) {
    VLMO_CHECK_ARG(src && dst && n >= 0, "vlmo_grad_pack: bad argument");
    VLMO_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "vlmo_grad_pack: 16-byte alignment");
    if (n == 0) return 0;
    pack_kernel<<<pack_grid(n), 256, 0, stream>>>(src, (bf16*)dst, n, scale);
    VLMO_CHECK_LAUNCH("vlmo_grad_pack");
    return 0;
}

int vlmo_grad_unpack(const void* src, float* dst, int64_t n, hipStream_t stream) {
    VLMO_CHECK_ARG(src && dst && n >= 0, "vlmo_grad_unpack: bad argument");
    VLMO_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "vlmo_grad_unpack: 16-byte alignment");
    if (n == 0) return 0;
    unpack_kernel<<<pack_grid(n), 256, 0, stream>>>((const bf16*)src, dst, n);
    VLMO_CHECK_LAUNCH("vlmo_grad_unpack");
    return 0;
}

}  // extern "C"
