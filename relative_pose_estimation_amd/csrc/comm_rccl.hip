// comm_rccl.hip -- the one collective of the sharded path: the final pose gather over RCCL / xGMI.
//
// Image pairs shard embarrassingly across the GPUs of a node (the reference never chains estimates:
// src/core/batch_processor.py:82-92 takes R_prev from ground truth), one process per GPU, no data-path
// exchange.  At the end of a step every rank contributes `per_rank` fixed-size 128-byte pose records
// and receives everybody's: one ncclAllGather of per_rank * 128 bytes per rank (C4: 4096 pairs per rank
// -> 512 KiB per rank, 4 MiB in all: latency bound over xGMI, no ring all-reduce needed).  The records are
// packed on the device straight from the result arrays of the last batch, so the gather needs no host
// round trip before the collective; the gathered block is copied to the host once.
//
// librccl.so (573 MB) is loaded lazily with dlopen on the first communicator call: single-GPU users of
// librpe_amd.so never pay for it and the library has no link-time dependency on it.  The host side needs
// no torch: ranks find each other through the unique id that rank 0 publishes (a file, see sharding.py).
#include "rpe_internal.h"
#include <dlfcn.h>
#include <string.h>

// ---- the few RCCL declarations used (rccl/rccl.h; ABI-stable C API) -------------------------------
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                   // ncclSuccess == 0
enum { rpeNcclChar = 0, rpeNcclFloat64 = 8 };      // ncclDataType_t: ncclInt8/ncclChar = 0, ncclFloat64/ncclDouble = 8
enum { rpeNcclMax = 2 };                           // ncclRedOp_t: ncclSum 0, ncclProd 1, ncclMax 2, ncclMin 3

struct RcclApi {
    void *dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static std::string g_comm_err;

static int load_rccl()
{
    if (g_rccl.dl) return RPE_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *dl = nullptr;
    for (const char *n : names) if ((dl = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!dl) { g_comm_err = std::string("cannot load librccl.so: ") + dlerror(); return RPE_ERR_HIP; }
    RcclApi a;
    a.dl = dl;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(dl, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(dl, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(dl, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(dl, "ncclAllGather");
    a.AllReduce = (decltype(a.AllReduce))dlsym(dl, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(dl, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.AllReduce || !a.GetErrorString) {
        g_comm_err = "librccl.so lacks an expected symbol"; dlclose(dl); return RPE_ERR_HIP;
    }
    g_rccl = a;
    return RPE_OK;
}

struct rpe_comm {
    ncclComm_t comm = nullptr;
    rpe_handle *h = nullptr;
    int rank = 0, world = 1;
    uint8_t *d_send = nullptr, *d_recv = nullptr;     // per_rank_cap / world * per_rank_cap records
    int per_rank_cap = 0;
    double *d_scalar = nullptr;                       // 2 doubles for the scalar all-reduce
};

#define NCHK(c, call)                                                                                   \
    do {                                                                                                \
        ncclResult_t r_ = (call);                                                                       \
        if (r_ != 0) { g_comm_err = std::string(#call " failed: ") + g_rccl.GetErrorString(r_); return RPE_ERR_HIP; } \
    } while (0)
#define CHIP(call)                                                                                      \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) { g_comm_err = std::string(#call " failed: ") + hipGetErrorString(e_); return RPE_ERR_HIP; } \
    } while (0)

extern "C" const char *rpe_comm_last_error(void) { return g_comm_err.c_str(); }

extern "C" int rpe_comm_unique_id(uint8_t id[RPE_COMM_ID_BYTES])
{
    if (!id) return RPE_ERR_INVALID;
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId u;
    NCHK(nullptr, g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, RPE_COMM_ID_BYTES);
    return RPE_OK;
}

// Set-up in two steps so that the ranks of a job can AGREE between them (sharding.PoseComm): everything that can fail
// on one rank alone (loading librccl, device memory) happens in rpe_comm_prepare; only rpe_comm_connect enters the
// collective ncclCommInitRank -- a rank that failed locally never leaves the others waiting inside it.
extern "C" int rpe_comm_prepare(rpe_handle *h, int rank, int world, rpe_comm **out)
{
    if (!h || !out || world < 1 || rank < 0 || rank >= world) return RPE_ERR_INVALID;
    *out = nullptr;
    int rc = load_rccl();
    if (rc) return rc;
    CHIP(hipSetDevice(h->cfg.device));
    rpe_comm *c = new rpe_comm();
    c->h = h; c->rank = rank; c->world = world;
    c->per_rank_cap = h->cfg.max_batch;
    const size_t rec = RPE_POSE_RECORD_BYTES;
    if (hipMalloc((void **)&c->d_send, rec * c->per_rank_cap) != hipSuccess ||
        hipMalloc((void **)&c->d_recv, rec * c->per_rank_cap * (size_t)world) != hipSuccess ||
        hipMalloc((void **)&c->d_scalar, sizeof(double) * 2) != hipSuccess) {
        g_comm_err = "rpe_comm_prepare: hipMalloc failed"; rpe_comm_destroy(c); return RPE_ERR_HIP;
    }
    *out = c;
    return RPE_OK;
}

extern "C" int rpe_comm_connect(rpe_comm *c, const uint8_t id[RPE_COMM_ID_BYTES])
{
    if (!c || !id || c->comm) return RPE_ERR_INVALID;
    CHIP(hipSetDevice(c->h->cfg.device));
    ncclUniqueId u;
    memcpy(u.internal, id, RPE_COMM_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, c->world, u, c->rank);
    if (r != 0) { c->comm = nullptr; g_comm_err = std::string("ncclCommInitRank failed: ") + g_rccl.GetErrorString(r); return RPE_ERR_HIP; }
    return RPE_OK;
}

extern "C" int rpe_comm_create(rpe_handle *h, int rank, int world, const uint8_t id[RPE_COMM_ID_BYTES], rpe_comm **out)
{
    if (!id || !out) return RPE_ERR_INVALID;
    int rc = rpe_comm_prepare(h, rank, world, out);
    if (rc) return rc;
    rc = rpe_comm_connect(*out, id);
    if (rc) { rpe_comm_destroy(*out); *out = nullptr; }
    return rc;
}

extern "C" int rpe_comm_destroy(rpe_comm *c)
{
    if (!c) return RPE_OK;
    if (c->h) hipSetDevice(c->h->cfg.device);
    if (c->h && c->h->stream) hipStreamSynchronize(c->h->stream);
    if (c->comm) g_rccl.CommDestroy(c->comm);
    if (c->d_send) hipFree(c->d_send);
    if (c->d_recv) hipFree(c->d_recv);
    if (c->d_scalar) hipFree(c->d_scalar);
    delete c;
    return RPE_OK;
}

// 128-byte pose record: R[9] f64, t[3] f64, inliers, status, n_matches, pair (global index; -1 = padding), 16 B pad
__global__ __launch_bounds__(256) void pack_records_kernel(const double *__restrict__ R, const double *__restrict__ t,
                                                            const int *__restrict__ inliers, const int *__restrict__ status,
                                                            const int *__restrict__ n_matches, int n_local, int per_rank, int first_pair,
                                                            uint8_t *__restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= per_rank) return;
    double *d = (double *)(out + (size_t)p * RPE_POSE_RECORD_BYTES);
    int *q = (int *)(d + 12);
    if (p < n_local) {
#pragma unroll
        for (int e = 0; e < 9; ++e) d[e] = R[(size_t)p * 9 + e];
#pragma unroll
        for (int e = 0; e < 3; ++e) d[9 + e] = t[(size_t)p * 3 + e];
        q[0] = inliers[p]; q[1] = status[p]; q[2] = n_matches[p]; q[3] = first_pair + p;
    } else {
#pragma unroll
        for (int e = 0; e < 12; ++e) d[e] = 0.;
        q[0] = 0; q[1] = 0; q[2] = 0; q[3] = -1;
    }
    q[4] = q[5] = q[6] = q[7] = 0;
}

extern "C" int rpe_gather_poses(rpe_handle *h, rpe_comm *c, int n_local, int per_rank, int first_pair, void *h_records)
{
    if (!h || !c || !c->comm || c->h != h || !h_records || n_local < 0 || per_rank < 1 || n_local > per_rank) return RPE_ERR_INVALID;
    if (per_rank > c->per_rank_cap || n_local > h->cfg.max_batch) { g_comm_err = "rpe_gather_poses: per_rank exceeds the handle's max_batch"; return RPE_ERR_CAPACITY; }
    CHIP(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(pack_records_kernel, dim3((per_rank + 255) / 256), dim3(256), 0, h->stream,
                       (const double *)h->d_R, (const double *)h->d_t, (const int *)h->d_inliers, (const int *)h->d_status,
                       (const int *)h->d_m_n, n_local, per_rank, first_pair, c->d_send);
    CHIP(hipGetLastError());
    const size_t bytes = (size_t)per_rank * RPE_POSE_RECORD_BYTES;
    NCHK(c, g_rccl.AllGather(c->d_send, c->d_recv, bytes, rpeNcclChar, c->comm, h->stream));
    CHIP(hipMemcpyAsync(h_records, c->d_recv, bytes * (size_t)c->world, hipMemcpyDeviceToHost, h->stream));
    CHIP(hipStreamSynchronize(h->stream));
    return RPE_OK;
}

// max over ranks of one host double (bench timing) -- doubles as the barrier of the step loop
extern "C" int rpe_comm_allreduce_max(rpe_comm *c, double *value)
{
    if (!c || !c->comm || !value) return RPE_ERR_INVALID;
    rpe_handle *h = c->h;
    CHIP(hipSetDevice(h->cfg.device));
    CHIP(hipMemcpyAsync(c->d_scalar, value, sizeof(double), hipMemcpyHostToDevice, h->stream));
    NCHK(c, g_rccl.AllReduce(c->d_scalar, c->d_scalar + 1, 1, rpeNcclFloat64, rpeNcclMax, c->comm, h->stream));
    CHIP(hipMemcpyAsync(value, c->d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CHIP(hipStreamSynchronize(h->stream));
    return RPE_OK;
}

extern "C" int rpe_comm_barrier(rpe_comm *c)
{
    double v = 0.;
    return rpe_comm_allreduce_max(c, &v);
}
