// urso_comm_*: the data-parallel exchange step of the path behind the C ABI -- bucketed gradient averaging over
// RCCL, one communicator per process (one GPU per process), collectives on the communicator's OWN stream so that they overlap the
// backward kernels still being enqueued on the compute stream:
//     urso_comm_allreduce_bucket(comm, bucket, n, dt, compute_stream)   after the kernels that finish a bucket were enqueued
//     ...more backward kernels, more buckets...
//     urso_comm_wait(comm, compute_stream)                             before the optimizer reads the gradients
// Replaces Keras' ParallelModel + the implicit gradient gather of multi_gpu_model (pose_estimator.py:28; net.py compile) by what
// DESIGN.md section 7 describes.  RCCL is bound at run time (dlopen): the library loads, and every other entry point works, on a host
// without RCCL; a process that already holds an RCCL (PyTorch's) gets that same copy.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

bool rccl_load() {
    if (g_rccl.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;       // the copy this process already uses, if any
    if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { urso_set_error("urso_comm: RCCL not found (%s)", dlerror()); return false; }
    RcclApi a; a.handle = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
        urso_set_error("urso_comm: RCCL symbols missing"); return false;
    }
    g_rccl = a;
    return true;
}
int rccl_fail(const char* what, ncclResult_t r) { urso_set_error("%s: %s", what, g_rccl.GetErrorString(r)); return URSO_ELAUNCH; }
int hip_fail(const char* what, hipError_t e) { urso_set_error("%s: %s", what, hipGetErrorString(e)); return URSO_ELAUNCH; }
}  // namespace

struct urso_comm {
    ncclComm_t comm;
    hipStream_t stream;          // the collectives' stream
    hipEvent_t ready, done;      // bucket complete on the compute stream / collectives complete
    int world, rank;
    bool pending;
};

extern "C" int urso_comm_unique_id(void* id_out) {
    if (!id_out) { urso_set_error("urso_comm_unique_id: null"); return URSO_EINVAL; }
    if (!rccl_load()) return URSO_ELAUNCH;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id_out, &id, URSO_COMM_ID_BYTES);
    return URSO_OK;
}

extern "C" int urso_comm_init(urso_comm** out, int world, int rank, const void* id) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) { urso_set_error("urso_comm_init: bad argument"); return URSO_EINVAL; }
    static_assert(URSO_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!rccl_load()) return URSO_ELAUNCH;
    urso_comm* c = new urso_comm();
    c->world = world; c->rank = rank; c->pending = false;
    ncclUniqueId uid; memcpy(&uid, id, URSO_COMM_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) { delete c; return rccl_fail("ncclCommInitRank", r); }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) { g_rccl.CommDestroy(c->comm); delete c; return hip_fail("urso_comm_init", e); }
    *out = c;
    return URSO_OK;
}

extern "C" int urso_comm_allreduce_bucket(urso_comm* c, void* buf_d, size_t count, int dt, void* compute_stream) {
    if (!c || !buf_d || count == 0 || (dt != URSO_F32 && dt != URSO_BF16 && dt != URSO_F16)) { urso_set_error("urso_comm_allreduce_bucket: bad argument"); return URSO_EINVAL; }
    hipStream_t cs = (hipStream_t)compute_stream;
    hipError_t e = hipEventRecord(c->ready, cs);                 // everything enqueued so far produced this bucket
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ready, 0);
    if (e != hipSuccess) return hip_fail("urso_comm_allreduce_bucket", e);
    const ncclDataType_t t = dt == URSO_F32 ? ncclFloat32 : (dt == URSO_BF16 ? ncclBfloat16 : ncclFloat16);
    ncclResult_t r = g_rccl.AllReduce(buf_d, buf_d, count, t, ncclAvg, c->comm, c->stream);
    if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    c->pending = true;
    return URSO_OK;
}

extern "C" int urso_comm_wait(urso_comm* c, void* compute_stream) {
    if (!c) { urso_set_error("urso_comm_wait: null"); return URSO_EINVAL; }
    if (!c->pending) return URSO_OK;
    hipError_t e = hipEventRecord(c->done, c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)compute_stream, c->done, 0);
    if (e != hipSuccess) return hip_fail("urso_comm_wait", e);
    c->pending = false;
    return URSO_OK;
}

extern "C" int urso_comm_destroy(urso_comm* c) {
    if (!c) return URSO_OK;
    hipStreamSynchronize(c->stream);
    g_rccl.CommDestroy(c->comm);
    hipEventDestroy(c->ready); hipEventDestroy(c->done); hipStreamDestroy(c->stream);
    delete c;
    return URSO_OK;
}

// ---------------------------------------------------------------- 16-bit gradient buckets with error feedback (ursonet_amd/dp.py GradReducer, compress = 'bf16')
// One pass instead of three torch elementwise launches per bucket:  t = g + resid;  c = bf16(t);  resid = t - float(c).  g itself is left
// alone: the averaged bucket is expanded over it by urso_bucket_expand_bf16 once the collective is done.  Round-to-nearest-even, the
// conversion torch's copy_ uses, so that the two forms agree bit for bit (tests/test_kernels_gpu.py).
__global__ void bucket_round_ef_kernel(size_t n, const float* __restrict__ g, float* __restrict__ resid, __bf16* __restrict__ c) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4_t gv = ((const f32x4_t*)g)[i], rv = ((const f32x4_t*)resid)[i];
        float t[4] = {gv.x + rv.x, gv.y + rv.y, gv.z + rv.z, gv.w + rv.w};
        __bf16 cb[4]; f32x4_t rn;
#pragma unroll
        for (int k = 0; k < 4; ++k) { cb[k] = Elem<__bf16>::from_f(t[k]); rn[k] = t[k] - Elem<__bf16>::to_f(cb[k]); }
        uint64_t pk; __builtin_memcpy(&pk, cb, 8);
        ((uint64_t*)c)[i] = pk;
        ((f32x4_t*)resid)[i] = rn;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) { const float t = g[i] + resid[i]; const __bf16 cb = Elem<__bf16>::from_f(t); c[i] = cb; resid[i] = t - Elem<__bf16>::to_f(cb); }
}
extern "C" int urso_bucket_round_ef(size_t n, const float* g_d, float* resid_d, void* c_bf16_d, void* stream) {
    if (!g_d || !resid_d || !c_bf16_d) { urso_set_error("urso_bucket_round_ef: null argument"); return URSO_EINVAL; }
    if (((((uintptr_t)g_d) | ((uintptr_t)resid_d)) & 15) || (((uintptr_t)c_bf16_d) & 7)) { urso_set_error("urso_bucket_round_ef: g / resid must be 16-byte, c 8-byte aligned"); return URSO_EINVAL; }
    if (n == 0) return URSO_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_OPTIM, 0, (double)n * 14);
    size_t blocks = (n / 4 + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    URSO_KLAUNCH(bucket_round_ef_kernel, dim3((int)blocks), dim3(256), 0, st, n, g_d, resid_d, (__bf16*)c_bf16_d);
    return urso_check_launch("urso_bucket_round_ef");
}

__global__ void bucket_expand_kernel(size_t n, const __bf16* __restrict__ c, float* __restrict__ g) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t pk = ((const uint64_t*)c)[i];
        __bf16 cb[4]; __builtin_memcpy(cb, &pk, 8);
        ((f32x4_t*)g)[i] = f32x4_t{Elem<__bf16>::to_f(cb[0]), Elem<__bf16>::to_f(cb[1]), Elem<__bf16>::to_f(cb[2]), Elem<__bf16>::to_f(cb[3])};
    }
    if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) g[i] = Elem<__bf16>::to_f(c[i]);
}
extern "C" int urso_bucket_expand_bf16(size_t n, const void* c_bf16_d, float* g_d, void* stream) {
    if (!g_d || !c_bf16_d) { urso_set_error("urso_bucket_expand_bf16: null argument"); return URSO_EINVAL; }
    if ((((uintptr_t)g_d) & 15) || (((uintptr_t)c_bf16_d) & 7)) { urso_set_error("urso_bucket_expand_bf16: g must be 16-byte, c 8-byte aligned"); return URSO_EINVAL; }
    if (n == 0) return URSO_OK;
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_OPTIM, 0, (double)n * 6);
    size_t blocks = (n / 4 + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    URSO_KLAUNCH(bucket_expand_kernel, dim3((int)blocks), dim3(256), 0, st, n, (const __bf16*)c_bf16_d, g_d);
    return urso_check_launch("urso_bucket_expand_bf16");
}
