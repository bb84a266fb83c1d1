// The one exchange step of the sharded path as a C entry point: an all-gather of the fixed-size per-image result records
// over RCCL (xGMI inside a node) - SURVEY 8b's `ecseg_allgather_records`, for hosts that bind the library without
// torch.distributed (the reference's TF environment).  The reference has no counterpart: its only parallelism is
// tf.distribute.MirroredStrategy around load_model (src/metaseg.py:33-36); every image is independent (src/metaseg.py:42),
// so whole images are sharded and only these 128-byte records ever cross devices.
//
// RCCL is resolved with dlopen at first use: the library has no link-time dependency on librccl.so and the single-GPU path
// never loads it.  Rendezvous is the caller's business: rank 0 calls ecseg_comm_unique_id and hands the 128 bytes to the
// other ranks by whatever channel the host has (a file next to config.yaml, an environment variable, torch's store).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "../../include/ecseg_hip.h"

struct ecseg_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    int64_t* d_send = nullptr;
    int64_t* d_recv = nullptr;
    size_t cap_records = 0;              // capacity of d_send in records (d_recv: world times that)
};

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

thread_local std::string g_comm_error;

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
    });
    return r;
}

int comm_fail(int code, const std::string& msg) { g_comm_error = msg; return code; }

int nccl_fail(ncclResult_t e, const char* what) {
    Rccl& r = rccl();
    return comm_fail(ECSEG_E_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}

int ensure(ecseg_comm* c, size_t n_records) {
    if (n_records <= c->cap_records) return ECSEG_OK;
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    c->d_send = c->d_recv = nullptr; c->cap_records = 0;
    const size_t bytes = n_records * ECSEG_RECORD_INT64 * sizeof(int64_t);
    if (hipMalloc(reinterpret_cast<void**>(&c->d_send), bytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_recv), bytes * (size_t)c->world) != hipSuccess)
        return comm_fail(ECSEG_E_NOMEM, "hipMalloc(record buffers)");
    c->cap_records = n_records;
    return ECSEG_OK;
}

}  // namespace

extern "C" {

const char* ecseg_comm_last_error(void) { return g_comm_error.c_str(); }

int ecseg_comm_unique_id(void* out, int out_bytes) {
    if (!out || out_bytes < ECSEG_COMM_ID_BYTES) return comm_fail(ECSEG_E_INVALID, "unique id buffer too small");
    Rccl& r = rccl();
    if (!r.ok) return comm_fail(ECSEG_E_UNSUPPORTED, "librccl.so could not be loaded");
    static_assert(sizeof(ncclUniqueId) == ECSEG_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    const ncclResult_t e = r.GetUniqueId(&id);
    if (e != ncclSuccess) return nccl_fail(e, "ncclGetUniqueId");
    std::memcpy(out, &id, sizeof id);
    return ECSEG_OK;
}

int ecseg_comm_create(ecseg_comm** out, const void* unique_id, int rank, int world, int device_id) {
    if (!out || !unique_id || world < 1 || rank < 0 || rank >= world) return comm_fail(ECSEG_E_INVALID, "bad communicator arguments");
    Rccl& r = rccl();
    if (!r.ok) return comm_fail(ECSEG_E_UNSUPPORTED, "librccl.so could not be loaded");
    if (hipSetDevice(device_id) != hipSuccess) return comm_fail(ECSEG_E_HIP, "hipSetDevice");
    ecseg_comm* c = new ecseg_comm;
    c->rank = rank; c->world = world; c->device = device_id;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    const ncclResult_t e = r.CommInitRank(&c->comm, world, id, rank);
    if (e != ncclSuccess) { delete c; return nccl_fail(e, "ncclCommInitRank"); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)r.CommDestroy(c->comm);
        delete c;
        return comm_fail(ECSEG_E_HIP, "hipStreamCreate");
    }
    *out = c;
    return ECSEG_OK;
}

void ecseg_comm_destroy(ecseg_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    delete c;
}

// Device pointers, caller's stream (hipStream_t as void*, null: the communicator's own stream); returns after the
// collective has been ENQUEUED when a stream is given, after it has completed otherwise.
int ecseg_allgather_records_dev(ecseg_comm* c, const int64_t* send_dev, int n_records, int64_t* recv_dev, void* stream) {
    if (!c || !send_dev || !recv_dev || n_records < 0) return comm_fail(ECSEG_E_INVALID, "bad all-gather arguments");
    if (n_records == 0) return ECSEG_OK;
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail(ECSEG_E_HIP, "hipSetDevice");
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const ncclResult_t e = rccl().AllGather(send_dev, recv_dev, (size_t)n_records * ECSEG_RECORD_INT64, ncclInt64, c->comm, s);
    if (e != ncclSuccess) return nccl_fail(e, "ncclAllGather");
    if (!stream && hipStreamSynchronize(s) != hipSuccess) return comm_fail(ECSEG_E_HIP, "hipStreamSynchronize");
    return ECSEG_OK;
}

// Host pointers: n_records records of ECSEG_RECORD_INT64 int64 each in, world * n_records out (rank-major); synchronous.
int ecseg_allgather_records(ecseg_comm* c, const int64_t* send, int n_records, int64_t* recv) {
    if (!c || !send || !recv || n_records < 0) return comm_fail(ECSEG_E_INVALID, "bad all-gather arguments");
    if (n_records == 0) return ECSEG_OK;
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail(ECSEG_E_HIP, "hipSetDevice");
    int rc = ensure(c, (size_t)n_records);
    if (rc != ECSEG_OK) return rc;
    const size_t bytes = (size_t)n_records * ECSEG_RECORD_INT64 * sizeof(int64_t);
    if (hipMemcpyAsync(c->d_send, send, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return comm_fail(ECSEG_E_HIP, "hipMemcpy H2D");
    const ncclResult_t e = rccl().AllGather(c->d_send, c->d_recv, (size_t)n_records * ECSEG_RECORD_INT64, ncclInt64, c->comm, c->stream);
    if (e != ncclSuccess) return nccl_fail(e, "ncclAllGather");
    if (hipMemcpyAsync(recv, c->d_recv, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return comm_fail(ECSEG_E_HIP, "hipMemcpy D2H");
    return ECSEG_OK;
}

}  // extern "C"
