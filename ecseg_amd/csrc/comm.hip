// The one exchange step of the sharded path as a C entry point: an all-gather of the fixed-size per-image result records
// over RCCL (xGMI inside a node) - SURVEY 8b's `ecseg_allgather_records`, for hosts that bind the library without
// torch.distributed (the reference's TF environment).  The reference has no counterpart: its only parallelism is
// tf.distribute.MirroredStrategy around load_model (src/metaseg.py:33-36); every image is independent (src/metaseg.py:42),
// so whole images are sharded and only these 128-byte records ever cross devices.
//
// Every wait is bounded (ECSEG_COMM_TIMEOUT_S seconds, default 300).  ncclCommInitRank runs on a helper thread that the
// caller waits for with a deadline (RCCL 2.27's "non-blocking" ncclCommInitRankConfig(blocking = 0) does not return either
// while a peer is missing - measured on MI355X, round 5); the collective's completion is polled with hipStreamQuery.  When
// the limit expires - a peer never arrived, or died between its U-Net and the exchange - the call returns ECSEG_E_HIP with
// a message instead of hanging for ever (the reference's analogue fails loudly: src/metaseg.py:33-36).  A stuck
// initialisation is told to give up through the communicator pointer RCCL publishes before it waits for its peers
// (ncclCommAbort; measured: the helper unwinds within 2 s and the process then exits cleanly - without it a process that
// exits with a thread still inside RCCL's bootstrap crashes in the library's static destructors); a stuck collective is
// aborted on a detached thread, so that an abort which cannot finish either does not take the caller along.
//
// RCCL is resolved with dlopen at first use: the library has no link-time dependency on librccl.so and the single-GPU path
// never loads it.  Rendezvous is the caller's business: rank 0 calls ecseg_comm_unique_id and hands the 128 bytes to the
// other ranks by whatever channel the host has (a file next to config.yaml, an environment variable, torch's store).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/ecseg_hip.h"

struct ecseg_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    int64_t* d_send = nullptr;
    int64_t* d_recv = nullptr;
    size_t cap_records = 0;              // capacity of d_send in records (d_recv: world times that)
    bool dead = false;                   // aborted after a timeout: every further call fails
};

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;     // optional
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
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
    });
    return r;
}

using Clock = std::chrono::steady_clock;

double timeout_s() {
    const char* v = std::getenv("ECSEG_COMM_TIMEOUT_S");
    if (v && *v) {
        char* end = nullptr;
        const double t = std::strtod(v, &end);
        if (end != v && t > 0.0) return t;
    }
    return 300.0;
}
Clock::time_point deadline_from_now() {
    return Clock::now() + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(timeout_s()));
}

int comm_fail(int code, const std::string& msg) { g_comm_error = msg; return code; }

int nccl_fail(ncclResult_t e, const char* what) {
    Rccl& r = rccl();
    return comm_fail(ECSEG_E_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}

// Abort a communicator whose peers did not show up in time; it is unusable afterwards.
std::string timeout_text(const std::string& what, int rank, int world) {
    char buf[64];
    snprintf(buf, sizeof buf, "%.0f", timeout_s());
    return what + ": no completion within " + buf + " s (ECSEG_COMM_TIMEOUT_S): a peer of rank " + std::to_string(rank) + " of " +
           std::to_string(world) + " is missing or stuck; the communicator was aborted";
}

// ncclCommAbort on a detached thread: an abort that blocks (peers gone in the middle of a collective) must not block the caller.
void abort_detached(ncclComm_t comm) {
    Rccl& r = rccl();
    if (!comm || !r.CommAbort) return;
    auto fn = r.CommAbort;
    std::thread([fn, comm] { (void)fn(comm); }).detach();
}

// Abort a communicator whose peers did not answer in time; it is unusable afterwards.
int abort_comm(ecseg_comm* c, const std::string& what) {
    abort_detached(c->comm);
    c->comm = nullptr;
    c->dead = true;
    return comm_fail(ECSEG_E_HIP, timeout_text(what, c->rank, c->world));
}

// ncclCommInitRank with a deadline: the call itself runs on a helper thread.  If it does not return in time it is aborted
// through the early communicator pointer; should even that fail, the helper is left behind (it owns its state through the
// shared_ptr) and, if it ever finishes, disposes of the communicator itself.
struct InitState {
    std::mutex m;
    std::condition_variable cv;
    bool done = false, abandoned = false, caller_aborted = false;      // caller_aborted: the waiting side has already called ncclCommAbort(early)
    volatile ncclComm_t early = nullptr;
    ncclComm_t comm = nullptr;
    ncclResult_t res = ncclSuccess;
};

int init_with_deadline(ecseg_comm* c, const ncclUniqueId& id) {
    Rccl& r = rccl();
    auto st = std::make_shared<InitState>();
    const int world = c->world, rank = c->rank, device = c->device;
    auto init = r.CommInitRank;
    auto abort_fn = r.CommAbort;
    auto destroy_fn = r.CommDestroy;
    std::thread([st, init, abort_fn, destroy_fn, id, world, rank, device] {
        // (RCCL stores the communicator through the pointer BEFORE it starts waiting for the peers: `early` lets the caller
        // abort an initialisation that is stuck in the bootstrap)
        ncclResult_t res = hipSetDevice(device) == hipSuccess ? init(const_cast<ncclComm_t*>(&st->early), world, id, rank) : ncclUnhandledCudaError;
        std::unique_lock<std::mutex> lk(st->m);
        st->comm = st->early; st->res = res; st->done = true;
        const bool orphan = st->abandoned && !st->caller_aborted;          // (never abort one communicator twice)
        lk.unlock();
        st->cv.notify_all();
        if (orphan && res == ncclSuccess && st->comm) { if (abort_fn) (void)abort_fn(st->comm); else (void)destroy_fn(st->comm); }
    }).detach();
    std::unique_lock<std::mutex> lk(st->m);
    const bool ok = st->cv.wait_until(lk, deadline_from_now(), [&] { return st->done; });
    if (!ok) {
        st->abandoned = true;
        c->dead = true;
        const ncclComm_t early = st->early;
        if (early && abort_fn) st->caller_aborted = true;
        lk.unlock();
        if (early && abort_fn) {
            // ask the stuck initialisation to give up and give the helper a moment to unwind: a process that exits while a
            // thread is still inside RCCL's bootstrap crashes in the library's static destructors
            (void)abort_fn(early);
            std::unique_lock<std::mutex> lk2(st->m);
            (void)st->cv.wait_for(lk2, std::chrono::seconds(5), [&] { return st->done; });
        }
        return comm_fail(ECSEG_E_HIP, timeout_text("ncclCommInitRank", rank, world));
    }
    if (st->res != ncclSuccess) return nccl_fail(st->res, "ncclCommInitRank");
    c->comm = st->comm;
    return ECSEG_OK;
}

// Completion of everything enqueued on `s`, polled: hipStreamSynchronize would wait for a missing peer for ever.
int wait_stream(ecseg_comm* c, hipStream_t s, Clock::time_point deadline, const char* what) {
    for (int spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return ECSEG_OK;
        if (q != hipErrorNotReady) return comm_fail(ECSEG_E_HIP, std::string(what) + ": " + hipGetErrorString(q));
        if (Clock::now() >= deadline) return abort_comm(c, what);
        if (spins < 2000) std::this_thread::yield();          // (a healthy exchange of a few KB completes within microseconds)
        else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
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
    const int rci = init_with_deadline(c, id);
    if (rci != ECSEG_OK) { delete c; return rci; }
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
    if (c->stream && !c->dead) { (void)hipStreamSynchronize(c->stream); }
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
    if (c->dead) return comm_fail(ECSEG_E_HIP, "the communicator was aborted after a timeout");
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    const Clock::time_point deadline = deadline_from_now();
    const ncclResult_t e = rccl().AllGather(send_dev, recv_dev, (size_t)n_records * ECSEG_RECORD_INT64, ncclInt64, c->comm, s);
    if (e != ncclSuccess) return nccl_fail(e, "ncclAllGather");
    // (a caller's stream: enqueued only - the caller owns the wait, and its bound)
    if (!stream) return wait_stream(c, s, deadline, "ncclAllGather");
    return ECSEG_OK;
}

// Host pointers: n_records records of ECSEG_RECORD_INT64 int64 each in, world * n_records out (rank-major); synchronous.
int ecseg_allgather_records(ecseg_comm* c, const int64_t* send, int n_records, int64_t* recv) {
    if (!c || !send || !recv || n_records < 0) return comm_fail(ECSEG_E_INVALID, "bad all-gather arguments");
    if (n_records == 0) return ECSEG_OK;
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail(ECSEG_E_HIP, "hipSetDevice");
    if (c->dead) return comm_fail(ECSEG_E_HIP, "the communicator was aborted after a timeout");
    int rc = ensure(c, (size_t)n_records);
    if (rc != ECSEG_OK) return rc;
    const size_t bytes = (size_t)n_records * ECSEG_RECORD_INT64 * sizeof(int64_t);
    const Clock::time_point deadline = deadline_from_now();
    if (hipMemcpyAsync(c->d_send, send, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return comm_fail(ECSEG_E_HIP, "hipMemcpy H2D");
    const ncclResult_t e = rccl().AllGather(c->d_send, c->d_recv, (size_t)n_records * ECSEG_RECORD_INT64, ncclInt64, c->comm, c->stream);
    if (e != ncclSuccess) return nccl_fail(e, "ncclAllGather");
    // the copy into the caller's memory is issued only after the collective has COMPLETED: a copy queued behind a collective that
    // times out would run once the abort unblocks the stream - into a buffer the caller may have freed by then
    rc = wait_stream(c, c->stream, deadline, "ncclAllGather");
    if (rc != ECSEG_OK) return rc;
    if (hipMemcpy(recv, c->d_recv, bytes * (size_t)c->world, hipMemcpyDeviceToHost) != hipSuccess) return comm_fail(ECSEG_E_HIP, "hipMemcpy D2H");
    return ECSEG_OK;
}

}  // extern "C"
