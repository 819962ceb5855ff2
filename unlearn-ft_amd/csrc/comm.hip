// pdmk_comm_t: the explicit communicator handle of the C ABI - the data-parallel gradient exchange of the
// bilevel step (DDP all-reduce inside accelerator.backward, pdm/training/trainer.py:117-129, 2782, 2808) as ONE RCCL
// all-reduce (sum, fp32, in place) per arena bucket on the caller's comm stream, over xGMI.
// RCCL is bound at run time (dlopen "librccl.so.1": the copy PyTorch-ROCm has already loaded when there is one, so that
// a process never runs two RCCL instances); libpdmk.so itself has no link-time dependency on it.
#include "common.h"

#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {

typedef struct { char internal[128]; } nccl_uid;                 // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* nccl_comm;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(nccl_comm*, int, nccl_uid, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_reduce_scatter)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, nccl_comm, hipStream_t);
typedef int (*fn_destroy)(nccl_comm);

struct Rccl {
    void* lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allreduce allreduce = nullptr;
    fn_reduce_scatter reduce_scatter = nullptr;
    fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr;
    bool ok = false;
};
std::mutex g_mu;
Rccl g_rccl;

const Rccl& rccl() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_rccl.lib) {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.lib) break;
        }
        if (g_rccl.lib) {
            g_rccl.get_uid = (fn_get_uid)dlsym(g_rccl.lib, "ncclGetUniqueId");
            g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.lib, "ncclCommInitRank");
            g_rccl.allreduce = (fn_allreduce)dlsym(g_rccl.lib, "ncclAllReduce");
            g_rccl.reduce_scatter = (fn_reduce_scatter)dlsym(g_rccl.lib, "ncclReduceScatter");
            g_rccl.allgather = (fn_allgather)dlsym(g_rccl.lib, "ncclAllGather");
            g_rccl.destroy = (fn_destroy)dlsym(g_rccl.lib, "ncclCommDestroy");
            g_rccl.ok = g_rccl.get_uid && g_rccl.init_rank && g_rccl.allreduce && g_rccl.reduce_scatter && g_rccl.allgather &&
                        g_rccl.destroy;
        }
    }
    return g_rccl;
}

}  // namespace

struct pdmk_comm {
    nccl_comm comm;
    int rank, world;
};

extern "C" int pdmk_comm_unique_id(void* out128) {
    if (!out128) return -1;
    const Rccl& r = rccl();
    if (!r.ok) return -2;
    nccl_uid id;
    const int rc = r.get_uid(&id);
    if (rc != 0) return -(2000 + rc);
    memcpy(out128, id.internal, 128);
    return 0;
}

extern "C" int pdmk_comm_create(const void* id128, int rank, int world, pdmk_comm_t* out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return -1;
    const Rccl& r = rccl();
    if (!r.ok) return -2;
    nccl_uid id;
    memcpy(id.internal, id128, 128);
    nccl_comm c = nullptr;
    const int rc = r.init_rank(&c, world, id, rank);          // uses the calling thread's current HIP device
    if (rc != 0 || !c) return -(2000 + rc);
    *out = new pdmk_comm{c, rank, world};
    return 0;
}

extern "C" int pdmk_comm_allreduce_sum_f32(pdmk_comm_t h, float* buf, int64_t n, pdmk_stream stream) {
    if (!h || !buf || n <= 0) return -1;
    const int rc = rccl().allreduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, h->comm, (hipStream_t)stream);
    return rc == 0 ? 0 : -(2000 + rc);
}

// The all-reduce split into its two halves: every rank first receives the SUM of its own 1/world share
// of the bucket directly from its 7 xGMI peers (reduce-scatter), may work on that share (the fused AdamW of exactly those
// parameters could run here), and the shares are then exchanged (all-gather).  n_per_rank elements per rank; the bucket is
// buf[0 .. world * n_per_rank) and rank r's share starts at buf + r * n_per_rank (in place: RCCL's in-place convention for
// both collectives).  A world of one rank is a no-op.
extern "C" int pdmk_comm_reduce_scatter_sum_f32(pdmk_comm_t h, float* buf, int64_t n_per_rank, pdmk_stream stream) {
    if (!h || !buf || n_per_rank <= 0) return -1;
    const int rc = rccl().reduce_scatter(buf, buf + (size_t)h->rank * (size_t)n_per_rank, (size_t)n_per_rank, /*ncclFloat32*/ 7,
                                         /*ncclSum*/ 0, h->comm, (hipStream_t)stream);
    return rc == 0 ? 0 : -(2000 + rc);
}

extern "C" int pdmk_comm_allgather_f32(pdmk_comm_t h, float* buf, int64_t n_per_rank, pdmk_stream stream) {
    if (!h || !buf || n_per_rank <= 0) return -1;
    const int rc = rccl().allgather(buf + (size_t)h->rank * (size_t)n_per_rank, buf, (size_t)n_per_rank, /*ncclFloat32*/ 7, h->comm,
                                    (hipStream_t)stream);
    return rc == 0 ? 0 : -(2000 + rc);
}

extern "C" int pdmk_comm_world(pdmk_comm_t h) { return h ? h->world : -1; }
extern "C" int pdmk_comm_rank(pdmk_comm_t h) { return h ? h->rank : -1; }

// Library-owned side streams ("a library-owned side stream for comm"): plain hipStreams created once per role by
// the host and kept for the life of the process - never handed out of a pool, so two roles of one step (teacher pass,
// streamed AdamW, dgrad-copy refresh, communication) can never alias one hipStream.
extern "C" int pdmk_stream_create(int high_priority, pdmk_stream* out) {
    if (!out) return -1;
    hipStream_t s = nullptr;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? hi : lo);
    if (e != hipSuccess) { (void)hipGetLastError(); return -(1000 + (int)e); }
    *out = (pdmk_stream)s;
    return 0;
}

extern "C" int pdmk_stream_destroy(pdmk_stream s) {
    if (!s) return -1;
    const hipError_t e = hipStreamDestroy((hipStream_t)s);
    if (e != hipSuccess) { (void)hipGetLastError(); return -(1000 + (int)e); }
    return 0;
}

extern "C" int pdmk_comm_destroy(pdmk_comm_t h) {
    if (!h) return -1;
    const int rc = rccl().destroy(h->comm);
    delete h;
    return rc == 0 ? 0 : -(2000 + rc);
}
