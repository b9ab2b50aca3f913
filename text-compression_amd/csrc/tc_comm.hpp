// tc_comm.hpp -- the one exchange of the multi-GPU path behind the C ABI: records are
// independent, one per GPU, nothing is exchanged during the encode; at the end the variable-size
// containers are gathered on one rank -- sizes by an all-gather of one word per rank, then ONE group in
// which the root posts a receive per peer and every peer one send, so that the root ingests on all of its
// xGMI links at once (a ring would be bound by a single link) -- and an FM-index is replicated by one
// broadcast.  The reference has no counterpart: its only parallelism is parListChunk over the pattern
// list inside one process (FMIndex.hs:417-423).
//
// RCCL is bound at run time (dlopen; the copy the process has already loaded, e.g. PyTorch's, is
// preferred), so libtextcomp.so itself does not depend on it: a single-GPU user never loads it, and a
// missing / failing RCCL is the status code TC_ERR_NCCL, not a load error.
#pragma once
#include <dlfcn.h>
#include <stdlib.h>

#include <mutex>

#include "tc_common.hpp"

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value */ struct RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
struct RcclId {
    char internal[TC_COMM_ID_BYTES];   // = ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
};
enum { kNcclUint8 = 1, kNcclUint64 = 5 };   // ncclDataType_t

// TC_RCCL_LIB (tests, unusual installs): the one library name to bind instead of the list below.
static RcclApi *rccl_api(std::string *why) {
    static RcclApi api;
    static std::string err;
    static std::once_flag once;      // tc_comm_* may be entered from several contexts / threads at once
    std::call_once(once, [] {
        const char *only = getenv("TC_RCCL_LIB");
        const char *dflt[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        std::vector<const char *> names;
        if (only && *only) names.push_back(only);
        else names.assign(dflt, dflt + 3);
        for (const char *n : names)
            if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // a copy already in the process
        for (const char *n : names)
            if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!api.lib) {
            const char *e = dlerror();    // (one call: dlerror() clears the message it returns)
            err = std::string("librccl not found: ") + (e ? e : "?");
            return;
        }
        auto sym = [&](const char *s) {
            void *p = dlsym(api.lib, s);
            if (!p && err.empty()) err = std::string("librccl lacks ") + s;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (!err.empty()) {
        if (why) *why = err;
        return nullptr;
    }
    return &api;
}

struct tc_comm {
    tc_ctx *ctx = nullptr;
    RcclApi *api = nullptr;
    void *comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;   // the exchange runs beside the encoder's stream
    hipEvent_t ev_ready = nullptr;  // recorded on the encoder's stream when a gather is posted: the exchange waits for it on the device
    u64 *d_words = nullptr;         // [1 + world] my size, all sizes
    u64 *h_words = nullptr;         // pinned mirror
    bool inflight = false;
    int cus = 0;                    // compute units the exchange is restricted to (0: any)
};

#define TC_NCCL(c, expr)                                                                         \
    do {                                                                                         \
        int r__ = (expr);                                                                        \
        if (r__ != 0) {                                                                          \
            char b__[512];                                                                       \
            snprintf(b__, sizeof b__, "%s -> %s", #expr, (c)->api->GetErrorString(r__));         \
            (c)->ctx->err = b__;                                                                 \
            throw TcFail{TC_ERR_NCCL};                                                           \
        }                                                                                        \
    } while (0)

static void comm_release(tc_comm *c) {
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && c->api) (void)c->api->CommDestroy(c->comm);
    if (c->d_words) (void)hipFree(c->d_words);
    if (c->h_words) (void)hipHostFree(c->h_words);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
