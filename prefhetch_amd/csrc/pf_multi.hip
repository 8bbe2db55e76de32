// pf_multi.hip -- device group: ONE process drives the G GPUs of a node.
//
// The reference server is a single process on a single device (/root/reference/src/server/server_lib.cpp:48-53).
// Its units of work -- one query of the IndexFlatL2 pre-filter, one ciphertext x plaintext product -- are
// independent, so the group splits a batch contiguously over its members, replicates the RNS tables and the fp32
// base matrix on every member, and exchanges exactly once: ONE all-gather of the packed per-member top-k blocks,
// which the selection kernel of each member has written IN PLACE at its offset of the gathered buffer
// (pf_flat_search_packed), so the collective is an in-place ncclAllGather with no packing pass before it.
//
// Members: one host thread + one non-blocking HIP stream per device.  A group call hands one job to every member
// thread; the threads enqueue on their streams and report back; the caller then issues the exchange
// (ncclGroupStart / ncclAllGather per member / ncclGroupEnd -- RCCL's single-thread multi-device form).
// RCCL is loaded with dlopen on first use: librccl.so is 570 MB and the single-device path never needs it.
// When a device is listed twice (a one-GPU machine rehearsing the control flow) RCCL refuses the communicator;
// the exchange is then direct device-to-device copies ordered by events (also selectable for distinct devices).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "pf_common.hpp"

using namespace pf;

namespace {

// ---- the few RCCL entry points used, resolved at run time (types as in <rccl/rccl.h>) ----------------------------
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;                      // ncclSuccess = 0
constexpr int kNcclUint32 = 3;                 // ncclDataType_t: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
    bool load() {
        if (handle) return true;
        const char *env = getenv("PF_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
            why = dlerror();
        }
        if (!handle) return false;
        auto sym = [&](const char *n) { void *p = dlsym(handle, n); if (!p) why = std::string("missing symbol ") + n; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) { dlclose(handle); handle = nullptr; return false; }
        return true;
    }
};
Rccl &rccl() { static Rccl r; return r; }

// ---- one member = one device, one stream, one host thread ----------------------------------------------------------
struct Member {
    int rank = 0, device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_done = nullptr;                        // peer-copy exchange: own pushes done
    hipEvent_t ev_free = nullptr;                        // peer-copy exchange: recorded on this member's stream at the start of an exchange --
                                                         // whatever was queued before (consumers of the PREVIOUS result in this member's gathered
                                                         // buffer) is ordered before the other members' pushes into that buffer
    pf_ctx *ctx = nullptr;
    pf_flat *flat = nullptr;
    ncclComm_t comm = nullptr;
    void *stage = nullptr; size_t stage_bytes = 0;       // host-buffer entry point: query shard + gathered block on the device
    // worker
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<pf_status()> job;
    bool has_job = false, quit = false, done = false;
    pf_status result = PF_OK;
    std::string err;
};

}  // namespace

struct pf_multi {
    std::vector<Member *> mem;
    int exchange = PF_MULTI_PEER_COPY;
    std::mutex call;                                     // group calls are serialised

    // run fn(member) on every member thread, wait for all; first failure wins (its message becomes the caller's)
    pf_status run_all(const std::function<pf_status(Member &)> &fn) {
        for (Member *w : mem) {
            std::lock_guard<std::mutex> lk(w->m);
            w->job = [w, &fn]() { return fn(*w); };
            w->has_job = true; w->done = false;
            w->cv.notify_all();
        }
        pf_status st = PF_OK;
        for (Member *w : mem) {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [w] { return w->done; });
            if (st == PF_OK && w->result != PF_OK) { st = w->result; last_error_ref() = "member " + std::to_string(w->rank) + " (device " + std::to_string(w->device) + "): " + w->err; }
        }
        return st;
    }
};

namespace {

void worker_main(Member *w) {
    (void)hipSetDevice(w->device);
    for (;;) {
        std::function<pf_status()> job;
        {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [w] { return w->has_job || w->quit; });
            if (w->quit) return;
            job = std::move(w->job);
            w->has_job = false;
        }
        last_error_ref().clear();
        const pf_status st = job();
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->result = st;
            w->err = st == PF_OK ? std::string() : last_error_ref();
            w->done = true;
        }
        w->cv.notify_all();
    }
}

// the ONE exchange: every member's block [n_local words] sits in place at gathered[r] + r * words
pf_status exchange_blocks(pf_multi *g, uint32_t *const *gathered, size_t words) {
    const int G = (int)g->mem.size();
    if (G == 1 && g->exchange != PF_MULTI_RCCL) return PF_OK;         // a one-member RCCL group still runs its collective
    if (g->exchange == PF_MULTI_RCCL) {
        Rccl &r = rccl();
        ncclResult_t rc = r.GroupStart();
        for (int i = 0; i < G && rc == 0; ++i)
            rc = r.AllGather(gathered[i] + (size_t)i * words, gathered[i], words, kNcclUint32, g->mem[i]->comm, g->mem[i]->stream);
        const ncclResult_t rc2 = r.GroupEnd();
        if (rc == 0) rc = rc2;
        if (rc != 0) return fail(PF_ERR_HIP, std::string("ncclAllGather: ") + r.GetErrorString(rc));
        return PF_OK;
    }
    // direct copies: member r pushes its block into every other member's buffer on its own stream, then every stream
    // waits for all pushes, so that whatever the caller enqueues next on a member's stream sees the complete result.
    // A push lands in ANOTHER member's buffer, which that member's stream may still be reading (a consumer of the previous
    // call's result, a device-to-host copy of it): every member first records "my buffer is free" behind everything already
    // queued on its stream, and a pusher waits for the target's event before it writes (write-after-read across streams).
    const pf_status freed = g->run_all([&](Member &m) -> pf_status {
        PF_GUARD(m.device);
        PF_HIP(hipEventRecord(m.ev_free, m.stream));
        return PF_OK;
    });
    if (freed != PF_OK) return freed;
    const pf_status pushed = g->run_all([&](Member &m) -> pf_status {
        PF_GUARD(m.device);
        const uint32_t *src = gathered[m.rank] + (size_t)m.rank * words;
        for (int t = 0; t < G; ++t) {
            if (t == m.rank) continue;
            PF_HIP(hipStreamWaitEvent(m.stream, g->mem[t]->ev_free, 0));
            uint32_t *dst = gathered[t] + (size_t)m.rank * words;
            if (g->mem[t]->device == m.device) PF_HIP(hipMemcpyAsync(dst, src, words * 4, hipMemcpyDeviceToDevice, m.stream));
            else PF_HIP(hipMemcpyPeerAsync(dst, g->mem[t]->device, src, m.device, words * 4, m.stream));
        }
        PF_HIP(hipEventRecord(m.ev_done, m.stream));
        return PF_OK;
    });
    if (pushed != PF_OK) return pushed;
    // run_all returned: every ev_done is recorded, so the waits below refer to this exchange's pushes
    return g->run_all([&](Member &m) -> pf_status {
        PF_GUARD(m.device);
        for (int t = 0; t < G; ++t)
            if (t != m.rank) PF_HIP(hipStreamWaitEvent(m.stream, g->mem[t]->ev_done, 0));
        return PF_OK;
    });
}

}  // namespace

extern "C" {

pf_status pf_multi_destroy(pf_multi *g) {
    if (!g) return PF_OK;
    for (Member *w : g->mem) {
        if (w->th.joinable()) {
            { std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
            w->cv.notify_all();
            w->th.join();
        }
        DeviceGuard guard(w->device);
        if (w->stream) (void)hipStreamSynchronize(w->stream);
        if (w->comm && rccl().handle) (void)rccl().CommDestroy(w->comm);
        if (w->ctx) (void)pf_ctx_destroy(w->ctx);
        if (w->flat) (void)pf_flat_destroy(w->flat);
        if (w->stage) (void)hipFree(w->stage);
        if (w->ev_done) (void)hipEventDestroy(w->ev_done);
        if (w->ev_free) (void)hipEventDestroy(w->ev_free);
        if (w->stream) (void)hipStreamDestroy(w->stream);
        delete w;
    }
    delete g;
    return PF_OK;
}

pf_status pf_multi_create(pf_multi **out, const int *devices, int n, int exchange) {
    if (!out || !devices || n <= 0) return fail(PF_ERR_INVALID_ARG, "null argument or no devices");
    *out = nullptr;
    if (exchange != PF_MULTI_AUTO && exchange != PF_MULTI_RCCL && exchange != PF_MULTI_PEER_COPY) return fail(PF_ERR_INVALID_ARG, "unknown exchange mode");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(PF_ERR_NO_DEVICE, "no HIP device");
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
        if (devices[i] < 0 || devices[i] >= ndev) return fail(PF_ERR_NO_DEVICE, "device " + std::to_string(devices[i]) + " does not exist");
        for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
    }
    if (exchange == PF_MULTI_RCCL && !distinct && n > 1) return fail(PF_ERR_INVALID_ARG, "RCCL needs distinct devices (a repeated device can only exchange by copies)");
    pf_multi *g = new pf_multi;
    g->exchange = exchange != PF_MULTI_AUTO ? exchange : (distinct && n > 1) ? PF_MULTI_RCCL : PF_MULTI_PEER_COPY;
    for (int i = 0; i < n; ++i) {
        Member *w = new Member;
        w->rank = i; w->device = devices[i];
        g->mem.push_back(w);
        DeviceGuard guard(w->device);
        hipError_t e = guard.err;
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev_free, hipEventDisableTiming);
        if (e != hipSuccess) { pf_multi_destroy(g); return fail(PF_ERR_HIP, std::string("pf_multi_create: ") + hipGetErrorString(e)); }
    }
    if (g->exchange == PF_MULTI_PEER_COPY) {           // let members write into each other's memory (no-op on one device)
        for (int i = 0; i < n; ++i) {
            DeviceGuard guard(devices[i]);
            for (int j = 0; j < n; ++j) {
                if (devices[i] == devices[j]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
                    const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { pf_multi_destroy(g); return fail(PF_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e)); }
                    (void)hipGetLastError();
                }
            }
        }
    } else {
        Rccl &r = rccl();
        if (!r.load()) { pf_multi_destroy(g); return fail(PF_ERR_UNSUPPORTED, "librccl could not be loaded: " + r.why); }
        std::vector<ncclComm_t> comms(n, nullptr);
        const ncclResult_t rc = r.CommInitAll(comms.data(), n, devices);
        if (rc != 0) { pf_multi_destroy(g); return fail(PF_ERR_HIP, std::string("ncclCommInitAll: ") + r.GetErrorString(rc)); }
        for (int i = 0; i < n; ++i) g->mem[i]->comm = comms[i];
    }
    for (Member *w : g->mem) w->th = std::thread(worker_main, w);
    *out = g;
    return PF_OK;
}

pf_status pf_multi_info(const pf_multi *g, int *n, int *devices, int *exchange) {
    if (!g) return fail(PF_ERR_INVALID_ARG, "null group");
    if (n) *n = (int)g->mem.size();
    if (devices) for (size_t i = 0; i < g->mem.size(); ++i) devices[i] = g->mem[i]->device;
    if (exchange) *exchange = g->exchange;
    return PF_OK;
}

pf_status pf_multi_ring(pf_multi *g, uint32_t N, uint32_t L, const uint64_t *moduli) {
    if (!g || !moduli) return fail(PF_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(g->call);
    return g->run_all([&](Member &m) -> pf_status {
        if (m.ctx) { (void)pf_ctx_destroy(m.ctx); m.ctx = nullptr; }
        return pf_ctx_create(&m.ctx, m.device, N, L, moduli);
    });
}

pf_status pf_multi_flat(pf_multi *g, const float *xb_host, size_t nb, uint32_t d) {
    if (!g || (!xb_host && nb)) return fail(PF_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(g->call);
    return g->run_all([&](Member &m) -> pf_status {
        if (m.flat) { (void)pf_flat_destroy(m.flat); m.flat = nullptr; }
        return pf_flat_create(&m.flat, m.device, xb_host, nb, d);
    });
}

pf_status pf_multi_reserve(pf_multi *g, size_t nq_local_max, uint32_t k_max) {
    if (!g) return fail(PF_ERR_INVALID_ARG, "null group");
    std::lock_guard<std::mutex> lk(g->call);
    return g->run_all([&](Member &m) -> pf_status {
        if (!m.flat) return fail(PF_ERR_INVALID_ARG, "pf_multi_flat has not run");
        return pf_flat_reserve(m.flat, nq_local_max, k_max);
    });
}

pf_status pf_multi_member(pf_multi *g, int rank, int *device, pf_stream *stream, pf_ctx **ctx, pf_flat **flat) {
    if (!g || rank < 0 || rank >= (int)g->mem.size()) return fail(PF_ERR_INVALID_ARG, "no such member");
    const Member *m = g->mem[rank];
    if (device) *device = m->device;
    if (stream) *stream = m->stream;
    if (ctx) *ctx = m->ctx;
    if (flat) *flat = m->flat;
    return PF_OK;
}

pf_status pf_multi_flat_search(pf_multi *g, const float *const *xq_dev, size_t nq_local, uint32_t k, uint32_t *const *gathered_dev) {
    if (!g || !xq_dev || !gathered_dev) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (nq_local == 0) return PF_OK;
    const size_t words = nq_local * (size_t)k * 3;
    for (size_t i = 0; i < g->mem.size(); ++i)
        if (!xq_dev[i] || !gathered_dev[i]) return fail(PF_ERR_INVALID_ARG, "null member buffer");
    std::lock_guard<std::mutex> lk(g->call);
    const pf_status st = g->run_all([&](Member &m) -> pf_status {
        if (!m.flat) return fail(PF_ERR_INVALID_ARG, "pf_multi_flat has not run");
        return pf_flat_search_packed(m.flat, xq_dev[m.rank], nq_local, k, nullptr, nullptr, gathered_dev[m.rank] + (size_t)m.rank * words, m.stream);
    });
    if (st != PF_OK) return st;
    return exchange_blocks(g, gathered_dev, words);
}

pf_status pf_multi_ct_pt_mul(pf_multi *g, const uint64_t *const *ct_dev, const uint64_t *const *pt_dev, size_t pt_count,
                             uint64_t *const *out_dev, size_t B_local, int flags) {
    if (!g || !ct_dev || !pt_dev || !out_dev) return fail(PF_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> lk(g->call);
    return g->run_all([&](Member &m) -> pf_status {
        if (!m.ctx) return fail(PF_ERR_INVALID_ARG, "pf_multi_ring has not run");
        return pf_ct_pt_mul(m.ctx, ct_dev[m.rank], pt_dev[m.rank], pt_count, out_dev[m.rank], B_local, flags, m.stream);
    });
}

pf_status pf_multi_synchronize(pf_multi *g) {
    if (!g) return fail(PF_ERR_INVALID_ARG, "null group");
    std::lock_guard<std::mutex> lk(g->call);
    return g->run_all([&](Member &m) -> pf_status {
        PF_GUARD(m.device);
        PF_HIP(hipStreamSynchronize(m.stream));
        return PF_OK;
    });
}

pf_status pf_multi_flat_search_host(pf_multi *g, const float *xq_host, size_t nq, uint32_t k, float *D_host, int64_t *I_host) {
    if (!g) return fail(PF_ERR_INVALID_ARG, "null group");
    if (nq == 0) return PF_OK;
    if (!xq_host || !D_host || !I_host) return fail(PF_ERR_INVALID_ARG, "null argument");
    const size_t G = g->mem.size();
    if (!g->mem[0]->flat) return fail(PF_ERR_INVALID_ARG, "pf_multi_flat has not run");
    size_t nb = 0; uint32_t d = 0;
    (void)pf_flat_info(g->mem[0]->flat, &nb, &d);
    const size_t n_local = (nq + G - 1) / G;                               // every member's block has room for the largest shard
    const size_t words = n_local * (size_t)k * 3, xq_bytes = (n_local * d * 4 + 255) / 256 * 256, gat_bytes = G * words * 4;
    std::vector<const float *> xq_dev(G);
    std::vector<uint32_t *> gat_dev(G);
    {
        std::lock_guard<std::mutex> lk(g->call);
        const pf_status st = g->run_all([&](Member &m) -> pf_status {
            PF_GUARD(m.device);
            if (m.stage_bytes < xq_bytes + gat_bytes) {
                if (m.stage) { PF_HIP(hipFree(m.stage)); m.stage = nullptr; m.stage_bytes = 0; }
                PF_HIP(hipMalloc(&m.stage, xq_bytes + gat_bytes));
                m.stage_bytes = xq_bytes + gat_bytes;
            }
            // contiguous shards; earlier members take the remainder
            const size_t base = nq / G, rem = nq % G;
            const size_t lo = m.rank * base + ((size_t)m.rank < rem ? m.rank : rem), cnt = base + ((size_t)m.rank < rem ? 1 : 0);
            float *xq = static_cast<float *>(m.stage);
            uint32_t *gat = reinterpret_cast<uint32_t *>(static_cast<char *>(m.stage) + xq_bytes);
            xq_dev[m.rank] = xq; gat_dev[m.rank] = gat;
            if (cnt < n_local) PF_HIP(hipMemsetAsync(gat + (size_t)m.rank * words + cnt * (size_t)k * 3, 0, (n_local - cnt) * (size_t)k * 12, m.stream));
            if (cnt) {
                PF_HIP(hipMemcpyAsync(xq, xq_host + lo * d, cnt * (size_t)d * 4, hipMemcpyHostToDevice, m.stream));
                return pf_flat_search_packed(m.flat, xq, cnt, k, nullptr, nullptr, gat + (size_t)m.rank * words, m.stream);
            }
            return PF_OK;
        });
        if (st != PF_OK) return st;
        const pf_status se = exchange_blocks(g, gat_dev.data(), words);
        if (se != PF_OK) return se;
    }
    // member 0's copy comes back (every member holds the same)
    std::vector<uint32_t> host(G * words);
    {
        Member &m0 = *g->mem[0];
        PF_GUARD(m0.device);
        PF_HIP(hipMemcpyAsync(host.data(), gat_dev[0], gat_bytes, hipMemcpyDeviceToHost, m0.stream));
        PF_HIP(hipStreamSynchronize(m0.stream));
    }
    const pf_status sy = pf_multi_synchronize(g);                          // the other members' pushes and waits are done too
    if (sy != PF_OK) return sy;
    const size_t base = nq / G, rem = nq % G;
    for (size_t r = 0; r < G; ++r) {
        const size_t lo = r * base + (r < rem ? r : rem), cnt = base + (r < rem ? 1 : 0);
        const uint32_t *blk = host.data() + r * words;
        for (size_t i = 0; i < cnt * k; ++i) {
            const uint64_t id = (uint64_t)blk[3 * i] | ((uint64_t)blk[3 * i + 1] << 32);
            I_host[lo * k + i] = (int64_t)id;
            std::memcpy(&D_host[lo * k + i], &blk[3 * i + 2], 4);
        }
    }
    return PF_OK;
}

}  // extern "C"
