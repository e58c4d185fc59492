// mirge_native.hip -- host runtime + C ABI of libmirge_native.so (include/mirge_native.h).
//
// Host side of the MI355X hot path: device context with a stream-ordered buffer pool (no
// hipMalloc/hipFree once warm), library packing + k-mer table construction, and the launch
// sequences for pack / collapse / cascade / count join.  Kernels: mirge_kernels.hpp.
// There is no CPU implementation of any of the compute in this file or anywhere in the product.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mirge_native.h"
#include "mirge_kernels.hpp"
#include "mirge_libbuild.hpp"

#ifndef MIRGE_BITMAP_MAXK
#define MIRGE_BITMAP_MAXK 10
#endif
static_assert(sizeof(mirge_policy) == sizeof(MirgePolicy), "policy layout");
static_assert(MIRGE_MAX_PASSES == MIRGE_MAX_PASSES_K, "pass cap");

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPOK(expr)                                                                          \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e));              \
    } while (0)
#define CHECK(expr)            \
    do {                       \
        int _c = (expr);       \
        if (_c != 0) return _c; \
    } while (0)

extern "C" const char* mirge_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------
// context: device, stream, pooled device memory, profiler
// ------------------------------------------------------------------------------------------
struct ProfRec {
    std::string name;
    int64_t launches = 0;
    double total_ms = 0.0;
    double units = 0.0;
};
struct PendingEvt {
    int rec;
    hipEvent_t a, b;
};
struct ProfUnits {  // "units" of a cascade pass = sum of the per-workgroup survivor counts of the stage before
    int rec, stage;
    uint32_t grid;
    const uint32_t* host;  // pinned copy of seg_n[stage][grid]
    double n_first;
};
#define MIRGE_PROF_PINNED_WORDS (1u << 18)

struct PassStep {
    int32_t p0 = 0, np = 1;       // passes p0 .. p0+np-1 run as one launch
    const mirge_lib* lib = nullptr;
    MergeInfo mi;
    const MirgePlanTable* dplan = nullptr;  // device copy of the tabulated probe plan
};

struct mirge_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second stream: the small read groups (long reads, reads with N) run beside the big one.
    // cur = the stream the launch helpers currently target.
    hipStream_t aux = nullptr, cur = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int n_cu = 256;
    // pool
    std::multimap<size_t, void*> free_blocks;
    std::unordered_map<void*, size_t> sizes;
    size_t pool_bytes = 0;
    // profiler
    bool profiling = false;
    std::string prof_only;  // non-empty: only launches whose name contains it are bracketed
    std::vector<ProfRec> recs;
    std::unordered_map<std::string, int> rec_of;
    std::vector<PendingEvt> pending;
    std::vector<hipEvent_t> evt_pool;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // pinned scratch for small D2H
    uint32_t* pinned = nullptr;
    uint32_t* prof_pinned = nullptr;
    unsigned long long* join_pinned = nullptr;
    size_t join_pinned_bytes = 0;
    struct PlanEntry { uint64_t uid; MirgePolicy pol; MirgePlanTable* dplan; };
    std::vector<PlanEntry> plans;
    struct FusedEntry { std::unique_ptr<FusedSteps> host; FusedSteps* dev; };
    std::vector<FusedEntry> fused;  // step lists of k_cascade_fused already on the device
    std::string casc_key;           // configuration of the previous mirge_cascade_run ...
    std::vector<PassStep> casc_steps;  // ... and what was prepared for it
    ResolveTable casc_rt;
    const FusedSteps* casc_dsteps = nullptr;
    size_t prof_used = 0;
    std::vector<ProfUnits> prof_pending;

    int alloc(void** out, size_t bytes) {
        bytes = (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
        auto it = free_blocks.lower_bound(bytes);
        if (it != free_blocks.end() && it->first <= bytes * 2 + (1u << 20)) {
            *out = it->second;
            free_blocks.erase(it);
            return 0;
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {  // give cached blocks back to the driver and retry once
            for (auto& kv : free_blocks) { (void)hipFree(kv.second); pool_bytes -= kv.first; sizes.erase(kv.second); }
            free_blocks.clear();
            e = hipMalloc(&p, bytes);
            if (e != hipSuccess) return fail(-3, "hipMalloc(" + std::to_string(bytes) + "): " + hipGetErrorString(e));
        }
        sizes[p] = bytes;
        pool_bytes += bytes;
        *out = p;
        return 0;
    }
    void release(void* p) {
        if (!p) return;
        auto it = sizes.find(p);
        if (it == sizes.end()) return;
        free_blocks.emplace(it->second, p);  // stream-ordered reuse: one stream per ctx
    }
    // libraries merged for runs of passes that share one policy (see mirge_cascade_run)
    struct Merged { std::vector<uint64_t> uids; struct mirge_lib* lib; };
    std::vector<Merged> merged;
    // inside a fork/join region a buffer must not go back to the pool before the join: the other
    // stream could be handed it while this stream's kernels still use it
    std::vector<void*> deferred;
    void defer(void* p) { if (p) deferred.push_back(p); }
    void flush_deferred() { for (void* p : deferred) release(p); deferred.clear(); }
    int rec_index(const char* name) {
        auto it = rec_of.find(name);
        if (it != rec_of.end()) return it->second;
        recs.push_back(ProfRec{name});
        rec_of[name] = (int)recs.size() - 1;
        return (int)recs.size() - 1;
    }
    hipEvent_t get_evt() {
        if (!evt_pool.empty()) { hipEvent_t e = evt_pool.back(); evt_pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    void drain() {  // resolve pending event pairs (caller has synchronised the stream)
        for (auto& u : prof_pending) {
            double units = u.n_first;
            if (u.stage > 0) { units = 0; for (uint32_t b = 0; b < u.grid; b++) units += u.host[(size_t)u.grid * (u.stage - 1) + b]; }
            recs[u.rec].units += units;
        }
        prof_pending.clear();
        prof_used = 0;
        for (auto& p : pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) recs[p.rec].total_ms += ms;
            evt_pool.push_back(p.a);
            evt_pool.push_back(p.b);
        }
        pending.clear();
    }
};

template <typename T>
static int dalloc(mirge_ctx* c, T** out, size_t count) { return c->alloc((void**)out, count * sizeof(T)); }

// launch bracket: HIP events on the ctx stream around each kernel when profiling is on
struct LaunchScope {
    mirge_ctx* c; int rec = -1; hipEvent_t a = nullptr, b = nullptr;
    LaunchScope(mirge_ctx* ctx, const char* name, double units) : c(ctx) {
        if (!c->profiling) return;
        if (!c->prof_only.empty() && !std::strstr(name, c->prof_only.c_str())) return;
        rec = c->rec_index(name);
        c->recs[rec].launches++;
        c->recs[rec].units += units;
        a = c->get_evt(); b = c->get_evt();
        (void)hipEventRecord(a, c->cur);
    }
    ~LaunchScope() {
        if (rec < 0) return;
        (void)hipEventRecord(b, c->cur);
        c->pending.push_back(PendingEvt{rec, a, b});
    }
};

// fork: work queued on `aux` from now on starts after everything already queued on the main stream;
// join: the main stream continues only after `aux` has drained.  Buffers handed back to the pool
// between the two are reused only by work queued after the join, so stream-ordered reuse still holds.
static int stream_fork(mirge_ctx* c) {
    HIPOK(hipEventRecord(c->ev_fork, c->stream));
    HIPOK(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
    return 0;
}
static int stream_join(mirge_ctx* c) {
    c->cur = c->stream;
    hipError_t e = hipEventRecord(c->ev_join, c->aux);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
    c->flush_deferred();  // reused only by work queued on the main stream after the wait
    if (e != hipSuccess) return fail(-2, std::string("stream join: ") + hipGetErrorString(e));
    return 0;
}
static int largest_group(const struct mirge_reads* R);

static inline int grid_for(const mirge_ctx* c, size_t n, int per_block = MIRGE_BLOCK) {
    size_t blocks = (n + per_block - 1) / per_block;
    static const size_t per_cu = std::getenv("MIRGE_GRID_PER_CU") ? (size_t)std::atoi(std::getenv("MIRGE_GRID_PER_CU")) : 8;
    size_t cap = (size_t)c->n_cu * per_cu;
    return (int)std::max<size_t>(1, std::min(blocks, cap));
}

// MIRGE_HOST_TIMING=1: host microseconds spent in the stages of a call, to stderr (enqueue-bound phases)
static std::chrono::steady_clock::time_point g_last_exit = std::chrono::steady_clock::now();
struct HostClock {
    const char* what;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit HostClock(const char* w) : what(w), t0(std::chrono::steady_clock::now()) {
        static const bool e = std::getenv("MIRGE_HOST_TIMING") != nullptr;
        on = e;
        if (on) std::fprintf(stderr, "[host] %s entered %.1f us after the previous call returned\n", what,
                             std::chrono::duration<double, std::micro>(t0 - g_last_exit).count());
    }
    ~HostClock() { if (on) g_last_exit = std::chrono::steady_clock::now(); }
    void lap(const char* stage) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[host] %s/%s %.1f us\n", what, stage, std::chrono::duration<double, std::micro>(t - t0).count());
        t0 = t;
    }
};

extern "C" int mirge_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int mirge_ctx_create(int device, void* hip_stream, mirge_ctx** out) {
    if (!out) return fail(-1, "mirge_ctx_create: out is NULL");
    int n = mirge_device_count();
    if (n <= 0) return fail(-4, "no HIP device visible: the hot path has no CPU fallback");
    if (device < 0 || device >= n) return fail(-1, "device index out of range");
    HIPOK(hipSetDevice(device));
    auto c = std::make_unique<mirge_ctx>();
    c->device = device;
    hipDeviceProp_t prop;
    HIPOK(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; }
    else { HIPOK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    c->cur = c->stream;
    HIPOK(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    HIPOK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPOK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIPOK(hipEventCreate(&c->t0));
    HIPOK(hipEventCreate(&c->t1));
    HIPOK(hipHostMalloc((void**)&c->pinned, 4096, hipHostMallocDefault));
    HIPOK(hipHostMalloc((void**)&c->prof_pinned, MIRGE_PROF_PINNED_WORDS * 4, hipHostMallocDefault));
    *out = c.release();
    return 0;
}

extern "C" void mirge_lib_destroy(mirge_lib* L);
extern "C" void mirge_ctx_destroy(mirge_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& m : c->merged) mirge_lib_destroy(m.lib);
    c->merged.clear();
    c->drain();
    for (auto& kv : c->sizes) (void)hipFree(kv.first);
    for (auto e : c->evt_pool) (void)hipEventDestroy(e);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->prof_pinned) (void)hipHostFree(c->prof_pinned);
    if (c->join_pinned) (void)hipHostFree(c->join_pinned);
    for (auto& e : c->plans) (void)hipFree(e.dplan);
    for (auto& e : c->fused) (void)hipFree(e.dev);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int mirge_ctx_sync(mirge_ctx* c) {
    if (!c) return fail(-1, "ctx is NULL");
    HIPOK(hipSetDevice(c->device));
    HIPOK(hipStreamSynchronize(c->stream));
    c->drain();
    return 0;
}

extern "C" int mirge_ctx_timer_start(mirge_ctx* c) {
    if (!c) return fail(-1, "ctx is NULL");
    HIPOK(hipEventRecord(c->t0, c->stream));
    return 0;
}
extern "C" int mirge_ctx_timer_stop(mirge_ctx* c, double* ms_out) {
    if (!c || !ms_out) return fail(-1, "NULL argument");
    HIPOK(hipEventRecord(c->t1, c->stream));
    HIPOK(hipEventSynchronize(c->t1));
    float ms = 0.f;
    HIPOK(hipEventElapsedTime(&ms, c->t0, c->t1));
    *ms_out = ms;
    return 0;
}
extern "C" int mirge_ctx_profile_enable(mirge_ctx* c, int32_t on) {
    if (!c) return fail(-1, "ctx is NULL");
    c->profiling = on != 0;
    return 0;
}
extern "C" int mirge_ctx_profile_only(mirge_ctx* c, const char* substr) {
    if (!c) return fail(-1, "ctx is NULL");
    c->prof_only = substr ? substr : "";
    return 0;
}
extern "C" int mirge_ctx_profile_reset(mirge_ctx* c) {
    if (!c) return fail(-1, "ctx is NULL");
    CHECK(mirge_ctx_sync(c));
    c->recs.clear();
    c->rec_of.clear();
    return 0;
}
extern "C" int32_t mirge_ctx_profile_count(mirge_ctx* c) {
    if (!c) return 0;
    if (mirge_ctx_sync(c) != 0) return 0;
    return (int32_t)c->recs.size();
}
extern "C" int mirge_ctx_profile_get(mirge_ctx* c, int32_t i, char* name_out, int32_t name_cap,
                                     int64_t* launches, double* total_ms, double* units) {
    if (!c || i < 0 || i >= (int32_t)c->recs.size()) return fail(-1, "profile index out of range");
    const ProfRec& r = c->recs[i];
    if (name_out && name_cap > 0) { std::snprintf(name_out, (size_t)name_cap, "%s", r.name.c_str()); }
    if (launches) *launches = r.launches;
    if (total_ms) *total_ms = r.total_ms;
    if (units) *units = r.units;
    return 0;
}

// ------------------------------------------------------------------------------------------
// library: 2-bit text + invalid bitmap + per-k tables
// ------------------------------------------------------------------------------------------
static std::atomic<uint64_t> g_lib_uid{1};

struct mirge_lib {
    mirge_ctx* ctx = nullptr;
    uint64_t uid = 0;  // never reused: identifies a member of a merged library after its pointer is gone
    MirgeHostLib h;  // host image (table construction)
    // device
    uint64_t* dT = nullptr;
    uint64_t* dinv = nullptr;
    uint32_t* dref_start = nullptr;
    MirgeKTable* dtables = nullptr;           // [MIRGE_SHAPE_SLOTS] on the device
    std::vector<MirgeKTable> htables;          // host mirror (device pointers), by mirge_shape_id
    size_t device_bytes = 0;
    std::mutex mu;
    int64_t n_refs = 0;
    int kmax = 8;

    MirgeLibView view() const {
        MirgeLibView v;
        v.T = dT; v.inv = dinv; v.ref_start = dref_start; v.tables = dtables;
        v.total = h.total; v.n_refs = (uint32_t)n_refs; v.kmax = kmax;
        return v;
    }
};

extern "C" int mirge_lib_create(mirge_ctx* c, const char* seq, const int64_t* off, int64_t n_refs, mirge_lib** out) {
    if (!c || !out || (!seq && n_refs > 0) || !off || n_refs < 0) return fail(-1, "mirge_lib_create: bad argument");
    HIPOK(hipSetDevice(c->device));
    auto L = std::make_unique<mirge_lib>();
    L->ctx = c;
    L->uid = g_lib_uid.fetch_add(1);
    std::string err;
    int rc = mirge_hostlib_build(L->h, seq, off, n_refs, err);
    if (rc) return fail(rc, "mirge_lib_create: " + err);
    L->n_refs = n_refs;
    L->kmax = L->h.kmax;
    L->htables.assign(MIRGE_SHAPE_SLOTS, MirgeKTable{nullptr, nullptr, nullptr});
    const size_t nT = L->h.T.size() * 8, nI = L->h.inv.size() * 8, nR = ((size_t)n_refs + 1) * 4;
    HIPOK(hipMalloc((void**)&L->dT, nT));
    HIPOK(hipMalloc((void**)&L->dinv, nI));
    HIPOK(hipMalloc((void**)&L->dref_start, nR));
    HIPOK(hipMalloc((void**)&L->dtables, sizeof(MirgeKTable) * MIRGE_SHAPE_SLOTS));
    HIPOK(hipMemcpy(L->dT, L->h.T.data(), nT, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(L->dinv, L->h.inv.data(), nI, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(L->dref_start, L->h.ref_start.data(), nR, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(L->dtables, L->htables.data(), sizeof(MirgeKTable) * MIRGE_SHAPE_SLOTS, hipMemcpyHostToDevice));
    L->device_bytes = nT + nI + nR + sizeof(MirgeKTable) * MIRGE_SHAPE_SLOTS;
    *out = L.release();
    return 0;
}

extern "C" void mirge_lib_destroy(mirge_lib* L) {
    if (!L) return;
    (void)hipSetDevice(L->ctx->device);
    (void)hipStreamSynchronize(L->ctx->stream);
    (void)hipFree(L->dT); (void)hipFree(L->dinv); (void)hipFree(L->dref_start); (void)hipFree(L->dtables);
    for (auto& t : L->htables) { (void)hipFree((void*)t.bucket); (void)hipFree((void*)t.pos); (void)hipFree((void*)t.bits); }
    delete L;
}
extern "C" int64_t mirge_lib_n_refs(const mirge_lib* L) { return L ? L->n_refs : -1; }
extern "C" int64_t mirge_lib_device_bytes(const mirge_lib* L) { return L ? (int64_t)L->device_bytes : -1; }

// one probe shape (k1 bases, gap, k2 bases)
struct ShapeJob {
    int k1 = 0, gap = 0, k2 = 0;
};

// build one table on the device (k_table_pass: count, scan, fill) and publish it in the library's registry
static int lib_build_shape(mirge_lib* L, const ShapeJob& j) {
    mirge_ctx* c = L->ctx;
    const int sid = mirge_shape_id(j.k1, j.gap, j.k2);
    const uint64_t nb = 1ull << (2 * (j.k1 + j.k2));
    uint32_t *A = nullptr, *dpos = nullptr, *dbits = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    uint32_t npos = 0;
    const int grid = c->n_cu * 8;
    hipError_t e = hipMalloc((void**)&A, (nb + 2) * 4);
    if (e == hipSuccess) e = hipMemsetAsync(A, 0, (nb + 2) * 4, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_table_pass<false>, dim3(grid), dim3(MIRGE_BLOCK), 0, c->stream, L->dT, L->dinv, L->h.total, j.k1,
                           j.gap, j.k2, A, (uint32_t*)nullptr);
        e = hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, A, A, (int)(nb + 2), c->stream);  // size query
    }
    if (e == hipSuccess) e = hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16));
    if (e == hipSuccess) e = hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, A, A, (int)(nb + 2), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&npos, A + nb + 1, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMalloc((void**)&dpos, std::max<size_t>(npos, 1) * 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_table_pass<true>, dim3(grid), dim3(MIRGE_BLOCK), 0, c->stream, L->dT, L->dinv, L->h.total, j.k1,
                           j.gap, j.k2, A, dpos);
        if (j.k1 + j.k2 <= MIRGE_BITMAP_MAXK) {  // non-empty-bucket bitmap
            const size_t words = (size_t)((nb + 31) / 32);
            e = hipMalloc((void**)&dbits, words * 4);
            if (e == hipSuccess)
                hipLaunchKernelGGL(k_table_bits, dim3(grid_for(c, words)), dim3(MIRGE_BLOCK), 0, c->stream, A, nb, dbits);
        }
    }
    // table complete; no kernel may be reading the registry while it changes
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    (void)hipFree(tmp);
    if (e != hipSuccess) {
        (void)hipFree(A); (void)hipFree(dpos); (void)hipFree(dbits);
        return fail(-2, std::string("probe table construction: ") + hipGetErrorString(e));
    }
    L->device_bytes += (nb + 2) * 4 + (size_t)npos * 4 + (dbits ? (size_t)((nb + 31) / 32) * 4 : 0);
    L->htables[sid].bucket = A;
    L->htables[sid].pos = dpos;
    L->htables[sid].bits = dbits;
    HIPOK(hipMemcpy(L->dtables + sid, &L->htables[sid], sizeof(MirgeKTable), hipMemcpyHostToDevice));
    return 0;
}

// build the tables of the wanted probe shapes that do not exist yet
static int lib_prepare_shapes(mirge_lib* L, const std::vector<ShapeJob>& wanted) {
    std::lock_guard<std::mutex> lk(L->mu);
    bool first = true;
    for (const auto& w : wanted) {
        if (w.k1 < 1 || w.k2 < 0 || w.k1 + w.k2 > (w.k2 == 0 ? MIRGE_KMAX0 : MIRGE_KMAX) || w.gap < 0 || w.gap > 31 || (w.k2 == 0 && w.gap != 0))
            return fail(-1, "probe shape out of range");
        if (L->htables[mirge_shape_id(w.k1, w.gap, w.k2)].bucket) continue;
        if (first) { HIPOK(hipSetDevice(L->ctx->device)); first = false; }
        CHECK(lib_build_shape(L, w));
    }
    return 0;
}

extern "C" int mirge_lib_prepare(mirge_lib* L, int32_t k) {
    if (!L) return fail(-1, "lib is NULL");
    std::vector<ShapeJob> w(1);
    w[0].k1 = k;
    return lib_prepare_shapes(L, w);
}

// ------------------------------------------------------------------------------------------
// reads
// ------------------------------------------------------------------------------------------
// Six read groups: width class (<=31, <=64, <=128 nt) x (no ambiguous call | has an N).  Reads with
// an N are rare (~0.1 %); keeping them apart lets the big groups run without an nmask array and lets
// the <=31-nt group collapse on a 64-bit key (sequence bits + length sentinel).
#define MIRGE_NGROUPS 6
static const int kGroupW[MIRGE_NGROUPS] = {1, 2, 4, 1, 2, 4};
static inline int width_class(int64_t L) { return L <= 31 ? 0 : (L <= 64 ? 1 : 2); }

struct ReadGroup {
    int W = 1;
    uint32_t n = 0;
    uint64_t* seq = nullptr;
    uint8_t* len = nullptr;
    uint64_t* nmask = nullptr;
    uint32_t* orig = nullptr;    // handle-order index of each read (nullptr: base + j)
    uint32_t base = 0;
    uint32_t* counts = nullptr;  // [n][S]
    uint32_t* first = nullptr;   // [n] raw index of first appearance (collapse output)
};

struct mirge_reads {
    mirge_ctx* ctx = nullptr;
    int64_t n = 0;
    int64_t total_bases = 0;
    int32_t n_samples = 0;  // 0: no count matrix attached
    ReadGroup g[MIRGE_NGROUPS];
    int32_t len_hist[MIRGE_MAX_READ_LEN + 1];  // lengths present (host), for table preparation
    bool hist_valid = false;
};

static int largest_group(const mirge_reads* R) {
    int best = 0;
    for (int gi = 1; gi < MIRGE_NGROUPS; gi++) if (R->g[gi].n > R->g[best].n) best = gi;
    return best;
}

template <int W>
static GroupView<W> view_of(const ReadGroup& g) {
    GroupView<W> v; v.seq = g.seq; v.len = g.len; v.nmask = g.nmask; v.n = g.n; return v;
}

extern "C" void mirge_reads_destroy(mirge_reads* r) {
    if (!r) return;
    for (auto& g : r->g) {
        r->ctx->release(g.seq); r->ctx->release(g.len); r->ctx->release(g.nmask);
        r->ctx->release(g.orig); r->ctx->release(g.counts); r->ctx->release(g.first);
    }
    delete r;
}
// Several raw read sets (the samples of a run, each parsed from its own file) as one, in the order given: read j of
// part p gets handle index (reads of the parts before p) + j.  No count matrices; the parts stay valid.
extern "C" int mirge_reads_concat(mirge_ctx* c, const mirge_reads* const* parts, int32_t n_parts, mirge_reads** out) {
    if (!c || !parts || n_parts < 1 || !out) return fail(-1, "mirge_reads_concat: bad argument");
    HIPOK(hipSetDevice(c->device));
    int64_t total = 0;
    for (int p = 0; p < n_parts; p++) {
        if (!parts[p] || parts[p]->ctx != c) return fail(-1, "mirge_reads_concat: foreign or NULL read set");
        if (parts[p]->n_samples) return fail(-1, "mirge_reads_concat: collapsed read sets cannot be appended");
        total += parts[p]->n;
    }
    if (total >= 0xFFFFFFF0ll) return fail(-5, "more than 2^32 reads in one set is not supported");
    auto R = std::make_unique<mirge_reads>();
    R->ctx = c; R->n = total; R->hist_valid = true;
    std::memset(R->len_hist, 0, sizeof(R->len_hist));
    for (int p = 0; p < n_parts; p++) {
        R->total_bases += parts[p]->total_bases;
        R->hist_valid = R->hist_valid && parts[p]->hist_valid;
        for (int L = 0; L <= MIRGE_MAX_READ_LEN; L++) R->len_hist[L] += parts[p]->len_hist[L];
    }
    int rc = 0;
    for (int gi = 0; gi < MIRGE_NGROUPS && rc == 0; gi++) {
        ReadGroup& g = R->g[gi];
        g.W = kGroupW[gi];
        uint64_t n = 0;
        bool mask = false;
        for (int p = 0; p < n_parts; p++) { n += parts[p]->g[gi].n; mask = mask || parts[p]->g[gi].nmask; }
        g.n = (uint32_t)n;
        if (!g.n) continue;
        if ((rc = dalloc(c, &g.seq, (size_t)g.W * g.n))) break;
        if ((rc = dalloc(c, &g.len, (size_t)g.n))) break;
        if ((rc = dalloc(c, &g.orig, (size_t)g.n))) break;
        if (mask && (rc = dalloc(c, &g.nmask, (size_t)g.W * g.n))) break;
        uint32_t at = 0, before = 0;
        hipError_t e = hipSuccess;
        for (int p = 0; p < n_parts && e == hipSuccess; p++) {
            const ReadGroup& q = parts[p]->g[gi];
            if (q.n) {
                for (int w = 0; w < g.W && e == hipSuccess; w++) {  // word-major arrays: one copy per word plane
                    e = hipMemcpyAsync(g.seq + (size_t)w * g.n + at, q.seq + (size_t)w * q.n, (size_t)q.n * 8, hipMemcpyDeviceToDevice, c->stream);
                    if (e == hipSuccess && g.nmask) {
                        if (q.nmask) e = hipMemcpyAsync(g.nmask + (size_t)w * g.n + at, q.nmask + (size_t)w * q.n, (size_t)q.n * 8, hipMemcpyDeviceToDevice, c->stream);
                        else e = hipMemsetAsync(g.nmask + (size_t)w * g.n + at, 0, (size_t)q.n * 8, c->stream);
                    }
                }
                if (e == hipSuccess) e = hipMemcpyAsync(g.len + at, q.len, (size_t)q.n, hipMemcpyDeviceToDevice, c->stream);
                hipLaunchKernelGGL(k_index_shift, dim3(grid_for(c, q.n)), dim3(MIRGE_BLOCK), 0, c->stream, (const uint32_t*)q.orig, q.base,
                                   q.n, before, g.orig + at);
                at += q.n;
            }
            before += (uint32_t)parts[p]->n;
        }
        if (e != hipSuccess) rc = fail(-2, std::string("mirge_reads_concat: ") + hipGetErrorString(e));
    }
    if (rc == 0) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(-2, std::string("mirge_reads_concat: ") + hipGetErrorString(e));
    }
    if (rc) { mirge_reads_destroy(R.release()); return rc; }
    *out = R.release();
    return 0;
}

extern "C" int64_t mirge_reads_count(const mirge_reads* r) { return r ? r->n : -1; }
extern "C" int64_t mirge_reads_total_bases(const mirge_reads* r) { return r ? r->total_bases : -1; }
extern "C" int32_t mirge_reads_n_samples(const mirge_reads* r) { return r ? r->n_samples : -1; }

template <int W>
static void launch_pack(mirge_ctx* c, const uint8_t* dascii, const int64_t* dstart, const int64_t* dend, const uint32_t* didx,
                        ReadGroup& g, uint32_t* dflags) {
    LaunchScope ls(c, "k_pack", g.n);
    hipLaunchKernelGGL(k_pack<W>, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream,
                       dascii, dstart, dend, didx, g.n, g.seq, g.len, g.nmask, dflags);
}

extern "C" int mirge_reads_pack(mirge_ctx* c, const char* ascii, const int64_t* off, int64_t n, mirge_reads** out) {
    if (!c || !out || !off || n < 0 || (n > 0 && !ascii)) return fail(-1, "mirge_reads_pack: bad argument");
    if (n >= 0xFFFFFFF0ll) return fail(-5, "more than 2^32 reads in one set is not supported");
    HIPOK(hipSetDevice(c->device));
    auto R = std::make_unique<mirge_reads>();
    R->ctx = c; R->n = n;
    std::memset(R->len_hist, 0, sizeof(R->len_hist));
    std::vector<uint32_t> idx[MIRGE_NGROUPS];
    bool is_acgt[256] = {false};
    for (const char* q = "ACGTUacgtu"; *q; q++) is_acgt[(unsigned char)*q] = true;
    {
        // classify the reads (width class x has-an-ambiguous-call) on all host cores: this byte scan is the
        // largest host cost of the PCIe-inclusive path
        const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        const int T = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n / 65536));
        std::vector<std::vector<uint32_t>> part((size_t)T * MIRGE_NGROUPS);
        std::vector<std::vector<int32_t>> hist((size_t)T, std::vector<int32_t>(MIRGE_MAX_READ_LEN + 1, 0));
        std::vector<int64_t> bad((size_t)T, -1), badlen((size_t)T, 0);
        auto work = [&](int t) {
            const int64_t lo = n * t / T, hi = n * (t + 1) / T;
            for (int64_t i = lo; i < hi; i++) {
                const int64_t L = off[i + 1] - off[i];
                if (L < 0 || L > MIRGE_MAX_READ_LEN) { if (bad[t] < 0) { bad[t] = i; badlen[t] = L; } continue; }
                hist[t][L]++;
                bool amb = false;
                for (int64_t b = off[i]; b < off[i + 1]; b++) amb |= !is_acgt[(unsigned char)ascii[b]];
                part[(size_t)t * MIRGE_NGROUPS + width_class(L) + (amb ? 3 : 0)].push_back((uint32_t)i);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
        for (int t = 0; t < T; t++) {
            if (bad[t] >= 0) {
                if (badlen[t] < 0) return fail(-1, "mirge_reads_pack: offsets not monotone");
                return fail(-6, "read " + std::to_string(bad[t]) + " is " + std::to_string(badlen[t]) + " nt; the limit is " +
                                std::to_string(MIRGE_MAX_READ_LEN));
            }
            for (int L = 0; L <= MIRGE_MAX_READ_LEN; L++) R->len_hist[L] += hist[t][L];
        }
        for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {  // thread ranges are consecutive: order is preserved
            size_t tot = 0;
            for (int t = 0; t < T; t++) tot += part[(size_t)t * MIRGE_NGROUPS + gi].size();
            idx[gi].reserve(tot);
            for (int t = 0; t < T; t++) {
                auto& v = part[(size_t)t * MIRGE_NGROUPS + gi];
                idx[gi].insert(idx[gi].end(), v.begin(), v.end());
            }
        }
    }
    R->hist_valid = true;
    R->total_bases = n ? off[n] - off[0] : 0;
    const int64_t nbytes = R->total_bases;
    uint8_t* dascii = nullptr; int64_t* doff = nullptr; uint32_t* dflags = nullptr;
    CHECK(dalloc(c, &dascii, (size_t)std::max<int64_t>(nbytes, 1)));
    CHECK(dalloc(c, &doff, (size_t)n + 1));
    CHECK(dalloc(c, &dflags, 16));
    std::vector<int64_t> rel((size_t)n + 1);
    for (int64_t i = 0; i <= n; i++) rel[(size_t)i] = off[i] - off[0];
    if (nbytes) HIPOK(hipMemcpyAsync(dascii, ascii + off[0], (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemcpyAsync(doff, rel.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPOK(hipMemsetAsync(dflags, 0, 64, c->stream));
    uint32_t* didx[MIRGE_NGROUPS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        ReadGroup& g = R->g[gi];
        g.W = kGroupW[gi];
        g.n = (uint32_t)idx[gi].size();
        if (!g.n) continue;
        CHECK(dalloc(c, &g.seq, (size_t)g.W * g.n));
        CHECK(dalloc(c, &g.nmask, (size_t)g.W * g.n));
        CHECK(dalloc(c, &g.len, (size_t)g.n));
        CHECK(dalloc(c, &g.orig, (size_t)g.n));
        HIPOK(hipMemcpyAsync(g.orig, idx[gi].data(), (size_t)g.n * 4, hipMemcpyHostToDevice, c->stream));
        didx[gi] = g.orig;
        if (kGroupW[gi] == 1) launch_pack<1>(c, dascii, doff, doff + 1, didx[gi], g, dflags + 2 * gi);
        else if (kGroupW[gi] == 2) launch_pack<2>(c, dascii, doff, doff + 1, didx[gi], g, dflags + 2 * gi);
        else launch_pack<4>(c, dascii, doff, doff + 1, didx[gi], g, dflags + 2 * gi);
    }
    HIPOK(hipMemcpyAsync(c->pinned, dflags, 64, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));  // idx/rel host vectors are read by the async copies
    c->release(dascii); c->release(doff); c->release(dflags);
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        if (c->pinned[2 * gi + 1]) {
            mirge_reads_destroy(R.release());
            return fail(-7, "a read contains a character other than A/C/G/T/U/N");
        }
        if (!c->pinned[2 * gi] && R->g[gi].nmask) { c->release(R->g[gi].nmask); R->g[gi].nmask = nullptr; }
    }
    *out = R.release();
    return 0;
}

// Sequence text -> packed reads, parsed on the device (k_nl_count / k_nl_mark / k_seq_class / k_seq_place, then
// k_pack straight from the text).  format: 1 = FASTQ (4-line records), 2 = FASTA (one sequence line per record),
// 3 = one sequence per line, 0 = by the first byte ('@', '>', else 3).  Reads shorter than min_len are dropped
// (digest.py:348,368); *n_records = records seen before the filter (digest.py:326 `count`).
extern "C" int mirge_reads_parse(mirge_ctx* c, const char* text, int64_t nbytes, int32_t format, int32_t min_len,
                                 mirge_reads** out, int64_t* n_records) {
    if (!c || !out || nbytes < 0 || (nbytes > 0 && !text) || format < 0 || format > 3)
        return fail(-1, "mirge_reads_parse: bad argument");
    HIPOK(hipSetDevice(c->device));
    if (format == 0) format = nbytes == 0 ? 3 : (text[0] == '@' ? 1 : (text[0] == '>' ? 2 : 3));
    const int period = format == 1 ? 4 : (format == 2 ? 2 : 1), sphase = format == 3 ? 0 : 1;
    auto R = std::make_unique<mirge_reads>();
    R->ctx = c; R->n = 0;
    std::memset(R->len_hist, 0, sizeof(R->len_hist));
    R->hist_valid = true;
    R->total_bases = 0;
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) R->g[gi].W = kGroupW[gi];
    if (n_records) *n_records = 0;
    if (nbytes == 0) { *out = R.release(); return 0; }
    // the text, with a final newline if the file has none
    const bool add_nl = text[nbytes - 1] != '\n';
    const uint64_t n = (uint64_t)nbytes + (add_nl ? 1 : 0);
    const uint32_t ntile = (uint32_t)((n + MIRGE_PARSE_TILE - 1) / MIRGE_PARSE_TILE);
    uint8_t* dtext = nullptr;
    uint32_t *tile_cnt = nullptr, *tile_off = nullptr;
    CHECK(dalloc(c, &dtext, (size_t)n + 16));
    CHECK(dalloc(c, &tile_cnt, (size_t)ntile + 1));
    CHECK(dalloc(c, &tile_off, (size_t)ntile + 1));
    HIPOK(hipMemcpyAsync(dtext, text, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    if (add_nl) HIPOK(hipMemsetAsync(dtext + nbytes, '\n', 1, c->stream));
    HIPOK(hipMemsetAsync(tile_cnt + ntile, 0, 4, c->stream));
    hipLaunchKernelGGL(k_nl_count, dim3(ntile), dim3(MIRGE_BLOCK), 0, c->stream, dtext, n, tile_cnt);
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    HIPOK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, tile_cnt, tile_off, (int)(ntile + 1), c->stream));
    size_t tmp_cap = std::max<size_t>(tmp_bytes, 1 << 16);
    CHECK(dalloc(c, (uint8_t**)&tmp, tmp_cap));
    HIPOK(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, tile_cnt, tile_off, (int)(ntile + 1), c->stream));
    uint32_t n_lines = 0;
    HIPOK(hipMemcpyAsync(&n_lines, tile_off + ntile, 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    // sequence lines: li in [0, n_lines) with li % period == sphase
    const uint64_t n_seq64 = n_lines > (uint32_t)sphase ? ((uint64_t)n_lines - sphase + period - 1) / period : 0;
    int rc = n_seq64 >= 0xFFFFFFF0ull ? fail(-5, "more than 2^32 reads in one set is not supported") : 0;
    const uint32_t n_seq = rc ? 0u : (uint32_t)n_seq64;
    if (n_records) *n_records = n_seq;
    int64_t *dstart = nullptr, *dend = nullptr;
    uint8_t* dcls = nullptr;
    uint32_t *blk = nullptr, *blk_off = nullptr, *keep = nullptr, *keep_off = nullptr, *dmeta = nullptr, *src_all = nullptr,
             *orig_all = nullptr;
    const uint32_t nblk = std::max<uint32_t>(1, (n_seq + MIRGE_BLOCK - 1) / MIRGE_BLOCK);
    const size_t meta_words = 8 + MIRGE_MAX_READ_LEN + 1;  // [0..2] flags, [8..] length histogram
    do {
        if (!n_seq) break;
        if ((rc = dalloc(c, &dstart, (size_t)n_seq))) break;
        if ((rc = dalloc(c, &dend, (size_t)n_seq))) break;
        if ((rc = dalloc(c, &dcls, (size_t)n_seq))) break;
        if ((rc = dalloc(c, &blk, (size_t)6 * nblk + 1))) break;
        if ((rc = dalloc(c, &blk_off, (size_t)6 * nblk + 1))) break;
        if ((rc = dalloc(c, &keep, (size_t)nblk + 1))) break;
        if ((rc = dalloc(c, &keep_off, (size_t)nblk + 1))) break;
        if ((rc = dalloc(c, &dmeta, meta_words))) break;
        hipError_t e = hipMemsetAsync(dmeta, 0, meta_words * 4, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(blk + (size_t)6 * nblk, 0, 4, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(keep + nblk, 0, 4, c->stream);
        if (e != hipSuccess) { rc = fail(-2, hipGetErrorString(e)); break; }
        hipLaunchKernelGGL(k_nl_mark, dim3(ntile), dim3(MIRGE_BLOCK), 0, c->stream, dtext, n, tile_off, period, sphase, dstart, dend,
                           (uint64_t)n_seq);
        hipLaunchKernelGGL(k_seq_class, dim3(nblk), dim3(MIRGE_BLOCK), 0, c->stream, dtext, dstart, dend, n_seq, min_len, dcls, blk,
                           keep, nblk, dmeta + 8, dmeta);
        size_t need = 0;
        e = hipcub::DeviceScan::ExclusiveSum(nullptr, need, blk, blk_off, (int)(6 * nblk + 1), c->stream);
        if (e == hipSuccess && need > tmp_cap) { c->release(tmp); tmp = nullptr; tmp_cap = need; if ((rc = dalloc(c, (uint8_t**)&tmp, tmp_cap))) break; }
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, need, blk, blk_off, (int)(6 * nblk + 1), c->stream);
        size_t need2 = tmp_cap;
        if (e == hipSuccess) e = hipcub::DeviceScan::ExclusiveSum(tmp, need2, keep, keep_off, (int)(nblk + 1), c->stream);
        // group bounds = blk_off at the first block of every class, and the total
        uint32_t bounds[7];
        for (int q = 0; q < 6 && e == hipSuccess; q++)
            e = hipMemcpyAsync(&bounds[q], blk_off + (size_t)q * nblk, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&bounds[6], blk_off + (size_t)6 * nblk, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->pinned, dmeta, meta_words * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = fail(-2, std::string("mirge_reads_parse: ") + hipGetErrorString(e)); break; }
        if (c->pinned[1]) { rc = fail(-6, "a read is " + std::to_string(c->pinned[2]) + " nt; the limit is " + std::to_string(MIRGE_MAX_READ_LEN)); break; }
        if (c->pinned[0]) { rc = fail(-7, "a read contains a character other than A/C/G/T/U/N"); break; }
        const uint32_t kept = bounds[6];
        R->n = kept;
        for (int L = 0; L <= MIRGE_MAX_READ_LEN; L++) {
            R->len_hist[L] = (int32_t)c->pinned[8 + L];
            R->total_bases += (int64_t)L * c->pinned[8 + L];
        }
        if (!kept) break;
        if ((rc = dalloc(c, &src_all, (size_t)kept))) break;
        if ((rc = dalloc(c, &orig_all, (size_t)kept))) break;
        hipLaunchKernelGGL(k_seq_place, dim3(nblk), dim3(MIRGE_BLOCK), 0, c->stream, dcls, n_seq, blk_off, keep_off, nblk, src_all,
                           orig_all);
        uint32_t* dflags = dmeta;  // reused: k_pack's per-group (saw N, bad byte) pairs
        e = hipMemsetAsync(dflags, 0, 64, c->stream);
        for (int gi = 0; gi < MIRGE_NGROUPS && rc == 0 && e == hipSuccess; gi++) {
            ReadGroup& g = R->g[gi];
            g.n = bounds[gi + 1] - bounds[gi];
            if (!g.n) continue;
            if ((rc = dalloc(c, &g.seq, (size_t)g.W * g.n))) break;
            if ((rc = dalloc(c, &g.nmask, (size_t)g.W * g.n))) break;
            if ((rc = dalloc(c, &g.len, (size_t)g.n))) break;
            if ((rc = dalloc(c, &g.orig, (size_t)g.n))) break;
            e = hipMemcpyAsync(g.orig, orig_all + bounds[gi], (size_t)g.n * 4, hipMemcpyDeviceToDevice, c->stream);
            const uint32_t* src = src_all + bounds[gi];
            if (kGroupW[gi] == 1) launch_pack<1>(c, dtext, dstart, dend, src, g, dflags + 2 * gi);
            else if (kGroupW[gi] == 2) launch_pack<2>(c, dtext, dstart, dend, src, g, dflags + 2 * gi);
            else launch_pack<4>(c, dtext, dstart, dend, src, g, dflags + 2 * gi);
        }
        if (rc == 0 && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (rc == 0 && e != hipSuccess) rc = fail(-2, std::string("mirge_reads_parse: ") + hipGetErrorString(e));
        if (rc == 0)
            for (int gi = 0; gi < 3; gi++)  // the groups without an ambiguous call carry no mask
                if (R->g[gi].nmask) { c->release(R->g[gi].nmask); R->g[gi].nmask = nullptr; }
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    c->release(dtext); c->release(tile_cnt); c->release(tile_off); c->release(tmp); c->release(dstart); c->release(dend);
    c->release(dcls); c->release(blk); c->release(blk_off); c->release(keep); c->release(keep_off); c->release(dmeta);
    c->release(src_all); c->release(orig_all);
    if (rc) { mirge_reads_destroy(R.release()); return rc; }
    *out = R.release();
    return 0;
}


extern "C" int mirge_reads_unpack(mirge_ctx* c, const mirge_reads* R, char* ascii_out, int64_t* off_out) {
    if (!c || !R || !off_out || (R->total_bases > 0 && !ascii_out)) return fail(-1, "mirge_reads_unpack: bad argument");
    HIPOK(hipSetDevice(c->device));
    const int64_t n = R->n;
    int32_t* dlen = nullptr;
    CHECK(dalloc(c, &dlen, (size_t)std::max<int64_t>(n, 1)));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ReadGroup& g = R->g[gi];
        if (!g.n) continue;
        LaunchScope ls(c, "k_scatter_len", g.n);
        hipLaunchKernelGGL(k_scatter_len, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream,
                           g.len, g.n, g.base, g.orig, dlen);
    }
    std::vector<int32_t> hlen((size_t)std::max<int64_t>(n, 1));
    if (n) HIPOK(hipMemcpyAsync(hlen.data(), dlen, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    off_out[0] = 0;
    for (int64_t i = 0; i < n; i++) off_out[i + 1] = off_out[i] + hlen[(size_t)i];
    const int64_t total = off_out[n];
    int64_t* doff = nullptr; uint8_t* dout = nullptr;
    CHECK(dalloc(c, &doff, (size_t)n + 1));
    CHECK(dalloc(c, &dout, (size_t)std::max<int64_t>(total, 1)));
    HIPOK(hipMemcpyAsync(doff, off_out, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ReadGroup& g = R->g[gi];
        if (!g.n) continue;
        LaunchScope ls(c, "k_unpack", g.n);
        if (kGroupW[gi] == 1) hipLaunchKernelGGL(k_unpack<1>, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, view_of<1>(g), doff, g.base, g.orig, dout);
        else if (kGroupW[gi] == 2) hipLaunchKernelGGL(k_unpack<2>, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, view_of<2>(g), doff, g.base, g.orig, dout);
        else hipLaunchKernelGGL(k_unpack<4>, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, view_of<4>(g), doff, g.base, g.orig, dout);
    }
    if (total) HIPOK(hipMemcpyAsync(ascii_out, dout, (size_t)total, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    c->release(dlen); c->release(doff); c->release(dout);
    return 0;
}

extern "C" int mirge_reads_set_counts(mirge_ctx* c, mirge_reads* R, const uint32_t* counts, int32_t S) {
    if (!c || !R || !counts || S < 1) return fail(-1, "mirge_reads_set_counts: bad argument");
    HIPOK(hipSetDevice(c->device));
    // counts are in handle order; each group wants its rows contiguous -> gather on the host
    // through the group's orig list (small: U x S)
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        ReadGroup& g = R->g[gi];
        if (!g.n) continue;
        std::vector<uint32_t> horig(g.n);
        if (g.orig) { HIPOK(hipMemcpyAsync(horig.data(), g.orig, (size_t)g.n * 4, hipMemcpyDeviceToHost, c->stream)); HIPOK(hipStreamSynchronize(c->stream)); }
        else for (uint32_t j = 0; j < g.n; j++) horig[j] = g.base + j;
        std::vector<uint32_t> rows((size_t)g.n * S);
        for (uint32_t j = 0; j < g.n; j++)
            std::memcpy(&rows[(size_t)j * S], &counts[(size_t)horig[j] * S], (size_t)S * 4);
        c->release(g.counts); g.counts = nullptr;
        CHECK(dalloc(c, &g.counts, (size_t)g.n * S));
        HIPOK(hipMemcpyAsync(g.counts, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPOK(hipStreamSynchronize(c->stream));
    }
    R->n_samples = S;
    return 0;
}

// ------------------------------------------------------------------------------------------
// collapse
// ------------------------------------------------------------------------------------------
// Collapse runs in two phases so that the whole call synchronises with the host ONCE: phase A
// (insert, head flags + block sums, scan) for every read group, one copy of {U per group, length
// histogram of the uniques} to the host, then phase B (output allocation sized by U, scatter).
struct CollapseTmp {
    uint32_t *rep = nullptr, *firstj = nullptr, *cnt = nullptr, *slot_of = nullptr, *blocksum = nullptr;
    KeySlot* slots = nullptr;
    uint8_t* flag = nullptr;
    const uint32_t* cnt_base = nullptr;
    uint32_t cnt_stride = 1, nb = 0;
    // partitioned key path
    bool partitioned = false;
    uint32_t *hist = nullptr, *off = nullptr, *btotal = nullptr, *nrec = nullptr;
    uint4 *part = nullptr, *recs = nullptr;
    uint32_t G = 0, chunk = 0, bshift = 0, B = 0, cap = MIRGE_PART_CAP;
};
// dmeta: [0..5] U of each group, [6] partition overflow flag, [8 .. 8+128] length histogram
#define MIRGE_META_OVERFLOW 6
#define MIRGE_META_HIST 8
#define MIRGE_META_WORDS (MIRGE_META_HIST + MIRGE_MAX_READ_LEN + 1)

static const char* group_tag(int gi) {
    static const char* t[MIRGE_NGROUPS] = {".w1", ".w2", ".w4", ".w1n", ".w2n", ".w4n"};
    return t[gi];
}

// partitioned key path after k_part_agg: bucket offsets, scatter, per-bucket de-duplication
static int collapse_part_rest(mirge_ctx* c, int gi, const ReadGroup& in, ReadGroup& out, CollapseTmp& t, uint32_t* dmeta) {
    const uint32_t G = t.G, B = t.B;
    {
        LaunchScope ls(c, "k_part_prefix.w1", (double)G * B);
        hipLaunchKernelGGL(k_part_prefix, dim3((B + 63) / 64), dim3(64), 0, c->cur, t.hist, G, B, t.off, t.btotal);
    }
    {
        LaunchScope ls(c, "k_scan_blocksums", B);
        hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(MIRGE_BLOCK), 0, c->cur, t.btotal, B, t.btotal + B);
    }
    {
        LaunchScope ls(c, "k_part_scatter.w1", in.n);
        hipLaunchKernelGGL(k_part_scatter, dim3(G), dim3(MIRGE_PART_THREADS), B * 4, c->cur, t.recs, t.nrec, t.chunk, t.bshift, B, t.off,
                           t.btotal, t.part);
    }
    {
        LaunchScope ls(c, "k_part_dedup.w1", in.n);
        if (t.cap == 2048)
            hipLaunchKernelGGL(k_part_dedup<2048>, dim3(B), dim3(MIRGE_DEDUP_THREADS), 2048 * 16 + 1024, c->cur, t.part, t.btotal,
                               out.seq, out.len, out.counts, out.first, dmeta + gi, dmeta + MIRGE_META_HIST, dmeta + MIRGE_META_OVERFLOW);
        else
            hipLaunchKernelGGL(k_part_dedup<MIRGE_PART_CAP>, dim3(B), dim3(MIRGE_DEDUP_THREADS), MIRGE_PART_CAP * 16 + 1024, c->cur, t.part,
                               t.btotal, out.seq, out.len, out.counts, out.first, dmeta + gi, dmeta + MIRGE_META_HIST,
                               dmeta + MIRGE_META_OVERFLOW);
    }
    return 0;
}

template <int W>
static int collapse_phase_a(mirge_ctx* c, int gi, const ReadGroup& in, ReadGroup& out, CollapseTmp& t,
                            const int32_t* dsample, int32_t S, uint32_t* dmeta, bool force_atomic, int stage = 0) {
    // stage 0 = everything; 1 = only the first kernel of the partitioned path; 2 = what stage 1 left
    if (!in.n) return 0;
    if (stage == 2 && t.partitioned) return collapse_part_rest(c, gi, in, out, t, dmeta);
    // key path: <=31 nt, no ambiguous call, one sample -> the slot holds the 64-bit key itself
    const bool key_path = (W == 1) && !in.nmask && S == 1;
    uint32_t tsize = 1024;
    while (tsize < (key_path ? in.n + in.n / 2 : 2ull * in.n)) tsize <<= 1;
    const uint32_t per_block = MIRGE_BLOCK * MIRGE_SCAN_ITEMS;
    t.nb = (in.n + per_block - 1) / per_block;
    GroupView<W> v = view_of<W>(in);
    char name[48];
    const uint32_t* first_base;
    uint32_t first_stride;
    if (key_path && !force_atomic && in.n >= 65536) {
        // partition by hash -> de-duplicate each bucket in LDS: no global atomics (see mirge_kernels.hpp)
        t.partitioned = true;
        uint32_t B = 64;
        // test hook: MIRGE_TEST_SMALL_PART=1 keeps 64 buckets so that big inputs overflow the LDS tables and
        // exercise the fallback to the global-atomic path (tests/test_gpu_parity.py)
        static const bool small_part = std::getenv("MIRGE_TEST_SMALL_PART") != nullptr;
        while (!small_part && B < 32768 && (uint64_t)B * 2048 < in.n) B <<= 1;  // ~1-2 k reads per bucket (up to 64 M reads)
        // buckets of <= 1024 records get a 2048-slot LDS table in k_part_dedup (4 workgroups per CU instead of 2).
        // Forcing that by doubling B was measured slower overall: k_part_agg/k_part_scatter pay for the larger B
        t.cap = (!small_part && (uint64_t)B * 1024 >= in.n) ? 2048u : (uint32_t)MIRGE_PART_CAP;
        const uint32_t CS = B > 16384 ? 1024 : 2048;  // chunk-level LDS cache slots (16 B each)
        const int agg_lds = (int)(CS * 16 + (B + 1) * 4 + 64);
        // dynamic-LDS ceilings, raised once per process and device to the largest configuration (B = 32768)
        static std::mutex attr_mu;
        static std::vector<int> attr_done;
        {
            std::lock_guard<std::mutex> lk(attr_mu);
            if (std::find(attr_done.begin(), attr_done.end(), c->device) == attr_done.end()) {
                HIPOK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_agg), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          1024 * 16 + (32768 + 1) * 4 + 64));  // CS = 1024 at B = 32768; 2048 * 16 + 16385 * 4 + 64 is smaller
                HIPOK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
                HIPOK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_dedup<MIRGE_PART_CAP>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, MIRGE_PART_CAP * 16 + 1024));
                attr_done.push_back(c->device);
            }
        }
        int lg = 0; while ((1u << lg) < B) lg++;
        const uint32_t bshift = 64 - lg;
        const uint32_t G = std::min<uint32_t>(256, (in.n + 2047) / 2048);
        uint32_t chunk = (in.n + G - 1) / G;
        chunk = (chunk + MIRGE_BLOCK - 1) / MIRGE_BLOCK * MIRGE_BLOCK;
        CHECK(dalloc(c, &t.hist, (size_t)G * B));
        CHECK(dalloc(c, &t.off, (size_t)G * B));
        CHECK(dalloc(c, &t.btotal, (size_t)B + 1));
        CHECK(dalloc(c, &t.part, (size_t)in.n));
        CHECK(dalloc(c, &t.recs, (size_t)G * chunk));
        CHECK(dalloc(c, &t.nrec, (size_t)G));
        // outputs at capacity n (U is not known yet): the bucket workgroups emit the unique reads themselves
        out.W = 1;
        CHECK(dalloc(c, &out.seq, (size_t)in.n));
        CHECK(dalloc(c, &out.len, (size_t)in.n));
        CHECK(dalloc(c, &out.counts, (size_t)in.n));
        CHECK(dalloc(c, &out.first, (size_t)in.n));
        GroupView<1> v1 = view_of<1>(in);
        {
            LaunchScope ls(c, "k_part_agg.w1", in.n);
            hipLaunchKernelGGL(k_part_agg, dim3(G), dim3(MIRGE_PART_THREADS), agg_lds, c->cur, v1, in.orig, in.base, chunk, bshift, B, CS, t.recs, t.nrec, t.hist);
        }
        t.G = G; t.chunk = chunk; t.bshift = bshift; t.B = B;
        if (stage == 1) return 0;
        return collapse_part_rest(c, gi, in, out, t, dmeta);
    }
    if (stage == 1) return 0;
    CHECK(dalloc(c, &t.slot_of, in.n));
    CHECK(dalloc(c, &t.flag, (size_t)t.nb * per_block));
    CHECK(dalloc(c, &t.blocksum, t.nb));
    if (key_path) {
        CHECK(dalloc(c, &t.slots, tsize));
        HIPOK(hipMemsetAsync(t.slots, 0, (size_t)tsize * sizeof(KeySlot), c->cur));
        LaunchScope ls(c, "k_collapse_insert_key.w1", in.n);
        hipLaunchKernelGGL(k_collapse_insert_key, dim3(grid_for(c, in.n)), dim3(MIRGE_BLOCK), 0, c->cur,
                           view_of<1>(in), t.slots, t.slot_of, tsize - 1);
        first_base = reinterpret_cast<const uint32_t*>(t.slots) + 2; first_stride = 4;
        t.cnt_base = reinterpret_cast<const uint32_t*>(t.slots) + 3; t.cnt_stride = 4;
    } else {
        CHECK(dalloc(c, &t.rep, tsize));
        CHECK(dalloc(c, &t.firstj, tsize));
        CHECK(dalloc(c, &t.cnt, (size_t)tsize * S));
        HIPOK(hipMemsetAsync(t.rep, 0xFF, (size_t)tsize * 4, c->cur));
        HIPOK(hipMemsetAsync(t.firstj, 0xFF, (size_t)tsize * 4, c->cur));
        HIPOK(hipMemsetAsync(t.cnt, 0, (size_t)tsize * S * 4, c->cur));
        std::snprintf(name, sizeof(name), "k_collapse_insert%s", group_tag(gi));
        LaunchScope ls(c, name, in.n);
        // at most 2 workgroups per CU: each sees enough of the group for its LDS cell cache to merge hot reads
        const int ins_grid = std::min(grid_for(c, in.n), c->n_cu * 2);
        hipLaunchKernelGGL(k_collapse_insert<W>, dim3(ins_grid), dim3(MIRGE_BLOCK), 0, c->cur,
                           v, t.rep, t.firstj, t.cnt, t.slot_of, tsize - 1, dsample, in.orig, in.base, S);
        first_base = t.firstj; first_stride = 1;
        t.cnt_base = t.cnt; t.cnt_stride = (uint32_t)S;
    }
    {
        std::snprintf(name, sizeof(name), "k_heads_blocksum%s", group_tag(gi));
        LaunchScope ls(c, name, in.n);
        hipLaunchKernelGGL(k_heads_blocksum, dim3(t.nb), dim3(MIRGE_BLOCK), 0, c->cur, t.slot_of, first_base, first_stride,
                           key_path ? 1u : 0u, in.n, in.len, t.flag, t.blocksum, dmeta + MIRGE_META_HIST);
    }
    {
        LaunchScope ls(c, "k_scan_blocksums", t.nb);
        hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(MIRGE_BLOCK), 0, c->cur, t.blocksum, t.nb, dmeta + gi);
    }
    return 0;
}

template <int W>
static int collapse_phase_b(mirge_ctx* c, int gi, const ReadGroup& in, ReadGroup& out, CollapseTmp& t, int32_t S,
                            uint32_t U, uint32_t out_base, const uint32_t* dmeta) {
    out.W = W; out.n = U; out.base = out_base;
    if (in.n && !t.partitioned) {
        CHECK(dalloc(c, &out.seq, (size_t)W * U));
        CHECK(dalloc(c, &out.len, (size_t)U));
        if (in.nmask) CHECK(dalloc(c, &out.nmask, (size_t)W * U));
        CHECK(dalloc(c, &out.counts, (size_t)U * S));
        CHECK(dalloc(c, &out.first, (size_t)U));
        char name[48];
        std::snprintf(name, sizeof(name), "k_collapse_scatter%s", group_tag(gi));
        LaunchScope ls(c, name, in.n);
        hipLaunchKernelGGL(k_collapse_scatter<W>, dim3(t.nb), dim3(MIRGE_BLOCK), 0, c->cur, view_of<W>(in), t.slot_of,
                           t.flag, t.cnt_base, t.cnt_stride, t.blocksum, dmeta + gi, in.orig, in.base, S, out.seq,
                           out.len, out.nmask, out.counts, out.first);
    }
    return 0;  // the temporaries go back to the pool in mirge_collapse, after the join
}

static void collapse_tmp_release(mirge_ctx* c, CollapseTmp& t) {
    c->defer(t.rep); c->defer(t.firstj); c->defer(t.cnt); c->defer(t.slots); c->defer(t.slot_of);
    c->defer(t.flag); c->defer(t.blocksum);
    c->defer(t.hist); c->defer(t.off); c->defer(t.btotal); c->defer(t.part); c->defer(t.recs); c->defer(t.nrec);
    t = CollapseTmp();
}

extern "C" int mirge_collapse(mirge_ctx* c, const mirge_reads* raw, const int32_t* sample_ids, int32_t S,
                              mirge_reads** uniq, int64_t* n_uniq) {
    HostClock hc("collapse");
    if (!c || !raw || !uniq || S < 1 || (S > 1 && !sample_ids)) return fail(-1, "mirge_collapse: bad argument");
    HIPOK(hipSetDevice(c->device));
    int32_t* dsample = nullptr;
    if (sample_ids && raw->n) {
        for (int64_t i = 0; i < raw->n; i++)
            if (sample_ids[i] < 0 || sample_ids[i] >= S) return fail(-1, "sample id out of range");
        CHECK(dalloc(c, &dsample, (size_t)raw->n));
        HIPOK(hipMemcpyAsync(dsample, sample_ids, (size_t)raw->n * 4, hipMemcpyHostToDevice, c->stream));
    }
    auto R = std::make_unique<mirge_reads>();
    R->ctx = c; R->n_samples = S;
    uint32_t* dmeta = nullptr;
    CHECK(dalloc(c, &dmeta, MIRGE_META_WORDS));
    CollapseTmp tmp[MIRGE_NGROUPS];
    int rc = 0;
    const int big = largest_group(raw);
    // attempt 0 may use the partitioned LDS path; if one of its buckets overflows its LDS table
    // (pathological hash skew) everything is redone with the global-atomic tables
    for (int attempt = 0; attempt < 2 && rc == 0; attempt++) {
        hipError_t e0 = hipMemsetAsync(dmeta, 0, MIRGE_META_WORDS * 4, c->stream);
        if (e0 != hipSuccess) { rc = fail(-2, std::string("mirge_collapse: ") + hipGetErrorString(e0)); break; }
        rc = stream_fork(c);
        // Order of enqueue (profiles/r01_timeline.txt): the bulk group's first kernel (k_part_agg: one workgroup
        // per CU, ~0.2 ms) goes first, the small groups' ~15 short launches are enqueued while it runs and share
        // the CUs with it, then the bulk group's wide kernels.  Small groups entirely first left the GPU idle for
        // the ~0.2 ms their enqueue takes; entirely last, each of their kernels waits behind 2048-8192-workgroup
        // launches for CUs to drain and the join at the end waits for them (5.3 vs 3.8 ms).
        for (int k = -1; k <= MIRGE_NGROUPS && rc == 0; k++) {
            const int gi = (k < 0 || k == MIRGE_NGROUPS) ? big : k;
            if (k >= 0 && k < MIRGE_NGROUPS && gi == big) continue;
            const int stage = k < 0 ? 1 : (k == MIRGE_NGROUPS ? 2 : 0);
            c->cur = gi == big ? c->stream : c->aux;
            if (kGroupW[gi] == 1) rc = collapse_phase_a<1>(c, gi, raw->g[gi], R->g[gi], tmp[gi], dsample, S, dmeta, attempt == 1, stage);
            else if (kGroupW[gi] == 2) rc = collapse_phase_a<2>(c, gi, raw->g[gi], R->g[gi], tmp[gi], dsample, S, dmeta, attempt == 1, stage);
            else rc = collapse_phase_a<4>(c, gi, raw->g[gi], R->g[gi], tmp[gi], dsample, S, dmeta, attempt == 1, stage);
        }
        { int jr = stream_join(c); if (rc == 0) rc = jr; }
        hc.lap("enqueue A");
        if (rc == 0) {  // the one host synchronisation of the call: U sizes the outputs
            hipError_t e = hipMemcpyAsync(c->pinned, dmeta, MIRGE_META_WORDS * 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) rc = fail(-2, std::string("mirge_collapse: ") + hipGetErrorString(e));
        }
        if (rc == 0 && c->pinned[MIRGE_META_OVERFLOW] && attempt == 0) {
            for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
                collapse_tmp_release(c, tmp[gi]);
                ReadGroup& og = R->g[gi];  // outputs the partitioned attempt had allocated at capacity n
                c->release(og.seq); c->release(og.len); c->release(og.counts); c->release(og.first);
                og = ReadGroup();
            }
            c->flush_deferred();
            continue;
        }
        break;
    }
    hc.lap("sync");
    uint32_t base = 0;
    if (rc == 0) {
        uint32_t U[MIRGE_NGROUPS];
        for (int gi = 0; gi < MIRGE_NGROUPS; gi++) U[gi] = c->pinned[gi];
        R->total_bases = 0;
        for (int L = 0; L <= MIRGE_MAX_READ_LEN; L++) {
            R->len_hist[L] = (int32_t)c->pinned[MIRGE_META_HIST + L];
            R->total_bases += (int64_t)L * c->pinned[MIRGE_META_HIST + L];
        }
        R->hist_valid = true;
        rc = stream_fork(c);
        for (int gi = 0; gi < MIRGE_NGROUPS && rc == 0; gi++) {
            c->cur = gi == big ? c->stream : c->aux;
            if (kGroupW[gi] == 1) rc = collapse_phase_b<1>(c, gi, raw->g[gi], R->g[gi], tmp[gi], S, U[gi], base, dmeta);
            else if (kGroupW[gi] == 2) rc = collapse_phase_b<2>(c, gi, raw->g[gi], R->g[gi], tmp[gi], S, U[gi], base, dmeta);
            else rc = collapse_phase_b<4>(c, gi, raw->g[gi], R->g[gi], tmp[gi], S, U[gi], base, dmeta);
            base += R->g[gi].n;
        }
        { int jr = stream_join(c); if (rc == 0) rc = jr; }
    }
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) collapse_tmp_release(c, tmp[gi]);
    c->flush_deferred();
    c->release(dsample); c->release(dmeta);
    if (rc) { mirge_reads_destroy(R.release()); return rc; }
    R->n = base;
    *uniq = R.release();
    if (n_uniq) *n_uniq = base;
    hc.lap("phase B + release");
    return 0;
}

extern "C" int mirge_collapse_fetch(mirge_ctx* c, const mirge_reads* U, uint32_t* counts_out, int64_t* first_out) {
    if (!c || !U || !counts_out) return fail(-1, "mirge_collapse_fetch: bad argument");
    if (U->n_samples < 1) return fail(-1, "read set has no count matrix");
    HIPOK(hipSetDevice(c->device));
    const int32_t S = U->n_samples;
    std::vector<uint32_t> tmp;
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ReadGroup& g = U->g[gi];
        if (!g.n) continue;
        if (g.orig) return fail(-1, "mirge_collapse_fetch: handle is not a collapse result");
        HIPOK(hipMemcpyAsync(counts_out + (size_t)g.base * S, g.counts, (size_t)g.n * S * 4, hipMemcpyDeviceToHost, c->stream));
        if (first_out && g.first) {
            tmp.resize(g.n);
            HIPOK(hipMemcpyAsync(tmp.data(), g.first, (size_t)g.n * 4, hipMemcpyDeviceToHost, c->stream));
            HIPOK(hipStreamSynchronize(c->stream));
            for (uint32_t j = 0; j < g.n; j++) first_out[g.base + j] = tmp[j];
        }
    }
    HIPOK(hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------
// cascade
// ------------------------------------------------------------------------------------------
struct ResGroup {
    uint32_t n = 0;
    int8_t* pass = nullptr;
    uint32_t* pos = nullptr;
    int8_t* mm = nullptr;
    int32_t* ref = nullptr;
    int32_t* off = nullptr;
};
struct mirge_result {
    mirge_ctx* ctx = nullptr;
    int64_t n = 0;
    int32_t n_pass = 0;
    ResGroup g[MIRGE_NGROUPS];
    const mirge_reads* reads = nullptr;  // borrowed: orig/base mapping (must outlive the fetch)
};

extern "C" void mirge_result_destroy(mirge_result* r) {
    if (!r) return;
    for (auto& g : r->g) {
        r->ctx->release(g.pass); r->ctx->release(g.pos); r->ctx->release(g.mm);
        r->ctx->release(g.ref); r->ctx->release(g.off);
    }
    delete r;
}

// One library = the members' references in order (same bases, same separators), so that a position in
// the merged text minus the member's start is the position in the member's own text.
static int merged_library(mirge_ctx* c, const mirge_lib* const* members, int n, mirge_lib** out) {
    std::vector<uint64_t> uids;
    for (int i = 0; i < n; i++) uids.push_back(members[i]->uid);
    for (auto& m : c->merged)
        if (m.uids == uids) { *out = m.lib; return 0; }
    std::string seq;
    std::vector<int64_t> off{0};
    for (int i = 0; i < n; i++) {
        const MirgeHostLib& h = members[i]->h;
        for (int64_t r = 0; r < h.n_refs; r++) {
            for (uint64_t g = h.ref_start[(size_t)r]; g + 1 < h.ref_start[(size_t)r + 1]; g++) {
                const bool bad = (h.inv[g >> 6] >> (g & 63)) & 1ull;
                seq.push_back(bad ? 'N' : "ACGT"[(h.T[g >> 5] >> (2 * (g & 31))) & 3ull]);
            }
            off.push_back((int64_t)seq.size());
        }
    }
    mirge_lib* L = nullptr;
    CHECK(mirge_lib_create(c, seq.data(), off.data(), (int64_t)off.size() - 1, &L));
    c->merged.push_back(mirge_ctx::Merged{uids, L});
    *out = L;
    return 0;
}

// build every probe table pass `p` can ask for, given the read lengths present
static int prepare_tables(mirge_lib* lib, const mirge_policy& pol, const int32_t* hist) {
    MirgePolicy p;
    std::memcpy(&p, &pol, sizeof(p));
    std::vector<ShapeJob> wanted;
    std::vector<bool> seen(MIRGE_SHAPE_SLOTS, false);
    for (int L = 1; L <= MIRGE_MAX_READ_LEN; L++) {
        if (!hist[L]) continue;
        if (p.len_lt > 0 && !(L < p.len_lt)) continue;
        if (p.len_gt > 0 && !(L > p.len_gt)) continue;
        int lo = L, hi = L;
        if (p.ttail) { lo = 1; hi = L - 3; }  // any head length once the T run is gone
        for (int l0 = lo; l0 <= hi; l0++) {
            const int l = l0 - p.trim5 - p.trim3;
            if (l < 1 || l <= p.mm) continue;
            const int np = mirge_probe_count(p, l, lib->kmax, lib->h.total);
            for (int q = 0; q < np; q++) {
                MirgeProbe pr;
                mirge_probe_at(p, l, lib->kmax, lib->h.total, q, pr);
                if (pr.k1 <= 0) continue;
                const int sid = mirge_shape_id(pr.k1, pr.gap, pr.k2);
                if (seen[sid]) continue;
                seen[sid] = true;
                wanted.emplace_back();
                wanted.back().k1 = pr.k1; wanted.back().gap = pr.gap; wanted.back().k2 = pr.k2;
            }
        }
    }
    return lib_prepare_shapes(lib, wanted);
}

template <int W>
static int cascade_group(mirge_ctx* c, const ReadGroup& rg, ResGroup& out, const std::vector<PassStep>& steps,
                         const mirge_policy* pol, const ResolveTable& rt, const char* gtag) {
    out.n = rg.n;
    if (!rg.n) return 0;
    const uint32_t n = rg.n;
    CHECK(dalloc(c, &out.pass, n));
    CHECK(dalloc(c, &out.pos, n));
    CHECK(dalloc(c, &out.mm, n));
    CHECK(dalloc(c, &out.ref, n));
    CHECK(dalloc(c, &out.off, n));
    // every workgroup keeps its own survivor segment through all passes: no global cursor.  All workgroups
    // must be resident at once: a grid of 8 per CU ran the workgroups that did not fit as a second round on an
    // almost empty machine (average occupancy 47 %, SQ_WAVE_CYCLES; -20 % kernel time with the right grid).
    // Measured on MI355X (tools/occ_sweep.sh, profiles/README.md): kernel time falls up to 6 workgroups per CU
    // and jumps back by 30 % at 7 and beyond -- for the 69-VGPR build and for 57/63-VGPR builds alike, so the
    // cliff is not the register file although hipOccupancyMaxActiveBlocksPerMultiprocessor reports 7.  The grid
    // is therefore min(occupancy query, register bound, 6) per CU.  MIRGE_WG_PER_CU overrides (sweeps).
    static int wg_per_cu[5] = {0, 0, 0, 0, 0};
    if (!wg_per_cu[W]) {
        int nb = 0;
        hipFuncAttributes fa;
        HIPOK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_pass<W, 15>), MIRGE_BLOCK, 0));
        HIPOK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_pass<W, 15>)));
        const int by_regs = 512 / std::max(16, (fa.numRegs + 15) / 16 * 16);  // waves per SIMD = 4-wave workgroups per CU
        wg_per_cu[W] = std::max(1, std::min({nb, by_regs, 6}));
        if (std::getenv("MIRGE_WG_PER_CU")) wg_per_cu[W] = std::max(1, std::atoi(std::getenv("MIRGE_WG_PER_CU")));
        if (std::getenv("MIRGE_HOST_TIMING"))
            std::fprintf(stderr, "[host] k_pass<%d>: %d VGPRs, occupancy query %d -> %d workgroups per CU\n", W, fa.numRegs, nb, wg_per_cu[W]);
    }
    const uint32_t grid = (uint32_t)std::min<size_t>((size_t)grid_for(c, n), (size_t)c->n_cu * wg_per_cu[W]);
    uint32_t cap = (n + grid - 1) / grid;
    cap = (cap + MIRGE_BLOCK - 1) / MIRGE_BLOCK * MIRGE_BLOCK;
    uint32_t *actA = nullptr, *actB = nullptr, *seg_n = nullptr;
    CHECK(dalloc(c, &actA, (size_t)grid * cap));
    CHECK(dalloc(c, &actB, (size_t)grid * cap));
    CHECK(dalloc(c, &seg_n, (size_t)grid * (MIRGE_MAX_PASSES + 1)));
    HIPOK(hipMemsetAsync(out.pass, 0xFF, n, c->cur));
    HIPOK(hipMemsetAsync(out.mm, 0xFF, n, c->cur));
    GroupView<W> v = view_of<W>(rg);
    const uint32_t* act_in = nullptr;
    uint32_t* act_out = actA;
    int stage = 0;
    char name[32];
    std::vector<std::pair<int, int>> stage_of_pass;  // (profile record, stage) for unit accounting
    for (const PassStep& st : steps) {
        const int32_t p = st.p0;
        MirgePolicy mp;
        std::memcpy(&mp, &pol[p], sizeof(mp));
        {
            if (st.np > 1) std::snprintf(name, sizeof(name), "k_pass[%d-%d]%s", (int)p, (int)(p + st.np - 1), gtag);
            else std::snprintf(name, sizeof(name), "k_pass[%d]%s", (int)p, gtag);
            LaunchScope ls(c, name, 0.0);
            if (ls.rec >= 0) stage_of_pass.emplace_back(ls.rec, stage);
            const uint32_t* sn_in = seg_n + (size_t)grid * (stage > 0 ? stage - 1 : 0);
            uint32_t* sn_out = seg_n + (size_t)grid * stage;
#define MIRGE_LAUNCH_PASS(SLOT)                                                                                       \
    hipLaunchKernelGGL((k_pass<W, SLOT>), dim3(grid), dim3(MIRGE_BLOCK), 0, c->cur, st.lib->view(), mp, st.mi, st.dplan, v, act_in, \
                       sn_in, act_out, sn_out, cap, p, out.pass, out.pos, out.mm)
            switch (p) {
                case 0: MIRGE_LAUNCH_PASS(0); break;
                case 1: MIRGE_LAUNCH_PASS(1); break;
                case 2: MIRGE_LAUNCH_PASS(2); break;
                case 3: MIRGE_LAUNCH_PASS(3); break;
                case 4: MIRGE_LAUNCH_PASS(4); break;
                case 5: MIRGE_LAUNCH_PASS(5); break;
                case 6: MIRGE_LAUNCH_PASS(6); break;
                case 7: MIRGE_LAUNCH_PASS(7); break;
                case 8: MIRGE_LAUNCH_PASS(8); break;
                case 9: MIRGE_LAUNCH_PASS(9); break;
                default: MIRGE_LAUNCH_PASS(15); break;
            }
#undef MIRGE_LAUNCH_PASS
        }
        act_in = act_out;
        act_out = (act_out == actA) ? actB : actA;
        stage++;
    }
    {
        std::snprintf(name, sizeof(name), "k_resolve%s", gtag);
        LaunchScope ls(c, name, n);
        hipLaunchKernelGGL(k_resolve, dim3(grid_for(c, n)), dim3(MIRGE_BLOCK), 0, c->cur, rt, out.pass, out.pos, n, out.ref, out.off);
    }
    if (c->profiling && stage > 0) {  // units of a pass = reads it was handed = survivors of the stage before
        // copied now (stream-ordered), summed after the one synchronisation at the end of the call
        const size_t words = (size_t)grid * stage;
        if (c->prof_used + words <= MIRGE_PROF_PINNED_WORDS) {
            uint32_t* dst = c->prof_pinned + c->prof_used;
            HIPOK(hipMemcpyAsync(dst, seg_n, words * 4, hipMemcpyDeviceToHost, c->cur));
            for (auto& sp : stage_of_pass) c->prof_pending.push_back(ProfUnits{sp.first, sp.second, grid, dst, (double)n});
            c->prof_used += words;
        }
    }
    c->defer(actA); c->defer(actB); c->defer(seg_n);
    return 0;
}

// a small group's whole cascade as one launch (k_cascade_fused)
template <int W>
static int cascade_group_fused(mirge_ctx* c, const ReadGroup& rg, ResGroup& out, const FusedSteps* dsteps,
                               const ResolveTable& rt, const char* gtag) {
    out.n = rg.n;
    if (!rg.n) return 0;
    const uint32_t n = rg.n;
    CHECK(dalloc(c, &out.pass, n));
    CHECK(dalloc(c, &out.pos, n));
    CHECK(dalloc(c, &out.mm, n));
    CHECK(dalloc(c, &out.ref, n));
    CHECK(dalloc(c, &out.off, n));
    char name[32];
    std::snprintf(name, sizeof(name), "k_cascade_fused%s", gtag);
    LaunchScope ls(c, name, n);
    const uint32_t grid = std::min<uint32_t>((n + MIRGE_BLOCK - 1) / MIRGE_BLOCK, (uint32_t)c->n_cu * 8);
    hipLaunchKernelGGL(k_cascade_fused<W>, dim3(grid), dim3(MIRGE_BLOCK), 0, c->cur, dsteps, rt, view_of<W>(rg), out.pass,
                       out.pos, out.mm, out.ref, out.off);
    return 0;
}

// steps (merged runs, probe tables, plan tables), resolve table and the fused kernel's device step list for
// one (libraries, policies, read-length set) configuration
static int cascade_prepare(mirge_ctx* c, const mirge_lib* const* libs, const mirge_policy* pol, int32_t n_pass,
                           const int32_t* hist, std::vector<PassStep>& steps, ResolveTable& rt, const FusedSteps** dsteps_out) {
    for (int p = 0; p < MIRGE_MAX_PASSES; p++) { rt.ref_start[p] = nullptr; rt.n_refs[p] = 0; }
    steps.clear();
    for (int32_t p = 0; p < n_pass; p++) {
        if (!libs[p]) continue;
        if (libs[p]->ctx->device != c->device) return fail(-1, "library lives on another device");
        if (pol[p].mm < 0 || pol[p].mm > 3 || pol[p].trim5 < 0 || pol[p].trim5 > 31 || pol[p].trim3 < 0)
            return fail(-1, "unsupported policy");
        rt.ref_start[p] = libs[p]->dref_start;
        rt.n_refs[p] = (uint32_t)libs[p]->n_refs;
    }
    for (int32_t p = 0; p < n_pass;) {
        if (!libs[p]) { p++; continue; }
        // a run of consecutive passes with one and the same policy over distinct libraries becomes ONE
        // launch over their concatenation (k_pass ranks candidates by member library first, so the
        // cascade's "first library with a hit wins" is unchanged): human set -> passes 4,5,6
        int np = 1;
        uint64_t total = libs[p]->h.total;
        static const bool merge_on = !(std::getenv("MIRGE_MERGE_PASSES") && std::getenv("MIRGE_MERGE_PASSES")[0] == '0');
        while (merge_on && np < 4 && p + np < n_pass && libs[p + np] && std::memcmp(&pol[p + np], &pol[p], sizeof(mirge_policy)) == 0 &&
               total + libs[p + np]->h.total < 0xFFFFFFF0ull) {
            bool distinct = true;
            for (int q = 0; q < np; q++) distinct &= libs[p + q] != libs[p + np];
            if (!distinct) break;
            total += libs[p + np]->h.total;
            np++;
        }
        PassStep st;
        st.p0 = p; st.np = np;
        st.mi.n = np;
        for (int i = 0; i < 4; i++) st.mi.bound[i] = 0;
        if (np == 1) st.lib = libs[p];
        else {
            mirge_lib* m = nullptr;
            CHECK(merged_library(c, libs + p, np, &m));
            st.lib = m;
            uint64_t b = 0;
            for (int i = 0; i < np; i++) { st.mi.bound[i] = (uint32_t)b; b += libs[p + i]->h.total; }
        }
        CHECK(prepare_tables(const_cast<mirge_lib*>(st.lib), pol[p], hist));
        steps.push_back(st);
        p += np;
    }
    // tabulated probe plans: built and uploaded once per (library, policy), then reused by every call
    for (auto& st : steps) {
        MirgePolicy mp;
        std::memcpy(&mp, &pol[st.p0], sizeof(mp));
        const MirgePlanTable* dp = nullptr;
        for (auto& e : c->plans)
            if (e.uid == st.lib->uid && std::memcmp(&e.pol, &mp, sizeof(mp)) == 0) { dp = e.dplan; break; }
        if (!dp) {
            auto h = std::make_unique<MirgePlanTable>();
            mirge_plan_table_fill(mp, st.lib->kmax, st.lib->h.total, *h);
            MirgePlanTable* d = nullptr;
            HIPOK(hipMalloc((void**)&d, sizeof(MirgePlanTable)));
            HIPOK(hipMemcpy(d, h.get(), sizeof(MirgePlanTable), hipMemcpyHostToDevice));
            c->plans.push_back(mirge_ctx::PlanEntry{st.lib->uid, mp, d});
            dp = d;
        }
        st.dplan = dp;
    }
    // the step list of the fused small-group kernel lives in device memory; uploaded when it changes
    auto fs = std::make_unique<FusedSteps>();
    std::memset(fs.get(), 0, sizeof(FusedSteps));
    fs->n = (int32_t)steps.size();
    for (size_t i = 0; i < steps.size(); i++) {
        FusedStep& f = fs->s[i];
        f.lib = steps[i].lib->view();
        std::memcpy(&f.pol, &pol[steps[i].p0], sizeof(MirgePolicy));
        f.mi = steps[i].mi;
        f.plan = steps[i].dplan;
        f.pass_id = steps[i].p0;
    }
    const FusedSteps* dsteps = nullptr;
    for (auto& e : c->fused)
        if (std::memcmp(e.host.get(), fs.get(), sizeof(FusedSteps)) == 0) { dsteps = e.dev; break; }
    if (!dsteps) {
        if (c->fused.size() >= 64) {  // callers cycling through libraries: start over
            HIPOK(hipDeviceSynchronize());
            for (auto& e : c->fused) (void)hipFree(e.dev);
            c->fused.clear();
        }
        FusedSteps* d = nullptr;
        HIPOK(hipMalloc((void**)&d, sizeof(FusedSteps)));
        HIPOK(hipMemcpy(d, fs.get(), sizeof(FusedSteps), hipMemcpyHostToDevice));
        c->fused.push_back(mirge_ctx::FusedEntry{std::move(fs), d});
        dsteps = d;
    }
    *dsteps_out = dsteps;
    return 0;
}

extern "C" int mirge_cascade_run(mirge_ctx* c, const mirge_reads* R, const mirge_lib* const* libs,
                                 const mirge_policy* pol, int32_t n_pass, mirge_result** out) {
    HostClock hc("cascade");
    if (!c || !R || !libs || !pol || !out || n_pass < 1 || n_pass > MIRGE_MAX_PASSES)
        return fail(-1, "mirge_cascade_run: bad argument");
    HIPOK(hipSetDevice(c->device));
    // read lengths present (host histogram from pack; a collapse result asks the device once)
    int32_t hist[MIRGE_MAX_READ_LEN + 1];
    if (R->hist_valid) std::memcpy(hist, R->len_hist, sizeof(hist));
    else {
        std::memset(hist, 0, sizeof(hist));
        for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
            const ReadGroup& g = R->g[gi];
            if (!g.n) continue;
            // conservative: every length the width group can hold is assumed present
            int lo = kGroupW[gi] == 1 ? 1 : (kGroupW[gi] == 2 ? 32 : 65), hi = kGroupW[gi] == 1 ? 31 : (kGroupW[gi] == 2 ? 64 : 128);
            for (int L = lo; L <= hi; L++) hist[L] = 1;
        }
    }
    // Everything below up to the launches depends only on (libraries, policies, read lengths present): it is
    // kept from the previous call and reused when those are unchanged (~45 us of host time per call otherwise,
    // on the critical path between the collapse's synchronisation and the first pass)
    std::string key;
    key.append(reinterpret_cast<const char*>(&n_pass), sizeof(n_pass));
    for (int32_t p = 0; p < n_pass; p++) {
        const uint64_t uid = libs[p] ? libs[p]->uid : 0;
        key.append(reinterpret_cast<const char*>(&uid), sizeof(uid));
        key.append(reinterpret_cast<const char*>(&pol[p]), sizeof(mirge_policy));
    }
    for (int L = 0; L <= MIRGE_MAX_READ_LEN; L++) key.push_back(hist[L] ? 1 : 0);
    if (c->casc_key != key) {
        c->casc_key.clear();
        CHECK(cascade_prepare(c, libs, pol, n_pass, hist, c->casc_steps, c->casc_rt, &c->casc_dsteps));
        c->casc_key = key;
    }
    const std::vector<PassStep>& steps = c->casc_steps;
    const ResolveTable& rt = c->casc_rt;
    const FusedSteps* dsteps = c->casc_dsteps;
    hc.lap("plans+fused");
    // MIRGE_FUSED_MAX: largest group (reads) that takes the one-launch path; 0 = always staged (tests)
    static const uint32_t fused_max = std::getenv("MIRGE_FUSED_MAX") ? (uint32_t)std::strtoul(std::getenv("MIRGE_FUSED_MAX"), nullptr, 10) : (1u << 20);
    auto res = std::make_unique<mirge_result>();
    res->ctx = c; res->n = R->n; res->n_pass = n_pass; res->reads = R;
    int rc = 0;
    const int big = largest_group(R);
    CHECK(stream_fork(c));
    // enqueue order: the small groups first (one fused launch each, or the staged launches if a group is too
    // large for that), the bulk group last: measured, its 2048-workgroup launches otherwise hold every CU and the
    // small kernels squeeze in between them, stretching single passes of the bulk group by 30 %
    int order[MIRGE_NGROUPS], no = 0;
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) if (gi != big) order[no++] = gi;
    order[no++] = big;
    for (int k = 0; k < MIRGE_NGROUPS && rc == 0; k++) {
        const int gi = order[k];
        c->cur = gi == big ? c->stream : c->aux;
        if (gi != big && R->g[gi].n <= fused_max) {
            if (kGroupW[gi] == 1) rc = cascade_group_fused<1>(c, R->g[gi], res->g[gi], dsteps, rt, group_tag(gi));
            else if (kGroupW[gi] == 2) rc = cascade_group_fused<2>(c, R->g[gi], res->g[gi], dsteps, rt, group_tag(gi));
            else rc = cascade_group_fused<4>(c, R->g[gi], res->g[gi], dsteps, rt, group_tag(gi));
            continue;
        }
        if (kGroupW[gi] == 1) rc = cascade_group<1>(c, R->g[gi], res->g[gi], steps, pol, rt, group_tag(gi));
        else if (kGroupW[gi] == 2) rc = cascade_group<2>(c, R->g[gi], res->g[gi], steps, pol, rt, group_tag(gi));
        else rc = cascade_group<4>(c, R->g[gi], res->g[gi], steps, pol, rt, group_tag(gi));
    }
    hc.lap("enqueue");
    { int jr = stream_join(c); if (rc == 0) rc = jr; }
    hc.lap("join");
    if (rc) { mirge_result_destroy(res.release()); return rc; }
    *out = res.release();
    return 0;
}

template <typename T>
static int fetch_field(mirge_ctx* c, const mirge_result* res, T* host_out, T* ResGroup::*field) {
    if (!host_out || !res->n) return 0;
    T* dfull = nullptr;
    CHECK(dalloc(c, &dfull, (size_t)res->n));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ResGroup& g = res->g[gi];
        const ReadGroup& rg = res->reads->g[gi];
        if (!g.n) continue;
        hipLaunchKernelGGL(k_scatter_out<T>, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream,
                           (const T*)(g.*field), g.n, rg.base, (const uint32_t*)rg.orig, dfull);
    }
    HIPOK(hipMemcpyAsync(host_out, dfull, (size_t)res->n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    c->release(dfull);
    return 0;
}

extern "C" int mirge_result_fetch(mirge_ctx* c, const mirge_result* res, int8_t* pass_out, int32_t* ref_out,
                                  int32_t* off_out, int8_t* mm_out) {
    if (!c || !res) return fail(-1, "mirge_result_fetch: bad argument");
    HIPOK(hipSetDevice(c->device));
    CHECK(fetch_field<int8_t>(c, res, pass_out, &ResGroup::pass));
    CHECK(fetch_field<int32_t>(c, res, ref_out, &ResGroup::ref));
    CHECK(fetch_field<int32_t>(c, res, off_out, &ResGroup::off));
    CHECK(fetch_field<int8_t>(c, res, mm_out, &ResGroup::mm));
    return 0;
}

// ------------------------------------------------------------------------------------------
// count join
// ------------------------------------------------------------------------------------------
extern "C" int mirge_count_join(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, int32_t exact_pass,
                                int32_t iso_pass, int64_t n_mirna, int64_t* class_sums, int64_t* exact, int64_t* iso) {
    if (!c || !U || !res || !class_sums || !exact || !iso || n_mirna < 0) return fail(-1, "mirge_count_join: bad argument");
    if (U->n_samples < 1) return fail(-1, "read set has no count matrix (collapse it or mirge_reads_set_counts)");
    if (res->n != U->n) return fail(-1, "result and read set differ in size");
    HIPOK(hipSetDevice(c->device));
    const int32_t S = U->n_samples, P = res->n_pass;
    const size_t n_cls = (size_t)P * S, n_tab = (size_t)std::max<int64_t>(n_mirna, 1) * S;
    unsigned long long* d = nullptr;
    CHECK(dalloc(c, &d, n_cls + 2 * n_tab));
    HIPOK(hipMemsetAsync(d, 0, (n_cls + 2 * n_tab) * 8, c->stream));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ResGroup& g = res->g[gi];
        if (!g.n) continue;
        LaunchScope ls(c, "k_join", g.n);
        hipLaunchKernelGGL(k_join, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, g.pass, g.ref,
                           U->g[gi].counts, g.n, S, P, exact_pass, iso_pass, d, d + n_cls, d + n_cls + n_tab);
    }
    // one device-to-host copy through pinned memory for all three tables (they are contiguous)
    const size_t words = n_cls + 2 * n_tab;
    if (words * 8 > c->join_pinned_bytes) {
        if (c->join_pinned) (void)hipHostFree(c->join_pinned);
        c->join_pinned = nullptr; c->join_pinned_bytes = 0;
        HIPOK(hipHostMalloc((void**)&c->join_pinned, words * 8 * 2, hipHostMallocDefault));
        c->join_pinned_bytes = words * 8 * 2;
    }
    HIPOK(hipMemcpyAsync(c->join_pinned, d, words * 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    std::memcpy(class_sums, c->join_pinned, n_cls * 8);
    if (n_mirna) {
        std::memcpy(exact, c->join_pinned + n_cls, (size_t)n_mirna * S * 8);
        std::memcpy(iso, c->join_pinned + n_cls + n_tab, (size_t)n_mirna * S * 8);
    }
    c->drain();
    c->release(d);
    return 0;
}

extern "C" int mirge_count_join_host(mirge_ctx* c, const int8_t* pass, const int32_t* ref, const uint32_t* counts,
                                     int64_t n, int32_t S, int32_t P, int32_t exact_pass, int32_t iso_pass,
                                     int64_t n_mirna, int64_t* class_sums, int64_t* exact, int64_t* iso) {
    if (!c || !class_sums || !exact || !iso || n < 0 || S < 1 || P < 1 || P > MIRGE_MAX_PASSES || n_mirna < 0 ||
        (n > 0 && (!pass || !ref || !counts)))
        return fail(-1, "mirge_count_join_host: bad argument");
    if (n >= 0xFFFFFFF0ll) return fail(-5, "more than 2^32 rows");
    HIPOK(hipSetDevice(c->device));
    for (int64_t i = 0; i < n; i++) {
        if (pass[i] >= P) return fail(-1, "pass index out of range");
        if ((pass[i] == exact_pass || pass[i] == iso_pass) && (ref[i] < 0 || ref[i] >= n_mirna))
            return fail(-1, "miRNA reference index out of range");
    }
    const size_t n_cls = (size_t)P * S, n_tab = (size_t)std::max<int64_t>(n_mirna, 1) * S;
    unsigned long long* d = nullptr; int8_t* dp = nullptr; int32_t* dr = nullptr; uint32_t* dc = nullptr;
    CHECK(dalloc(c, &d, n_cls + 2 * n_tab));
    CHECK(dalloc(c, &dp, (size_t)std::max<int64_t>(n, 1)));
    CHECK(dalloc(c, &dr, (size_t)std::max<int64_t>(n, 1)));
    CHECK(dalloc(c, &dc, (size_t)std::max<int64_t>(n, 1) * S));
    HIPOK(hipMemsetAsync(d, 0, (n_cls + 2 * n_tab) * 8, c->stream));
    if (n) {
        HIPOK(hipMemcpyAsync(dp, pass, (size_t)n, hipMemcpyHostToDevice, c->stream));
        HIPOK(hipMemcpyAsync(dr, ref, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        HIPOK(hipMemcpyAsync(dc, counts, (size_t)n * S * 4, hipMemcpyHostToDevice, c->stream));
        LaunchScope ls(c, "k_join", (double)n);
        hipLaunchKernelGGL(k_join, dim3(grid_for(c, (size_t)n)), dim3(MIRGE_BLOCK), 0, c->stream, dp, dr, dc, (uint32_t)n,
                           S, P, exact_pass, iso_pass, d, d + n_cls, d + n_cls + n_tab);
    }
    // one device-to-host copy through pinned memory for all three tables (they are contiguous)
    const size_t words = n_cls + 2 * n_tab;
    if (words * 8 > c->join_pinned_bytes) {
        if (c->join_pinned) (void)hipHostFree(c->join_pinned);
        c->join_pinned = nullptr; c->join_pinned_bytes = 0;
        HIPOK(hipHostMalloc((void**)&c->join_pinned, words * 8 * 2, hipHostMallocDefault));
        c->join_pinned_bytes = words * 8 * 2;
    }
    HIPOK(hipMemcpyAsync(c->join_pinned, d, words * 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    std::memcpy(class_sums, c->join_pinned, n_cls * 8);
    if (n_mirna) {
        std::memcpy(exact, c->join_pinned + n_cls, (size_t)n_mirna * S * 8);
        std::memcpy(iso, c->join_pinned + n_cls + n_tab, (size_t)n_mirna * S * 8);
    }
    c->drain();
    c->release(d); c->release(dp); c->release(dr); c->release(dc);
    return 0;
}

extern "C" int mirge_variant_tally(mirge_ctx* c, const mirge_reads* U, const mirge_result* res, const mirge_lib* mirna,
                                   int32_t exact_pass, int32_t iso_pass, int32_t iso_trim5, int64_t n_mirna,
                                   int64_t* accepted, int64_t* canonical, int64_t* census) {
    static_assert(MIRGE_TALLY_POSITIONS == MIRGE_TALLY_MAXPOS, "tally positions");
    if (!c || !U || !res || !mirna || !accepted || !canonical || !census || n_mirna != mirna->n_refs)
        return fail(-1, "mirge_variant_tally: bad argument");
    if (U->n_samples < 1) return fail(-1, "read set has no count matrix");
    if (res->n != U->n) return fail(-1, "result and read set differ in size");
    HIPOK(hipSetDevice(c->device));
    const int32_t S = U->n_samples;
    const size_t n_rs = (size_t)std::max<int64_t>(n_mirna, 1) * S, n_cen = n_rs * MIRGE_TALLY_MAXPOS * 16;
    unsigned long long* d = nullptr;
    CHECK(dalloc(c, &d, 2 * n_rs + n_cen));
    HIPOK(hipMemsetAsync(d, 0, (2 * n_rs + n_cen) * 8, c->stream));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        if (kGroupW[gi] != 1) continue;  // a read annotated to a miRNA is at most 3 nt longer than it
        const ResGroup& g = res->g[gi];
        if (!g.n) continue;
        LaunchScope ls(c, "k_tally", g.n);
        hipLaunchKernelGGL(k_tally, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, view_of<1>(U->g[gi]), g.pass, g.ref,
                           g.off, U->g[gi].counts, S, mirna->view(), exact_pass, iso_pass, iso_trim5, d, d + n_rs, d + 2 * n_rs);
    }
    std::vector<unsigned long long> h(2 * n_rs + n_cen);
    HIPOK(hipMemcpyAsync(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    c->drain();
    std::memcpy(accepted, h.data(), (size_t)n_mirna * S * 8);
    std::memcpy(canonical, h.data() + n_rs, (size_t)n_mirna * S * 8);
    std::memcpy(census, h.data() + 2 * n_rs, (size_t)n_mirna * S * MIRGE_TALLY_MAXPOS * 16 * 8);
    c->release(d);
    return 0;
}
