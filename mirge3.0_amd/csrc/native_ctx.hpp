// native_ctx.hpp -- part of mirge_native.hip (one translation unit): errors, device context: streams, stream-ordered buffer pool, profiler, host timing hook.
#pragma once
// (errors -- g_err, fail, CHECK, mirge_last_error -- and HostClock: native_host.hpp)
#define HIPOK(expr)                                                                          \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e));              \
    } while (0)

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
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_meta = nullptr, ev_meta_small = nullptr, ev_bulk_counted = nullptr;
    // the small read groups' one-launch cascades each on a stream of their own (round 3): with the bulk group's passes in one
    // launch they only get the chip when its workgroups retire, and on ONE stream three of them then ran one after the other
    // (56 + 69 + 49 us behind the bulk kernel, with the join waiting); side by side they take what the longest takes
#define MIRGE_N_XAUX 3
#define MIRGE_SPEC_TICKET_ROUNDS 8192  // groups of up to 2 M reads pick inside k_cascade_spec
    hipStream_t xaux[MIRGE_N_XAUX] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_xfork = nullptr, ev_xjoin[MIRGE_N_XAUX] = {nullptr, nullptr, nullptr};
    bool xaux_used = false;
    bool join_pending = false;  // a cascade's side streams are not joined yet: whoever touches the ctx next does it (join_pending_now)
    // mirge_collapse_cascade: the bulk group's cascade is already queued on the main stream while the small groups'
    // collapse tail and cascades are still being enqueued on the second one.  They do not depend on it: no fork
    // wait (it would serialise them behind ~1.3 ms of kernels), and no pool block goes back into circulation
    // before the final join (a block still in use on one stream could be handed to the other).
    bool overlap_mode = false;
    int n_cu = 256;
    // pool
    std::multimap<size_t, void*> free_blocks;
    std::unordered_map<void*, size_t> sizes;
    size_t pool_bytes = 0;
    // profiler
    bool profiling = false;
    bool prof_units = true;  // read the cascade's per-pass survivor counts back (mirge_ctx_profile_units)
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
    // page-locked staging of mirge_annotation_csv_device (the text of mapped.csv + unmapped.csv), kept between samples
    uint8_t* csv_pinned = nullptr;
    size_t csv_pinned_bytes = 0;
    // start / end clocks of the workgroups of the last profiled k_cascade_bulk launch (mirge_cascade_wg_times)
    // k_cascade_heavy (kernels_cascade.hpp): per read group {reads listed, workgroups done}, zero between launches; the threshold the
    // current configuration's steps carry (0: no library holds a bucket that large -- no deferral, no extra launch)
    uint32_t* heavy_cnt = nullptr;
    // (round 6) the join of a mirge_collapse_cascade step: extra stream 0 collected the other extra streams when the cascades were queued
    // (x0_gathered), and the host has since waited for the last thing queued on `aux` (aux_drained): the main stream then waits for
    // extra stream 0 alone -- one dependency packet and four runtime calls instead of two and eight in the step's tail
    bool x0_gathered = false, aux_drained = false;
    bool xaux_forked = false;  // collapse_impl -> cascade_launch_groups: the extra streams were put behind `aux` for the scatter kernels,
    int small_slot[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};  // ... group gi's on extra stream small_slot[gi] (-1: on `aux`): its cascade goes there too
    uint32_t* spec_tickets = nullptr;  // [MIRGE_NGROUPS][MIRGE_SPEC_TICKET_ROUNDS], see k_cascade_spec
    uint32_t casc_big_t = 0;
    bool casc_rep = false;  // the configuration's libraries repeat themselves: the cascade kernels' repeat-aware build (align_hybrid<.., REP>)
    uint32_t* wg_pinned = nullptr;
    size_t wg_pinned_words = 0;
    uint32_t wg_grid = 0;
    // the count join's device tables, kept between calls and cleared right AFTER a call's read-back: the next sample's
    // k_join starts behind its cascade without two fill kernels and their launch gaps in front of it
    unsigned long long* join_dev = nullptr;
    size_t join_dev_words = 0;
    bool join_dev_clean = false;
    struct PlanEntry { uint64_t uid; MirgePolicy pol; MirgePlanTable* dplan; };
    std::vector<PlanEntry> plans;
    struct FusedEntry { std::unique_ptr<FusedSteps> host; FusedSteps* dev; };
    std::vector<FusedEntry> fused;  // step lists of k_cascade_fused already on the device
    std::string casc_key;           // configuration of the previous mirge_cascade_run ...
    std::vector<PassStep> casc_steps;  // ... and what was prepared for it
    ResolveTable casc_rt;
    const FusedSteps* casc_dsteps = nullptr;
    struct WalksEntry { std::unique_ptr<BulkWalks> host; BulkWalks* dev; };
    std::vector<WalksEntry> walks;      // walk lists of k_cascade_bulk already on the device
    const BulkWalks* casc_dwalks[2] = {nullptr, nullptr};  // [0] the one-word group's list (exact steps ride along), [1] wider groups
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
    // (a no-op while overlap_mode holds every block back, see below)
    void flush_deferred() { if (overlap_mode) return; for (void* p : deferred) release(p); deferred.clear(); }
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
    if (c->overlap_mode) return 0;
    HIPOK(hipEventRecord(c->ev_fork, c->stream));
    HIPOK(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
    return 0;
}
// the extra streams take over from `aux` (whatever was queued there -- the small groups' collapse tail -- comes first)
static int xaux_fork(mirge_ctx* c) {
    HIPOK(hipEventRecord(c->ev_xfork, c->aux));
    for (int k = 0; k < MIRGE_N_XAUX; k++) HIPOK(hipStreamWaitEvent(c->xaux[k], c->ev_xfork, 0));
    c->xaux_used = true;
    return 0;
}
static int stream_join(mirge_ctx* c) {
    c->cur = c->stream;
    hipError_t e = hipSuccess;
    const bool x = c->xaux_used;
    static const bool short_join = !(std::getenv("MIRGE_SHORT_JOIN") && std::atoi(std::getenv("MIRGE_SHORT_JOIN")) == 0);  // A/B
    const bool only_x0 = short_join && x && c->x0_gathered && c->aux_drained;
    c->x0_gathered = c->aux_drained = false;
    if (only_x0) {
        c->xaux_used = false;
        e = hipEventRecord(c->ev_xjoin[0], c->xaux[0]);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_xjoin[0], 0);
        c->flush_deferred();
        if (e != hipSuccess) return fail(-2, std::string("stream join: ") + hipGetErrorString(e));
        return 0;
    }
    if (x) {
        // Every wait is a barrier packet its queue works through in order (~7 us each even when already satisfied), and a wait
        // that is NOT yet satisfied costs a queue-to-queue hop (~15 us) once it is.  `aux` (idle by then) collects the extra
        // streams 1.. early; the main stream waits for `aux` and then, directly, for extra stream 0 -- the one the largest
        // small group runs on, the last to finish (one hop behind it; collected by `aux` too it was two: 35 us between the
        // end of that group's cascade and k_join, profiles/r03_timeline.txt)
        for (int k = 1; k < MIRGE_N_XAUX && e == hipSuccess; k++) {
            e = hipEventRecord(c->ev_xjoin[k], c->xaux[k]);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->aux, c->ev_xjoin[k], 0);
        }
        c->xaux_used = false;
    }
    if (e == hipSuccess) e = hipEventRecord(c->ev_join, c->aux);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
    if (x && e == hipSuccess) {
        e = hipEventRecord(c->ev_xjoin[0], c->xaux[0]);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_xjoin[0], 0);
    }
    c->flush_deferred();  // reused only by work queued on the main stream after the wait
    if (e != hipSuccess) return fail(-2, std::string("stream join: ") + hipGetErrorString(e));
    return 0;
}
// mirge_cascade_run / mirge_collapse_cascade leave their side streams unjoined; every entry point that enqueues on the main
// stream, hands buffers back to the pool or synchronises joins them first.  mirge_count_join uses the slack: the bulk
// group's part of the join runs before the wait, beside the small groups' cascades.
static int join_pending_now(mirge_ctx* c) {
    if (!c->join_pending) return 0;
    c->join_pending = false;
    return stream_join(c);
}
static int largest_group(const struct mirge_reads* R);

static inline int grid_for(const mirge_ctx* c, size_t n, int per_block = MIRGE_BLOCK) {
    size_t blocks = (n + per_block - 1) / per_block;
    static const size_t per_cu = std::getenv("MIRGE_GRID_PER_CU") ? (size_t)std::atoi(std::getenv("MIRGE_GRID_PER_CU")) : 8;
    size_t cap = (size_t)c->n_cu * per_cu;
    return (int)std::max<size_t>(1, std::min(blocks, cap));
}

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
    // MIRGE_STREAM_PRIORITY=1 (round 6 experiment): the main stream at the device's highest stream priority, the side streams at its lowest
    static const bool prio = std::getenv("MIRGE_STREAM_PRIORITY") && std::atoi(std::getenv("MIRGE_STREAM_PRIORITY")) == 1;
    int p_low = 0, p_high = 0;
    if (prio) HIPOK(hipDeviceGetStreamPriorityRange(&p_low, &p_high));
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; }
    else { HIPOK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, p_high)); c->own_stream = true; }
    c->cur = c->stream;
    HIPOK(hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, p_low));
    for (int k = 0; k < MIRGE_N_XAUX; k++) {
        HIPOK(hipStreamCreateWithPriority(&c->xaux[k], hipStreamNonBlocking, p_low));
        HIPOK(hipEventCreateWithFlags(&c->ev_xjoin[k], hipEventDisableTiming));
    }
    HIPOK(hipEventCreateWithFlags(&c->ev_xfork, hipEventDisableTiming));
    HIPOK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPOK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIPOK(hipEventCreateWithFlags(&c->ev_meta, hipEventDisableTiming));
    HIPOK(hipEventCreateWithFlags(&c->ev_meta_small, hipEventDisableTiming));
    HIPOK(hipEventCreateWithFlags(&c->ev_bulk_counted, hipEventDisableTiming));
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
    if (c->csv_pinned) (void)hipHostFree(c->csv_pinned);
    if (c->wg_pinned) (void)hipHostFree(c->wg_pinned);
    if (c->heavy_cnt) (void)hipFree(c->heavy_cnt);
    if (c->spec_tickets) (void)hipFree(c->spec_tickets);
    if (c->join_dev) (void)hipFree(c->join_dev);
    for (auto& e : c->plans) (void)hipFree(e.dplan);
    for (auto& e : c->fused) (void)hipFree(e.dev);
    for (auto& e : c->walks) (void)hipFree(e.dev);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_meta) (void)hipEventDestroy(c->ev_meta);
    if (c->ev_meta_small) (void)hipEventDestroy(c->ev_meta_small);
    if (c->ev_bulk_counted) (void)hipEventDestroy(c->ev_bulk_counted);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    for (int k = 0; k < MIRGE_N_XAUX; k++) {
        if (c->xaux[k]) (void)hipStreamDestroy(c->xaux[k]);
        if (c->ev_xjoin[k]) (void)hipEventDestroy(c->ev_xjoin[k]);
    }
    if (c->ev_xfork) (void)hipEventDestroy(c->ev_xfork);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int mirge_ctx_sync(mirge_ctx* c) {
    if (!c) return fail(-1, "ctx is NULL");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    HIPOK(hipStreamSynchronize(c->stream));
    c->drain();
    return 0;
}

extern "C" int mirge_ctx_timer_start(mirge_ctx* c) {
    if (c) CHECK(join_pending_now(c));
    if (!c) return fail(-1, "ctx is NULL");
    HIPOK(hipEventRecord(c->t0, c->stream));
    return 0;
}
extern "C" int mirge_ctx_timer_stop(mirge_ctx* c, double* ms_out) {
    if (c) CHECK(join_pending_now(c));
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
extern "C" int mirge_ctx_profile_units(mirge_ctx* c, int32_t on) {
    if (!c) return fail(-1, "ctx is NULL");
    c->prof_units = on != 0;
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
