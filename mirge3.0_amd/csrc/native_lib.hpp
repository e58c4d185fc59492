// native_lib.hpp -- part of mirge_native.hip (one translation unit): libraries: 2-bit text + invalid bitmap on the device, probe tables built on the device.
#pragma once
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
    uint32_t* dcoarse = nullptr;               // [(total >> MIRGE_COARSE_SHIFT) + 2][2]: see ResolveTable
    MirgeKTable* dtables = nullptr;           // [MIRGE_SHAPE_SLOTS] on the device
    std::vector<MirgeKTable> htables;          // host mirror (device pointers), by mirge_shape_id
    size_t device_bytes = 0;
    std::mutex mu;
    int64_t n_refs = 0;
    int kmax = 8;
    uint32_t max_bucket = 0;  // windows in the fullest bucket of any probe table built so far (k_table_heavy_list): > casc_big_t = reads may be deferred
    // whole-read tables of the exact passes (kernels_cascade.hpp, ExactStep), by the set of lengths they hold (bit l of the key)
    struct ExactTab { uint64_t* slots = nullptr; uint32_t mask = 0; uint64_t windows = 0; };
    std::map<uint32_t, ExactTab> exact;

    MirgeLibView view() const {
        MirgeLibView v;
        v.T = dT; v.inv = dinv; v.ref_start = dref_start; v.tables = dtables;
        v.total = h.total; v.n_refs = (uint32_t)n_refs; v.kmax = kmax;
        return v;
    }
};

static int lib_upload(mirge_ctx* c, std::unique_ptr<mirge_lib>& L, int64_t n_refs, mirge_lib** out);

extern "C" int mirge_lib_create(mirge_ctx* c, const char* seq, const int64_t* off, int64_t n_refs, mirge_lib** out) {
    if (!c || !out || (!seq && n_refs > 0) || !off || n_refs < 0) return fail(-1, "mirge_lib_create: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    auto L = std::make_unique<mirge_lib>();
    L->ctx = c;
    L->uid = g_lib_uid.fetch_add(1);
    std::string err;
    int rc = mirge_hostlib_build(L->h, seq, off, n_refs, err);
    if (rc) return fail(rc, "mirge_lib_create: " + err);
    return lib_upload(c, L, n_refs, out);
}

// The packed image of a library -- what mirge_lib_create derives from the sequences -- handed back in: the one-time
// conversion a library cache keeps next to the index (hard part 2), so that a process does not parse 150 MB of
// FASTA / decode an .ebwt and pack it again before its first kernel.  T: (total + 31) / 32 + 8 words, inv:
// (total + 63) / 64 + 4 words (padding ones), ref_start: n_refs + 1; the probe tables are still built on the device.
extern "C" int mirge_lib_create_packed(mirge_ctx* c, const uint64_t* T, int64_t n_T, const uint64_t* inv, int64_t n_inv,
                                       const uint32_t* ref_start, int64_t n_refs, uint64_t total, int32_t kmax,
                                       uint64_t valid_positions, mirge_lib** out) {
    if (!c || !out) return fail(-1, "mirge_lib_create_packed: bad argument");
    CHECK(lib_packed_args_check(T, n_T, inv, n_inv, ref_start, n_refs, total, kmax));  // (native_host.hpp)
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    auto L = std::make_unique<mirge_lib>();
    L->ctx = c;
    L->uid = g_lib_uid.fetch_add(1);
    L->h.n_refs = n_refs; L->h.total = total; L->h.kmax = kmax; L->h.valid_positions = valid_positions;
    L->h.T.assign(T, T + n_T);
    L->h.inv.assign(inv, inv + n_inv);
    L->h.ref_start.assign(ref_start, ref_start + n_refs + 1);
    return lib_upload(c, L, n_refs, out);
}
// sizes[4] = {words of T, words of inv, total, valid positions}; then the arrays themselves (caller-allocated)
extern "C" int mirge_lib_packed_sizes(const mirge_lib* L, int64_t* sizes, int32_t* kmax) {
    if (!L || !sizes || !kmax) return fail(-1, "mirge_lib_packed_sizes: bad argument");
    sizes[0] = (int64_t)L->h.T.size(); sizes[1] = (int64_t)L->h.inv.size(); sizes[2] = (int64_t)L->h.total; sizes[3] = (int64_t)L->h.valid_positions;
    *kmax = L->h.kmax;
    return 0;
}
extern "C" int mirge_lib_packed_copy(const mirge_lib* L, uint64_t* T, uint64_t* inv, uint32_t* ref_start) {
    if (!L || !T || !inv || !ref_start) return fail(-1, "mirge_lib_packed_copy: bad argument");
    std::memcpy(T, L->h.T.data(), L->h.T.size() * 8);
    std::memcpy(inv, L->h.inv.data(), L->h.inv.size() * 8);
    std::memcpy(ref_start, L->h.ref_start.data(), L->h.ref_start.size() * 4);
    return 0;
}

static int lib_upload(mirge_ctx* c, std::unique_ptr<mirge_lib>& L, int64_t n_refs, mirge_lib** out) {
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
    size_t nC = 0;
    if (n_refs > 0) {  // the granule table of ResolveTable (kernels_cascade.hpp)
        const size_t n_gran = (size_t)(L->h.total >> MIRGE_COARSE_SHIFT) + 2;
        if ((uint64_t)n_refs >= (1ull << 27)) return fail(-1, "mirge_lib_create: more than 2^27 references");
        std::vector<uint32_t> coarse(2 * n_gran);
        const std::vector<uint32_t>& rs = L->h.ref_start;
        uint32_t t = 0;
        for (size_t b = 0; b < n_gran; b++) {
            const uint64_t x = (uint64_t)b << MIRGE_COARSE_SHIFT, xe = x + (1ull << MIRGE_COARSE_SHIFT);
            while ((int64_t)t + 1 < n_refs && rs[(size_t)t + 1] <= x) t++;
            uint32_t inside = 0;  // references that start in (x, xe)
            while ((int64_t)t + 1 + inside < n_refs && rs[(size_t)t + 1 + inside] < xe) inside++;
            const uint32_t code = inside == 0 ? 16u : inside == 1 ? (uint32_t)(rs[(size_t)t + 1] - x) : 17u;
            coarse[2 * b] = t << 5 | code;
            coarse[2 * b + 1] = rs[t];
        }
        nC = coarse.size() * 4;
        HIPOK(hipMalloc((void**)&L->dcoarse, nC));
        HIPOK(hipMemcpy(L->dcoarse, coarse.data(), nC, hipMemcpyHostToDevice));
    }
    L->device_bytes = nT + nI + nR + nC + sizeof(MirgeKTable) * MIRGE_SHAPE_SLOTS;
    *out = L.release();
    return 0;
}

extern "C" void mirge_lib_destroy(mirge_lib* L) {
    if (!L) return;
    (void)hipSetDevice(L->ctx->device);
    (void)hipStreamSynchronize(L->ctx->stream);
    (void)hipFree(L->dT); (void)hipFree(L->dinv); (void)hipFree(L->dref_start); (void)hipFree(L->dcoarse); (void)hipFree(L->dtables);
    for (auto& t : L->htables) { (void)hipFree((void*)t.bucket); (void)hipFree((void*)t.pos); (void)hipFree((void*)((uintptr_t)t.bits & ~(uintptr_t)3)); }
    for (auto& e : L->exact) (void)hipFree(e.second.slots);
    delete L;
}
extern "C" int64_t mirge_lib_n_refs(const mirge_lib* L) { return L ? L->n_refs : -1; }
extern "C" int64_t mirge_lib_device_bytes(const mirge_lib* L) { return L ? (int64_t)L->device_bytes : -1; }

// one probe shape (k1 bases, gap, k2 bases)
struct ShapeJob {
    int k1 = 0, gap = 0, k2 = 0;
};

// Build the tables of `jobs` on the device (k_table_pass: count, scan, fill; bitmap or self-contained entries) and publish them
// in the library's registry.  Everything is queued on the ctx stream and the host waits ONCE, at the end: a table at a time
// with two synchronisations, its own scan area and CSR bounds and a blocking registry update was 0.09 s per process for the
// human set's 56 tables against 0.04 s of kernels (bench.py cli_path: probe_tables_s).  (One allocation for a whole batch is
// NOT the way: a single hipMalloc of several GB took longer than all of this, 0.4-0.5 s.)
static int lib_build_shapes(mirge_lib* L, const std::vector<ShapeJob>& jobs) {
    mirge_ctx* c = L->ctx;
    struct Built { int sid; uint64_t nb; uint32_t *A, *dpos, *dbits; uint64_t* dentry; };
    std::vector<Built> built;
    // what only construction needs, once for the batch (the launches of consecutive tables are ordered by the stream): the scan's
    // work area and the CSR bounds of the tables that end up as self-contained entries (up to 1 GB each)
    void* tmp = nullptr;
    uint32_t* A_scratch = nullptr;
    size_t tmp_cap = 0;
    uint64_t nb_max = 0, nb_scratch = 0;
    for (const ShapeJob& j : jobs) {
        const uint64_t nb = 1ull << (2 * (j.k1 + j.k2));
        nb_max = std::max(nb_max, nb);
        if (j.k1 + j.k2 > MIRGE_BITMAP_MAXK) nb_scratch = std::max(nb_scratch, nb);
    }
    hipError_t e = hipcub::DeviceScan::InclusiveSum(nullptr, tmp_cap, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(nb_max + 2), c->stream);  // size query
    if (e == hipSuccess) e = hipMalloc(&tmp, std::max<size_t>(tmp_cap, 16));
    if (e == hipSuccess && nb_scratch) e = hipMalloc((void**)&A_scratch, (nb_scratch + 2) * 4);
    const int grid = c->n_cu * 8;
    // (round 6) the heavy buckets' position lists in ascending order: their list, the segmented sort's second array and work area
    const uint32_t heavy_cap = (uint32_t)std::min<uint64_t>(L->h.total / (MIRGE_LIGHT_MAX + 1) + 16, 0x7FFFFFF0ull);
    uint32_t *seg_begin = nullptr, *seg_end = nullptr, *n_heavy = nullptr, *pos_sorted = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_cap = 0;
    const bool sort_heavy = true;  // (not a switch: verify_heavy's early exit and k_cascade_heavy rely on ascending lists)
    int pos_bits = 1;
    while (pos_bits < 32 && (1ull << pos_bits) < L->h.total + 1) pos_bits++;
    if (e == hipSuccess && sort_heavy && !jobs.empty()) {
        e = hipMalloc((void**)&seg_begin, (size_t)heavy_cap * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&seg_end, (size_t)heavy_cap * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&n_heavy, 64);
        if (e == hipSuccess) e = hipMalloc((void**)&pos_sorted, std::max<size_t>((size_t)L->h.total, 1) * 4);
    }
    for (size_t k = 0; k < jobs.size() && e == hipSuccess; k++) {
        const ShapeJob& j = jobs[k];
        Built t{mirge_shape_id(j.k1, j.gap, j.k2), 1ull << (2 * (j.k1 + j.k2)), nullptr, nullptr, nullptr, nullptr};
        const uint64_t nb = t.nb;
        const bool entries = j.k1 + j.k2 > MIRGE_BITMAP_MAXK;  // a table too large for a bitmap: self-contained entries (MirgeKTable)
        if (!entries) e = hipMalloc((void**)&t.A, (nb + 2) * 4);
        // pos[] for every position of the text: the exact number (a few k-mers short of it) is only known after the scan
        if (e == hipSuccess) e = hipMalloc((void**)&t.dpos, std::max<size_t>((size_t)L->h.total, 1) * 4);
        if (e == hipSuccess && (!entries || MIRGE_PRESENCE_FILTER)) e = hipMalloc((void**)&t.dbits, (size_t)((nb + 31) / 32) * 4);
        if (e == hipSuccess && entries) e = hipMalloc((void**)&t.dentry, nb * 8);
        built.push_back(t);  // (whatever was allocated is freed below on failure)
        uint32_t* A = entries ? A_scratch : t.A;
        if (e == hipSuccess) e = hipMemsetAsync(A, 0, (nb + 2) * 4, c->stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_table_pass<false>, dim3(grid), dim3(MIRGE_BLOCK), 0, c->stream, L->dT, L->dinv, L->h.total, j.k1,
                           j.gap, j.k2, A, (uint32_t*)nullptr);
        size_t tb = tmp_cap;
        e = hipcub::DeviceScan::InclusiveSum(tmp, tb, A, A, (int)(nb + 2), c->stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_table_pass<true>, dim3(grid), dim3(MIRGE_BLOCK), 0, c->stream, L->dT, L->dinv, L->h.total, j.k1,
                           j.gap, j.k2, A, t.dpos);
        if (sort_heavy && L->h.total > MIRGE_LIGHT_MAX) {
            // buckets of more than MIRGE_LIGHT_MAX windows (what verify_heavy scans): listed, sorted by position, written back.
            // One host wait per table for the number of segments (a launch over every possible segment would be 8 M workgroups
            // for the human mRNA table): ~56 tables, a few ms per process on top of the tables' 40.
            uint32_t nh = 0;
            e = hipMemsetAsync(n_heavy, 0, 8, c->stream);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(k_table_heavy_list, dim3(grid_for(c, (size_t)std::min<uint64_t>(nb, 1ull << 24))), dim3(MIRGE_BLOCK), 0, c->stream,
                               (const uint32_t*)A, nb, (uint32_t)MIRGE_LIGHT_MAX + 1u, n_heavy, heavy_cap, seg_begin, seg_end,
                               (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, std::max<uint64_t>(1024, 32 * (L->h.total / nb + 1))));
            uint32_t nh2[2] = {0, 0};
            e = hipMemcpyAsync(nh2, n_heavy, 8, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) break;
            nh = std::min(nh2[0], heavy_cap);
            L->max_bucket = std::max(L->max_bucket, nh2[1]);
            if (nh) {
                size_t need = 0;
                e = hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, need, (const uint32_t*)t.dpos, pos_sorted, (int)std::min<uint64_t>(L->h.total, 0x7FFFFFFFull),
                                                               (int)nh, seg_begin, seg_end, 0, pos_bits, c->stream);
                if (e == hipSuccess && need > sort_tmp_cap) {
                    (void)hipFree(sort_tmp); sort_tmp = nullptr; sort_tmp_cap = 0;
                    e = hipMalloc(&sort_tmp, need + need / 4 + 256);
                    if (e == hipSuccess) sort_tmp_cap = need + need / 4 + 256;
                }
                size_t tb2 = sort_tmp_cap;
                if (e == hipSuccess)
                    e = hipcub::DeviceSegmentedRadixSort::SortKeys(sort_tmp, tb2, (const uint32_t*)t.dpos, pos_sorted, (int)std::min<uint64_t>(L->h.total, 0x7FFFFFFFull),
                                                                   (int)nh, seg_begin, seg_end, 0, pos_bits, c->stream);
                if (e != hipSuccess) break;
                hipLaunchKernelGGL(k_table_heavy_copy, dim3((unsigned)std::min<uint32_t>(nh, (uint32_t)c->n_cu * 16u)), dim3(MIRGE_BLOCK), 0, c->stream,
                                   (const uint32_t*)seg_begin, (const uint32_t*)seg_end, nh, (const uint32_t*)pos_sorted, t.dpos);
            }
        }
        if (!entries || MIRGE_PRESENCE_FILTER)  // non-empty-bucket bitmap (MIRGE_PRESENCE_FILTER: in front of the entries of a large table too)
            hipLaunchKernelGGL(k_table_bits, dim3(grid_for(c, (size_t)((nb + 31) / 32))), dim3(MIRGE_BLOCK), 0, c->stream, A, nb, t.dbits);
        if (entries)
            hipLaunchKernelGGL(k_table_entries, dim3(grid_for(c, nb)), dim3(MIRGE_BLOCK), 0, c->stream, A, t.dpos, nb, t.dentry);
    }
    // tables complete; no kernel may be reading the registry while it changes
    { const hipError_t e2 = hipStreamSynchronize(c->stream); if (e == hipSuccess) e = e2; }
    if (e == hipSuccess) e = hipGetLastError();
    (void)hipFree(tmp);
    (void)hipFree(A_scratch);
    (void)hipFree(seg_begin); (void)hipFree(seg_end); (void)hipFree(n_heavy); (void)hipFree(pos_sorted); (void)hipFree(sort_tmp);
    if (e != hipSuccess) {
        for (auto& t : built) { (void)hipFree(t.A); (void)hipFree(t.dpos); (void)hipFree(t.dbits); (void)hipFree(t.dentry); }
        return fail(e == hipErrorOutOfMemory ? -3 : -2, std::string("probe table construction: ") + hipGetErrorString(e));
    }
    for (const Built& t : built) {
        L->device_bytes += (t.dentry ? t.nb * 8 : (t.nb + 2) * 4) + (size_t)L->h.total * 4 + (t.dbits ? (size_t)((t.nb + 31) / 32) * 4 : 0);
        L->htables[t.sid].bucket = t.dentry ? (const void*)t.dentry : (const void*)t.A;
        L->htables[t.sid].pos = t.dpos;
        L->htables[t.sid].bits = (t.dentry && t.dbits) ? (const uint32_t*)((uintptr_t)t.dbits | 1u) : t.dbits;  // (the mark: entries behind a filter)
    }
    if (!built.empty())
        HIPOK(hipMemcpy(L->dtables, L->htables.data(), sizeof(MirgeKTable) * MIRGE_SHAPE_SLOTS, hipMemcpyHostToDevice));
    return 0;
}

// build the tables of the wanted probe shapes that do not exist yet
static int lib_prepare_shapes(mirge_lib* L, const std::vector<ShapeJob>& wanted) {
    std::lock_guard<std::mutex> lk(L->mu);
    std::vector<ShapeJob> jobs;
    std::vector<int> seen;
    for (const auto& w : wanted) {
        if (w.k1 < 1 || w.k2 < 0 || w.k1 + w.k2 > (w.k2 == 0 ? MIRGE_KMAX0 : MIRGE_KMAX) || w.gap < 0 || w.gap > 31 || (w.k2 == 0 && w.gap != 0))
            return fail(-1, "probe shape out of range");
        const int sid = mirge_shape_id(w.k1, w.gap, w.k2);
        if (L->htables[sid].bucket || std::find(seen.begin(), seen.end(), sid) != seen.end()) continue;
        seen.push_back(sid);
        jobs.push_back(w);
    }
    if (jobs.empty()) return 0;
    HIPOK(hipSetDevice(L->ctx->device));
    return lib_build_shapes(L, jobs);
}

// The whole-read table of an exact pass over this library for the read lengths in `lmask` (bit l, 1 <= l <= 31): every valid
// window of those lengths, keyed by its sequence, lowest position kept (k_exact_table).  *out stays empty (slots == nullptr)
// when the table would hold more than MIRGE_EXACT_MAX_WINDOWS windows: the pass then keeps its probe path.
#ifndef MIRGE_EXACT_MAX_WINDOWS
#define MIRGE_EXACT_MAX_WINDOWS (4u << 20)  // 4 M windows -> at most 16 M slots = 128 MB; the human miRNA / pre-tRNA sets hold 0.1 / 1.3 M
#endif
// A library keeps the tables of its last few length sets (a batch of samples has one or two; a caller cycling through many
// would otherwise collect up to 128 MB each).  Called by cascade_prepare BEFORE it asks for any table of a new configuration:
// beyond 6 everything goes, behind a synchronisation (walk lists on the device may still name them; the caller rebuilds its
// lists, and a dropped table is built again when its lengths come back).  Returns true when tables were dropped.
static bool lib_exact_trim(mirge_lib* L) {
    std::lock_guard<std::mutex> lk(L->mu);
    if (L->exact.size() < 6) return false;
    (void)hipSetDevice(L->ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& e : L->exact)
        if (e.second.slots) { L->device_bytes -= ((size_t)e.second.mask + 1) * 8; (void)hipFree(e.second.slots); }
    L->exact.clear();
    return true;
}

static int lib_exact_table(mirge_lib* L, uint32_t lmask, mirge_lib::ExactTab* out) {
    *out = mirge_lib::ExactTab();
    lmask &= 0xFFFFFFFEu;
    if (!lmask) return 0;
    std::lock_guard<std::mutex> lk(L->mu);
    auto it = L->exact.find(lmask);
    if (it != L->exact.end()) { *out = it->second; return 0; }
    mirge_ctx* c = L->ctx;
    HIPOK(hipSetDevice(c->device));
    mirge_lib::ExactTab tab;
    if ((uint64_t)__builtin_popcount(lmask) * L->h.valid_positions <= 4ull * MIRGE_EXACT_MAX_WINDOWS) {  // (an upper bound first: no kernel over a 130 Mb text for nothing)
        unsigned long long* dcount = nullptr;
        HIPOK(hipMalloc((void**)&dcount, 8));
        HIPOK(hipMemsetAsync(dcount, 0, 8, c->stream));
        const int grid = (int)std::min<uint64_t>((uint64_t)c->n_cu * 8, std::max<uint64_t>(1, (L->h.total + MIRGE_BLOCK - 1) / MIRGE_BLOCK));
        hipLaunchKernelGGL(k_exact_table<false>, dim3(grid), dim3(MIRGE_BLOCK), 0, c->stream, L->dT, L->dinv, L->h.total, lmask,
                           (uint64_t*)nullptr, 0u, dcount);
        unsigned long long n = 0;
        hipError_t e = hipMemcpyAsync(&n, dcount, 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(dcount);
        if (e != hipSuccess) return fail(-2, std::string("exact table (count): ") + hipGetErrorString(e));
        tab.windows = n;
        if (n <= MIRGE_EXACT_MAX_WINDOWS) {
            uint64_t ns = 1024;
            while (ns < 3 * n) ns <<= 1;  // load factor <= 1/3 (duplicates make it less): a miss ends at an empty slot after ~1.2 probes
            e = hipMalloc((void**)&tab.slots, ns * 8);
            if (e == hipSuccess) e = hipMemsetAsync(tab.slots, 0, ns * 8, c->stream);
            if (e == hipSuccess) {
                tab.mask = (uint32_t)(ns - 1);
                hipLaunchKernelGGL(k_exact_table<true>, dim3(grid), dim3(MIRGE_BLOCK), 0, c->stream, L->dT, L->dinv, L->h.total, lmask,
                                   tab.slots, tab.mask, (unsigned long long*)nullptr);
                e = hipStreamSynchronize(c->stream);
                if (e == hipSuccess) e = hipGetLastError();
            }
            if (e != hipSuccess) {
                (void)hipFree(tab.slots);
                return fail(e == hipErrorOutOfMemory ? -3 : -2, std::string("exact table: ") + hipGetErrorString(e));
            }
            L->device_bytes += ns * 8;
        }
    }
    L->exact[lmask] = tab;
    *out = tab;
    return 0;
}

extern "C" int mirge_lib_prepare(mirge_lib* L, int32_t k) {
    if (!L) return fail(-1, "lib is NULL");
    std::vector<ShapeJob> w(1);
    w[0].k1 = k;
    return lib_prepare_shapes(L, w);
}
