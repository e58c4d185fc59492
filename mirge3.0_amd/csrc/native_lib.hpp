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
    if (!c || !out || !T || !inv || !ref_start || n_refs < 0 || total >= 0xFFFFFFF0ull || kmax < 8 || kmax > MIRGE_KMAX ||
        n_T != (int64_t)((total + 31) / 32) + 8 || n_inv != (int64_t)((total + 63) / 64) + 4 || ref_start[n_refs] != (uint32_t)total)
        return fail(-1, "mirge_lib_create_packed: bad argument (a cache of another layout?)");
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
    uint64_t* dentry = nullptr;
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
    if (e == hipSuccess && !dbits) {  // a table too large for a bitmap: self-contained entries (MirgeKTable)
        e = hipMalloc((void**)&dentry, nb * 8);
        if (e == hipSuccess)
            hipLaunchKernelGGL(k_table_entries, dim3(grid_for(c, nb)), dim3(MIRGE_BLOCK), 0, c->stream, A, dpos, nb, dentry);
    }
    // table complete; no kernel may be reading the registry while it changes
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    (void)hipFree(tmp);
    if (e != hipSuccess) {
        (void)hipFree(A); (void)hipFree(dpos); (void)hipFree(dbits); (void)hipFree(dentry);
        return fail(-2, std::string("probe table construction: ") + hipGetErrorString(e));
    }
    if (dentry) { (void)hipFree(A); A = nullptr; }  // the CSR bounds were only the way to the entries
    L->device_bytes += (dentry ? nb * 8 : (nb + 2) * 4) + (size_t)npos * 4 + (dbits ? (size_t)((nb + 31) / 32) * 4 : 0);
    L->htables[sid].bucket = dentry ? (const void*)dentry : (const void*)A;
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
