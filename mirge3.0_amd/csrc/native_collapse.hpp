// native_collapse.hpp -- part of mirge_native.hip (one translation unit): collapse: partitioned key path, general path, one host synchronisation.
#pragma once
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
    // partitioned key path: level-1 regions rec1[(b1 * G + g) * cap1 + i], level-2 slabs rec2[(b1 * W2 + w) * slab + off2[..] + i]
    bool partitioned = false;
    uint4 *rec1 = nullptr, *rec2 = nullptr;
    uint32_t *cnt1 = nullptr, *cnt2 = nullptr, *off2 = nullptr, *hist = nullptr;
    // k_part_dedup's sharded output (attempt 0): cursors, capacity of a shard's stretch, the staging arrays k_part_compact reads
    uint32_t *shard_cur = nullptr, shard_cap = 0, *s_cnt = nullptr, *s_first = nullptr, rec_thresh = 0;
    uint64_t* s_seq = nullptr;
    uint8_t* s_len = nullptr;
    uint32_t G = 0, B = 0, NB1 = 0, NB2 = 1, W2 = 1, RPW = 0, cap1 = 0, shift2 = 0, cap = MIRGE_PART_CAP;
    uint64_t slab = 0;
};
// dmeta: [0..9] U of each group, [10] partition overflow flag, [12..13] bases of the unique reads of the long class (64 bits),
// [16 .. 16+255] length histogram of the unique reads of the other classes
#ifndef MIRGE_PART_CACHE
#define MIRGE_PART_CACHE 4096  // slots of the chunk cache of k_part_agg (2048: 3 % more records for split and dedup on the default sample)
#endif
#ifndef MIRGE_PART_RETRY_BYTES
#define MIRGE_PART_RETRY_BYTES (64ull << 30)  // memory the second partitioned attempt may take (2 KiB per read)
#endif
#define MIRGE_META_OVERFLOW 10
#define MIRGE_META_LONG_BASES 12
#define MIRGE_META_HIST 16
// [MIRGE_META_RECORDS] records k_part_agg emitted -- on a 256-byte line of its own: every k_part_dedup workgroup reads it while the
// buckets' returning adds hammer the cursor in word 0..9; on the cursor's line those reads queued behind the adds (k_part_dedup
// 0.117 -> 0.175 ms on the default draw, measured)
#define MIRGE_META_RECORDS (((MIRGE_META_HIST + MIRGE_MAX_READ_LEN + 1 + 63) & ~63) + 64)
#define MIRGE_META_WORDS (MIRGE_META_RECORDS + 64)

static const char* group_tag(int gi) {
    static const char* t[MIRGE_NGROUPS] = {".w1", ".w2", ".w4", ".w8", ".wl", ".w1n", ".w2n", ".w4n", ".w8n", ".wln"};
    return t[gi];
}

// capacity of a level-1 region for `mean` expected records: a quarter above the mean for reads the chunk cache
// missed, 8 sigma of a uniform hash, slack; a multiple of 4 records
static uint32_t part_region_cap(double mean) {
    const double c = 1.25 * mean + 8.0 * std::sqrt(mean) + 64.0;
    return ((uint32_t)c + 3u) & ~3u;
}

// the unique reads' length histogram of a group that took the partitioned path: k_len_hist on the side stream, in front of the
// copy that takes counts and histogram to the host (k_part_dedup no longer counts it; see the kernel)
static void part_len_hist(mirge_ctx* c, const ReadGroup& out, const CollapseTmp& t, uint32_t* dmeta, int gi) {
    if (!t.partitioned || !out.len) return;
    hipStream_t const was = c->cur;
    c->cur = c->aux;  // (LaunchScope records its events on c->cur: the kernel's stream, not the caller's)
    {
        LaunchScope ls(c, "k_len_hist", 0.0);
        hipLaunchKernelGGL(k_len_hist, dim3((unsigned)c->n_cu * 2), dim3(256), 0, c->aux, (const uint8_t*)out.len, (const uint32_t*)(dmeta + gi),
                           dmeta + MIRGE_META_HIST);
    }
    c->cur = was;
}

// partitioned key path after k_part_agg.  `part` 1: the second radix level (k_part_split is one workgroup per CU like
// k_part_agg: the small groups' kernels, enqueued while these run, find room beside them); 2: the per-bucket
// de-duplication (8192 workgroups: takes the machine); 0: both.  The main queue used to idle ~0.1 ms behind
// k_part_agg while the host enqueued the small groups' launches (profiles/r02_timeline.txt).
static int collapse_part_rest(mirge_ctx* c, int gi, const ReadGroup& in, ReadGroup& out, CollapseTmp& t, uint32_t* dmeta, int part = 0) {
    const bool two = t.NB2 > 1;
    if (part != 2 && two) {
        LaunchScope ls(c, "k_part_split.w1", in.n);
        hipLaunchKernelGGL(k_part_split, dim3(t.NB1 * t.W2), dim3(MIRGE_PART_THREADS), (MIRGE_PART_MAXREG + t.NB2) * 4, c->cur,
                           t.rec1, t.cnt1, t.hist, t.G, t.B, t.cap1, t.W2, t.RPW, t.shift2, t.NB2, t.slab, t.rec2, t.off2, t.cnt2);
    }
    if (part != 1) {
        LaunchScope ls(c, "k_part_dedup.w1", in.n);
        const uint4* rec = two ? t.rec2 : t.rec1;
        const uint32_t* cnt = two ? t.cnt2 : t.cnt1;
        const uint32_t* off = two ? t.off2 : nullptr;
        const uint32_t R = two ? t.W2 : t.G;
        const uint64_t rcap = two ? t.slab : (uint64_t)t.cap1;
        // (round 6) with the staging arrays there (t.shard_cur) the kernel itself chooses, from the sample's record count, between
        // eight cursors into them -- k_part_compact then makes the output dense -- and the one cursor into the output arrays
        const bool sh = t.shard_cur != nullptr;
        uint64_t* const oseq = reinterpret_cast<uint64_t*>(out.seq);
        uint8_t* const olen = reinterpret_cast<uint8_t*>(out.len);
        if (t.cap == 2048)
            hipLaunchKernelGGL(k_part_dedup<2048>, dim3(t.B), dim3(MIRGE_DEDUP_THREADS), 2048 * 16 + 4096, c->cur, rec, cnt, off, R, rcap, t.NB2,
                               oseq, olen, out.counts, out.first, dmeta + gi, (uint32_t*)nullptr, dmeta + MIRGE_META_OVERFLOW, t.shard_cur, t.shard_cap,
                               (const uint32_t*)(dmeta + MIRGE_META_RECORDS), t.rec_thresh, t.s_seq, t.s_len, t.s_cnt, t.s_first);
        else
            hipLaunchKernelGGL(k_part_dedup<MIRGE_PART_CAP>, dim3(t.B), dim3(MIRGE_DEDUP_THREADS), MIRGE_PART_CAP * 16 + 4096, c->cur, rec,
                               cnt, off, R, rcap, t.NB2, oseq, olen, out.counts, out.first, dmeta + gi, (uint32_t*)nullptr,
                               dmeta + MIRGE_META_OVERFLOW, t.shard_cur, t.shard_cap, (const uint32_t*)(dmeta + MIRGE_META_RECORDS), t.rec_thresh,
                               t.s_seq, t.s_len, t.s_cnt, t.s_first);
        if (sh) {
            LaunchScope ls2(c, "k_part_compact.w1", in.n);
            hipLaunchKernelGGL(k_part_compact, dim3((unsigned)c->n_cu * 8), dim3(256), 0, c->cur, (const uint32_t*)t.shard_cur, t.shard_cap,
                               (const uint64_t*)t.s_seq, (const uint8_t*)t.s_len, (const uint32_t*)t.s_cnt, (const uint32_t*)t.s_first,
                               oseq, olen, out.counts, out.first, dmeta + gi, (const uint32_t*)(dmeta + MIRGE_META_RECORDS), t.rec_thresh);
        }
    }
    return 0;
}

template <int W>
static int collapse_phase_a(mirge_ctx* c, int gi, const ReadGroup& in, ReadGroup& out, CollapseTmp& t,
                            const int32_t* dsample, int32_t S, uint32_t* dmeta, int attempt, int stage = 0,
                            const uint32_t* dweight = nullptr) {
    // attempt 0: partitioned key path, level-1 regions sized for a uniform hash; 1: the same with regions as large as a
    // writer's chunk (nothing can overflow them: a burst of one sequence that arrives after the chunk cache has filled
    // then costs memory -- 1 KiB per read -- not the global-atomic path); 2: global-atomic tables
    const bool force_atomic = attempt >= 2 || (attempt == 1 && (uint64_t)in.n * 2048ull > MIRGE_PART_RETRY_BYTES);
    // stage 0 = everything; 1 = only the first kernel of the partitioned path; 2 = what stage 1 left
    if (!in.n) return 0;
    if (stage == 2 && t.partitioned) return collapse_part_rest(c, gi, in, out, t, dmeta, 2);
    // key path: <=31 nt, no ambiguous call, one sample -> the slot holds the 64-bit key itself
    const bool key_path = (W == 1) && !in.nmask && S == 1 && !dweight;
    // slots of the open-addressing table: next power of two above 1.5 n (key path) / 2 n, computed in 64 bits --
    // a read set near the 2^32 limit of mirge_reads_pack would wrap a 32-bit size to 0 and never terminate
    const uint64_t need = key_path ? (uint64_t)in.n + in.n / 2 : 2ull * in.n;
    uint64_t tsize64 = 1024;
    while (tsize64 < need) tsize64 <<= 1;
    if (tsize64 > (1ull << 31)) return fail(-5, "collapse: a read group of " + std::to_string(in.n) + " reads needs a hash table beyond 2^31 slots");
    const uint32_t tsize = (uint32_t)tsize64;
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
        // (a bucket of up to MIRGE_PART_SMALL reads + 8 sigma stays below the 2048 - 64 distinct keys that table takes)
        t.cap = (!small_part && (uint64_t)B * MIRGE_PART_SMALL >= in.n) ? 2048u : (uint32_t)MIRGE_PART_CAP;
        // level-1 bins: MIRGE_PART_B1 (64) up to 8192 final buckets (~17 M reads), 128 at 16384, 256 at 32768 -- a splitter then
        // appends to 128 final buckets at most.  Measured (round 5, interleaved): 20 M reads 2.337 -> 2.304 ms per step, 50 M reads
        // 6.68 -> 6.49; at 10 M reads 128 bins were 1.7 % slower than 64 (k_part_agg pays for every level-1 stream it keeps open).
        // MIRGE_PART_NB1 overrides at run time (sweeps: a power of two up to 256)
        static const uint32_t nb1_env = std::getenv("MIRGE_PART_NB1") ? (uint32_t)std::max(1, std::min(256, std::atoi(std::getenv("MIRGE_PART_NB1")))) : 0u;
        const uint32_t nb1_cap = nb1_env ? nb1_env : B >= 32768 ? 256u : B >= 16384 ? 128u : (uint32_t)MIRGE_PART_B1;
        const uint32_t NB1 = std::min<uint32_t>(B, nb1_cap), NB2 = B / NB1;
        const uint32_t CS = B > 16384 ? 1024 : MIRGE_PART_CACHE;  // chunk-level LDS cache slots (16 B each)
        const int agg_lds = (int)(CS * 16 + NB1 * 4 + (NB2 > 1 ? B * 4 : 0) + 64);
        // dynamic-LDS ceilings, raised once per process and device to the largest configuration (B = 32768)
        static std::mutex attr_mu;
        static std::vector<int> attr_done;
        {
            std::lock_guard<std::mutex> lk(attr_mu);
            if (std::find(attr_done.begin(), attr_done.end(), c->device) == attr_done.end()) {
                HIPOK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_agg), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          1024 * 16 + 256 * 4 + 32768 * 4 + 64));  // CS = 1024 at B = 32768; 2048 * 16 + 16384 * 4 is smaller
                HIPOK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_dedup<MIRGE_PART_CAP>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, MIRGE_PART_CAP * 16 + 4096));
                attr_done.push_back(c->device);
            }
        }
        int lg = 0; while ((1u << lg) < B) lg++;
        int lg1 = 0; while ((1u << lg1) < NB1) lg1++;
        const uint32_t G = std::min<uint32_t>(256, (in.n + 2047) / 2048);
        uint32_t chunk = (in.n + G - 1) / G;
        chunk = (chunk + MIRGE_BLOCK - 1) / MIRGE_BLOCK * MIRGE_BLOCK;
        // the level-1 bin is the top lg1 bits of the hash, the level-2 bin the next lg - lg1 bits
        const uint32_t shift1 = 64 - lg1, shift2 = 64 - lg;
        // (splitter workgroups: MIRGE_PART_W2 per level-1 bin at 64 bins = one per CU; the same 256 in all at other bin counts)
        const uint32_t w2_want = std::max<uint32_t>(1, (uint32_t)MIRGE_PART_W2 * (uint32_t)MIRGE_PART_B1 / NB1);
        const uint32_t W2 = NB2 > 1 ? std::min<uint32_t>(w2_want, G) : 1, RPW = (G + W2 - 1) / W2;
        // test hook: MIRGE_TEST_SMALL_REGION=1 halves the level-1 regions so that they overflow and the call falls back
        static const bool small_region = std::getenv("MIRGE_TEST_SMALL_REGION") != nullptr;
        const uint32_t cap1 = attempt == 1 ? chunk
                                           : small_region ? std::max<uint32_t>(4, chunk / NB1 / 2 & ~3u) : part_region_cap((double)chunk / NB1);
        const uint64_t slab = (uint64_t)RPW * cap1;  // a splitter's input can never exceed the capacity of its regions
        // test hook: MIRGE_TEST_PART_OOM=1 makes the roomy second attempt fail like an exhausted device would
        static const bool part_oom = std::getenv("MIRGE_TEST_PART_OOM") != nullptr;
        if (part_oom && attempt == 1) return fail(-3, "hipMalloc: out of memory (MIRGE_TEST_PART_OOM)");
        CHECK(dalloc(c, &t.rec1, (size_t)NB1 * G * cap1));
        CHECK(dalloc(c, &t.cnt1, (size_t)NB1 * G));
        // sharded output of k_part_dedup (the first attempt only): a shard's stretch holds an eighth of the reads plus an eighth of
        // that and 64 k -- a hash that fills it sets the overflow flag.  Measured (round 5, interleaved): a sample with 3 % unique reads
        // 0.529 -> 0.488 ms per step (the kernel 0.103 -> 0.066 + 0.008 for k_part_compact: it was waiting for the one cursor), the
        // default draw with 42 % unique reads 1.184 -> 1.190 (0.116 -> 0.123 + 0.028: there the kernel's time is its records).  So:
        // when the context's previous collapse found under a fifth of its reads unique -- the samples of a batch are alike; the
        // first one takes the cursor.  MIRGE_DEDUP_SHARDED=1 / 0: always / never (tests, A/B).
        // Round 6: the choice is made ON THE DEVICE, per sample, from the record count k_part_agg leaves behind the overflow flag --
        // a sample whose chunks merge into few records (Zipf over 1.25 M templates: 0.33 records per raw read) has few unique reads,
        // the default draw (0.61) does not; the staging arrays are there either way (pool blocks, 19 B per raw read).  Until round 5
        // the host chose from the context's PREVIOUS sample: a batch's first sample, or alternating kinds, got the other mode.
        // MIRGE_DEDUP_SHARDED=1 / 0: always / never (tests, A/B); MIRGE_DEDUP_REC_SHARE: the threshold (records per raw read).
        static const int sharded_env = std::getenv("MIRGE_DEDUP_SHARDED") ? std::atoi(std::getenv("MIRGE_DEDUP_SHARDED")) : -1;
        static const double rec_share = std::getenv("MIRGE_DEDUP_REC_SHARE") ? std::atof(std::getenv("MIRGE_DEDUP_REC_SHARE")) : 0.45;
        const bool sharded = sharded_env != 0;
        t.rec_thresh = sharded_env > 0 ? 0xFFFFFFFFu : (uint32_t)std::min<double>(4.0e9, rec_share * (double)in.n);
        if (sharded && attempt == 0 && !small_part) {
            t.shard_cap = in.n / MIRGE_DEDUP_SHARDS + in.n / (8 * MIRGE_DEDUP_SHARDS) + 65536u;
            const size_t sn = (size_t)t.shard_cap * MIRGE_DEDUP_SHARDS;
            CHECK(dalloc(c, &t.shard_cur, (size_t)MIRGE_DEDUP_SHARDS * MIRGE_DEDUP_SHARD_STRIDE));
            CHECK(dalloc(c, &t.s_seq, sn));
            CHECK(dalloc(c, &t.s_len, sn));
            CHECK(dalloc(c, &t.s_cnt, sn));
            CHECK(dalloc(c, &t.s_first, sn));
            // (the eight cursors are zeroed by k_part_agg's first workgroup: a memset in front of it was a launch of its own, 2 us of
            //  kernel behind 17 us of queue gap at the head of every step)
        }
        if (NB2 > 1) {
            CHECK(dalloc(c, &t.hist, (size_t)G * B));
            CHECK(dalloc(c, &t.rec2, (size_t)NB1 * W2 * slab));
            CHECK(dalloc(c, &t.cnt2, (size_t)B * W2));
            CHECK(dalloc(c, &t.off2, (size_t)B * W2));
        }
        // outputs at capacity n (U is not known yet): the bucket workgroups emit the unique reads themselves
        out.W = 1;
        CHECK(dalloc(c, &out.seq, (size_t)in.n));
        CHECK(dalloc(c, &out.len, (size_t)in.n));
        CHECK(dalloc(c, &out.counts, (size_t)in.n));
        CHECK(dalloc(c, &out.first, (size_t)in.n));
        GroupView<1> v1 = view_of<1>(in);
        {
            LaunchScope ls(c, "k_part_agg.w1", in.n);
            hipLaunchKernelGGL(k_part_agg, dim3(G), dim3(MIRGE_PART_THREADS), agg_lds, c->cur, v1, in.orig, in.base, chunk, shift1, NB1,
                               shift2, B, CS, cap1, t.rec1, t.cnt1, t.hist, dmeta + MIRGE_META_OVERFLOW, dmeta + MIRGE_META_RECORDS, t.shard_cur);
        }
        t.G = G; t.B = B; t.NB1 = NB1; t.NB2 = NB2; t.W2 = W2; t.RPW = RPW; t.cap1 = cap1; t.slab = slab; t.shift2 = shift2;
        if (stage == 1) return collapse_part_rest(c, gi, in, out, t, dmeta, 1);
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
        hipLaunchKernelGGL(k_collapse_init, dim3(grid_for(c, (size_t)tsize * S)), dim3(MIRGE_BLOCK), 0, c->cur, t.rep, t.firstj, t.cnt,
                           tsize, (uint64_t)tsize * (uint64_t)S);
        std::snprintf(name, sizeof(name), "k_collapse_insert%s", group_tag(gi));
        LaunchScope ls(c, name, in.n);
        // at most 2 workgroups per CU: each sees enough of the group for its LDS cell cache to merge hot reads
        // (round 5: up to 8 per CU for a group of under 2 M reads.  Beside the bulk group's k_part_agg / k_part_split a memory
        // access of this kernel takes ~10 us; with 2 workgroups per CU every thread walked four reads one after the other,
        // 166 us for the 0.5 M reads of a sample's 32-64-nt group -- the head of the small groups' path, which ends the step.
        // A large group keeps 2: its hot sequences merge in fewer, fuller caches.)
        static const int ins_per_cu = std::getenv("MIRGE_INSERT_WG_PER_CU") ? std::max(1, std::atoi(std::getenv("MIRGE_INSERT_WG_PER_CU"))) : 0;
        const int ins_grid = std::min(grid_for(c, in.n), c->n_cu * (ins_per_cu ? ins_per_cu : (in.n < (2u << 20) ? 8 : 2)));
        // the first reads go in ahead of the rest, by a few workgroups: a sequence that makes up a percent of the group is
        // among them, so when the whole chip arrives its slot is already claimed and found by a plain load -- otherwise
        // every copy in the first wave of threads sees the slot empty and they all compare-and-swap ONE address
        const uint32_t seed = std::min<uint32_t>(in.n, MIRGE_COLLAPSE_SEED);
        hipLaunchKernelGGL(k_collapse_insert<W>, dim3((seed + MIRGE_BLOCK - 1) / MIRGE_BLOCK), dim3(MIRGE_BLOCK), 0, c->cur,
                           v, t.rep, t.firstj, t.cnt, t.slot_of, tsize - 1, dsample, in.orig, in.base, S, 0u, seed, dweight);
        if (in.n > seed)
            hipLaunchKernelGGL(k_collapse_insert<W>, dim3(ins_grid), dim3(MIRGE_BLOCK), 0, c->cur,
                               v, t.rep, t.firstj, t.cnt, t.slot_of, tsize - 1, dsample, in.orig, in.base, S, seed, in.n, dweight);
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

// the long class (reads beyond 255 nt, kernels_long.hpp): the general path's steps with kernels that take any length
static int collapse_phase_a_long(mirge_ctx* c, int gi, const ReadGroup& in, CollapseTmp& t, const int32_t* dsample, int32_t S,
                                 uint32_t* dmeta, int stage, const uint32_t* dweight) {
    if (!in.n || stage == 1) return 0;
    uint64_t tsize64 = 1024;
    while (tsize64 < 2ull * in.n) tsize64 <<= 1;
    if (tsize64 > (1ull << 31)) return fail(-5, "collapse: a read group of " + std::to_string(in.n) + " reads needs a hash table beyond 2^31 slots");
    const uint32_t tsize = (uint32_t)tsize64;
    const uint32_t per_block = MIRGE_BLOCK * MIRGE_SCAN_ITEMS;
    t.nb = (in.n + per_block - 1) / per_block;
    CHECK(dalloc(c, &t.slot_of, in.n));
    CHECK(dalloc(c, &t.flag, (size_t)t.nb * per_block));
    CHECK(dalloc(c, &t.blocksum, t.nb));
    CHECK(dalloc(c, &t.rep, tsize));
    CHECK(dalloc(c, &t.firstj, tsize));
    CHECK(dalloc(c, &t.cnt, (size_t)tsize * S));
    hipLaunchKernelGGL(k_collapse_init, dim3(grid_for(c, (size_t)tsize * S)), dim3(MIRGE_BLOCK), 0, c->cur, t.rep, t.firstj, t.cnt,
                       tsize, (uint64_t)tsize * (uint64_t)S);
    const LongView v = long_view_of(in);
    char name[48];
    {
        std::snprintf(name, sizeof(name), "k_collapse_insert%s", group_tag(gi));
        LaunchScope ls(c, name, in.n);
        hipLaunchKernelGGL(k_collapse_insert_long, dim3(grid_for(c, in.n)), dim3(MIRGE_BLOCK), 0, c->cur, v, t.rep, t.firstj, t.cnt, t.slot_of,
                           tsize - 1, dsample, in.orig, in.base, S, dweight);
    }
    t.cnt_base = t.cnt; t.cnt_stride = (uint32_t)S;
    {
        std::snprintf(name, sizeof(name), "k_heads_blocksum%s", group_tag(gi));
        LaunchScope ls(c, name, in.n);
        // (no length histogram: it stops at MIRGE_MAX_READ_LEN; the heads' bases are summed instead)
        hipLaunchKernelGGL(k_heads_blocksum, dim3(t.nb), dim3(MIRGE_BLOCK), 0, c->cur, t.slot_of, (const uint32_t*)t.firstj, 1u, 0u, in.n,
                           (const uint8_t*)nullptr, t.flag, t.blocksum, dmeta + MIRGE_META_HIST);
        hipLaunchKernelGGL(k_long_heads_bases, dim3(grid_for(c, in.n)), dim3(MIRGE_BLOCK), 0, c->cur, t.slot_of, (const uint32_t*)t.firstj, v.len, in.n,
                           reinterpret_cast<unsigned long long*>(dmeta + MIRGE_META_LONG_BASES));
    }
    {
        LaunchScope ls(c, "k_scan_blocksums", t.nb);
        hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(MIRGE_BLOCK), 0, c->cur, t.blocksum, t.nb, dmeta + gi);
    }
    return 0;
}

static int collapse_phase_b_long(mirge_ctx* c, int gi, const ReadGroup& in, ReadGroup& out, CollapseTmp& t, int32_t S, uint32_t U,
                                 uint32_t out_base, const uint32_t* dmeta) {
    out.W = in.W; out.n = U; out.base = out_base;
    if (!in.n) return 0;
    CHECK(dalloc(c, &out.seq, (size_t)out.W * U));
    CHECK(dalloc(c, &out.len, (size_t)U * 2));
    if (in.nmask) CHECK(dalloc(c, &out.nmask, (size_t)out.W * U));
    CHECK(dalloc(c, &out.counts, (size_t)U * S));
    CHECK(dalloc(c, &out.first, (size_t)U));
    char name[48];
    std::snprintf(name, sizeof(name), "k_collapse_scatter%s", group_tag(gi));
    LaunchScope ls(c, name, in.n);
    hipLaunchKernelGGL(k_collapse_scatter_long, dim3(t.nb), dim3(MIRGE_BLOCK), 0, c->cur, long_view_of(in), t.slot_of, t.flag, t.cnt_base,
                       t.cnt_stride, t.blocksum, dmeta + gi, in.orig, in.base, S, out.seq, reinterpret_cast<uint16_t*>(out.len), out.nmask,
                       out.counts, out.first);
    return 0;
}

static void collapse_tmp_release(mirge_ctx* c, CollapseTmp& t) {
    c->defer(t.rep); c->defer(t.firstj); c->defer(t.cnt); c->defer(t.slots); c->defer(t.slot_of);
    c->defer(t.flag); c->defer(t.blocksum);
    c->defer(t.rec1); c->defer(t.cnt1); c->defer(t.rec2); c->defer(t.cnt2); c->defer(t.off2); c->defer(t.hist);
    c->defer(t.shard_cur); c->defer(t.s_seq); c->defer(t.s_len); c->defer(t.s_cnt); c->defer(t.s_first);
    t = CollapseTmp();
}

// Work a caller wants on the GPU BEHIND the collapse kernels but before the host has read the unique counts back
// (mirge_collapse_cascade: the bulk group's cascade).  pre_sync runs after the count read-back has been enqueued;
// discard must undo it when the partitioned attempt overflowed and everything is redone.
struct CollapseHook {
    std::function<int(mirge_reads* partial, const CollapseTmp* tmp, uint32_t* dmeta, int big)> pre_sync;
    // (round 5) the small groups' unique reads exist (their counts came back ahead of the bulk group's, their scatter kernels are
    // queued on `aux`): whatever the caller wants behind them -- their cascades -- while the bulk group's collapse still runs
    std::function<int(mirge_reads* partial, int big)> small_ready;
    std::function<void()> discard;
    bool ran = false;         // pre_sync was called and its work stands
    bool small_ran = false;   // small_ready was called and its work stands (the caller's side streams are left unjoined)
    uint32_t* dmeta = nullptr;  // handed over: the hook's kernels read the counts from it
};

static int collapse_impl(mirge_ctx* c, const mirge_reads* raw, const int32_t* sample_ids, int32_t S, mirge_reads** uniq,
                         int64_t* n_uniq, CollapseHook* hook, const uint32_t* weights = nullptr, int32_t* dsample_in = nullptr,
                         uint32_t* dweight_in = nullptr);

// How many of the extra streams the small groups are dealt over (MIRGE_XAUX_SLOTS, default 2).  The runtime maps this context's streams
// onto four hardware queues: a third extra stream shares a queue with the second, and two streams taking turns on one queue pay a
// dependency packet per turn (~12 us, profiles/r06_timeline_zipf.txt).  Two: 0.382 vs 0.409 ms on the sample with few unique reads,
// 1.197 vs 1.195 ms on the default draw (profiles/r06_ab_xaux_slots.txt; largest group alone on a stream, the others chained: the same).
static int xaux_slots() {
    static const int v = std::getenv("MIRGE_XAUX_SLOTS") ? std::max(1, std::min(MIRGE_N_XAUX, std::atoi(std::getenv("MIRGE_XAUX_SLOTS")))) : 2;
    return v;
}
static int xaux_slot_of(int k) { return k % xaux_slots(); }  // the k-th small group by size, 0 = the largest
static uint32_t small_fused_max() {  // (cascade_launch_groups: the largest group that takes the one-launch cascade)
    static const uint32_t v = std::getenv("MIRGE_FUSED_MAX") ? (uint32_t)std::strtoul(std::getenv("MIRGE_FUSED_MAX"), nullptr, 10) : (1u << 19);
    return v;
}
// Which extra stream the one-launch cascade of every small read group takes (cascade_launch_groups assigns them this way): the groups
// other than `big` by size, largest first; one slot each for those that take the one-launch route, when there is more than one; -1:
// the second stream.  n[gi] = reads of group gi.
static void small_group_slots(const uint32_t* n, int big, int* slot) {
    const uint32_t fused_max = small_fused_max();
    static const bool xaux_on = !(std::getenv("MIRGE_XAUX") && std::atoi(std::getenv("MIRGE_XAUX")) == 0);
    int order[MIRGE_NGROUPS], no = 0;
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) { slot[gi] = -1; if (gi != big) order[no++] = gi; }
    std::stable_sort(order, order + no, [&](int a, int b) { return n[a] > n[b]; });
    int n_small = 0;
    for (int k = 0; k < no; k++) if (n[order[k]] && n[order[k]] <= fused_max) n_small++;
    if (!xaux_on || n_small <= 1) return;
    int next = 0;
    for (int k = 0; k < no; k++) {
        const int gi = order[k];
        if (is_long_group(gi) || !n[gi] || n[gi] > fused_max) continue;
        slot[gi] = xaux_slot_of(next++);
    }
}

extern "C" int mirge_collapse(mirge_ctx* c, const mirge_reads* raw, const int32_t* sample_ids, int32_t S,
                              mirge_reads** uniq, int64_t* n_uniq) {
    return collapse_impl(c, raw, sample_ids, S, uniq, n_uniq, nullptr);
}
// every raw read stands for weights[i] copies: the merge of several already collapsed dictionaries -- the per-sample
// results of a sharded run -- into the sample matrix (digest.py:243) without expanding them again
extern "C" int mirge_collapse_weighted(mirge_ctx* c, const mirge_reads* raw, const int32_t* sample_ids, int32_t S,
                                       const uint32_t* weights, mirge_reads** uniq, int64_t* n_uniq) {
    return collapse_impl(c, raw, sample_ids, S, uniq, n_uniq, nullptr, weights);
}

// dsample_in / dweight_in (device, [raw->n], pool blocks this call releases): sample ids and weights that are on the device already
// (mirge_collapse_merge) instead of the host arrays
static int collapse_impl(mirge_ctx* c, const mirge_reads* raw, const int32_t* sample_ids, int32_t S, mirge_reads** uniq,
                         int64_t* n_uniq, CollapseHook* hook, const uint32_t* weights, int32_t* dsample_in, uint32_t* dweight_in) {
    HostClock hc("collapse");
    if (!c || !raw || !uniq || S < 1 || (S > 1 && !sample_ids && !dsample_in)) return fail(-1, "mirge_collapse: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    int32_t* dsample = dsample_in;
    if (!dsample_in && sample_ids && raw->n) {
        for (int64_t i = 0; i < raw->n; i++)
            if (sample_ids[i] < 0 || sample_ids[i] >= S) return fail(-1, "sample id out of range");
        CHECK(dalloc(c, &dsample, (size_t)raw->n));
        HIPOK(hipMemcpyAsync(dsample, sample_ids, (size_t)raw->n * 4, hipMemcpyHostToDevice, c->stream));
    }
    uint32_t* dweight = dweight_in;
    if (!dweight_in && weights && raw->n) {
        CHECK(dalloc(c, &dweight, (size_t)raw->n));
        HIPOK(hipMemcpyAsync(dweight, weights, (size_t)raw->n * 4, hipMemcpyHostToDevice, c->stream));
    }
    auto R = std::make_unique<mirge_reads>();
    R->ctx = c; R->n_samples = S;
    uint32_t* dmeta = nullptr;
    CHECK(dalloc(c, &dmeta, MIRGE_META_WORDS));
    CollapseTmp tmp[MIRGE_NGROUPS];
    int rc = 0;
    const int big = largest_group(raw);
    // attempts 0 and 1 may use the partitioned LDS path (collapse_phase_a); if a level-1 region (attempt 0) or a bucket's
    // LDS table (pathological hash skew) overflows, everything is redone -- at last with the global-atomic tables
    bool small_done = false;  // the small groups' phase B ran ahead of the bulk group's count (their cascades may be queued too)
    for (int attempt = 0; attempt < 3 && rc == 0; attempt++) {
        hipError_t e0 = hipMemsetAsync(dmeta, 0, MIRGE_META_WORDS * 4, c->stream);
        if (e0 != hipSuccess) { rc = fail(-2, std::string("mirge_collapse: ") + hipGetErrorString(e0)); break; }
        rc = stream_fork(c);
        // Order of enqueue (profiles/r01_timeline.txt): the bulk group's first kernel (k_part_agg: one workgroup
        // per CU, ~0.2 ms) goes first, the small groups' ~15 short launches are enqueued while it runs and share
        // the CUs with it, then the bulk group's wide kernels.  Small groups entirely first left the GPU idle for
        // the ~0.2 ms their enqueue takes; entirely last, each of their kernels waits behind 2048-8192-workgroup
        // launches for CUs to drain and the join at the end waits for them (5.3 vs 3.8 ms).
        // MIRGE_SPREAD_SMALL=1 (round 5 experiment, OFF by default): the small groups' chains SIDE BY SIDE -- the largest on
        // `aux`, the others on extra streams of their own -- instead of one after the other on `aux`.  Measured worse: 1.29 vs
        // 1.21 ms per C3 step (profiles/r05_ab_spread_small.txt): more kernels in flight beside k_part_split / k_part_dedup slow
        // those, and the small groups' cascades end when the bulk kernel's workgroups retire whenever they start.
        static const bool spread_on = std::getenv("MIRGE_SPREAD_SMALL") && std::atoi(std::getenv("MIRGE_SPREAD_SMALL")) == 1;
        int small_stream[MIRGE_NGROUPS];  // -1: aux, k: xaux[k]
        int n_spread = 0;
        {
            int largest = -1;
            for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
                small_stream[gi] = -1;
                if (gi != big && raw->g[gi].n && (largest < 0 || raw->g[gi].n > raw->g[largest].n)) largest = gi;
            }
            if (spread_on && !c->overlap_mode && rc == 0)
                for (int gi = 0; gi < MIRGE_NGROUPS; gi++)
                    if (gi != big && gi != largest && raw->g[gi].n) small_stream[gi] = n_spread++ % MIRGE_N_XAUX;
            if (n_spread && rc == 0) rc = xaux_fork(c);
        }
        // MIRGE_DEDUP_FIRST=1 (round 6 experiment): the bulk group's k_part_dedup enqueued with its first two kernels, ahead of the small
        // groups' launches, instead of behind them
        static const bool dedup_first = std::getenv("MIRGE_DEDUP_FIRST") && std::atoi(std::getenv("MIRGE_DEDUP_FIRST")) == 1;
        for (int k = -1; k <= MIRGE_NGROUPS && rc == 0; k++) {
            const int gi = (k < 0 || k == MIRGE_NGROUPS) ? big : k;
            if (k >= 0 && k < MIRGE_NGROUPS && gi == big) continue;
            if (dedup_first && k == MIRGE_NGROUPS) continue;
            const int stage = k < 0 ? (dedup_first ? 0 : 1) : (k == MIRGE_NGROUPS ? 2 : 0);
            c->cur = gi == big ? c->stream : (small_stream[gi] >= 0 ? c->xaux[small_stream[gi]] : c->aux);
            if (is_long_group(gi)) rc = collapse_phase_a_long(c, gi, raw->g[gi], tmp[gi], dsample, S, dmeta, stage, dweight);
            else MIRGE_BY_WIDTH(gi, rc, collapse_phase_a<W>(c, gi, raw->g[gi], R->g[gi], tmp[gi], dsample, S, dmeta, attempt, stage, dweight));
            if (k < 0) hc.lap("first kernel of the bulk group enqueued");
        }
        if (n_spread && rc == 0) {  // `aux` behind the extra streams' chains: the counts it copies next are theirs too
            hipError_t e = hipSuccess;
            for (int k = 0; k < std::min(n_spread, MIRGE_N_XAUX) && e == hipSuccess; k++) {
                e = hipEventRecord(c->ev_xjoin[k], c->xaux[k]);
                if (e == hipSuccess) e = hipStreamWaitEvent(c->aux, c->ev_xjoin[k], 0);
            }
            if (e != hipSuccess) rc = fail(-2, std::string("mirge_collapse: ") + hipGetErrorString(e));
        }
        // (no join here: the second stream waits for the main one below and carries the read-back of the counts)
        c->cur = c->stream;
        hc.lap("enqueue A");
        if (rc == -3 && attempt < 2) {
            // the partitioned attempts take up to 2 KiB of HBM per read (attempt 1); the global-atomic tables ~20 B.  A device
            // that cannot give the former (a second context in flight, a fragmented pool) is no reason to fail the call.
            for (int k = 0; k < MIRGE_N_XAUX; k++) (void)hipStreamSynchronize(c->xaux[k]);
            (void)hipStreamSynchronize(c->aux); (void)hipStreamSynchronize(c->stream);
            for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
                collapse_tmp_release(c, tmp[gi]);
                ReadGroup& og = R->g[gi];
                c->release(og.seq); c->release(og.len); c->release(og.nmask); c->release(og.counts); c->release(og.first);
                og = ReadGroup();
            }
            c->flush_deferred();  // (alloc() hands cached blocks back to the driver by itself when a request fails)
            rc = 0; attempt = 1;  // -> the loop's increment makes it 2
            continue;
        }
        bool hooked = false;
        // Round 5: with a caller that queues the cascades itself (mirge_collapse_cascade) the host does not wait for ALL counts at
        // once.  The small groups' counts are final when `aux` has run their kernels (~0.3 ms into a 10 M-read step) while the
        // bulk group's come out of k_part_dedup (~0.37 ms) -- and the bulk group does not need the host at all (its cascade takes
        // the count from device memory).  So: the small groups' counts are copied by `aux` with nothing of the main stream in
        // front, the host sizes their outputs, queues their scatter kernels and hands them to the caller (their cascades); only
        // then the bulk group's count.  Measured NEUTRAL on the 10 M-read step (1.235 vs 1.238 ms, profiles/README.md round 5):
        // beside k_part_agg / k_part_split the small groups' own kernels end when k_part_dedup does, and their cascades end when
        // the bulk kernel's workgroups retire, whenever they start.  Kept: it cannot lose, and the host returns no later.
        static const bool early_on = !(std::getenv("MIRGE_EARLY_SMALL") && std::atoi(std::getenv("MIRGE_EARLY_SMALL")) == 0);
        if (rc == 0 && early_on && hook && hook->small_ready && attempt == 0 && tmp[big].partitioned && !c->overlap_mode) {
            uint32_t* const small = c->pinned + 512;  // (the page-locked block holds 1024 words; the full copy takes the first 272)
            hipError_t e = hipMemcpyAsync(small, dmeta, MIRGE_NGROUPS * 4, hipMemcpyDeviceToHost, c->aux);
            if (e == hipSuccess) e = hipEventRecord(c->ev_meta_small, c->aux);
            if (e == hipSuccess) e = hipEventRecord(c->ev_bulk_counted, c->stream);  // the point behind k_part_dedup
            if (e == hipSuccess) {
                rc = hook->pre_sync(R.get(), tmp, dmeta, big);
                hooked = rc == 0;
            }
            static const bool scatter_x = !(std::getenv("MIRGE_SCATTER_ON_XAUX") && std::atoi(std::getenv("MIRGE_SCATTER_ON_XAUX")) == 0);
            if (e == hipSuccess && hooked) e = hipEventSynchronize(c->ev_meta_small);
            if (e == hipSuccess && hooked) {
                // (round 6) a small group's scatter kernel goes on the extra stream its one-launch cascade will take (small_ready ->
                // cascade_launch_groups): the three scatters ran one after the other on the second stream (74 us on a sample with few
                // unique reads) and every cascade then started a cross-stream hop (~22 us) behind the LAST of them -- now each group's
                // scatter and cascade are neighbours on one stream and the groups run side by side.  MIRGE_SCATTER_ON_XAUX=0: as before.
                int xslot[MIRGE_NGROUPS];
                uint32_t n_small_u[MIRGE_NGROUPS];
                for (int gi = 0; gi < MIRGE_NGROUPS; gi++) n_small_u[gi] = gi == big ? 0u : small[gi];
                small_group_slots(n_small_u, big, xslot);
                bool forked = false;
                for (int gi = 0; gi < MIRGE_NGROUPS; gi++) forked |= scatter_x && xslot[gi] >= 0;
                // Not when a small group is large enough for the staged cascade (beyond MIRGE_FUSED_MAX unique reads: a 20 M-read sample's
                // 32-64-nt group): its k_cascade_bulk<2> is a grid of resident workgroups like the bulk group's own, and whichever of the
                // two is resident first keeps the other's remaining workgroups waiting until it retires.  With the scatter kernels one
                // after the other on `aux` the bulk group's kernel has ~55 us of head start and is resident first in 83 of 87 steps; with
                // them side by side in 55 of 79, and a step that loses the race takes 2.85 instead of 2.33 ms (20 M reads: 2.65 vs 2.26 ms
                // per step on average, profiles/r06_ab_c4_shape.txt).  MIRGE_SCATTER_ON_XAUX=2: side by side regardless.
                static const bool scatter_always = std::getenv("MIRGE_SCATTER_ON_XAUX") && std::atoi(std::getenv("MIRGE_SCATTER_ON_XAUX")) == 2;
                for (int gi = 0; gi < MIRGE_NGROUPS && !scatter_always; gi++)
                    if (gi != big && !is_long_group(gi) && small[gi] > small_fused_max()) forked = false;
                // The extra streams take over from `aux` here, once, for the scatter kernels AND the cascades (cascade_launch_groups skips
                // its own fork: four runtime calls less in front of the cascades).  Queued BEFORE the host's wait instead -- the calls off
                // the chain -- it was slower: 0.403 vs 0.378 ms on the sample with few unique reads, 1.235 vs 1.217 ms on the default draw
                // (profiles/r06_ab_fork_early.txt): two more queues holding an unsatisfied dependency packet beside k_part_dedup /
                // k_part_compact cost those kernels 15 % / 60 % of their time.
                if (forked) rc = xaux_fork(c);
                for (int gi = 0; gi < MIRGE_NGROUPS && rc == 0; gi++) {
                    if (gi == big) continue;
                    c->cur = (forked && xslot[gi] >= 0) ? c->xaux[xslot[gi]] : c->aux;
                    if (is_long_group(gi)) rc = collapse_phase_b_long(c, gi, raw->g[gi], R->g[gi], tmp[gi], S, small[gi], 0u, dmeta);
                    else MIRGE_BY_WIDTH(gi, rc, collapse_phase_b<W>(c, gi, raw->g[gi], R->g[gi], tmp[gi], S, small[gi], 0u, dmeta));
                }
                c->cur = c->stream;
                c->xaux_forked = forked;  // (cascade_launch_groups: the extra streams already stand behind `aux`, and where each group went)
                static_assert(MIRGE_NGROUPS <= 16, "mirge_ctx::small_slot");
                for (int gi = 0; gi < MIRGE_NGROUPS; gi++) c->small_slot[gi] = forked ? xslot[gi] : -1;
                if (rc == 0) { rc = hook->small_ready(R.get(), big); small_done = rc == 0; }
                c->xaux_forked = false;
            }
            // the bulk group's count, the overflow flag and the length histogram: `aux` behind the main stream's k_part_dedup
            if (e == hipSuccess) e = hipStreamWaitEvent(c->aux, c->ev_bulk_counted, 0);
            if (e == hipSuccess)
                for (int gi = 0; gi < MIRGE_NGROUPS; gi++) part_len_hist(c, R->g[gi], tmp[gi], dmeta, gi);  // (a group beside the bulk one ran on `aux` itself)
            if (e == hipSuccess) e = hipMemcpyAsync(c->pinned, dmeta, MIRGE_META_WORDS * 4, hipMemcpyDeviceToHost, c->aux);
            if (e == hipSuccess) e = hipEventRecord(c->ev_meta, c->aux);
            if (e == hipSuccess) e = hipEventSynchronize(c->ev_meta);
            if (e != hipSuccess && rc == 0) rc = fail(-2, std::string("mirge_collapse: ") + hipGetErrorString(e));
        } else
        if (rc == 0) {  // the one host synchronisation of the call: U sizes the outputs
            // The counts are copied by the SECOND stream, behind its own groups' kernels and an event of the main stream (the bulk
            // group's count comes from k_part_dedup): on the main stream the 4 us copy sat between k_part_dedup and the bulk
            // group's cascade -- which does not need it -- with 6 + 11 us of queue gaps around it
            rc = stream_fork(c);
            if (rc == 0)
                for (int gi = 0; gi < MIRGE_NGROUPS; gi++) part_len_hist(c, R->g[gi], tmp[gi], dmeta, gi);
            hipError_t e = rc == 0 ? hipMemcpyAsync(c->pinned, dmeta, MIRGE_META_WORDS * 4, hipMemcpyDeviceToHost, c->aux) : hipErrorUnknown;
            if (e == hipSuccess) e = hipEventRecord(c->ev_meta, c->aux);
            if (e == hipSuccess && hook && attempt == 0 && tmp[big].partitioned) {
                rc = hook->pre_sync(R.get(), tmp, dmeta, big);
                hooked = rc == 0;
            }
            if (e == hipSuccess) e = hipEventSynchronize(c->ev_meta);  // the counts, not whatever was queued behind them
            if (e != hipSuccess && rc == 0) rc = fail(-2, std::string("mirge_collapse: ") + hipGetErrorString(e));
        }
        if (hooked && (rc != 0 || c->pinned[MIRGE_META_OVERFLOW])) { hook->discard(); hooked = false; small_done = false; }
        if (hook) { hook->ran = hooked; hook->small_ran = small_done; }
        if (rc == 0 && c->pinned[MIRGE_META_OVERFLOW] && attempt < 2) {
            for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
                collapse_tmp_release(c, tmp[gi]);
                ReadGroup& og = R->g[gi];  // outputs the partitioned attempt had allocated at capacity n -- and, when the small
                // groups' phase B ran ahead of the bulk group's count (small_ready), theirs: the `N` groups' nmask among them
                c->release(og.seq); c->release(og.len); c->release(og.nmask); c->release(og.counts); c->release(og.first);
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
        R->total_bases += (int64_t)((uint64_t)c->pinned[MIRGE_META_LONG_BASES] | ((uint64_t)c->pinned[MIRGE_META_LONG_BASES + 1] << 32));
        R->long_max = raw->long_max;
        R->hist_valid = true;
        rc = stream_fork(c);
        for (int gi = 0; gi < MIRGE_NGROUPS && rc == 0; gi++) {
            c->cur = gi == big ? c->stream : c->aux;
            if (small_done && gi != big) R->g[gi].base = base;  // its unique reads were written ahead (above): only their place in the handle's order was open
            else if (is_long_group(gi)) rc = collapse_phase_b_long(c, gi, raw->g[gi], R->g[gi], tmp[gi], S, U[gi], base, dmeta);
            else MIRGE_BY_WIDTH(gi, rc, collapse_phase_b<W>(c, gi, raw->g[gi], R->g[gi], tmp[gi], S, U[gi], base, dmeta));
            base += R->g[gi].n;
        }
        c->cur = c->stream;
        // (with the small groups' cascades already on their streams the join is the caller's: mirge_count_join puts work in front of it)
        if (!small_done) { int jr = stream_join(c); if (rc == 0) rc = jr; }
        else c->aux_drained = rc == 0;  // (the host waited for ev_meta, the last thing queued on `aux`; the loop above put nothing there)
    }
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) collapse_tmp_release(c, tmp[gi]);
    c->flush_deferred();
    c->release(dsample); c->release(dweight);
    if (rc && hook && hook->ran) { hook->discard(); hook->ran = false; }
    if (hook && hook->ran) hook->dmeta = dmeta;  // still read by the work queued in pre_sync
    else c->release(dmeta);
    if (rc) { mirge_reads_destroy(R.release()); return rc; }
    R->n = base;
    *uniq = R.release();
    if (n_uniq) *n_uniq = base;
    hc.lap("phase B + release");
    return 0;
}

// The sample matrix of SEVERAL samples from their per-sample dictionaries, all on the device (round 6).  The reference joins the samples'
// dictionaries on their sequences (the outer join of digest.py:243); mirge_collapse with sample ids does the same from the RAW reads
// of all samples through the general (global-atomic) path -- 40 M raw reads of four samples: 18-37 ms, the hot sequences of real
// samples contending for their table cells -- while each sample alone takes the partitioned path in 0.3 ms.  So a run of several
// samples collapses every sample by itself and merges the dictionaries here: their entries appended (mirge_reads_concat), sample
// index and count of every entry as sample id and weight (k_merge_fill), one weighted collapse over S x U entries in which no
// sequence occurs more than S times.  parts[s] = the collapse result of sample s (one count column each); the parts stay valid.
extern "C" int mirge_collapse_merge(mirge_ctx* c, const mirge_reads* const* parts, int32_t n_parts, mirge_reads** uniq, int64_t* n_uniq) {
    if (!c || !parts || n_parts < 1 || !uniq) return fail(-1, "mirge_collapse_merge: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    for (int p = 0; p < n_parts; p++) {
        if (!parts[p] || parts[p]->ctx != c || parts[p]->n_samples != 1) return fail(-1, "mirge_collapse_merge: every part must be a one-sample collapse result of this context");
        for (int gi = 0; gi < MIRGE_NGROUPS; gi++)
            if (parts[p]->g[gi].n && (parts[p]->g[gi].orig || !parts[p]->g[gi].counts)) return fail(-1, "mirge_collapse_merge: a part is not a collapse result");
    }
    mirge_reads* all = nullptr;
    CHECK(reads_concat_impl(c, parts, n_parts, &all, true));
    int32_t* dsample = nullptr;
    uint32_t* dweight = nullptr;
    int rc = dalloc(c, &dsample, (size_t)std::max<int64_t>(all->n, 1));
    if (rc == 0) rc = dalloc(c, &dweight, (size_t)std::max<int64_t>(all->n, 1));
    if (rc) { c->release(dsample); mirge_reads_destroy(all); return rc; }
    uint32_t before = 0;
    for (int p = 0; p < n_parts; p++) {
        for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
            const ReadGroup& g = parts[p]->g[gi];
            if (g.n) hipLaunchKernelGGL(k_merge_fill, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, (const uint32_t*)g.counts, g.n,
                                        before + g.base, (int32_t)p, dsample, dweight);
        }
        before += (uint32_t)parts[p]->n;
    }
    rc = collapse_impl(c, all, nullptr, n_parts, uniq, n_uniq, nullptr, nullptr, dsample, dweight);  // (releases dsample / dweight)
    mirge_reads_destroy(all);
    return rc;
}

extern "C" int mirge_collapse_fetch(mirge_ctx* c, const mirge_reads* U, uint32_t* counts_out, int64_t* first_out) {
    if (!c || !U || !counts_out) return fail(-1, "mirge_collapse_fetch: bad argument");
    if (U->n_samples < 1) return fail(-1, "read set has no count matrix");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
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

// order_out[k] = handle index of the unique read that appeared k-th in the raw reads: the row order of the reference's
// per-sample dictionary (digest.py:158-163) without a host-side ranking of U first-indices -- one device radix sort of
// (first index, handle index) pairs over all groups.
extern "C" int mirge_collapse_order(mirge_ctx* c, const mirge_reads* U, int64_t* order_out) {
    if (!c || !U || (!order_out && U->n)) return fail(-1, "mirge_collapse_order: bad argument");
    HIPOK(hipSetDevice(c->device)); CHECK(join_pending_now(c));
    const size_t n = (size_t)U->n;
    if (!n) return 0;
    if (n >= 0x7FFFFFFFull) return fail(-5, "mirge_collapse_order: 2^31 unique reads or more (hipCUB's sort takes an int count)");
    uint32_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr;
    CHECK(dalloc(c, &keys, n)); CHECK(dalloc(c, &vals, n)); CHECK(dalloc(c, &keys2, n)); CHECK(dalloc(c, &vals2, n));
    for (int gi = 0; gi < MIRGE_NGROUPS; gi++) {
        const ReadGroup& g = U->g[gi];
        if (!g.n) continue;
        if (g.orig || !g.first) return fail(-1, "mirge_collapse_order: handle is not a collapse result");
        HIPOK(hipMemcpyAsync(keys + g.base, g.first, (size_t)g.n * 4, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_index_shift, dim3(grid_for(c, g.n)), dim3(MIRGE_BLOCK), 0, c->stream, (const uint32_t*)nullptr, 0u,
                           (uint32_t)g.n, g.base, vals + g.base);
    }
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    HIPOK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (int)n, 0, 32, c->stream));
    CHECK(dalloc(c, (uint8_t**)&tmp, std::max<size_t>(tmp_bytes, 16)));
    HIPOK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (int)n, 0, 32, c->stream));
    std::vector<uint32_t> h(n);
    HIPOK(hipMemcpyAsync(h.data(), vals2, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPOK(hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < n; k++) order_out[k] = (int64_t)h[k];
    c->release(keys); c->release(vals); c->release(keys2); c->release(vals2); c->release(tmp);
    return 0;
}

