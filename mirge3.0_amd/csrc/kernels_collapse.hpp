// kernels_collapse.hpp -- part of mirge_kernels.hpp: collapse kernels (general hash path, partitioned key path, heads / scan / scatter).
#pragma once
// ------------------------------------------------------------------------------------------
// collapse (digest.py:141-163): open-addressing hash table of representative read indices.
//   insert : slot claimed by atomicCAS on rep[]; a later equal read finds the slot by comparing
//            its words with the representative's (the raw arrays are read-only during the kernel)
//            and adds 1 to cnt[slot][sample]; firstj[slot] = min index (first appearance).
//   heads  : read j is the head of its group iff firstj[slot_of[j]] == j; block sums of heads.
//   scatter: exclusive scan of heads = rank in order of first appearance; heads copy their read
//            and their slot's count row to the output.
// ------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ bool same_read(const GroupView<W>& g, uint32_t a, const MirgeRead<W>& r) {
    if (g.len[a] != (uint8_t)r.len) return false;
    bool eq = true;
#pragma unroll
    for (int w = 0; w < W; w++) {
        eq &= g.seq[(size_t)w * g.n + a] == r.w[w];
        if (g.nmask) eq &= g.nmask[(size_t)w * g.n + a] == r.nm[w];
    }
    return eq;
}

template <int W>
__global__ void k_collapse_insert(GroupView<W> g, uint32_t* __restrict__ rep, uint32_t* __restrict__ firstj,
                                  uint32_t* __restrict__ cnt, uint32_t* __restrict__ slot_of,
                                  uint32_t mask, const int32_t* __restrict__ sample_ids,
                                  const uint32_t* __restrict__ orig, uint32_t base, int32_t S, uint32_t j_lo, uint32_t j_hi,
                                  const uint32_t* __restrict__ weight) {
    __shared__ unsigned long long c_key[MIRGE_CELL_CACHE];
    __shared__ uint32_t c_min[MIRGE_CELL_CACHE];
    __shared__ uint32_t c_cnt[MIRGE_CELL_CACHE];
    for (uint32_t i = threadIdx.x; i < MIRGE_CELL_CACHE; i += blockDim.x) { c_key[i] = 0ull; c_min[i] = 0xFFFFFFFFu; c_cnt[i] = 0; }
    __syncthreads();
    for (uint32_t j = j_lo + blockIdx.x * blockDim.x + threadIdx.x; j < j_hi; j += gridDim.x * blockDim.x) {
        MirgeRead<W> r;
        load_read<W>(g, j, r);
        uint64_t h = mirge_mix64(r.w[0] ^ ((uint64_t)r.len << 56));
#pragma unroll
        for (int w = 1; w < W; w++) h = mirge_mix64(h ^ r.w[w]);
#pragma unroll
        for (int w = 0; w < W; w++) h ^= mirge_mix64(r.nm[w] + 0x9e3779b97f4a7c15ull * (w + 1));
        uint32_t s = (uint32_t)(h >> 20) & mask;
        while (true) {
            uint32_t cur = rep[s];
            if (cur == MIRGE_EMPTY) cur = atomicCAS(&rep[s], MIRGE_EMPTY, j);
            if (cur == MIRGE_EMPTY || cur == j || same_read<W>(g, cur, r)) break;
            s = (s + 1) & mask;
        }
        slot_of[j] = s;
        const uint32_t hidx = orig ? orig[j] : base + j;
        const int32_t sid = sample_ids ? sample_ids[hidx] : 0;
        const uint32_t wgt = weight ? weight[hidx] : 1u;  // a read that stands for `wgt` copies (mirge_collapse_weighted)
        // The slot now identifies the read's sequence.  Its (min index, count) update goes through a
        // workgroup cache in LDS keyed by the cell (slot, sample): a hot sequence -- adapter dimers
        // are millions of identical long reads -- then costs this workgroup one pair of global
        // atomics instead of one pair per copy (same-address device atomics run at ~90 per us).
        const unsigned long long cell = (unsigned long long)s * (unsigned)S + (unsigned)sid + 1ull;  // 0 = empty
        uint32_t cs = (uint32_t)(mirge_mix64(cell) >> 11) & (MIRGE_CELL_CACHE - 1);
        bool cached = false;
        for (int t = 0; t < 4; t++) {
            unsigned long long cur = c_key[cs];
            if (cur == 0ull) cur = atomicCAS(&c_key[cs], 0ull, cell);
            if (cur == 0ull || cur == cell) {
                atomicMin(&c_min[cs], j);
                atomicAdd(&c_cnt[cs], wgt);
                cached = true;
                break;
            }
            cs = (cs + 1) & (MIRGE_CELL_CACHE - 1);
        }
        if (!cached) {
            atomicMin(&firstj[s], j);
            atomicAdd(&cnt[(size_t)s * S + sid], wgt);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < MIRGE_CELL_CACHE; i += blockDim.x) {
        const unsigned long long cell = c_key[i];
        if (cell == 0ull) continue;
        const unsigned long long lin = cell - 1ull;
        atomicMin(&firstj[lin / (unsigned)S], c_min[i]);
        atomicAdd(&cnt[lin], c_cnt[i]);
    }
}

// Fast form for the <=31-nt group without ambiguous calls and one sample (the bulk of any run): the
// whole identity of a read -- its bits plus a length sentinel bit at 2*len -- fits one u64, so the
// table holds the key itself: a duplicate is recognised from the slot (no representative read to
// fetch), and key, first index and count share one 16-byte slot = one 64-byte sector per read
// instead of six.  first is kept as ~j under atomicMax so that a zero-filled table is "empty".
struct KeySlot {
    unsigned long long key;  // 0 = empty
    uint32_t first_inv;      // 0xFFFFFFFF - (smallest read index)
    uint32_t cnt;
};

__global__ void k_collapse_insert_key(GroupView<1> g, KeySlot* __restrict__ slots, uint32_t* __restrict__ slot_of,
                                      uint32_t mask) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < g.n; j += gridDim.x * blockDim.x) {
        const unsigned long long key = g.seq[j] | (1ull << (2 * g.len[j]));
        uint32_t s = (uint32_t)(mirge_mix64(key) >> 20) & mask;
        uint32_t seen_first = 0;
        while (true) {
            const uint4 v = *reinterpret_cast<const uint4*>(&slots[s]);  // key, first_inv, cnt in one load
            unsigned long long cur = ((unsigned long long)v.y << 32) | v.x;
            seen_first = v.z;
            if (cur == 0ull) { cur = atomicCAS(&slots[s].key, 0ull, key); seen_first = 0; }
            if (cur == 0ull || cur == key) break;
            s = (s + 1) & mask;
        }
        slot_of[j] = s;
        // first_inv only grows, so a (possibly stale) plain read that is already >= ours proves the
        // atomic would change nothing: most duplicates skip it (scattered atomics run at ~20 G/s
        // chip-wide and are what bounds this kernel)
        if (0xFFFFFFFFu - j > seen_first) atomicMax(&slots[s].first_inv, 0xFFFFFFFFu - j);
        atomicAdd(&slots[s].cnt, 1u);
    }
}

// rep[] / firstj[] = 0xFFFFFFFF, cnt[] = 0 in ONE launch (three hipMemsetAsync = six fill kernels per small group before)
__global__ void k_collapse_init(uint32_t* __restrict__ rep, uint32_t* __restrict__ firstj, uint32_t* __restrict__ cnt,
                                uint32_t tsize, uint64_t ncnt) {
    // ncnt = tsize * S cells of the count matrix: beyond 2^32 from ~24 samples x 4 M reads on, hence the 64-bit walk
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < ncnt; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i < tsize) { rep[i] = MIRGE_EMPTY; firstj[i] = MIRGE_EMPTY; }
        cnt[i] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// Partitioned collapse for the key path (<=31 nt, no N, one sample).  Scattered device-scope atomics
// run at ~20 G/s chip-wide, which is what bounds k_collapse_insert_key (2.4 atomics per read).  Here
// equal keys are first brought together: reads are partitioned by the top bits of their hash into
// buckets of ~1-2 k reads, then ONE workgroup de-duplicates a bucket entirely in LDS (ds_cmpst / ds_min /
// ds_add) and emits the bucket's distinct reads with their counts.
// The partition is a radix split in one or two levels whose writes are APPEND STREAMS: a writer workgroup appends
// 16-byte records to one stream per bin through an LDS cursor.  A workgroup keeps 64-512 such streams open, so a CU's
// open 128-byte lines (<= 64 KiB) stay in L2 until they are full and HBM sees whole lines -- round 1's one-level scatter
// had 256 workgroups x 8192 buckets = 2 M open streams, every 16-byte record its own sector write (the random-sector
// ceiling, 0.17-0.18 ms of the step), behind a column prefix over a 2 M-cell histogram and a scan.
//   k_part_agg   : per workgroup chunk g: LDS cache merges equal reads -> records {key, min j, count}, appended to the
//                  workgroup's region of the record's level-1 bin b1 (fixed capacity: mean x 1.25 + 8 sigma; the hash is
//                  uniform and the cache has taken the hot keys out; a region that would overflow raises the flag that
//                  already sends a call to the global-atomic path)              rec1[(b1 * G + g) * cap1 + i]
//                  and counted per FINAL bucket                                 hist[g][b]
//   k_part_split : (two levels, B > 64) workgroup (b1, w) owns G / W2 of bin b1's regions: their hist rows give the
//                  exact size of every bucket's share, so the records are appended at exact offsets inside the
//                  workgroup's slab -- no capacity to overflow, however many copies of a read there are
//                                                                               rec2[(b1 * W2 + w) * slab + off[b2] + i]
//   k_part_dedup : per bucket, LDS table (key -> min j, count) over the bucket's R shares (R = G regions of level 1, or W2
//                  slab ranges); the bucket's distinct reads are written to the output at a range reserved with one global
//                  atomicAdd per workgroup (so the order of the unique reads of this path is unspecified; first[]
//                  carries the first raw index)
// ------------------------------------------------------------------------------------------
#ifndef MIRGE_PART_CAP
#define MIRGE_PART_CAP 4096  // largest LDS table per bucket (16 B per slot = 64 KiB)
#endif
#ifndef MIRGE_PART_B1
#define MIRGE_PART_B1 64       // level-1 bins (and the largest one-level partition)
#endif
#ifndef MIRGE_PART_SMALL
#define MIRGE_DEDUP_SHARDS 8          // output cursors of k_part_dedup (shards of its staging area)
#define MIRGE_DEDUP_SHARD_STRIDE 1024  // words between two cursors: 4 KiB, a memory channel of its own each
#define MIRGE_PART_SMALL 1600  // reads per bucket up to which k_part_dedup uses its 2048-slot table
#endif
#ifndef MIRGE_PART_W2
#define MIRGE_PART_W2 4       // splitter workgroups per level-1 bin (64 x 4 = one per CU)
#endif
#define MIRGE_PART_MAXREG 256  // regions a consumer workgroup reads (G <= 256 writers, or W2 splitters)

__device__ __forceinline__ unsigned long long read_key64(const GroupView<1>& g, uint32_t j) {
    return g.seq[j] | (1ull << (2 * g.len[j]));
}

// item i of a list of regions whose fill counts have the inclusive prefix pre[0..R): region and offset inside it
__device__ __forceinline__ uint32_t region_of(const uint32_t* pre, uint32_t R, uint32_t i, uint32_t& within) {
    uint32_t lo = 0, hi = R;  // first r with pre[r] > i
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pre[mid] > i) hi = mid; else lo = mid + 1;
    }
    within = i - (lo ? pre[lo - 1] : 0u);
    return lo;
}

// inclusive prefix of up to MIRGE_PART_MAXREG region counts (clamped to cap) into LDS, by the first wave(s); returns total
__device__ __forceinline__ uint32_t region_prefix(const uint32_t* __restrict__ cnt, uint32_t R, uint32_t cap, uint32_t* pre) {
    for (uint32_t r = threadIdx.x; r < R; r += blockDim.x) pre[r] = min(cnt[r], cap);
    __syncthreads();
    if (threadIdx.x == 0) {  // R <= 256: a serial pass over LDS costs less than the barriers of a block scan
        uint32_t run = 0;
        for (uint32_t r = 0; r < R; r++) { run += pre[r]; pre[r] = run; }
    }
    __syncthreads();
    return R ? pre[R - 1] : 0u;
}

// k_part_agg: a workgroup walks its chunk of reads through a small LDS cache (key -> min index,
// count) before anything is partitioned.  Real small-RNA samples are extremely skewed (one miRNA can be
// a third of all reads): without this the hot key's bucket holds millions of records for ONE workgroup
// and every LDS atomic on it is a 64-way conflict (measured on a Zipf sample: 10.9 ms per step against
// 2.8 ms on unskewed reads).  With it a key contributes at most one record per workgroup.  The cache is
// best effort: a read that finds no slot within 4 probes is emitted as a record of count 1.
// One workgroup per CU (the LDS cache takes a good part of a CU's LDS), so the workgroup itself must bring
// the waves that hide its load and LDS latencies: 1024 threads = 16 waves per CU (256 threads: 0.26 ms, 2x slower)
#ifndef MIRGE_PART_THREADS
#define MIRGE_PART_THREADS 1024
#endif
#ifndef MIRGE_AGG_BUCKET4
#define MIRGE_AGG_BUCKET4 1
#endif
#ifndef MIRGE_AGG_UNROLL
#define MIRGE_AGG_UNROLL 2  // (measured, round 5: 0.131 -> 0.122 ms per 9.6 M reads; 4 reads per trip: the same)
#endif
__global__ void __launch_bounds__(MIRGE_PART_THREADS)
k_part_agg(GroupView<1> g, const uint32_t* __restrict__ orig, uint32_t base, uint32_t chunk, uint32_t shift1, uint32_t NB1,
           uint32_t bshift, uint32_t B, uint32_t CS, uint32_t cap1, uint4* __restrict__ rec1, uint32_t* __restrict__ cnt1,
           uint32_t* __restrict__ hist, uint32_t* __restrict__ overflow, uint32_t* __restrict__ n_records, uint32_t* __restrict__ shard_cur) {
    if (shard_cur && blockIdx.x == 0 && threadIdx.x < MIRGE_DEDUP_SHARDS) shard_cur[threadIdx.x * MIRGE_DEDUP_SHARD_STRIDE] = 0u;  // k_part_dedup's cursors
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds_a[];  // [CS] keys | [CS] minj | [CS] cnt | [NB1] cursors | [B] hist
    uint32_t* c_min = reinterpret_cast<uint32_t*>(lds_a + CS);
    uint32_t* c_cnt = c_min + CS;
    uint32_t* cur = c_cnt + CS;
    uint32_t* lds_h = cur + NB1;  // records per final bucket (two levels only: hist != nullptr)
    for (uint32_t i = threadIdx.x; i < CS; i += blockDim.x) { lds_a[i] = 0ull; c_min[i] = 0xFFFFFFFFu; c_cnt[i] = 0; }
    for (uint32_t b = threadIdx.x; b < NB1; b += blockDim.x) cur[b] = 0;
    if (hist) for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) lds_h[b] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * chunk, hi = min(lo + chunk, g.n);
    const uint32_t G = gridDim.x;
    auto append = [&](unsigned long long key, uint64_t h, uint32_t jr, uint32_t count) {
        const uint32_t b1 = (uint32_t)(h >> shift1);
        const uint32_t p = atomicAdd(&cur[b1], 1u);
        if (p < cap1) {
            rec1[((size_t)b1 * G + blockIdx.x) * cap1 + p] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), jr, count);
            if (hist) atomicAdd(&lds_h[(uint32_t)(h >> bshift)], 1u);
        }
    };
    // MIRGE_AGG_UNROLL reads per thread and trip, their loads (key, length, raw index) issued together: with one workgroup per CU
    // the kernel is a chain of load -> LDS-atomic latencies, and sixteen waves with one load each in flight do not cover HBM
    for (uint32_t j0 = lo; j0 < hi; j0 += blockDim.x * MIRGE_AGG_UNROLL) {
        unsigned long long keys[MIRGE_AGG_UNROLL];
        uint32_t jrs[MIRGE_AGG_UNROLL];
#pragma unroll
        for (int u = 0; u < MIRGE_AGG_UNROLL; u++) {
            const uint32_t j = j0 + u * blockDim.x + threadIdx.x;
            keys[u] = 0ull; jrs[u] = 0;
            if (j < hi) {
                keys[u] = read_key64(g, j);
                // index among ALL raw reads (orig[] ascends with j, so min commutes; read coalesced here
                // instead of gathered per unique read at the end)
                jrs[u] = orig ? orig[j] : base + j;
            }
        }
#pragma unroll
        for (int u = 0; u < MIRGE_AGG_UNROLL; u++) {
        const uint32_t j = j0 + u * blockDim.x + threadIdx.x;
        bool direct = false;
        unsigned long long key = keys[u];
        uint32_t jr = jrs[u];
        uint64_t h = 0;
        if (j < hi) {
            h = mirge_mix64(key);
            direct = true;
#if MIRGE_AGG_BUCKET4
            // the cache as 4-way buckets (round 5): the four keys of the read's home bucket come with ONE 32-byte read (two
            // ds_read_b128 issued together) instead of up to four dependent probes; a key lives in the first slot of its bucket
            // that was free when it arrived (slots never change once set, every inserter walks them in order and takes the
            // compare-and-swap's word for what a slot holds: no key can sit in two slots), so a later copy always finds it
            const uint32_t b0 = ((uint32_t)(h >> 9) & (CS / 4 - 1)) * 4;
            const ulonglong2 ka = *reinterpret_cast<const ulonglong2*>(&lds_a[b0]);
            const ulonglong2 kb = *reinterpret_cast<const ulonglong2*>(&lds_a[b0 + 2]);
            const unsigned long long k4[4] = {ka.x, ka.y, kb.x, kb.y};
            int slot = -1;
#pragma unroll
            for (int t = 0; t < 4; t++) if (slot < 0 && k4[t] == key) slot = t;
            if (slot < 0) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (slot >= 0 || k4[t] != 0ull) continue;  // (held by another key when read: it still is)
                    const unsigned long long c0 = atomicCAS(&lds_a[b0 + t], 0ull, key);
                    if (c0 == 0ull || c0 == key) slot = t;
                }
            }
            if (slot >= 0) {
                const uint32_t s = b0 + (uint32_t)slot;
                if (*(volatile uint32_t*)&c_min[s] > jr) atomicMin(&c_min[s], jr);
                atomicAdd(&c_cnt[s], 1u);
                direct = false;
            }
#else
            uint32_t s = (uint32_t)(h >> 9) & (CS - 1);
            // four probes, always: a hot key that was displaced from its home slot when it arrived must still be found by its
            // later copies (looking at the home slot only once the cache is full was 6 % faster on unskewed reads and sent
            // every copy of such a key to its bin as a record of its own on a Zipf sample)
            for (int t = 0; t < 4; t++) {
                unsigned long long c0 = lds_a[s];
                if (c0 == 0ull) c0 = atomicCAS(&lds_a[s], 0ull, key);
                if (c0 == 0ull || c0 == key) {
                    // the slot's minimum only falls: a plain read that is already below ours proves the atomic would
                    // change nothing (indices grow with the loop, so this skips it for every later copy of a read)
                    if (*(volatile uint32_t*)&c_min[s] > jr) atomicMin(&c_min[s], jr);
                    atomicAdd(&c_cnt[s], 1u);
                    direct = false;
                    break;
                }
                s = (s + 1) & (CS - 1);
            }
#endif
        }
        // cache full around this key: the read itself becomes a record.  Copies of ONE sequence that meet in a wave are
        // merged first (two rounds: the key of the first such lane, then of the first lane left) -- a burst of a sequence that
        // arrives after the cache has filled is then one record per wave, not 64 for one level-1 region
        uint32_t count = 1;
        bool open = direct;  // not yet looked at as a group's key
#pragma unroll
        for (int round = 0; round < 2; round++) {
            const unsigned long long ob = __ballot(open);
            if (!ob) break;
            const int lead = __ffsll(ob) - 1;
            const unsigned long long k0 = __shfl(key, lead, 64);
            const bool same = open && key == k0;
            const unsigned long long sb = __ballot(same);
            if (__popcll(sb) > 1) {
                uint32_t mn = same ? jr : 0xFFFFFFFFu;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
                if ((threadIdx.x & 63) == lead) { jr = mn; count = (uint32_t)__popcll(sb); }
                else if (same) direct = false;  // the leader carries the group
            }
            open = open && !same;
        }
        if (direct) append(key, h, jr, count);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < CS; i += blockDim.x) {  // flush the cache
        const unsigned long long key = lds_a[i];
        if (key != 0ull) append(key, mirge_mix64(key), c_min[i], c_cnt[i]);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < NB1; b += blockDim.x) {
        const uint32_t n = cur[b];
        cnt1[(size_t)b * G + blockIdx.x] = min(n, cap1);
        if (n > cap1) atomicOr(overflow, 1u);
    }
    // (round 6) the sample's record count: what k_part_dedup chooses its output mode from -- on the device, for THIS sample (until
    // round 5 the host chose from the context's previous sample).  ONE add per workgroup: an add per level-1 bin was 16 384 adds on one
    // address, served one after the other at 11.6 ns -- 0.19 ms of atomic traffic that was still draining under k_part_split and
    // k_part_dedup, whose own cursor adds queued behind it (k_part_dedup 0.116 -> 0.130 ms).
    if (threadIdx.x == 0) {
        uint32_t mine = 0;
        for (uint32_t b = 0; b < NB1; b++) mine += min(cur[b], cap1);
        atomicAdd(n_records, mine);
    }
    if (hist) for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) hist[(size_t)blockIdx.x * B + b] = lds_h[b];
}

// k_part_split: second radix level.  Workgroup (b1 = blockIdx / W2, w = blockIdx % W2) takes the regions
// [w * RPW, (w + 1) * RPW) of level-1 bin b1 (coalesced reads of ~0.6 k records each).  The hist rows of those writers
// give the exact number of records per final bucket b1 * NB2 + b2, so every record is appended at an exact offset of
// the workgroup's slab; the bucket's share is published as (offset, count) for k_part_dedup.
__global__ void __launch_bounds__(MIRGE_PART_THREADS)
k_part_split(const uint4* __restrict__ rec1, const uint32_t* __restrict__ cnt1, const uint32_t* __restrict__ hist, uint32_t G,
             uint32_t B, uint32_t cap1, uint32_t W2, uint32_t RPW, uint32_t shift2, uint32_t NB2, uint64_t slab,
             uint4* __restrict__ rec2, uint32_t* __restrict__ off2, uint32_t* __restrict__ cnt2) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_s[];  // [MAXREG] prefix | [NB2] cursors (start at the bucket's offset)
    uint32_t* pre = lds_s;
    uint32_t* cur = lds_s + MIRGE_PART_MAXREG;
    __shared__ uint32_t scan_tmp[MIRGE_PART_THREADS / 64 + 1];
    const uint32_t b1 = blockIdx.x / W2, w = blockIdx.x % W2;
    const uint32_t r0 = w * RPW, R = r0 < G ? min(RPW, G - r0) : 0u;
    // this workgroup's records per bucket, then their exclusive prefix = the buckets' offsets in the slab (NB2 <= 512 <= threads)
    uint32_t mine = 0;
    if (threadIdx.x < NB2)
        for (uint32_t r = 0; r < R; r++) mine += hist[(size_t)(r0 + r) * B + (size_t)b1 * NB2 + threadIdx.x];
    uint32_t tot_all;
    const uint32_t off = block_excl_scan<MIRGE_PART_THREADS / 64>(mine, tot_all, scan_tmp);
    if (threadIdx.x < NB2) {
        cur[threadIdx.x] = off;
        const size_t cell = ((size_t)b1 * NB2 + threadIdx.x) * W2 + w;
        off2[cell] = off;
        cnt2[cell] = mine;
    }
    for (uint32_t r = threadIdx.x; r < R; r += blockDim.x) pre[r] = min(cnt1[(size_t)b1 * G + r0 + r], cap1);
    __syncthreads();  // publishes cur[] and the regions' fill counts
    const uint4* in = rec1 + ((size_t)b1 * G + r0) * cap1;
    uint4* out = rec2 + (size_t)blockIdx.x * slab;
    // a wave takes whole regions: 64 consecutive records per load, no search for the region of an item
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_waves = blockDim.x >> 6;
    for (uint32_t r = wave; r < R; r += n_waves) {
        const uint32_t n = pre[r];
        const uint4* src = in + (size_t)r * cap1;
        // four loads in flight per lane: with 16 waves on a CU the kernel is a chain of load -> LDS atomic -> store
        // latencies, not bandwidth
        for (uint32_t i0 = lane; i0 < n; i0 += 256) {
            uint4 rec[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (i0 + 64 * u < n) rec[u] = src[i0 + 64 * u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (i0 + 64 * u >= n) continue;
                const unsigned long long key = ((unsigned long long)rec[u].y << 32) | rec[u].x;
                const uint32_t b2 = (uint32_t)(mirge_mix64(key) >> shift2) & (NB2 - 1);
                out[atomicAdd(&cur[b2], 1u)] = rec[u];
            }
        }
    }
}

// CAP = LDS table slots (16 B each): 2048 when the buckets hold <= 1024 records (4 workgroups per CU), else 4096
// The table takes 32-64 KiB of LDS, so only 2-4 workgroups fit a CU: 1024-thread workgroups bring the waves.
// The bucket's records lie in R shares: share r holds cnt[bucket * R + r] records (clamped to rcap) from
// rec[(bucket * R + r) * rcap]  (off == nullptr: level-1 regions), or from rec[((bucket / NB2) * R + r) * rcap + off[bucket * R + r]]
// (ranges inside the level-2 slabs of capacity rcap).
#ifndef MIRGE_DEDUP_THREADS
// 512 threads: four workgroups per CU are resident (a CU holds 2048 threads: with 1024 only two of the four the 2048-slot table
// leaves room for), their barrier phases overlap: 0.140 -> 0.126 ms per 9.6 M reads (256 threads: 0.132)
#define MIRGE_DEDUP_THREADS 512
#endif
template <int CAP>
__global__ void __launch_bounds__(MIRGE_DEDUP_THREADS)
k_part_dedup(const uint4* __restrict__ rec, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, uint32_t R,
             uint64_t rcap, uint32_t NB2, uint64_t* __restrict__ useq, uint8_t* __restrict__ ulen,
             uint32_t* __restrict__ ucnt, uint32_t* __restrict__ ufirst, uint32_t* __restrict__ cursor,
             uint32_t* __restrict__ hist, uint32_t* __restrict__ overflow, uint32_t* __restrict__ shard_cur_in, uint32_t shard_cap,
             const uint32_t* __restrict__ n_records, uint32_t rec_thresh, uint64_t* __restrict__ sseq, uint8_t* __restrict__ slen, uint32_t* __restrict__ scnt,
             uint32_t* __restrict__ sfirst) {
    // Output mode, chosen HERE from the sample's record count (*n_records, written by k_part_agg; a line of its own): few records = few unique reads =
    // the kernel's time is the one global cursor's returning adds (11.6 ns each, one per bucket) -> eight cursors into the staging
    // arrays (s*), k_part_compact closes the gaps; many records = the kernel's time is its records -> the one cursor, straight into
    // the output arrays (and k_part_compact returns at once).  Uniform over the launch: every workgroup reads the same word.
    // (the word is ASKED FOR here and LOOKED AT behind the table's read-out, where the output range is reserved: a workgroup is a
    // latency chain of ~12 us and eight rounds of them fill the chip -- waiting for this load at the top cost the kernel 13 us)
    // ... and asked for with a VECTOR load (an address the compiler cannot prove uniform): a scalar load would be waited for at the
    // kernel's first `s_waitcnt lgkmcnt(0)` -- scalar loads and LDS operations share that counter --, i.e. at the top after all; the
    // vector load is in flight together with the bucket's first records.
    // ... by ONE thread of the workgroup (every lane asking was 65 k requests for one address per launch: a hot spot on one L2 channel).
    uint32_t lane_zero = 0u;
    asm volatile("" : "+v"(lane_zero));  // (a zero in a vector register the compiler cannot see through)
    uint32_t n_rec_now = 0u;
    if (threadIdx.x == 0 && shard_cur_in) n_rec_now = n_records[lane_zero];
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds_k[];  // [CAP] keys, then [CAP] minj, [CAP] cnt
    uint32_t* lds_min = reinterpret_cast<uint32_t*>(lds_k + CAP);
    uint32_t* lds_cnt = lds_min + CAP;
    uint32_t* lds_x = lds_cnt + CAP;  // [0] distinct keys, [1] output base, [2..17] scan scratch, [32..63] lengths (<= 31 nt)
    uint32_t* pre = lds_x + 64;       // [MAXREG] inclusive prefix of the regions' fill counts
    uint32_t& n_distinct = lds_x[0];
    {   // clear with 16-byte stores: keys and counts to 0, first indices to ~0 (the table's set-up and read-out are most of
        // this kernel's LDS instructions: a bucket fills a tenth of its table)
        uint4* k4 = reinterpret_cast<uint4*>(lds_k);
        for (uint32_t i = threadIdx.x; i < CAP / 2; i += blockDim.x) k4[i] = make_uint4(0u, 0u, 0u, 0u);
        uint4* m4 = reinterpret_cast<uint4*>(lds_min);
        for (uint32_t i = threadIdx.x; i < CAP / 4; i += blockDim.x) m4[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        uint4* c4 = reinterpret_cast<uint4*>(lds_cnt);
        for (uint32_t i = threadIdx.x; i < CAP / 4; i += blockDim.x) c4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    for (uint32_t i = threadIdx.x; i < 32 + 32; i += blockDim.x) lds_x[i] = 0;  // key-path reads are <= 31 nt
    const uint4* in = rec + (size_t)(off ? blockIdx.x / NB2 : blockIdx.x) * R * rcap;
    uint32_t* roff = pre + MIRGE_PART_MAXREG;  // where share r starts inside its region / slab
    // up to four shares (the two-level partition): their bounds in registers, no search through LDS per record
    const bool few = R <= 4;
    const uint32_t cap32 = (uint32_t)min(rcap, (uint64_t)0xFFFFFFFFu);
    uint32_t total_in, e0, e1, e2;
    if (few) {
        // (round 5) the shares' counts and offsets are loaded TOGETHER, one barrier, and every thread adds the four counts up itself
        // instead of region_prefix's serial pass and second barrier.  Measured: 0.125 -> 0.123 ms -- a bucket costs ~12 us whatever
        // it holds (0.11 ms per 8192 buckets on a sample with 3 % unique reads as on one with 42 %; the waves wait 70 % of their
        // cycles), but this round trip was not what they wait for.  Kept for the barrier it saves.
        if (threadIdx.x < R) {
            pre[threadIdx.x] = min(cnt[(size_t)blockIdx.x * R + threadIdx.x], cap32);
            roff[threadIdx.x] = off ? off[(size_t)blockIdx.x * R + threadIdx.x] : 0u;
        }
        __syncthreads();  // (covers the clears too)
        e0 = pre[0]; e1 = e0 + (R > 1 ? pre[1] : 0u); e2 = e1 + (R > 2 ? pre[2] : 0u);
        total_in = e2 + (R > 3 ? pre[3] : 0u);
    } else {
        total_in = region_prefix(cnt + (size_t)blockIdx.x * R, R, cap32, pre);  // (its barriers cover the clears)
        for (uint32_t r = threadIdx.x; r < R; r += blockDim.x) roff[r] = off ? off[(size_t)blockIdx.x * R + r] : 0u;
        __syncthreads();
        e0 = pre[0]; e1 = R > 1 ? pre[1] : e0; e2 = R > 2 ? pre[2] : e1;
    }
    // a bucket with fewer records than the table has slots cannot fill it: the distinct-key counter (one LDS atomic on a
    // single address per new key, serialised over the lanes) is kept for the oversized buckets only
    const bool counted = total_in >= (uint32_t)(CAP - 64);
    const uint32_t o0 = roff[0], o1 = R > 1 ? roff[1] : 0u, o2 = R > 2 ? roff[2] : 0u, o3 = R > 3 ? roff[3] : 0u;
    for (uint32_t i = threadIdx.x; i < total_in; i += blockDim.x) {
        size_t at;
        if (few) {
            at = i < e0 ? (size_t)o0 + i : i < e1 ? rcap + o1 + (i - e0) : i < e2 ? 2 * rcap + o2 + (i - e1) : 3 * rcap + o3 + (i - e2);
        } else {
            uint32_t k;
            const uint32_t r = region_of(pre, R, i, k);
            at = (size_t)r * rcap + roff[r] + k;
        }
        const uint4 rc = in[at];
        const unsigned long long key = ((unsigned long long)rc.y << 32) | rc.x;
        const uint32_t j = rc.z;
        uint32_t s = (uint32_t)(mirge_mix64(key) >> 7) & (CAP - 1);
        while (true) {
            unsigned long long c0 = lds_k[s];
            if (c0 == 0ull) {
                c0 = atomicCAS(&lds_k[s], 0ull, key);
                if (counted && c0 == 0ull && atomicAdd(&n_distinct, 1u) >= CAP - 64) atomicOr(overflow, 1u);
            }
            if (c0 == 0ull || c0 == key) break;
            if (counted && *(volatile uint32_t*)&n_distinct >= CAP - 32) break;  // table full: flagged, results discarded
            s = (s + 1) & (CAP - 1);
        }
        if (*(volatile uint32_t*)&lds_min[s] > j) atomicMin(&lds_min[s], j);
        atomicAdd(&lds_cnt[s], rc.w);  // a record stands for rc.w identical reads of one workgroup's chunk
    }
    __syncthreads();
    // emit the bucket's distinct reads: one global cursor add per workgroup reserves their output range
    constexpr int PER = CAP / MIRGE_DEDUP_THREADS;
    const uint32_t s0 = threadIdx.x * PER;
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) mine += lds_k[s0 + i] != 0ull;
    uint32_t total;
    uint32_t rank = block_excl_scan<MIRGE_DEDUP_THREADS / 64>(mine, total, lds_x + 2);
    uint32_t* const shard_cur = (shard_cur_in && n_rec_now < rec_thresh) ? shard_cur_in : nullptr;  // (thread 0's: it reserves the range)
    // Where the bucket's unique reads go.  One global cursor for all buckets (shard_cur == nullptr) is a returning add on ONE
    // address per bucket, and such adds are served one after the other, 11.6 ns each: 8192 buckets = 0.095 of this kernel's
    // 0.116 ms whatever they hold (profiles/README.md, round 5).  With shards, bucket b adds to cursor b mod MIRGE_DEDUP_SHARDS
    // (each on a line and channel of its own) and writes into that shard's stretch of a staging area; k_part_compact closes
    // the gaps between the stretches afterwards.
    if (threadIdx.x == 0) {
        uint32_t at = 0;
        if (total) {
            if (shard_cur) {
                const uint32_t x = blockIdx.x % MIRGE_DEDUP_SHARDS;
                const uint32_t o = atomicAdd(&shard_cur[x * MIRGE_DEDUP_SHARD_STRIDE], total);
                if (o + total > shard_cap) { atomicOr(overflow, 1u); at = 0xFFFFFFFFu; }  // (no room: nothing is written, the call is redone)
                else at = x * shard_cap + o;
            } else at = atomicAdd(cursor, total);
        }
        lds_x[1] = at;
        lds_x[18] = shard_cur ? 1u : 0u;  // (the mode, for the other threads: [2..17] is the scan's scratch, [32..63] the lengths)
    }
    __syncthreads();
    if (lds_x[1] == 0xFFFFFFFFu) return;
    if (lds_x[18]) { useq = sseq; ulen = slen; ucnt = scnt; ufirst = sfirst; }
    rank += lds_x[1];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const unsigned long long key = lds_k[s0 + i];
        if (key == 0ull) continue;
        const int L = (63 - __clzll((long long)key)) >> 1;  // the sentinel bit sits at 2*len
        useq[rank] = key ^ (1ull << (2 * L));
        ulen[rank] = (uint8_t)L;
        ucnt[rank] = lds_cnt[s0 + i];
        ufirst[rank] = lds_min[s0 + i];
        if (hist) atomicAdd(&lds_x[32 + L], 1u);
        rank++;
    }
    if (!hist) return;  // (the histogram comes from k_len_hist on the side stream: see there)
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 32; i += blockDim.x)
        if (lds_x[32 + i]) atomicAdd(&hist[i], lds_x[32 + i]);
}

// The length histogram of the partitioned path's unique reads, by a kernel of its own on the side stream (round 5).  k_part_dedup
// counted it while it emitted: one LDS atomic per unique read on ~16 hot counters and ~16 global atomics per bucket on 16
// addresses -- 130 k of them per sample from 8192 workgroups.  Nothing on the GPU waits for the histogram (the host reads it
// with the counts), but the main stream waited for those atomics: k_part_dedup 0.126 -> 0.118 ms and the step -1.7 % without
// them (profiles/README.md, round 5).  len[0 .. *n_dev): 3.8 MB per 10 M-read sample.
// The shards' stretches of k_part_dedup's staging area, one behind the other: the group's unique reads, dense.  *n_out = their number.
__global__ void __launch_bounds__(256) k_part_compact(const uint32_t* __restrict__ shard_cur, uint32_t shard_cap,
                                                       const uint64_t* __restrict__ sseq, const uint8_t* __restrict__ slen,
                                                       const uint32_t* __restrict__ scnt, const uint32_t* __restrict__ sfirst,
                                                       uint64_t* __restrict__ useq, uint8_t* __restrict__ ulen, uint32_t* __restrict__ ucnt,
                                                       uint32_t* __restrict__ ufirst, uint32_t* __restrict__ n_out,
                                                       const uint32_t* __restrict__ n_rec, uint32_t rec_thresh) {
    if (!(*n_rec < rec_thresh)) return;  // k_part_dedup took the one cursor: the output is dense already, *n_out is that cursor
    uint32_t end[MIRGE_DEDUP_SHARDS];  // (compile-time indices only: registers)
    uint32_t run = 0;
    bool over = false;  // a shard that overflowed was left unwritten by some buckets (k_part_dedup set the overflow flag: the call is
                        // redone): the group is published EMPTY, so that the cascade queued behind never walks uninitialised records
#pragma unroll
    for (int x = 0; x < MIRGE_DEDUP_SHARDS; x++) {
        const uint32_t cx = shard_cur[x * MIRGE_DEDUP_SHARD_STRIDE];
        over |= cx > shard_cap;
        run += min(cx, shard_cap); end[x] = run;
    }
    if (over) run = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = run;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < run; e += gridDim.x * blockDim.x) {
        uint32_t x = 0, lo = 0;
#pragma unroll
        for (int y = 0; y < MIRGE_DEDUP_SHARDS - 1; y++) if (e >= end[y]) { x = (uint32_t)y + 1; lo = end[y]; }
        const size_t src = (size_t)x * shard_cap + (e - lo);
        useq[e] = sseq[src]; ulen[e] = slen[src]; ucnt[e] = scnt[src]; ufirst[e] = sfirst[src];
    }
}

__global__ void __launch_bounds__(256) k_len_hist(const uint8_t* __restrict__ len, const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n = *n_dev;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(&h[len[i] & 63u], 1u);
    __syncthreads();
    if (threadIdx.x < 64 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// heads: read j is the head of its group iff it is the group's smallest index.  `first` is addressed
// as first[slot * stride] (inv: stored as 0xFFFFFFFF - index).  Writes flag[j] and per-block sums.
#define MIRGE_SCAN_ITEMS 8  // per thread -> 2048 per block
__global__ void k_heads_blocksum(const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ first,
                                 uint32_t stride, uint32_t inv, uint32_t n, const uint8_t* __restrict__ len,
                                 uint8_t* __restrict__ flag, uint32_t* __restrict__ blocksum,
                                 uint32_t* __restrict__ hist) {
    __shared__ uint32_t lds4[4];
    __shared__ uint32_t h[MIRGE_MAX_READ_LEN + 1];  // lengths of the heads = lengths of the unique reads
    for (int i = threadIdx.x; i <= MIRGE_MAX_READ_LEN; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint32_t b0 = blockIdx.x * (MIRGE_BLOCK * MIRGE_SCAN_ITEMS) + threadIdx.x * MIRGE_SCAN_ITEMS;
    uint32_t c = 0;
    uint8_t fl[MIRGE_SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < MIRGE_SCAN_ITEMS; i++) {
        const uint32_t j = b0 + i;
        fl[i] = 0;
        if (j < n) {
            const uint32_t f = first[(size_t)slot_of[j] * stride];
            fl[i] = (inv ? 0xFFFFFFFFu - f : f) == j;
            if (fl[i]) {
                c++;
                if (len) {  // (nullptr: the long class, whose lengths the histogram has no room for)
                    const uint32_t L = len[j];
                    atomicAdd(&h[L > MIRGE_MAX_READ_LEN ? MIRGE_MAX_READ_LEN : L], 1u);
                }
            }
        }
    }
    if (b0 + MIRGE_SCAN_ITEMS <= n) {
        uint64_t packed = 0;
#pragma unroll
        for (int i = 0; i < MIRGE_SCAN_ITEMS; i++) packed |= (uint64_t)fl[i] << (8 * i);
        *reinterpret_cast<uint64_t*>(flag + b0) = packed;
    } else {
#pragma unroll
        for (int i = 0; i < MIRGE_SCAN_ITEMS; i++) if (b0 + i < n) flag[b0 + i] = fl[i];
    }
    uint32_t total;
    block_excl_scan(c, total, lds4);  // two barriers: the LDS histogram is complete after it
    if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
    for (int i = threadIdx.x; i <= MIRGE_MAX_READ_LEN; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// single block: exclusive scan of blocksum[0..nb) in place, total to *out_total
__global__ void k_scan_blocksums(uint32_t* __restrict__ blocksum, uint32_t nb, uint32_t* __restrict__ out_total) {
    __shared__ uint32_t lds4[4];
    uint32_t carry = 0;
    for (uint32_t b = 0; b < nb; b += MIRGE_BLOCK) {
        uint32_t i = b + threadIdx.x;
        uint32_t v = i < nb ? blocksum[i] : 0u;
        uint32_t total;
        uint32_t ex = block_excl_scan(v, total, lds4);
        if (i < nb) blocksum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *out_total = carry;
}

template <int W>
__global__ void k_collapse_scatter(GroupView<W> g, const uint32_t* __restrict__ slot_of,
                                   const uint8_t* __restrict__ flag, const uint32_t* __restrict__ cnt,
                                   uint32_t cnt_stride, const uint32_t* __restrict__ blockoff,
                                   const uint32_t* __restrict__ n_uniq_ptr, const uint32_t* __restrict__ orig,
                                   uint32_t base, int32_t S, uint64_t* __restrict__ useq,
                                   uint8_t* __restrict__ ulen, uint64_t* __restrict__ unmask,
                                   uint32_t* __restrict__ ucnt, uint32_t* __restrict__ ufirst) {
    __shared__ uint32_t lds4[4];
    const uint32_t U = *n_uniq_ptr;
    const uint32_t b0 = blockIdx.x * (MIRGE_BLOCK * MIRGE_SCAN_ITEMS) + threadIdx.x * MIRGE_SCAN_ITEMS;
    uint32_t heads = 0, c = 0;
    if (b0 + MIRGE_SCAN_ITEMS <= g.n) {
        const uint64_t packed = *reinterpret_cast<const uint64_t*>(flag + b0);
#pragma unroll
        for (int i = 0; i < MIRGE_SCAN_ITEMS; i++) if ((packed >> (8 * i)) & 1ull) { heads |= 1u << i; c++; }
    } else {
#pragma unroll
        for (int i = 0; i < MIRGE_SCAN_ITEMS; i++) if (b0 + i < g.n && flag[b0 + i]) { heads |= 1u << i; c++; }
    }
    uint32_t total;
    uint32_t rank = blockoff[blockIdx.x] + block_excl_scan(c, total, lds4);
#pragma unroll
    for (int i = 0; i < MIRGE_SCAN_ITEMS; i++) {
        if (heads & (1u << i)) {
            const uint32_t j = b0 + i;
            const uint32_t s = slot_of ? slot_of[j] : j;  // partitioned path: counts are stored per head read
#pragma unroll
            for (int w = 0; w < W; w++) {
                useq[(size_t)w * U + rank] = g.seq[(size_t)w * g.n + j];
                if (unmask) unmask[(size_t)w * U + rank] = g.nmask ? g.nmask[(size_t)w * g.n + j] : 0ull;
            }
            ulen[rank] = g.len[j];
            for (int32_t q = 0; q < S; q++) ucnt[(size_t)rank * S + q] = cnt[(size_t)s * cnt_stride + q];
            ufirst[rank] = orig ? orig[j] : base + j;
            rank++;
        }
    }
}

// mirge_collapse_merge (round 6): entry j of a per-sample dictionary's group becomes "read" number at + j of the joint set with its
// sample's index and its count as the weight (what mirge_collapse_weighted takes from the host, built where the dictionaries lie)
__global__ void k_merge_fill(const uint32_t* __restrict__ counts, uint32_t n, uint32_t at, int32_t sample, int32_t* __restrict__ dsample,
                             uint32_t* __restrict__ dweight) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        dsample[at + j] = sample;
        dweight[at + j] = counts[j];
    }
}
