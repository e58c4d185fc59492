// mirge_kernels.hpp -- the gfx950 kernels of the hot path (included by mirge_native.hip).
//
// All of them are integer / indexing kernels: no MFMA, wave64, 256-thread workgroups,
// grid-stride loops over a grid capped at CUs x 8 so that every launch fills the 256 CUs
// and per-launch overhead stays constant.  Reads are structure-of-arrays so that lane i of a
// wave loads word i of a contiguous 512-byte run (coalesced); everything that is random
// access (k-mer buckets, text windows) goes to L2 / Infinity Cache / HBM by design.
#pragma once
#include <hip/hip_runtime.h>
#include "mirge_core.hpp"

#define MIRGE_BLOCK 256
#define MIRGE_MAX_PASSES_K 16
#define MIRGE_EMPTY 0xFFFFFFFFu
#define MIRGE_CELL_CACHE 2048  // per-workgroup LDS cache of (slot, sample) cells in the general collapse path

template <int W>
struct GroupView {
    const uint64_t* seq;    // [W][n]
    const uint8_t* len;     // [n]
    const uint64_t* nmask;  // [W][n] or nullptr
    uint32_t n;
};

template <int W>
__device__ __forceinline__ void load_read(const GroupView<W>& g, uint32_t i, MirgeRead<W>& r) {
#pragma unroll
    for (int w = 0; w < W; w++) {
        r.w[w] = g.seq[(size_t)w * g.n + i];
        r.nm[w] = g.nmask ? g.nmask[(size_t)w * g.n + i] : 0ull;
    }
    r.len = g.len[i];
}

// ------------------------------------------------------------------------------------------
// k_pack: ASCII -> 2-bit.  One thread per read; the read's bytes are contiguous in `ascii`.
// flags[0] |= 1 if any N was seen, flags[1] |= 1 if a byte outside ACGTN (any case) was seen.
// ------------------------------------------------------------------------------------------
template <int W>
__global__ void k_pack(const uint8_t* __restrict__ ascii, const int64_t* __restrict__ starts, const int64_t* __restrict__ ends,
                       const uint32_t* __restrict__ idx, uint32_t n, uint64_t* __restrict__ seq,
                       uint8_t* __restrict__ len, uint64_t* __restrict__ nmask,
                       uint32_t* __restrict__ flags) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t src = idx[j];
        const int64_t b = starts[src];
        const int L = (int)(ends[src] - b);  // contiguous reads: ends = starts + 1
        uint64_t w[W], nm[W];
#pragma unroll
        for (int i = 0; i < W; i++) { w[i] = 0; nm[i] = 0; }
        uint32_t sawN = 0, bad = 0;
        for (int p = 0; p < L; p++) {
            uint8_t c = ascii[b + p] & 0xDF;  // upper-case
            uint64_t code = 0, isn = 0;
            switch (c) {
                case 'A': code = 0; break;
                case 'C': code = 1; break;
                case 'G': code = 2; break;
                case 'T': code = 3; break;
                case 'U': code = 3; break;
                case 'N': isn = 1; break;
                default: isn = 1; bad = 1; break;
            }
            sawN |= (uint32_t)isn;
#pragma unroll
            for (int i = 0; i < W; i++)
                if ((p >> 5) == i) { w[i] |= code << (2 * (p & 31)); nm[i] |= isn << (2 * (p & 31)); }
        }
#pragma unroll
        for (int i = 0; i < W; i++) {
            seq[(size_t)i * n + j] = w[i];
            nmask[(size_t)i * n + j] = nm[i];
        }
        len[j] = (uint8_t)L;
        if (sawN) atomicOr(&flags[0], 1u);
        if (bad) atomicOr(&flags[1], 1u);
    }
}

// dst[i] = (src ? src[i] : src_base + i) + add : handle-order indices of a read set appended to another
__global__ void k_index_shift(const uint32_t* __restrict__ src, uint32_t src_base, uint32_t n, uint32_t add,
                              uint32_t* __restrict__ dst) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = (src ? src[i] : src_base + i) + add;
}

template <int W>
__global__ void k_unpack(GroupView<W> g, const int64_t* __restrict__ out_off, uint32_t base,
                         const uint32_t* __restrict__ orig, uint8_t* __restrict__ ascii_out) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < g.n; j += gridDim.x * blockDim.x) {
        MirgeRead<W> r;
        load_read<W>(g, j, r);
        const uint32_t dst = orig ? orig[j] : base + j;
        uint8_t* o = ascii_out + out_off[dst];
        for (int p = 0; p < r.len; p++) {
            uint32_t code = (uint32_t)((r.w[p >> 5] >> (2 * (p & 31))) & 3ull);
            uint32_t isn = (uint32_t)((r.nm[p >> 5] >> (2 * (p & 31))) & 1ull);
            o[p] = isn ? 'N' : "ACGT"[code];
        }
    }
}

// lengths scattered to handle order (for unpack offsets / histograms)
__global__ void k_scatter_len(const uint8_t* __restrict__ len, uint32_t n, uint32_t base,
                              const uint32_t* __restrict__ orig, int32_t* __restrict__ out) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        out[orig ? orig[j] : base + j] = len[j];
}

// ------------------------------------------------------------------------------------------
// Probe tables are built where they live.  For shape (k1, gap, k2): every position p whose span
// [p, p+k1+gap+k2) holds no invalid base, keyed by block A | block B << 2*k1 (mirge_hostlib_table is the host
// twin, tests/hostsim).  Counting sort with the count array shifted by two: count into A[key+2], inclusive
// scan, then slot = atomicAdd(&A[key+1], 1) leaves A[0..nb] = the CSR bucket bounds.  Positions inside a bucket
// come out in arbitrary order; every consumer takes a minimum over the whole bucket.
// Human mRNA, k = 15 (130 M positions, 2^30 buckets): ~30 ms on the GPU against ~6 s on the host.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t text_kmer_dev(const uint64_t* __restrict__ T, uint64_t g, int k) {
    const uint64_t q = g >> 5;
    const int s = (int)(g & 31) * 2;
    uint64_t lo = T[q] >> s;
    if (s) lo |= T[q + 1] << (64 - s);
    return lo & mirge_lowmask2(k);
}

template <bool FILL>
__global__ void k_table_pass(const uint64_t* __restrict__ T, const uint64_t* __restrict__ inv, uint64_t total, int k1, int gap,
                             int k2, uint32_t* __restrict__ A, uint32_t* __restrict__ pos) {
    const int span = k1 + (k2 > 0 ? gap + k2 : 0);
    if (total < (uint64_t)span) return;
    const uint64_t n = total - (uint64_t)span + 1;
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
        if (mirge_window_invalid(inv, p, span)) continue;
        uint64_t key = text_kmer_dev(T, p, k1);
        if (k2 > 0) key |= text_kmer_dev(T, p + (uint64_t)(k1 + gap), k2) << (2 * k1);
        if (FILL) pos[atomicAdd(&A[key + 1], 1u)] = (uint32_t)p;
        else atomicAdd(&A[key + 2], 1u);
    }
}

__global__ void k_table_bits(const uint32_t* __restrict__ bucket, uint64_t nb, uint32_t* __restrict__ bits) {
    const uint64_t nw = (nb + 31) / 32;
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < nw; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t m = 0;
        for (int b = 0; b < 32; b++) {
            const uint64_t k = w * 32 + b;
            if (k < nb && bucket[k + 1] > bucket[k]) m |= 1u << b;
        }
        bits[w] = m;
    }
}

// ------------------------------------------------------------------------------------------
// block-wide exclusive scan of one value per thread (256 threads = 4 waves of 64)
// ------------------------------------------------------------------------------------------
template <int NW = MIRGE_BLOCK / 64>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t& total, uint32_t* lds4) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds4[wv] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint32_t s = lds4[i];
        if (i < wv) woff += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return woff + inc - v;
}

// ------------------------------------------------------------------------------------------
// Sequence text parsed on the device (digest.py:320-375 reads FASTQ records with dnaio on the host): the file's
// bytes go to HBM as they are.  Line li is ended by newline number li; the sequence lines are those with
// li % period == sphase (FASTQ 4/1, single-line FASTA 2/1, one sequence per line 1/0).
//   k_nl_count  : newlines per 4 KiB tile                       (then an exclusive scan over the tiles)
//   k_nl_mark   : start[] / end[] of every sequence line
//   k_seq_class : per record: strip '\r', length filter (--minimum-length), width class x has-an-N, byte check;
//                 per-block class counts                          (then scans -> stable positions per group)
//   k_seq_place : record -> slot of its group's index list, in input order; kept rank = index among kept reads
// The groups are then packed by k_pack straight from the text.
// ------------------------------------------------------------------------------------------
#define MIRGE_PARSE_TILE (MIRGE_BLOCK * 16)
#define MIRGE_CLS_DROP 6  // shorter than --minimum-length (or longer than the engine's limit: flagged)

__device__ __forceinline__ uint32_t tile_newlines(const uint8_t* __restrict__ text, uint64_t n, uint64_t b0, uint32_t& mask) {
    mask = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) if (b0 + i < n && text[b0 + i] == 10) mask |= 1u << i;
    return (uint32_t)__popc(mask);
}

__global__ void k_nl_count(const uint8_t* __restrict__ text, uint64_t n, uint32_t* __restrict__ tile_cnt) {
    __shared__ uint32_t lds4[MIRGE_BLOCK / 64];
    uint32_t mask;
    const uint32_t c = tile_newlines(text, n, (uint64_t)blockIdx.x * MIRGE_PARSE_TILE + threadIdx.x * 16ull, mask);
    uint32_t total;
    (void)block_excl_scan(c, total, lds4);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

__global__ void k_nl_mark(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ tile_off, int period,
                          int sphase, int64_t* __restrict__ start, int64_t* __restrict__ end, uint64_t n_seq) {
    __shared__ uint32_t lds4[MIRGE_BLOCK / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * MIRGE_PARSE_TILE + threadIdx.x * 16ull;
    uint32_t mask;
    const uint32_t c = tile_newlines(text, n, b0, mask);
    uint32_t total;
    uint64_t li = (uint64_t)tile_off[blockIdx.x] + block_excl_scan(c, total, lds4);
    if (blockIdx.x == 0 && threadIdx.x == 0 && sphase == 0 && n_seq) start[0] = 0;
    while (mask) {
        const int i = __ffs(mask) - 1;
        mask &= mask - 1;
        const uint64_t pos = b0 + i;
        if ((int)(li % (uint64_t)period) == sphase && li / period < n_seq) end[li / period] = (int64_t)pos;
        if ((int)((li + 1) % (uint64_t)period) == sphase && (li + 1) / period < n_seq) start[(li + 1) / period] = (int64_t)pos + 1;
        li++;
    }
}

// flags: [0] reads with N seen per group ... kept by k_pack; here [0] = byte outside ACGTUN seen, [1] = reads longer
// than the limit, [2] = longest such read
__global__ void k_seq_class(const uint8_t* __restrict__ text, const int64_t* __restrict__ start, int64_t* __restrict__ end,
                            uint32_t n_seq, int32_t min_len, uint8_t* __restrict__ cls, uint32_t* __restrict__ blk_cls,
                            uint32_t* __restrict__ blk_keep, uint32_t nblk, uint32_t* __restrict__ hist, uint32_t* __restrict__ flags) {
    __shared__ uint32_t s_cnt[8];
    __shared__ uint32_t s_hist[MIRGE_MAX_READ_LEN + 1];
    if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;
    for (int i = threadIdx.x; i <= MIRGE_MAX_READ_LEN; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    int c = 7;  // no record
    if (r < n_seq) {
        const int64_t b = start[r];
        int64_t e = end[r];
        if (e > b && text[e - 1] == 13) { e--; end[r] = e; }
        const int64_t L = e - b;
        if (L > MIRGE_MAX_READ_LEN) {
            atomicOr(&flags[1], 1u);
            atomicMax(&flags[2], (uint32_t)(L > 0xFFFFFFF ? 0xFFFFFFF : L));
            c = MIRGE_CLS_DROP;
        } else if (L < (int64_t)min_len) {
            c = MIRGE_CLS_DROP;
        } else {
            uint32_t amb = 0, bad = 0;
            for (int p = 0; p < (int)L; p++) {
                const uint8_t ch = text[b + p] & 0xDF;
                const bool acgt = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'U';
                amb |= !acgt;
                bad |= !acgt && ch != 'N';
            }
            if (bad) atomicOr(&flags[0], 1u);
            c = (L <= 31 ? 0 : (L <= 64 ? 1 : 2)) + (amb ? 3 : 0);
            atomicAdd(&s_hist[L], 1u);
        }
        cls[r] = (uint8_t)c;
    }
#pragma unroll
    for (int q = 0; q < 7; q++) {
        const unsigned long long bal = __ballot(c == q);
        if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&s_cnt[q], (uint32_t)__popcll(bal));
    }
    __syncthreads();
    if (threadIdx.x < 6) blk_cls[(size_t)threadIdx.x * nblk + blockIdx.x] = s_cnt[threadIdx.x];
    if (threadIdx.x == 6) blk_keep[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3] + s_cnt[4] + s_cnt[5];
    for (int i = threadIdx.x; i <= MIRGE_MAX_READ_LEN; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

__global__ void k_seq_place(const uint8_t* __restrict__ cls, uint32_t n_seq, const uint32_t* __restrict__ cls_off,
                            const uint32_t* __restrict__ keep_off, uint32_t nblk, uint32_t* __restrict__ src_all,
                            uint32_t* __restrict__ orig_all) {
    __shared__ uint32_t lds4[MIRGE_BLOCK / 64];
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = r < n_seq ? (int)cls[r] : 7;
    uint32_t total;
    const uint32_t kr = keep_off[blockIdx.x] + block_excl_scan(c < MIRGE_CLS_DROP ? 1u : 0u, total, lds4);
    uint32_t slot = 0;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const uint32_t rk = block_excl_scan(c == q ? 1u : 0u, total, lds4);
        if (c == q) slot = cls_off[(size_t)q * nblk + blockIdx.x] + rk;
    }
    if (c < MIRGE_CLS_DROP) { src_all[slot] = r; orig_all[slot] = kr; }
}

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
                                  const uint32_t* __restrict__ orig, uint32_t base, int32_t S) {
    __shared__ unsigned long long c_key[MIRGE_CELL_CACHE];
    __shared__ uint32_t c_min[MIRGE_CELL_CACHE];
    __shared__ uint32_t c_cnt[MIRGE_CELL_CACHE];
    for (uint32_t i = threadIdx.x; i < MIRGE_CELL_CACHE; i += blockDim.x) { c_key[i] = 0ull; c_min[i] = 0xFFFFFFFFu; c_cnt[i] = 0; }
    __syncthreads();
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < g.n; j += gridDim.x * blockDim.x) {
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
        const int32_t sid = sample_ids ? sample_ids[orig ? orig[j] : base + j] : 0;
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
                atomicAdd(&c_cnt[cs], 1u);
                cached = true;
                break;
            }
            cs = (cs + 1) & (MIRGE_CELL_CACHE - 1);
        }
        if (!cached) {
            atomicMin(&firstj[s], j);
            atomicAdd(&cnt[(size_t)s * S + sid], 1u);
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

// ------------------------------------------------------------------------------------------
// Partitioned collapse for the key path (<=31 nt, no N, one sample).  Scattered device-scope atomics
// run at ~20 G/s chip-wide, which is what bounds k_collapse_insert_key (2.4 atomics per read).  Here
// equal keys are first brought together: reads are partitioned by the top bits of their hash into
// buckets of ~1-2 k reads (histogram per workgroup -> column prefix -> scatter, no global atomics),
// then ONE workgroup de-duplicates a bucket entirely in LDS (ds_cmpst / ds_min / ds_add) and emits the
// bucket's distinct reads with their counts.
//   k_part_agg    : per workgroup chunk: LDS cache merges equal reads -> records {key, min j, count};
//                   hist[g][b] = records of chunk g that fall into bucket b
//   k_part_prefix : off[g][b]   = sum over g' < g of hist[g'][b];  total[b] = column sum
//   (k_scan_blocksums over total[] -> bucket_start[])
//   k_part_scatter: part[bucket_start[b] + off[g][b] + local cursor] = record (16 B)
//   k_part_dedup  : per bucket, LDS table (key -> min j, count); the bucket's distinct reads are written
//                   to the output at a range reserved with one global atomicAdd per workgroup (so the order
//                   of the unique reads of this path is unspecified; first[] carries the first raw index)
// ------------------------------------------------------------------------------------------
#define MIRGE_PART_CAP 4096  // largest LDS table per bucket (16 B per slot = 64 KiB)

__device__ __forceinline__ unsigned long long read_key64(const GroupView<1>& g, uint32_t j) {
    return g.seq[j] | (1ull << (2 * g.len[j]));
}

// k_part_agg: a workgroup walks its chunk of reads through a small LDS cache (key -> min index,
// count) before anything is partitioned.  Real small-RNA samples are extremely skewed (one miRNA can be
// a third of all reads): without this the hot key's bucket holds millions of records for ONE workgroup
// and every LDS atomic on it is a 64-way conflict (measured on a Zipf sample: 10.9 ms per step against
// 2.8 ms on unskewed reads).  With it a key contributes at most one record per workgroup.  The cache is
// best effort: a read that finds no slot within 4 probes is emitted as a record of count 1.
// Output: recs[blockIdx * chunk ...] (compacted, nrec[blockIdx] of them) and hist[blockIdx][bucket].
// One workgroup per CU (the LDS cache + histogram take most of a CU's LDS), so the workgroup itself must bring
// the waves that hide its load and LDS latencies: 1024 threads = 16 waves per CU (256 threads: 0.26 ms, 2x slower)
#ifndef MIRGE_PART_THREADS
#define MIRGE_PART_THREADS 1024
#endif
__global__ void __launch_bounds__(MIRGE_PART_THREADS)
k_part_agg(GroupView<1> g, const uint32_t* __restrict__ orig, uint32_t base, uint32_t chunk, uint32_t bshift, uint32_t B,
           uint32_t CS, uint4* __restrict__ recs, uint32_t* __restrict__ nrec, uint32_t* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds_a[];  // [CS] keys | [CS] minj | [CS] cnt | [B] hist | cursor
    uint32_t* c_min = reinterpret_cast<uint32_t*>(lds_a + CS);
    uint32_t* c_cnt = c_min + CS;
    uint32_t* lds_h = c_cnt + CS;
    uint32_t& cursor = lds_h[B];
    for (uint32_t i = threadIdx.x; i < CS; i += blockDim.x) { lds_a[i] = 0ull; c_min[i] = 0xFFFFFFFFu; c_cnt[i] = 0; }
    for (uint32_t b = threadIdx.x; b <= B; b += blockDim.x) lds_h[b] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * chunk, hi = min(lo + chunk, g.n);
    uint4* out = recs + (size_t)blockIdx.x * chunk;
    const int lane = threadIdx.x & 63;
    for (uint32_t j0 = lo; j0 < hi; j0 += blockDim.x) {
        const uint32_t j = j0 + threadIdx.x;
        bool direct = false;
        unsigned long long key = 0ull;
        uint64_t h = 0;
        uint32_t jr = 0;  // index among ALL raw reads (orig[] ascends with j, so min commutes; read coalesced here
                          // instead of gathered per unique read at the end)
        if (j < hi) {
            key = read_key64(g, j);
            jr = orig ? orig[j] : base + j;
            h = mirge_mix64(key);
            uint32_t s = (uint32_t)(h >> 9) & (CS - 1);
            direct = true;
            for (int t = 0; t < 4; t++) {
                unsigned long long cur = lds_a[s];
                if (cur == 0ull) cur = atomicCAS(&lds_a[s], 0ull, key);
                if (cur == 0ull || cur == key) {
                    atomicMin(&c_min[s], jr);
                    atomicAdd(&c_cnt[s], 1u);
                    direct = false;
                    break;
                }
                s = (s + 1) & (CS - 1);
            }
        }
        const unsigned long long bal = __ballot(direct);  // cache full around this key: emit the read itself
        if (bal) {
            uint32_t wb = 0;
            if (lane == 0) wb = atomicAdd(&cursor, (uint32_t)__popcll(bal));
            wb = __shfl(wb, 0, 64);
            if (direct) {
                out[wb + __popcll(bal & ((1ull << lane) - 1ull))] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), jr, 1u);
                atomicAdd(&lds_h[(uint32_t)(h >> bshift)], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i0 = 0; i0 < CS; i0 += blockDim.x) {  // flush the cache
        const uint32_t i = i0 + threadIdx.x;
        const unsigned long long key = i < CS ? lds_a[i] : 0ull;
        const bool has = key != 0ull;
        const unsigned long long bal = __ballot(has);
        if (bal) {
            uint32_t wb = 0;
            if (lane == 0) wb = atomicAdd(&cursor, (uint32_t)__popcll(bal));
            wb = __shfl(wb, 0, 64);
            if (has) {
                out[wb + __popcll(bal & ((1ull << lane) - 1ull))] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), c_min[i], c_cnt[i]);
                atomicAdd(&lds_h[(uint32_t)(mirge_mix64(key) >> bshift)], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) hist[(size_t)blockIdx.x * B + b] = lds_h[b];
    if (threadIdx.x == 0) nrec[blockIdx.x] = cursor;
}

__global__ void k_part_prefix(const uint32_t* __restrict__ hist, uint32_t G, uint32_t B, uint32_t* __restrict__ off,
                              uint32_t* __restrict__ total) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t run = 0;
#pragma unroll 8
    for (uint32_t gq = 0; gq < G; gq++) {
        const uint32_t v = hist[(size_t)gq * B + b];
        off[(size_t)gq * B + b] = run;
        run += v;
    }
    total[b] = run;
}

__global__ void __launch_bounds__(MIRGE_PART_THREADS)
k_part_scatter(const uint4* __restrict__ recs, const uint32_t* __restrict__ nrec, uint32_t chunk,
                               uint32_t bshift, uint32_t B, const uint32_t* __restrict__ off,
                               const uint32_t* __restrict__ bucket_start, uint4* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_c[];
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) lds_c[b] = bucket_start[b] + off[(size_t)blockIdx.x * B + b];
    __syncthreads();
    const uint4* in = recs + (size_t)blockIdx.x * chunk;
    const uint32_t n = nrec[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint4 rec = in[i];
        const unsigned long long key = ((unsigned long long)rec.y << 32) | rec.x;
        const uint32_t p = atomicAdd(&lds_c[(uint32_t)(mirge_mix64(key) >> bshift)], 1u);
        part[p] = rec;  // {key, min index, count}: one 16-B store per record
    }
}

// CAP = LDS table slots (16 B each): 2048 when the buckets hold <= 1024 records (4 workgroups per CU), else 4096
// The table takes 32-64 KiB of LDS, so only 2-4 workgroups fit a CU: 1024-thread workgroups bring the waves.
#define MIRGE_DEDUP_THREADS 1024
template <int CAP>
__global__ void __launch_bounds__(MIRGE_DEDUP_THREADS)
k_part_dedup(const uint4* __restrict__ part, const uint32_t* __restrict__ bucket_start,
             uint64_t* __restrict__ useq, uint8_t* __restrict__ ulen,
             uint32_t* __restrict__ ucnt, uint32_t* __restrict__ ufirst, uint32_t* __restrict__ cursor,
             uint32_t* __restrict__ hist, uint32_t* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds_k[];  // [CAP] keys, then [CAP] minj, [CAP] cnt
    uint32_t* lds_min = reinterpret_cast<uint32_t*>(lds_k + CAP);
    uint32_t* lds_cnt = lds_min + CAP;
    uint32_t* lds_x = lds_cnt + CAP;  // [0] distinct keys, [1] output base, [2..17] scan scratch, [32..32+128] lengths
    uint32_t& n_distinct = lds_x[0];
    for (uint32_t i = threadIdx.x; i < CAP; i += blockDim.x) { lds_k[i] = 0ull; lds_min[i] = 0xFFFFFFFFu; lds_cnt[i] = 0; }
    for (uint32_t i = threadIdx.x; i < 32 + MIRGE_MAX_READ_LEN + 1; i += blockDim.x) lds_x[i] = 0;
    __syncthreads();
    const uint32_t lo = bucket_start[blockIdx.x], hi = bucket_start[blockIdx.x + 1];
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint4 rec = part[i];
        const unsigned long long key = ((unsigned long long)rec.y << 32) | rec.x;
        const uint32_t j = rec.z;
        uint32_t s = (uint32_t)(mirge_mix64(key) >> 7) & (CAP - 1);
        while (true) {
            unsigned long long cur = lds_k[s];
            if (cur == 0ull) {
                cur = atomicCAS(&lds_k[s], 0ull, key);
                if (cur == 0ull && atomicAdd(&n_distinct, 1u) >= CAP - 64) atomicOr(overflow, 1u);
            }
            if (cur == 0ull || cur == key) break;
            if (*(volatile uint32_t*)&n_distinct >= CAP - 32) break;  // table full: flagged, results discarded
            s = (s + 1) & (CAP - 1);
        }
        atomicMin(&lds_min[s], j);
        atomicAdd(&lds_cnt[s], rec.w);  // a record stands for rec.w identical reads of one workgroup's chunk
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
    if (threadIdx.x == 0) lds_x[1] = total ? atomicAdd(cursor, total) : 0u;
    __syncthreads();
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
        atomicAdd(&lds_x[32 + L], 1u);
        rank++;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= MIRGE_MAX_READ_LEN; i += blockDim.x)
        if (lds_x[32 + i]) atomicAdd(&hist[i], lds_x[32 + i]);
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
                const uint32_t L = len[j];
                atomicAdd(&h[L > MIRGE_MAX_READ_LEN ? MIRGE_MAX_READ_LEN : L], 1u);
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

// ------------------------------------------------------------------------------------------
// align_hybrid: the device form of mirge_align_indexed (same probes, same verification, same
// minimum) with the candidate lists balanced over the wave.
//   A probe's bucket holds from 0 to thousands of candidate windows (a 16-nt read under -v 2 is
//   probed with 4-mers: ~220 candidates per probe in a 57 kb library, while a 28-nt read sees ~1).
//   Lane-serial evaluation makes the whole wave wait for its unluckiest lane and walks each list
//   as a chain of dependent loads.  Here a lane verifies only short lists (<= MIRGE_LIGHT) itself;
//   longer lists are taken one at a time by the whole wave: the owner's read is broadcast with
//   v_readlane, the 64 lanes stride through the bucket (coalesced pos[] loads, 64 windows
//   verified per step); the few lanes that found a valid window are read back with v_readlane and
//   their minimum goes to the owner.
//   All 64 lanes of the wave must call this together (inactive lanes pass active = false).
// ------------------------------------------------------------------------------------------
#ifndef MIRGE_LIGHT
#define MIRGE_LIGHT 4
#endif
#ifndef MIRGE_LIGHT_MAX
#define MIRGE_LIGHT_MAX 16
#endif
#define MIRGE_COOP_UNROLL 1

// pointers that came out of memory or a v_readlane have lost their address space; these casts keep
// the loads global_load_* (not flat_load_*, which also ties up lgkmcnt)
typedef const __attribute__((address_space(1))) uint32_t* gptr_u32;
typedef const __attribute__((address_space(1))) uint64_t* gptr_u64;

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src) {
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, src);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// Consecutive passes with one and the same policy (snoRNA / rRNA / other ncRNA: all "-n 1") run as ONE
// pass over the concatenation of their libraries.  The cascade's priority -- the first library with
// ANY valid hit wins, whatever its mismatch count -- is kept by ranking candidates on
// (library class, mismatches, position): class = which member's text the window lies in.
struct MergeInfo {
    uint32_t bound[4];  // global position where member c's text starts (bound[0] = 0)
    int32_t n;          // members (1 = ordinary pass)
};
__device__ __forceinline__ uint64_t class_key(const MergeInfo& mi, uint64_t g) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 1; i < 4; i++) c += (i < mi.n && g >= mi.bound[i]) ? 1u : 0u;
    return c << 40;
}

// the two text words under a window: issued for several candidates before any is consumed
struct TextWin { uint64_t w[5]; };

template <int W>
__device__ __forceinline__ void load_window(gptr_u64 T, uint64_t g, int L, TextWin& tw) {
    const uint64_t q = g >> 5;
#pragma unroll
    for (int i = 0; i <= W; i++) tw.w[i] = (i == 0 || 32 * (i - 1) < L) ? T[q + i] : 0ull;
}

// same arithmetic as mirge_window_mm, on words that are already in registers
template <int W>
__device__ __forceinline__ int window_mm_regs(const TextWin& tw, uint64_t g, const MirgeRead<W>& r,
                                              const MirgePolicy& p) {
    const int L = r.len;
    const int s = (int)(g & 31) * 2;
    int tot = 0, seedmm = 0;
    const int seed = p.mode == 0 ? (L < p.seedlen ? L : p.seedlen) : L;
#pragma unroll
    for (int i = 0; i < W; i++) {
        if (32 * i < L) {
            const uint64_t t = s ? ((tw.w[i] >> s) | (tw.w[i + 1] << (64 - s))) : tw.w[i];
            const uint64_t x = r.w[i] ^ t;
            uint64_t m = (x | (x >> 1)) & 0x5555555555555555ull;
            const int rem = L - 32 * i;
            m &= mirge_lowmask2(rem > 32 ? 32 : rem);
            m |= r.nm[i];
            tot += mirge_popc(m);
            const int srem = seed - 32 * i;
            if (srem > 0) seedmm += mirge_popc(m & mirge_lowmask2(srem > 32 ? 32 : srem));
        }
    }
    if (tot > p.maxtotal || seedmm > p.mm) return -1;
    return tot;
}

// verify up to N candidate positions at once: all pos loads, then all text loads, then arithmetic
template <int W, int N>
__device__ __forceinline__ uint64_t eval_batch(const MirgeLibView& lib, const MirgePolicy& pol, const MergeInfo& mi,
                                               const MirgeRead<W>& r, gptr_u32 pos, const uint32_t (&c)[N],
                                               uint32_t hi, int a) {
    uint32_t pz[N];
    bool ok[N];
#pragma unroll
    for (int u = 0; u < N; u++) {
        ok[u] = c[u] < hi;
        pz[u] = ok[u] ? pos[c[u]] : 0u;
    }
    TextWin tw[N];
    uint64_t g[N];
#pragma unroll
    for (int u = 0; u < N; u++) {
        ok[u] = ok[u] && pz[u] >= (uint32_t)a;
        g[u] = ok[u] ? (uint64_t)pz[u] - (uint64_t)a : 0ull;
        load_window<W>((gptr_u64)lib.T, g[u], ok[u] ? r.len : 0, tw[u]);
    }
    uint64_t best = MIRGE_NO_HIT;
#pragma unroll
    for (int u = 0; u < N; u++) {
        if (!ok[u]) continue;
        const int m = window_mm_regs<W>(tw[u], g[u], r, pol);
        if (m < 0) continue;
        if (mirge_window_invalid(lib.inv, g[u], r.len)) continue;
        const uint64_t cand = class_key(mi, g[u]) | ((uint64_t)m << 32) | g[u];
        if (cand < best) best = cand;
    }
    return best;
}

template <int W>
__device__ __forceinline__ void align_hybrid(const MirgeLibView& lib, const MirgePolicy& pol, const MergeInfo& mi,
                                             const MirgePlanTable* __restrict__ plan, const MirgeRead<W>& r,
                                             bool active, uint64_t& best) {
    best = MIRGE_NO_HIT;
    const int lane = threadIdx.x & 63;
    const int np = active ? (int)plan->np[r.len] : 0;
    // wave-uniform bound on the probe count: (mm+1) plain segments or (mm+1)^2 recursive probes
    const int npmax = (pol.mm >= 1 && pol.mm <= 2) ? (pol.mm + 1) * (pol.mm + 1) : pol.mm + 1;
#pragma unroll 1
    for (int q = 0; q < npmax; q++) {
        uint32_t lo = 0, hi = 0;
        int a = 0;
        gptr_u32 pos = nullptr;
        if (active && q < np) {
            const MirgeProbe pr = plan->pr[r.len][q];  // tabulated mirge_probe_at(pol, len, K, q)
            uint64_t key;
            if (mirge_probe_key<W>(r, pr, key)) {  // no ambiguous call inside the probe
                const MirgeKTable tb = lib.tables[mirge_shape_id(pr.k1, pr.gap, pr.k2)];
                gptr_u32 bits = (gptr_u32)tb.bits;
                if (!bits || ((bits[key >> 5] >> (key & 31)) & 1u)) {  // L2-resident "bucket is non-empty" bit
                    gptr_u32 bucket = (gptr_u32)tb.bucket;
                    pos = (gptr_u32)tb.pos;
                    lo = bucket[key];
                    hi = bucket[key + 1];
                    a = pr.a1;
                }
            }
        }
        // lists of up to MIRGE_LIGHT_MAX windows are verified by their own lane, MIRGE_LIGHT per batch
        // (a cooperative hand-over costs the whole wave ~150 instructions per list; with 5-15 windows
        // per list and many such lanes per probe the lane-serial batches are several times cheaper)
        const bool heavy = (hi - lo) > MIRGE_LIGHT_MAX;
        if (!heavy) {
            for (uint32_t c0 = lo; c0 < hi; c0 += MIRGE_LIGHT) {
                uint32_t c[MIRGE_LIGHT];
#pragma unroll
                for (int u = 0; u < MIRGE_LIGHT; u++) c[u] = c0 + u;
                const uint64_t cand = eval_batch<W, MIRGE_LIGHT>(lib, pol, mi, r, pos, c, hi, a);
                if (cand < best) best = cand;
            }
        }
        unsigned long long hb = __ballot(heavy);
        while (hb) {
            const int src = __ffsll(hb) - 1;
            hb &= hb - 1;
            MirgeRead<W> rr;
#pragma unroll
            for (int w = 0; w < W; w++) {
                rr.w[w] = readlane_u64(r.w[w], src);
                rr.nm[w] = readlane_u64(r.nm[w], src);
            }
            rr.len = __builtin_amdgcn_readlane(r.len, src);
            const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)lo, src);
            const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)hi, src);
            const int ba = __builtin_amdgcn_readlane(a, src);
            gptr_u32 bpos = (gptr_u32)readlane_u64((uint64_t)pos, src);
            uint64_t lbest = MIRGE_NO_HIT;
            for (uint32_t c0 = blo + lane; c0 < bhi; c0 += 64 * MIRGE_COOP_UNROLL) {
                uint32_t c[MIRGE_COOP_UNROLL];
#pragma unroll
                for (int u = 0; u < MIRGE_COOP_UNROLL; u++) c[u] = c0 + 64 * u;
                const uint64_t cand = eval_batch<W, MIRGE_COOP_UNROLL>(lib, pol, mi, rr, bpos, c, bhi, ba);
                if (cand < lbest) lbest = cand;
            }
            // almost every candidate fails verification: instead of a shuffle tree, visit the few
            // lanes that hold a hit (v_readlane -> scalar min)
            unsigned long long hits = __ballot(lbest != MIRGE_NO_HIT);
            uint64_t tbest = MIRGE_NO_HIT;
            while (hits) {
                const int hl = __ffsll(hits) - 1;
                hits &= hits - 1;
                const uint64_t v = readlane_u64(lbest, hl);
                if (v < tbest) tbest = v;
            }
            if (lane == src && tbest < best) best = tbest;
        }
        // a 0-mismatch window (of the first member library) is in probe 0's bucket and buckets ascend:
        // nothing later can beat it
        if (q == 0 && (best >> 32) == 0) active = false;
    }
}

// ------------------------------------------------------------------------------------------
// k_pass: one cascade pass over the still-unannotated reads of one width group.
//   Every workgroup owns a fixed segment of `cap` slots.  First pass (act_in == nullptr):
//   workgroup b takes the contiguous reads [b*cap, (b+1)*cap).  Later passes: workgroup b takes
//   the seg_n_in[b] survivors its own previous pass left in act_in[b*cap ...].
//   A hit writes (pass, global position, mismatches) at the read's slot; every other read (not
//   selected by the pass's subset rule, skipped by bowtie, or unaligned) is appended to the
//   workgroup's segment of act_out, so the next pass sees exactly the rows with annotFlag == 0
//   (manifoldAlign.py:120,129).  The append needs no global atomic: one LDS counter per
//   workgroup, one ds_add per wave (ballot + prefix popcount).  A single global cursor was
//   measured at ~0.35 ms per pass for 2 M reads (33 k same-address returning atomics).
// ------------------------------------------------------------------------------------------
// SLOT is the pass index and only names the symbol (k_pass<1,6> ...), so that rocprofv3's per-kernel
// statistics separate the passes; the policy itself stays a run-time argument.
#ifndef MIRGE_PASS_MIN_WAVES
#define MIRGE_PASS_MIN_WAVES 1
#endif
template <int W, int SLOT>
__global__ void __launch_bounds__(MIRGE_BLOCK, MIRGE_PASS_MIN_WAVES)
k_pass(MirgeLibView lib, MirgePolicy pol, MergeInfo mi, const MirgePlanTable* __restrict__ plan, GroupView<W> g,
       const uint32_t* __restrict__ act_in,
       const uint32_t* __restrict__ seg_n_in, uint32_t* __restrict__ act_out, uint32_t* __restrict__ seg_n_out,
       uint32_t cap, int32_t pass_id, int8_t* __restrict__ res_pass, uint32_t* __restrict__ res_pos,
       int8_t* __restrict__ res_mm) {
    __shared__ uint32_t s_count;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const size_t seg = (size_t)blockIdx.x * cap;
    uint32_t n_in;
    if (act_in) n_in = seg_n_in[blockIdx.x];
    else n_in = seg < g.n ? (uint32_t)((g.n - seg) < cap ? (g.n - seg) : cap) : 0u;
    const int lane = threadIdx.x & 63;
    for (uint32_t base = 0; base < n_in; base += MIRGE_BLOCK) {
        const uint32_t t = base + threadIdx.x;
        const bool valid = t < n_in;
        bool survivor = false;
        uint32_t idx = 0;
        if (valid) {
            idx = act_in ? act_in[seg + t] : (uint32_t)seg + t;
            survivor = true;
        }
        // the wave aligns its 64 reads together (align_hybrid balances the candidate lists)
        MirgeRead<W> r2;
        bool elig = false;
        if (valid) {
            load_read<W>(g, idx, r2);
            elig = mirge_effective_read<W>(r2, pol);
        } else {
#pragma unroll
            for (int w = 0; w < W; w++) { r2.w[w] = 0; r2.nm[w] = 0; }
            r2.len = 0;
        }
        uint64_t best;
        align_hybrid<W>(lib, pol, mi, plan, r2, elig, best);
        if (elig && best != MIRGE_NO_HIT) {
            const int cls = (int)(best >> 40);  // member library of a merged pass (0 otherwise)
            res_pass[idx] = (int8_t)(pass_id + cls);
            uint32_t b0 = 0;
#pragma unroll
            for (int i = 1; i < 4; i++) if (i == cls) b0 = mi.bound[i];
            res_pos[idx] = (uint32_t)best - b0;  // position in that member's own text
            res_mm[idx] = (int8_t)((best >> 32) & 0xFF);
            survivor = false;
        }
        const unsigned long long bal = __ballot(survivor);
        if (bal) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&s_count, (uint32_t)__popcll(bal));
            wbase = __shfl(wbase, 0, 64);
            if (survivor) act_out[seg + wbase + __popcll(bal & ((1ull << lane) - 1ull))] = idx;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) seg_n_out[blockIdx.x] = s_count;
}

// global position -> (reference index, offset) by binary search in ref_start of the pass's library
struct ResolveTable {
    const uint32_t* ref_start[MIRGE_MAX_PASSES_K];
    uint32_t n_refs[MIRGE_MAX_PASSES_K];
};

__device__ __forceinline__ void resolve_one(const ResolveTable& tb, int p, uint32_t g, int32_t& ref, int32_t& off) {
    const uint32_t* rs = nullptr;
    uint32_t nr = 0;
#pragma unroll
    for (int q = 0; q < MIRGE_MAX_PASSES_K; q++)
        if (q == p) { rs = tb.ref_start[q]; nr = tb.n_refs[q]; }
    uint32_t lo = 0, hi = nr;  // last t with rs[t] <= g
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (rs[mid] <= g) lo = mid; else hi = mid;
    }
    ref = (int32_t)lo;
    off = (int32_t)(g - rs[lo]);
}

__global__ void k_resolve(ResolveTable tb, const int8_t* __restrict__ res_pass, const uint32_t* __restrict__ res_pos,
                          uint32_t n, int32_t* __restrict__ res_ref, int32_t* __restrict__ res_off) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int p = res_pass[i];
        int32_t ref = -1, off = -1;
        if (p >= 0) resolve_one(tb, p, res_pos[i], ref, off);
        res_ref[i] = ref;
        res_off[i] = off;
    }
}

// ------------------------------------------------------------------------------------------
// k_cascade_fused: the whole cascade of a SMALL read group in one launch.  The groups beside the bulk
// (reads with an N, 32-128 nt reads: a few hundred to a few 100 k reads) cost one launch per pass plus
// memsets and a resolve each -- ~30 launches whose host-side enqueue time, not their GPU time, kept the
// bulk group's first pass waiting (profiles/r01_timeline.txt).  Here a wave keeps its 64 reads through
// every step; a read that is annotated simply stops being eligible (no compaction: the group is small).
// Same device functions as k_pass, so the same answers.
// ------------------------------------------------------------------------------------------
struct FusedStep {
    MirgeLibView lib;
    MirgePolicy pol;
    MergeInfo mi;
    const MirgePlanTable* plan;
    int32_t pass_id;
};
struct FusedSteps {
    int32_t n;
    FusedStep s[MIRGE_MAX_PASSES_K];
};

template <int W>
__global__ void __launch_bounds__(MIRGE_BLOCK)
k_cascade_fused(const FusedSteps* __restrict__ steps, ResolveTable tb, GroupView<W> g, int8_t* __restrict__ res_pass,
                uint32_t* __restrict__ res_pos, int8_t* __restrict__ res_mm, int32_t* __restrict__ res_ref,
                int32_t* __restrict__ res_off) {
    const uint32_t nrounds = (g.n + MIRGE_BLOCK - 1) / MIRGE_BLOCK;
    const int nsteps = steps->n;
    for (uint32_t round = blockIdx.x; round < nrounds; round += gridDim.x) {
        const uint32_t idx = round * MIRGE_BLOCK + threadIdx.x;
        const bool valid = idx < g.n;
        MirgeRead<W> r0;
        if (valid) load_read<W>(g, idx, r0);
        else {
#pragma unroll
            for (int w = 0; w < W; w++) { r0.w[w] = 0; r0.nm[w] = 0; }
            r0.len = 0;
        }
        bool open = valid;
        int8_t o_pass = -1, o_mm = -1;
        uint32_t o_pos = 0;
        for (int si = 0; si < nsteps; si++) {
            if (!__ballot(open)) break;  // wave-uniform
            const FusedStep& st = steps->s[si];
            MirgeRead<W> r2 = r0;
            const bool elig = open && mirge_effective_read<W>(r2, st.pol);
            uint64_t best;
            align_hybrid<W>(st.lib, st.pol, st.mi, st.plan, r2, elig, best);
            if (elig && best != MIRGE_NO_HIT) {
                const int cls = (int)(best >> 40);
                o_pass = (int8_t)(st.pass_id + cls);
                uint32_t b0 = 0;
#pragma unroll
                for (int i = 1; i < 4; i++) if (i == cls) b0 = st.mi.bound[i];
                o_pos = (uint32_t)best - b0;
                o_mm = (int8_t)((best >> 32) & 0xFF);
                open = false;
            }
        }
        if (valid) {
            int32_t ref = -1, off = -1;
            if (o_pass >= 0) resolve_one(tb, o_pass, o_pos, ref, off);
            res_pass[idx] = o_pass; res_pos[idx] = o_pos; res_mm[idx] = o_mm;
            res_ref[idx] = ref; res_off[idx] = off;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_join: count join (summary.py:686-698,749-752): class_sums[pass][s] += counts[i][s],
// exact/iso[ref][s] += counts[i][s] for the two miRNA passes.  Class sums are accumulated in LDS
// per workgroup and flushed with one atomic per (pass, sample) cell.
// ------------------------------------------------------------------------------------------
#define MIRGE_JOIN_LDS 2048
__global__ void k_join(const int8_t* __restrict__ res_pass, const int32_t* __restrict__ res_ref,
                       const uint32_t* __restrict__ counts, uint32_t n, int32_t S, int32_t n_pass,
                       int32_t exact_pass, int32_t iso_pass, unsigned long long* __restrict__ class_sums,
                       unsigned long long* __restrict__ exact, unsigned long long* __restrict__ iso) {
    __shared__ unsigned long long acc[MIRGE_JOIN_LDS];
    const int cells = n_pass * S;
    const bool use_lds = cells <= MIRGE_JOIN_LDS;
    if (use_lds) {
        for (int c = threadIdx.x; c < cells; c += blockDim.x) acc[c] = 0ull;
        __syncthreads();
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int p = res_pass[i];
        if (p < 0) continue;
        const int32_t ref = res_ref[i];
        for (int32_t s = 0; s < S; s++) {
            const unsigned long long c = counts[(size_t)i * S + s];
            if (!c) continue;
            if (use_lds) atomicAdd(&acc[p * S + s], c);
            else atomicAdd(&class_sums[p * S + s], c);
            if (p == exact_pass) atomicAdd(&exact[(size_t)ref * S + s], c);
            else if (p == iso_pass) atomicAdd(&iso[(size_t)ref * S + s], c);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int c = threadIdx.x; c < cells; c += blockDim.x)
            if (acc[c]) atomicAdd(&class_sums[c], acc[c]);
    }
}

// out[orig[j] or base+j] = in[j]
template <typename T>
__global__ void k_scatter_out(const T* __restrict__ in, uint32_t n, uint32_t base,
                              const uint32_t* __restrict__ orig, T* __restrict__ out) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        out[orig ? orig[j] : base + j] = in[j];
}



// ------------------------------------------------------------------------------------------
// k_tally (BASELINE config 5): per-position base-change tally of the reads
// annotated to a miRNA (exact pass or isomiR pass) against that miRNA's canonical sequence -- the
// arithmetic of A2IEditing / judgeAllign (mirge/libs/mirge2_tRF_a2i.py:298-366) on the cascade's
// ungapped alignment instead of Bio.pairwise2's.
//   d = offset of read base 0 relative to canonical base 0 (negative: the read starts before it).
//   judgeAllign (:298-332), literally: reject if d > 1; walk the aligned columns from the canonical's
//   first base to min(end_pos1, end_pos2) (end_pos1 = aligned length - head dashes of the target - 1 - 3,
//   end_pos2 = last read base), count matches and mismatches (a column past the canonical's end is a
//   mismatch, a column before the read's first base is skipped); accept iff mismatches <= 1 and
//   matches >= Lc - 4 (- 1 more if d == 1).
//   Accepted reads add their counts to accepted[ref][s], to canonical[ref][s] when the read is an
//   exact substring of the canonical (:350-351), and to census[ref][q][canon base*4 + read base][s]
//   for every canonical position q they cover (A->G at q < Lc-5 is the A-to-I count, :358-366).
// ------------------------------------------------------------------------------------------
#define MIRGE_TALLY_MAXPOS 32
__global__ void k_tally(GroupView<1> g, const int8_t* __restrict__ res_pass, const int32_t* __restrict__ res_ref,
                        const int32_t* __restrict__ res_off, const uint32_t* __restrict__ counts, int32_t S,
                        MirgeLibView lib, int32_t exact_pass, int32_t iso_pass, int32_t iso_trim5,
                        unsigned long long* __restrict__ accepted, unsigned long long* __restrict__ canonical,
                        unsigned long long* __restrict__ census) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x) {
        const int p = res_pass[i];
        if (p != exact_pass && p != iso_pass) continue;
        const int32_t r = res_ref[i];
        const int d = res_off[i] - (p == iso_pass ? iso_trim5 : 0);
        const uint32_t rs = lib.ref_start[r];
        const int Lc = (int)(lib.ref_start[r + 1] - rs) - 1;  // minus the separator
        const int Lr = g.len[i];
        const uint64_t rw = g.seq[i];
        const uint64_t rn = g.nmask ? g.nmask[i] : 0ull;
        if (d > 1) continue;
        const int hd_t = d < 0 ? -d : 0, hd_s = d > 0 ? d : 0;
        const int A = max(hd_t + Lc, hd_s + Lr);
        const int end1 = A - hd_t - 1 - 3, end2 = hd_s + Lr - 1;
        const int last = min(end1, end2);
        int mism = 0, match = 0;
        for (int pos = hd_t; pos <= last; pos++) {
            const int ri = pos - hd_s, q = pos - hd_t;
            if (ri < 0) continue;
            bool eq = false;
            if (q < Lc && !((rn >> (2 * ri)) & 1ull)) {
                const uint64_t gq = (uint64_t)rs + (uint64_t)q;
                eq = ((lib.T[gq >> 5] >> (2 * (gq & 31))) & 3ull) == ((rw >> (2 * ri)) & 3ull);
            }
            if (eq) match++; else mism++;
        }
        const int match_limit = Lc - 3 - 1 - (d == 1 ? 1 : 0);
        if (mism > 1 || match < match_limit) continue;
        // exact substring of the canonical?
        bool sub = d >= 0 && d + Lr <= Lc && rn == 0ull;
        for (int ri = 0; sub && ri < Lr; ri++) {
            const uint64_t gq = (uint64_t)rs + (uint64_t)(d + ri);
            sub = ((lib.T[gq >> 5] >> (2 * (gq & 31))) & 3ull) == ((rw >> (2 * ri)) & 3ull);
        }
        for (int32_t s = 0; s < S; s++) {
            const unsigned long long c = counts[(size_t)i * S + s];
            if (!c) continue;
            atomicAdd(&accepted[(size_t)r * S + s], c);
            if (sub) atomicAdd(&canonical[(size_t)r * S + s], c);
            for (int ri = max(0, -d); ri < Lr; ri++) {
                const int q = d + ri;
                if (q >= Lc || q >= MIRGE_TALLY_MAXPOS) break;
                if ((rn >> (2 * ri)) & 1ull) continue;  // an N call is no base change
                const uint64_t gq = (uint64_t)rs + (uint64_t)q;
                const int cb = (int)((lib.T[gq >> 5] >> (2 * (gq & 31))) & 3ull);
                const int rb = (int)((rw >> (2 * ri)) & 3ull);
                atomicAdd(&census[(((size_t)r * MIRGE_TALLY_MAXPOS + q) * 16 + cb * 4 + rb) * S + s], c);
            }
        }
    }
}
