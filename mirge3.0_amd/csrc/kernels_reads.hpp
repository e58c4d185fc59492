// kernels_reads.hpp -- part of mirge_kernels.hpp: pack / unpack, probe-table construction, block scan, text parsing.
#pragma once
// ------------------------------------------------------------------------------------------
// k_pack: ASCII -> 2-bit.  One thread per read; the read's bytes are contiguous in `ascii`.
// flags[0] |= 1 if any ambiguous call was seen (|= 2: an IUPAC code other than N among them), flags[1] |= 1 if a byte
// that is no nucleotide code was seen.
// ------------------------------------------------------------------------------------------
// IUPAC ambiguity codes other than N (upper case): bowtie turns every one of them into N, and so does the packing --
// the collapsed sequence then prints N where the read had the code (flagged, so that the host can say so)
// letter classes of an upper-cased sequence character as one bit test each (bit = letter - 'A'):
// A C G T U | N | the ten other IUPAC ambiguity codes (R Y S W K M B D H V)
#define MIRGE_LETTERS_ACGTU ((1u << 0) | (1u << 2) | (1u << 6) | (1u << 19) | (1u << 20))
#define MIRGE_LETTERS_N (1u << 13)
#define MIRGE_LETTERS_IUPAC ((1u << 17) | (1u << 24) | (1u << 18) | (1u << 22) | (1u << 10) | (1u << 12) | (1u << 1) | (1u << 3) | (1u << 7) | (1u << 21))
// ('.' -- the no-call of Illumina's old pipelines, which bowtie reads as N -- arrives here as 0x0E after the upper-casing mask and is
// taken for an ambiguity code: aligned and printed as N, the run warns as it does for IUPAC codes)
__device__ __forceinline__ uint32_t letter_bit(uint8_t upper) {
    const uint32_t d = (uint32_t)upper - 'A';
    return d < 26u ? 1u << d : (upper == ('.' & 0xDF) ? 1u << 17 : 0u);
}
__device__ __forceinline__ bool is_iupac_code(uint8_t c) { return (letter_bit(c) & MIRGE_LETTERS_IUPAC) != 0; }

template <int W>
__global__ void k_pack(const uint8_t* __restrict__ ascii, const int64_t* __restrict__ starts, const int64_t* __restrict__ ends,
                       const uint32_t* __restrict__ idx, uint32_t n, uint64_t* __restrict__ seq,
                       uint8_t* __restrict__ len, uint64_t* __restrict__ nmask,
                       uint32_t* __restrict__ flags, const int64_t* __restrict__ s2start = nullptr,
                       const int32_t* __restrict__ s2len = nullptr) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t src = idx[j];
        const int64_t b = starts[src];
        const int L1 = (int)(ends[src] - b);  // contiguous reads: ends = starts + 1
        // a record may continue in a second stretch of the text (--qiagenumi: insert + the UMI behind the adapter)
        const int L2 = s2len ? (int)s2len[src] : 0;
        const int64_t b2 = s2len ? s2start[src] - L1 : 0;
        const int L = L1 + L2;
        uint64_t w[W], nm[W];
#pragma unroll
        for (int i = 0; i < W; i++) { w[i] = 0; nm[i] = 0; }
        uint32_t sawN = 0, bad = 0, iupac = 0;
        for (int p = 0; p < L; p++) {
            const uint8_t c = (p < L1 ? ascii[b + p] : ascii[b2 + p]) & 0xDF;  // upper-case
            const uint32_t bit = letter_bit(c);
            const uint32_t x = (c >> 1) & 3u;       // A 0, C 1, G 3, T / U 2 ...
            const uint64_t isn = (bit & MIRGE_LETTERS_ACGTU) ? 0ull : 1ull;
            const uint64_t code = isn ? 0ull : (uint64_t)(x ^ (x >> 1));  // ... -> A 0, C 1, G 2, T / U 3
            iupac |= (bit & MIRGE_LETTERS_IUPAC) != 0;
            bad |= isn && !(bit & (MIRGE_LETTERS_N | MIRGE_LETTERS_IUPAC));
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
        if (sawN) atomicOr(&flags[0], 1u | (iupac << 1));
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

// Round 6: the position list of every bucket the wave-cooperative scan takes (more than MIRGE_LIGHT_MAX = 16 windows) in ASCENDING
// order.  With real libraries -- poly-A tails, Alu-derived elements, simple repeats -- such a bucket holds 10^3 .. 10^6 windows,
// and a scan in position order may stop at the first window nothing later can beat (verify_heavy: a hit without a mismatch is final,
// since candidates rank by class, mismatches, position and both class and position only grow from there).  The counting sort's
// fill leaves a bucket in the order its atomics happened to be served: the heavy buckets are listed here (begin / end into pos[])
// and sorted by one segmented radix sort per table.
__global__ void k_table_heavy_list(const uint32_t* __restrict__ bucket, uint64_t nb, uint32_t min_count, uint32_t* __restrict__ n_heavy,
                                   uint32_t cap, uint32_t* __restrict__ seg_begin, uint32_t* __restrict__ seg_end, uint32_t outlier_min) {
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < nb; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t lo = bucket[k], hi = bucket[k + 1];
        if (hi - lo < min_count) continue;
        const uint32_t at = atomicAdd(n_heavy, 1u);
        if (at < cap) { seg_begin[at] = lo; seg_end[at] = hi; }
        // the fullest OUTLIER bucket -- far above what a uniform text of this size puts into a bucket of this table (a table of
        // 1- or 2-base keys holds 10^4 windows per bucket by construction; a poly-A bucket of a 15-base table holds 10^6 because
        // the library repeats itself): what decides whether reads may be deferred to k_cascade_heavy (mirge_lib::max_bucket)
        if (hi - lo >= outlier_min) atomicMax(n_heavy + 1, hi - lo);
    }
}
// sorted stretches back into the table's own list (the segmented sort writes to a second array and leaves what lies outside the
// listed segments unwritten): one workgroup per segment in turn, coalesced
__global__ void k_table_heavy_copy(const uint32_t* __restrict__ seg_begin, const uint32_t* __restrict__ seg_end, uint32_t n_seg,
                                   const uint32_t* __restrict__ sorted, uint32_t* __restrict__ pos) {
    for (uint32_t sg = blockIdx.x; sg < n_seg; sg += gridDim.x)
        for (uint32_t i = seg_begin[sg] + threadIdx.x; i < seg_end[sg]; i += blockDim.x) pos[i] = sorted[i];
}

// CSR bounds + position list -> self-contained entries (MirgeKTable): {count, the position itself | list start}
__global__ void k_table_entries(const uint32_t* __restrict__ bucket, const uint32_t* __restrict__ pos, uint64_t nb,
                                uint64_t* __restrict__ entry) {
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < nb; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t lo = bucket[k], cnt = bucket[k + 1] - lo;
        entry[k] = MIRGE_ENTRY(cnt, cnt == 1 ? pos[lo] : lo);
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
#define MIRGE_NCLS 10      // width class (<= 31, 64, 128, 255 nt, longer: kernels_long.hpp) x (no ambiguous call | has one)
#define MIRGE_CLS_DROP 10  // shorter than --minimum-length (or longer than MIRGE_LONG_MAX_LEN: flagged)
#define MIRGE_LONG_MAX_LEN 65535  // the long class keeps a read's length in 16 bits

// bit i of `mask` = byte b0 + i is a newline (b0 is a multiple of 16 and the text buffer is 256-byte aligned: one 16-byte
// load per lane; a byte-wise loop with its bound test per byte kept the newline kernels at 1 TB/s)
__device__ __forceinline__ uint32_t tile_newlines(const uint8_t* __restrict__ text, uint64_t n, uint64_t b0, uint32_t& mask) {
    mask = 0;
    if (b0 + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + b0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t x = w[k] ^ 0x0A0A0A0Au;                                       // a newline byte becomes 0
            const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 in exactly the zero bytes
            mask |= ((((z >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * k);                // their four flags, packed
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) if (b0 + i < n && text[b0 + i] == 10) mask |= 1u << i;
    }
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

// fmt 1 (FASTQ): a record's first line starts with '@', its third with '+'; fmt 2 (FASTA): with '>' -- anything else
// (a truncated record, a blank line, a wrapped FASTA sequence) sets flags[3]: a parser that finds records by line
// number alone would otherwise return shifted garbage for the rest of the file
__global__ void k_nl_mark(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ tile_off, int period,
                          int sphase, int64_t* __restrict__ start, int64_t* __restrict__ end, uint64_t n_seq, int fmt,
                          uint32_t* __restrict__ flags, int64_t* __restrict__ qstart, int64_t* __restrict__ qend) {
    __shared__ uint32_t lds4[MIRGE_BLOCK / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * MIRGE_PARSE_TILE + threadIdx.x * 16ull;
    uint32_t mask;
    const uint32_t c = tile_newlines(text, n, b0, mask);
    uint32_t total;
    uint64_t li = (uint64_t)tile_off[blockIdx.x] + block_excl_scan(c, total, lds4);
    if (blockIdx.x == 0 && threadIdx.x == 0 && sphase == 0 && n_seq) start[0] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && n && ((fmt == 1 && text[0] != '@') || (fmt == 2 && text[0] != '>'))) atomicOr(&flags[3], 1u);
    while (mask) {
        const int i = __ffs(mask) - 1;
        mask &= mask - 1;
        const uint64_t pos = b0 + i;
        if (pos + 1 < n && fmt != 3) {
            const int ph = (int)((li + 1) % (uint64_t)period);
            const uint8_t nx = text[pos + 1];
            if ((ph == 0 && nx != (fmt == 1 ? '@' : '>')) || (fmt == 1 && ph == 2 && nx != '+')) atomicOr(&flags[3], 1u);
        }
        if ((int)(li % (uint64_t)period) == sphase && li / period < n_seq) end[li / period] = (int64_t)pos;
        if ((int)((li + 1) % (uint64_t)period) == sphase && (li + 1) / period < n_seq) start[(li + 1) / period] = (int64_t)pos + 1;
        if (qstart && fmt == 1 && (li + 1) % 4 == 3 && (li + 1) / 4 < n_seq) qstart[(li + 1) / 4] = (int64_t)pos + 1;  // quality line (k_trim)
        if (qend && fmt == 1 && li % 4 == 3 && li / 4 < n_seq) qend[li / 4] = (int64_t)pos;
        li++;
    }
}

// What k_seq_class keeps of a record [start, end) (+ second stretch): the worker's length test on the read as the
// modifiers left it (min_len_pre), then the UMI slice read[cut_front : len - cut_back] (`UMIParser`, digest.py:305-315:
// Python's s[f:-b], an empty string when the cuts meet), then --minimum-length on what is left (min_len).
struct SliceOpts {
    int32_t min_len_pre;
    int32_t min_len;
    int32_t cut_front, cut_back;
    int32_t exact_bounds;  // 1: [start, end) are the caller's own bounds (mirge_reads_pack) -- no '\r' is stripped from the end
};

// flags: [0] reads with N seen per group ... kept by k_pack; here [0] = byte outside ACGTUN seen, [1] = reads longer
// than MIRGE_LONG_MAX_LEN, [2] = longest such read, [4] = an IUPAC code seen, [5] = longest kept read of the long class,
// [6..7] = bases of the kept reads of the long class (64 bits; the length histogram stops at MIRGE_MAX_READ_LEN)
__global__ void k_seq_class(const uint8_t* __restrict__ text, int64_t* __restrict__ start, int64_t* __restrict__ end,
                            int64_t* __restrict__ s2start, int32_t* __restrict__ s2len,
                            uint32_t n_seq, SliceOpts so, uint8_t* __restrict__ cls, uint32_t* __restrict__ blk_cls,
                            uint32_t* __restrict__ blk_keep, uint32_t nblk, uint32_t* __restrict__ hist, uint32_t* __restrict__ flags) {
    __shared__ uint32_t s_cnt[MIRGE_NCLS + 2];
    __shared__ uint32_t s_hist[MIRGE_MAX_READ_LEN + 1];
    if (threadIdx.x < MIRGE_NCLS + 2) s_cnt[threadIdx.x] = 0;
    for (int i = threadIdx.x; i <= MIRGE_MAX_READ_LEN; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    int c = MIRGE_NCLS + 1;  // no record
    if (r < n_seq) {
        int64_t b = start[r];
        int64_t e = end[r];
        if (!so.exact_bounds && e > b && text[e - 1] == 13) e--;
        const int64_t L1 = e - b, L2 = s2len ? (int64_t)s2len[r] : 0, tot = L1 + L2;
        const int64_t x = so.cut_front < tot ? so.cut_front : tot;
        int64_t y = tot - so.cut_back;
        if (y < x) y = x;
        const int64_t nb = b + (x < L1 ? x : L1), ne = b + (y < L1 ? y : L1);
        const int64_t x2 = x > L1 ? x - L1 : 0, y2 = y > L1 ? y - L1 : 0;
        const int64_t b2 = s2len ? s2start[r] + x2 : 0;
        start[r] = nb; end[r] = ne;
        if (s2len) { s2start[r] = b2; s2len[r] = (int32_t)(y2 - x2); }
        const int64_t La = ne - nb, L = La + (y2 - x2);
        if (L1 < (int64_t)so.min_len_pre || L < (int64_t)so.min_len) {
            c = MIRGE_CLS_DROP;
        } else if (L > MIRGE_LONG_MAX_LEN) {
            atomicOr(&flags[1], 1u);
            atomicMax(&flags[2], (uint32_t)(L > 0xFFFFFFF ? 0xFFFFFFF : L));
            c = MIRGE_CLS_DROP;
        } else {
            uint32_t amb = 0, bad = 0, iu = 0;
            for (int p = 0; p < (int)L; p++) {
                const uint32_t bit = letter_bit((p < (int)La ? text[nb + p] : text[b2 + p - La]) & 0xDF);
                const bool acgt = (bit & MIRGE_LETTERS_ACGTU) != 0;
                amb |= !acgt;
                bad |= !acgt && !(bit & (MIRGE_LETTERS_N | MIRGE_LETTERS_IUPAC));
                iu |= (bit & MIRGE_LETTERS_IUPAC) != 0;
            }
            if (bad) atomicOr(&flags[0], 1u);
            if (iu) atomicOr(&flags[4], 1u);
            c = (L <= 31 ? 0 : (L <= 64 ? 1 : (L <= 128 ? 2 : (L <= MIRGE_MAX_READ_LEN ? 3 : 4)))) + (amb ? MIRGE_NCLS / 2 : 0);
            if (L <= MIRGE_MAX_READ_LEN) atomicAdd(&s_hist[L], 1u);
            else {
                atomicMax(&flags[5], (uint32_t)L);
                atomicAdd(reinterpret_cast<unsigned long long*>(flags + 6), (unsigned long long)L);
            }
        }
        cls[r] = (uint8_t)c;
    }
#pragma unroll
    for (int q = 0; q <= MIRGE_NCLS; q++) {
        const unsigned long long bal = __ballot(c == q);
        if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&s_cnt[q], (uint32_t)__popcll(bal));
    }
    __syncthreads();
    if (threadIdx.x < MIRGE_NCLS) blk_cls[(size_t)threadIdx.x * nblk + blockIdx.x] = s_cnt[threadIdx.x];
    if (threadIdx.x == MIRGE_NCLS) {
        uint32_t kept = 0;
        for (int q = 0; q < MIRGE_NCLS; q++) kept += s_cnt[q];
        blk_keep[blockIdx.x] = kept;
    }
    for (int i = threadIdx.x; i <= MIRGE_MAX_READ_LEN; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

__global__ void k_seq_place(const uint8_t* __restrict__ cls, uint32_t n_seq, const uint32_t* __restrict__ cls_off,
                            const uint32_t* __restrict__ keep_off, uint32_t nblk, uint32_t* __restrict__ src_all,
                            uint32_t* __restrict__ orig_all, uint32_t* __restrict__ rec_of_kept) {
    __shared__ uint32_t lds4[MIRGE_BLOCK / 64];
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = r < n_seq ? (int)cls[r] : MIRGE_NCLS + 1;
    uint32_t total;
    const uint32_t kr = keep_off[blockIdx.x] + block_excl_scan(c < MIRGE_CLS_DROP ? 1u : 0u, total, lds4);
    uint32_t slot = 0;
#pragma unroll
    for (int q = 0; q < MIRGE_NCLS; q++) {
        const uint32_t rk = block_excl_scan(c == q ? 1u : 0u, total, lds4);
        if (c == q) slot = cls_off[(size_t)q * nblk + blockIdx.x] + rk;
    }
    if (c < MIRGE_CLS_DROP) {
        src_all[slot] = r; orig_all[slot] = kr;
        if (rec_of_kept) rec_of_kept[kr] = r;  // kept rank (= handle index of the read) -> record of the text
    }
}

// --qiagenumi (digest.py:334-352): the worker takes `currentSeq.split(trimmed)[1]` -- what stands between the FIRST
// occurrence of the trimmed read in the untrimmed line and its next non-overlapping occurrence (or the line's end) --,
// keeps the first len(adapter) + b characters of it and of those the last b (all of them when b == 0: Python's s[-0:]):
// the UMI that follows the 3' adapter.  An empty trimmed read makes split() raise ValueError: no UMI.  The stretch is
// returned as the record's second segment; the dictionary key is trimmed + UMI.
__global__ void k_qiagen_umi(const uint8_t* __restrict__ text, const int64_t* __restrict__ lstart, const int64_t* __restrict__ lend,
                             const int64_t* __restrict__ tstart, const int64_t* __restrict__ tend, uint32_t n_raw, int32_t alen,
                             int32_t back, int64_t* __restrict__ s2start, int32_t* __restrict__ s2len) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_raw; r += gridDim.x * blockDim.x) {
        const int64_t lb = lstart[r];
        int64_t le = lend[r];
        if (le > lb && text[le - 1] == 13) le--;
        const int64_t tb = tstart[r], Lt = tend[r] - tb;
        if (Lt <= 0) { s2start[r] = le; s2len[r] = 0; continue; }
        auto occurs = [&](int64_t q) {
            for (int64_t i = 0; i < Lt; i++) if (text[q + i] != text[tb + i]) return false;
            return true;
        };
        int64_t p = tb;
        for (int64_t q = lb; q < tb; q++) if (occurs(q)) { p = q; break; }
        const int64_t rs = p + Lt;
        int64_t re = le;
        for (int64_t q = rs; q + Lt <= le; q++) if (occurs(q)) { re = q; break; }
        const int64_t rest = re - rs;
        const int64_t ulen = rest < (int64_t)alen + back ? rest : (int64_t)alen + back;
        const int64_t take = back > 0 ? (ulen < back ? ulen : (int64_t)back) : ulen;
        s2start[r] = rs + ulen - take;
        s2len[r] = (int32_t)take;
    }
}

// record bounds of the reads listed by kept rank (the first appearances of the distinct UMI-tagged reads, in the order of
// those appearances): -udd's second collapse takes their inserts
__global__ void k_gather_records(const uint32_t* __restrict__ rank, uint32_t n, const uint32_t* __restrict__ rec_of_kept,
                                 const int64_t* __restrict__ start, const int64_t* __restrict__ end,
                                 const int64_t* __restrict__ s2start, const int32_t* __restrict__ s2len,
                                 int64_t* __restrict__ ostart, int64_t* __restrict__ oend, int64_t* __restrict__ os2start,
                                 int32_t* __restrict__ os2len) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t r = rec_of_kept[rank[i]];
        ostart[i] = start[r]; oend[i] = end[r];
        if (os2len) { os2start[i] = s2start[r]; os2len[i] = s2len[r]; }
    }
}
