// kernels_join.hpp -- part of mirge_kernels.hpp: k_join, k_tally.
#pragma once
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

// the small read groups of a result in ONE launch (each its own launch was 4-5 us of a step's tail, four times)
struct JoinGroups {
    const int8_t* pass[MIRGE_NCLS];
    const int32_t* ref[MIRGE_NCLS];
    const uint32_t* counts[MIRGE_NCLS];
    uint32_t start[MIRGE_NCLS + 1];  // group k holds the items [start[k], start[k + 1]) of the launch
    int32_t n_groups;
};
__global__ void k_join_multi(JoinGroups gs, int32_t S, int32_t n_pass, int32_t exact_pass, int32_t iso_pass,
                             unsigned long long* __restrict__ class_sums, unsigned long long* __restrict__ exact,
                             unsigned long long* __restrict__ iso) {
    __shared__ unsigned long long acc[MIRGE_JOIN_LDS];
    const int cells = n_pass * S;
    const bool use_lds = cells <= MIRGE_JOIN_LDS;
    if (use_lds) {
        for (int c = threadIdx.x; c < cells; c += blockDim.x) acc[c] = 0ull;
        __syncthreads();
    }
    const uint32_t total = gs.start[gs.n_groups];
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < MIRGE_NCLS; q++) if (q < gs.n_groups && t >= gs.start[q]) k = q;
        const uint32_t i = t - gs.start[k];
        const int8_t* gp = nullptr; const int32_t* gr = nullptr; const uint32_t* gc = nullptr;
#pragma unroll
        for (int q = 0; q < MIRGE_NCLS; q++) if (q == k) { gp = gs.pass[q]; gr = gs.ref[q]; gc = gs.counts[q]; }
        const int p = gp[i];
        if (p < 0) continue;
        const int32_t ref = gr[i];
        for (int32_t s = 0; s < S; s++) {
            const unsigned long long c = gc[(size_t)i * S + s];
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

// k_join_rows / k_join_reduce (round 3): the same sums without a global atomic per read.  k_join's exact / isomiR tables take
// one device-scope atomic per miRNA read -- 2.5 M of them on 2.9 k hot addresses for a 10 M-read sample, 74 us at the
// chip's ~20-30 G scattered atomics/s: the last kernel of the step and 5 % of it.  Here a 1024-thread workgroup keeps ALL
// tables (class sums + exact + isomiR, 64-bit cells) in LDS while it walks its share of the reads, writes them out as
// one plain row of partial[workgroup][cell], and a second small kernel sums the rows column-wise into the ctx's tables.
// Used when the tables fit 48 KiB of LDS (one sample: up to ~3 000 miRNA references); otherwise k_join.
#define MIRGE_JOIN_ROWS_THREADS 1024
#define MIRGE_JOIN_ROWS_CELLS 6144  // 64-bit LDS cells
__global__ void __launch_bounds__(MIRGE_JOIN_ROWS_THREADS)
k_join_rows(JoinGroups gs, int32_t S, int32_t n_pass, int32_t exact_pass, int32_t iso_pass, uint32_t n_tab,
            unsigned long long* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long j_acc[];  // [n_pass * S] | [n_tab] exact | [n_tab] iso
    const uint32_t n_cls = (uint32_t)(n_pass * S), W = n_cls + 2 * n_tab;
    for (uint32_t c = threadIdx.x; c < W; c += blockDim.x) j_acc[c] = 0ull;
    __syncthreads();
    const uint32_t total = gs.start[gs.n_groups];
    if (gs.n_groups == 1 && S == 1) {
        // the bulk group of one sample: four reads per thread and sweep, their loads issued together (16 dependent
        // pass -> reference -> count walks per thread before: 23 us for 38 MB)
        const int8_t* __restrict__ gp = gs.pass[0]; const int32_t* __restrict__ gr = gs.ref[0]; const uint32_t* __restrict__ gc = gs.counts[0];
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;  // 64-bit: t0 + 3 * stride passes 2^32 for a group near that size
        for (uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t0 < total; t0 += 4 * stride) {
            int p[4]; uint32_t r[4]; unsigned long long c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint64_t i = t0 + (uint64_t)u * stride;
                const bool in = i < total;
                p[u] = in ? (int)gp[i] : -1;
                r[u] = in ? (uint32_t)gr[i] : 0u;
                c[u] = in ? (unsigned long long)gc[i] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (p[u] < 0 || !c[u]) continue;
                atomicAdd(&j_acc[p[u]], c[u]);
                if (p[u] == exact_pass) atomicAdd(&j_acc[n_cls + r[u]], c[u]);
                else if (p[u] == iso_pass) atomicAdd(&j_acc[n_cls + n_tab + r[u]], c[u]);
            }
        }
    } else
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        int k = 0;  // one group (the bulk) or the few small ones, back to back
#pragma unroll
        for (int q = 1; q < MIRGE_NCLS; q++) if (q < gs.n_groups && t >= gs.start[q]) k = q;
        const int8_t* gp = gs.pass[0]; const int32_t* gr = gs.ref[0]; const uint32_t* gc = gs.counts[0];
#pragma unroll
        for (int q = 1; q < MIRGE_NCLS; q++) if (q == k) { gp = gs.pass[q]; gr = gs.ref[q]; gc = gs.counts[q]; }
        const uint32_t i = t - gs.start[k];
        const int p = gp[i];
        if (p < 0) continue;
        const uint32_t ref = (uint32_t)gr[i];
        for (int32_t s = 0; s < S; s++) {
            const unsigned long long c = gc[(size_t)i * S + s];
            if (!c) continue;
            atomicAdd(&j_acc[p * S + s], c);
            if (p == exact_pass) atomicAdd(&j_acc[n_cls + (size_t)ref * S + s], c);
            else if (p == iso_pass) atomicAdd(&j_acc[n_cls + n_tab + (size_t)ref * S + s], c);
        }
    }
    __syncthreads();
    unsigned long long* row = partial + (size_t)blockIdx.x * W;
    for (uint32_t c = threadIdx.x; c < W; c += blockDim.x) row[c] = j_acc[c];
}
// tables[c] += sum over rows of partial[row][c].  A workgroup owns 64 consecutive cells; its 16 waves take every 16th row each
// (a wave's load of a row's 64 cells is one coalesced 512-byte access), then the waves' sums are added up through LDS.
// (One thread per cell walking all 256 rows by itself took 46 us for 5.9 k cells: a chain of 256 dependent-latency loads.)
// host_out != nullptr (page-locked host memory): the sums -- plus whatever `tables` already holds -- are written THERE and
// `tables` stays as it is: no device-to-host copy behind the kernel, and tables that were zero need no clearing afterwards.
__global__ void __launch_bounds__(1024)
k_join_reduce(const unsigned long long* __restrict__ partial, uint32_t rows, uint32_t W, unsigned long long* __restrict__ tables,
              unsigned long long* __restrict__ host_out) {
    __shared__ unsigned long long s_part[16][64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * 64 + lane;
    unsigned long long sum = 0ull;
    if (c < W)
        for (uint32_t r = wave; r < rows; r += 16) sum += partial[(size_t)r * W + c];
    s_part[wave][lane] = sum;
    __syncthreads();
    if (wave == 0 && c < W) {
        unsigned long long tot = 0ull;
#pragma unroll
        for (int w = 0; w < 16; w++) tot += s_part[w][lane];
        if (host_out) host_out[c] = tables[c] + tot;
        else if (tot) tables[c] += tot;
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
// k_tally (BASELINE config 5): the arithmetic of align2TargetSeq / judgeAllign /
// A2IEditing / mismatchCountAnalysis (mirge/libs/mirge2_tRF_a2i.py:246-518) for every read the cascade annotated to
// a miRNA, against the canonical sequence of that miRNA's FAMILY (the merged name's entry of
// <org>_mirna_SNP_pseudo_<db>.fa, :1037-1044 -- not necessarily the reference the read aligned to).
//   member of its family's list: an exact-miRNA read, or an isomiR read with count * freq[s] >= 1 in some sample
//     (freq[s] = 1e6 / Filtered miRNA Reads, :988-1016);
//   alignment: pairwise2.align.localms(target, read, 2, -1, -20, -20) (:254) = the best-scoring ungapped diagonal
//     (a gap costs 20, more than any annotated read can win back); ties: the alignment that ends first in the
//     target, then first in the read;  d = position of read base 0 relative to target base 0;
//   judgeAllign (:298-332), literally: reject if d > 1; walk the columns from the target's first base to
//     min(end_pos1, end_pos2) (end_pos1 = aligned length - head dashes of the target - 1 - 3, end_pos2 = last read
//     base), count matches and mismatches (a column past the target's end is a mismatch, a column before the read's
//     first base is skipped); accept iff mismatches <= 1 and matches >= Lt - 4 (- 1 more if d == 1);
//   per (family, sample) with count > 0: n_seqs (list length, :1133-1137), and over accepted AND retained reads
//     (retained = the genome filter's answer, :1085-1096): seq_true, count_true, canon (the read is a substring of the
//     target, :350), kept_exact (kept reads that are exact-miRNA rows: checkSeqList, :964-976);
//   census[family][q][target base * 4 + read base][variant][s] for q < Lt - 5 and read base != target base:
//     variant 0 = every member (raw), 1 = accepted, 2 = accepted and retained (mismatchCountAnalysis, :440-517;
//     A -> G of variant 2 is A2IEditing's position count, :358-366).
// ------------------------------------------------------------------------------------------
#define MIRGE_TALLY_MAXPOS 32
struct TallyOut {
    unsigned long long *n_seqs, *seq_true, *count_true, *canon, *kept_exact, *census;
    int8_t *diag, *state;  // per read, handle order: diagonal; 1 accepted / 0 rejected / -1 not a member
};

// The reads a heavy per-read kernel works on, compacted.  k_tally and k_isotype run ~2 000 and more instructions of loops
// per read whose trip counts differ, on the minority of a sample's unique reads that are miRNA rows: a wave costs what
// its busiest lane costs, and skipping the other reads lane by lane left k_tally at 10 % lane utilisation with 66 % of its
// time in VALU issue.  Workgroup b compacts the reads [b * chunk, (b + 1) * chunk) that satisfy `pred` in place:
// list[b * chunk ...], n_list[b] of them -- no global cursor (one returning atomic per 256 reads on one address cost
// more than the work it saved); the consumer runs one workgroup per chunk.
struct TallyMember {  // a member of its family's list: an exact-miRNA read, or an isomiR read with count * freq[s] >= 1 somewhere
    const int8_t* res_pass; const int32_t* res_ref; const uint32_t* counts; int32_t S, exact_pass, iso_pass;
    const int32_t* fam_of_ref; const double* freq;
    __device__ __forceinline__ bool operator()(uint32_t i) const {
        const int p = res_pass[i];
        if (p != exact_pass && p != iso_pass) return false;
        const int32_t f = fam_of_ref[res_ref[i]];
        bool member = f >= 0 && p == exact_pass;
        if (f >= 0 && p == iso_pass)
            for (int32_t s = 0; s < S; s++) member |= (double)counts[(size_t)i * S + s] * freq[s] >= 1.0;
        return member;
    }
};
struct IsoMember {  // a miRNA row of the GFF: annotated by one of the two miRNA passes and given an output slot by the host
    const int8_t* res_pass; int32_t exact_pass, iso_pass; const uint32_t* orig; uint32_t base; const int32_t* slot_of_read;
    __device__ __forceinline__ bool operator()(uint32_t i) const {
        const int p = res_pass[i];
        return (p == exact_pass || p == iso_pass) && slot_of_read[orig ? orig[i] : base + i] >= 0;
    }
};

template <class Pred, int BLOCK>
__global__ void k_member_list(uint32_t n, uint32_t chunk, Pred pred, uint32_t* __restrict__ list, uint32_t* __restrict__ n_list) {
    __shared__ uint32_t s_wave[BLOCK / 64 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    uint32_t run = 0;
    for (uint32_t i0 = lo; i0 < hi; i0 += BLOCK) {
        const uint32_t i = i0 + threadIdx.x;
        const bool member = i < hi && pred(i);
        const unsigned long long bal = __ballot(member);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) { const uint32_t c = s_wave[w]; if (w < wave) before += c; tot += c; }
        if (member) list[lo + run + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = i;
        run += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_list[blockIdx.x] = run;
}

__global__ void k_tally(GroupView<1> g, uint32_t base, const uint32_t* __restrict__ orig, const int8_t* __restrict__ res_pass,
                        const int32_t* __restrict__ res_ref, const uint32_t* __restrict__ counts, int32_t S,
                        int32_t exact_pass, int32_t iso_pass, const int32_t* __restrict__ fam_of_ref,
                        const uint64_t* __restrict__ tgt_bits, const uint8_t* __restrict__ tgt_len,
                        const uint8_t* __restrict__ retained, const double* __restrict__ freq, TallyOut o,
                        const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list, uint32_t chunk) {
    const uint32_t n_members = n_list[blockIdx.x];
    for (uint32_t k = threadIdx.x; k < n_members; k += blockDim.x) {
        const uint32_t i = list[(size_t)blockIdx.x * chunk + k];  // a member of its family's list (TallyMember); every other read keeps state -1
        const int p = res_pass[i];
        const uint32_t h = orig ? orig[i] : base + i;
        const int32_t f = fam_of_ref[res_ref[i]];
        const uint64_t tw = tgt_bits[f];
        const int Lt = tgt_len[f], Lr = g.len[i];
        const uint64_t rw = g.seq[i];
        const uint64_t rn = g.nmask ? g.nmask[i] : 0ull;
        // ---- best ungapped local alignment: Kadane along every diagonal, +2 / -1
        // (a diagonal whose overlap cannot reach the best score so far is skipped, and the diagonals around the read's
        // own go first so that the bound is tight from the start: ~5-10 of ~55 diagonals are walked for an annotated
        // read.  The update rule is a total order on (score, end in target, end in read): the visiting order is free.
        // Per diagonal the mismatches of the whole overlap come from one XOR of the shifted words, folded to one bit
        // per base, so that the walk itself is 32-bit shifts.)
        int best = 0, best_i = 0, best_j = 0, d = 0;
        const int nd = Lr + Lt - 1;
        for (int k = 0; k < nd + 3; k++) {
            const int dd = k < 3 ? k - 1 : k - 3 - (Lr - 1);  // -1, 0, +1 first, then every diagonal in order
            if (dd <= -Lr || dd >= Lt) continue;
            const int i0 = dd > 0 ? dd : 0, i1 = min(Lt, dd + Lr), j0 = i0 - dd;
            if (2 * (i1 - i0) < best) continue;
            const uint64_t x = (tw >> (2 * i0)) ^ (rw >> (2 * j0));
            uint64_t m2 = ((x | (x >> 1)) & 0x5555555555555555ull) | ((rn >> (2 * j0)) & 0x5555555555555555ull);
            m2 = (m2 | (m2 >> 1)) & 0x3333333333333333ull;  // compress the even bits: one mismatch bit per base
            m2 = (m2 | (m2 >> 2)) & 0x0F0F0F0F0F0F0F0Full;
            m2 = (m2 | (m2 >> 4)) & 0x00FF00FF00FF00FFull;
            m2 = (m2 | (m2 >> 8)) & 0x0000FFFF0000FFFFull;
            const uint32_t mis = (uint32_t)(m2 | (m2 >> 16));
            int run = 0;
            for (int t = 0; t < i1 - i0; t++) {
                run = max(0, run + (((mis >> t) & 1u) ? -1 : 2));
                const int ti = i0 + t, rj = j0 + t;
                if (run > best || (run == best && run > 0 && (ti < best_i || (ti == best_i && rj < best_j)))) {
                    best = run; best_i = ti; best_j = rj; d = dd;
                }
            }
        }
        // ---- judgeAllign
        bool state = d <= 1;
        if (state) {
            const int hd_t = d < 0 ? -d : 0, hd_s = d > 0 ? d : 0;
            const int A = max(hd_t + Lt, hd_s + Lr);
            const int last = min(A - hd_t - 1 - 3, hd_s + Lr - 1);
            int mism = 0, match = 0;
            for (int pos = hd_t; pos <= last; pos++) {
                const int rj = pos - hd_s, q = pos - hd_t;
                if (rj < 0) continue;
                const bool eq = q < Lt && !((rn >> (2 * rj)) & 1ull) && (((tw >> (2 * q)) & 3ull) == ((rw >> (2 * rj)) & 3ull));
                if (eq) match++; else mism++;
            }
            state = !(mism > 1 || match < Lt - 3 - 1 - (d == 1 ? 1 : 0));
        }
        o.diag[h] = (int8_t)d;
        o.state[h] = state ? 1 : 0;
        const bool keep = state && (!retained || retained[h]);
        // the read is a substring of the target (`seqList[j] in targetSeq`, :350): any offset, no N
        bool sub = false;
        if (keep && rn == 0ull && Lr <= Lt) {
            const uint64_t m = mirge_lowmask2(Lr);
            for (int a = 0; a + Lr <= Lt && !sub; a++) sub = ((tw >> (2 * a)) & m) == rw;
        }
        for (int32_t s = 0; s < S; s++) {
            const unsigned long long c = counts[(size_t)i * S + s];
            if (!c) continue;
            const size_t fs = (size_t)f * S + s;
            atomicAdd(&o.n_seqs[fs], 1ull);
            if (keep) {
                atomicAdd(&o.seq_true[fs], 1ull);
                atomicAdd(&o.count_true[fs], c);
                if (sub) atomicAdd(&o.canon[fs], c);
                if (p == exact_pass) atomicAdd(&o.kept_exact[fs], 1ull);
            }
            for (int q = max(0, d); q < Lt - 5 && q - d < Lr; q++) {
                const int rj = q - d;
                if ((rn >> (2 * rj)) & 1ull) continue;  // an N call is none of the twelve base changes
                const int cb = (int)((tw >> (2 * q)) & 3ull), rb = (int)((rw >> (2 * rj)) & 3ull);
                if (cb == rb) continue;
                unsigned long long* cell = &o.census[((((size_t)f * MIRGE_TALLY_MAXPOS + q) * 16 + cb * 4 + rb) * 3) * S + s];
                atomicAdd(cell, c);
                if (state) atomicAdd(cell + S, c);
                if (keep) atomicAdd(cell + 2 * S, c);
            }
        }
    }
}
