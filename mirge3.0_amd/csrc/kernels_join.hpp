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
