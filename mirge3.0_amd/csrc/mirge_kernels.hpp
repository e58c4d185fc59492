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
#ifndef MIRGE_COLLAPSE_SEED
#define MIRGE_COLLAPSE_SEED 2048  // reads of a group inserted ahead of the rest (general collapse path)
#endif
#define MIRGE_CELL_CACHE 2048  // per-workgroup LDS cache of (slot, sample) cells in the general collapse path

template <int W>
struct GroupView {
    const uint64_t* seq;    // [W][n]
    const uint8_t* len;     // [n]
    const uint64_t* nmask;  // [W][n] or nullptr
    uint32_t n;
};

// HASN = false: the group holds no ambiguous call (its nmask is nullptr) and the caller says so at compile time -- the
// zero masks then fold away in everything inlined behind the load (trimming, probe keys, verification)
template <int W, bool HASN = true>
__device__ __forceinline__ void load_read(const GroupView<W>& g, uint32_t i, MirgeRead<W>& r) {
#pragma unroll
    for (int w = 0; w < W; w++) {
        r.w[w] = g.seq[(size_t)w * g.n + i];
        r.nm[w] = (HASN && g.nmask) ? g.nmask[(size_t)w * g.n + i] : 0ull;
    }
    r.len = g.len[i];
}

#include "kernels_reads.hpp"
#include "kernels_trim.hpp"
#include "kernels_collapse.hpp"
#include "kernels_cascade.hpp"
#include "kernels_long.hpp"
#include "kernels_join.hpp"
#include "kernels_iso.hpp"
#include "kernels_csv.hpp"
