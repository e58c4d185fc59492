/*
 * mirge_oracle.c -- CPU restatement of the miRge3.0 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  The
 * product (mirge3.0_amd/, libmirge_native.so) never links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the alignment predicate.  The arithmetic of the
 * cascade lives in bowtie 1.x (third-party C++, not vendored by the reference; allow-list
 * 1.0.0..1.3.2 at mirge/libs/miRgeEssential.py:17, docs install 1.3.0), which is absent from
 * this image and the reference holds no test or golden vector for it.  What is restated here
 * is bowtie-1's published (manual) semantics for the ten argument strings of
 * mirge/libs/manifoldAlign.py:85.  Everything around that predicate IS pinned against the
 * real reference code: the pass order / subset rules / overwrite rule (manifoldAlign.py:12-146)
 * and the count join (summary.py) are checked by tests/golden fixtures produced by running the
 * reference's own bwtAlign + summarize over this oracle (tests/golden/make_golden.py).
 *
 * Restated semantics (bowtie 1 manual; FASTA input => every base has Phred 40, Maq rounding
 * caps a mismatch's cost at 30, default -e 70 => at most 2 mismatches in total in -n mode):
 *   -n N : <= N mismatches in the first min(28, len) bases (the seed, -l 28), <= 2 overall
 *   -v V : <= V mismatches end to end, qualities ignored
 *   -5 a -3 b : a bases removed from the 5' end and b from the 3' end before aligning
 *   --norc : forward reference strand only;  a read aligns inside ONE reference sequence
 *   a read base that is not A/C/G/T mismatches every reference base
 *   a window that overlaps a reference base that is not A/C/G/T is never a valid alignment
 *   a read whose (trimmed) length is 0 or <= the mode's mismatch budget is skipped
 * Which of several valid hits bowtie reports (-k 1 without --best; "last SAM line wins" under
 * -a, manifoldAlign.py:55) depends on bowtie's FM-index traversal and PRNG and is not
 * reproducible from first principles.  Documented deterministic rule used by oracle AND product:
 * fewest total mismatches, then lowest reference index in the library, then leftmost offset.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* seconds spent building k-mer tables since the last reset: bench.py subtracts them from its
 * cpu_baseline timing (they stand for `bowtie-build`, which is not part of a run) */
static double g_build_seconds = 0.0;
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
double oracle_build_seconds(int reset) { double v = g_build_seconds; if (reset) g_build_seconds = 0.0; return v; }

typedef struct {
    int32_t mode;        /* 0: -n (seeded), 1: -v (end to end)                              */
    int32_t mm;          /* N of -n / V of -v                                               */
    int32_t seedlen;     /* -l, 28                                                          */
    int32_t maxtotal;    /* -n: floor(70/30)=2 ; -v: V                                      */
    int32_t trim5;       /* -5                                                              */
    int32_t trim3;       /* -3                                                              */
    int32_t ttail;       /* 1: pass 3 -- only reads matching T{3,}$, aligned without the run */
    int32_t len_lt;      /* >0: only reads with len <  len_lt  (pass 0: 26, manifoldAlign.py:93)  */
    int32_t len_gt;      /* >0: only reads with len >  len_gt  (pass 1: 25, manifoldAlign.py:104) */
    int32_t need_unannotated; /* 1: only rows with annotFlag==0 (passes >=2, manifoldAlign.py:120,129) */
} oracle_policy;

typedef struct {
    const char *data;       /* concatenated reference sequences, ASCII */
    const int64_t *offsets; /* n+1 */
    int64_t n;
} oracle_lib;

static inline int is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

/* mismatch test of read[0..L) against window w[0..L); returns total mismatches or -1 if the
 * window is invalid or the policy is violated */
static inline int window_mm(const char *read, int L, const char *w, const oracle_policy *p) {
    int seed = p->mode == 0 ? (L < p->seedlen ? L : p->seedlen) : L;
    int tot = 0, sd = 0;
    for (int i = 0; i < L; i++) {
        char rc = w[i];
        if (!is_acgt(rc)) return -1;
        if (read[i] != rc) {
            tot++;
            if (i < seed) sd++;
            if (tot > p->maxtotal || sd > p->mm) return -1;
        }
    }
    return tot;
}

/* what is aligned for this read under this policy: pointer/length after -5/-3 trimming or
 * T-tail stripping; returns 0 if the read is not submitted / skipped by bowtie */
static int effective_read(const char *read, int L, const oracle_policy *p, const char **out, int *outL) {
    if (p->len_lt > 0 && !(L < p->len_lt)) return 0;
    if (p->len_gt > 0 && !(L > p->len_gt)) return 0;
    const char *r = read;
    int l = L;
    if (p->ttail) { /* re.search('T{3,}$', seq): strip the whole terminal T run, needs >= 3 */
        int run = 0;
        while (run < L && read[L - 1 - run] == 'T') run++;
        if (run < 3) return 0;
        l = L - run;
    }
    r += p->trim5;
    l -= p->trim5 + p->trim3;
    if (l < 1 || l <= p->mm) return 0;
    *out = r;
    *outL = l;
    return 1;
}

/* ---- brute force: every window of every reference ---- */
static int align_brute(const char *r, int l, const oracle_lib *lib, const oracle_policy *p,
                       int32_t *ref, int32_t *off, int32_t *mm) {
    int best = 1 << 30;
    for (int64_t t = 0; t < lib->n; t++) {
        const char *s = lib->data + lib->offsets[t];
        int64_t sl = lib->offsets[t + 1] - lib->offsets[t];
        for (int64_t o = 0; o + l <= sl; o++) {
            int m = window_mm(r, l, s + o, p);
            if (m >= 0 && m < best) {
                best = m; *ref = (int32_t)t; *off = (int32_t)o; *mm = m;
                if (m == 0) return 1; /* nothing beats (0, lowest t, lowest o) */
            }
        }
    }
    return best < (1 << 30);
}

/* ---- indexed: pigeonhole seeds over a direct-addressed k-mer table (for sizes where brute
 * force does not finish); validated against align_brute by tests/test_host_logic.py::test_oracle_bruteforce_vs_indexed_ci_scale ---- */
#define OR_KMAX 12
typedef struct {
    int k;
    uint32_t *start; /* 4^k + 1 */
    uint32_t *pos;   /* global positions in lib->data */
} kindex;

typedef struct {
    const oracle_lib *lib;
    kindex idx[OR_KMAX + 1]; /* built before the parallel loop for every k the pass needs */
} lib_index;

static inline int code_of(char c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

static void build_kindex(const oracle_lib *lib, int k, kindex *ix) {
    uint64_t nb = 1ull << (2 * k);
    uint64_t mask = nb - 1;
    ix->k = k;
    ix->start = (uint32_t *)calloc(nb + 1, sizeof(uint32_t));
    for (int phase = 0; phase < 2; phase++) {
        if (phase == 1) {
            uint32_t acc = 0;
            for (uint64_t b = 0; b <= nb; b++) { uint32_t c = ix->start[b]; ix->start[b] = acc; acc += c; }
            ix->pos = (uint32_t *)malloc(sizeof(uint32_t) * (acc ? acc : 1));
        }
        for (int64_t t = 0; t < lib->n; t++) {
            int64_t s = lib->offsets[t], e = lib->offsets[t + 1];
            uint64_t key = 0; int valid = 0;
            for (int64_t i = s; i < e; i++) {
                int c = code_of(lib->data[i]);
                if (c < 0) { valid = 0; key = 0; continue; }
                key = ((key << 2) | (uint64_t)c) & mask; /* first base most significant */
                valid++;
                if (valid >= k) {
                    if (phase == 0) ix->start[key]++;
                    else ix->pos[ix->start[key]++] = (uint32_t)(i - k + 1);
                }
            }
        }
    }
    /* phase 1 advanced start[b] to the end of bucket b: shift back */
    for (uint64_t b = nb; b > 0; b--) ix->start[b] = ix->start[b - 1];
    ix->start[0] = 0;
}

static int ref_of_pos(const oracle_lib *lib, int64_t gpos) {
    int64_t lo = 0, hi = lib->n; /* last t with offsets[t] <= gpos */
    while (hi - lo > 1) { int64_t mid = (lo + hi) / 2; if (lib->offsets[mid] <= gpos) lo = mid; else hi = mid; }
    return (int)lo;
}

static int align_indexed(const char *r, int l, lib_index *li, const oracle_policy *p,
                         int32_t *ref, int32_t *off, int32_t *mm) {
    const oracle_lib *lib = li->lib;
    int seed = p->mode == 0 ? (l < p->seedlen ? l : p->seedlen) : l;
    int nseg = p->mm + 1;
    int h = seed / nseg;
    int k = h < OR_KMAX ? h : OR_KMAX;
    if (k < 4) return align_brute(r, l, lib, p, ref, off, mm);
    kindex *ix = &li->idx[k];
    if (!ix->start) return align_brute(r, l, lib, p, ref, off, mm); /* not prebuilt: cannot happen */
    int64_t best_pos = -1; int best = 1 << 30;
    for (int sgi = 0; sgi < nseg; sgi++) {
        int a = sgi * h; /* segment [a, a+h) (the last one may be longer; a k-prefix suffices) */
        uint64_t key = 0; int ok = 1;
        for (int i = 0; i < k; i++) { int c = code_of(r[a + i]); if (c < 0) { ok = 0; break; } key = (key << 2) | (uint64_t)c; }
        if (!ok) continue;
        for (uint32_t c = ix->start[key]; c < ix->start[key + 1]; c++) {
            int64_t g = (int64_t)ix->pos[c] - a;
            if (g < 0) continue;
            int t = ref_of_pos(lib, g);
            if (g + l > lib->offsets[t + 1]) continue; /* a hit lies inside ONE reference */
            int m = window_mm(r, l, lib->data + g, p);
            if (m < 0) continue;
            if (m < best || (m == best && g < best_pos)) { best = m; best_pos = g; }
        }
    }
    if (best_pos < 0) return 0;
    int t = ref_of_pos(lib, best_pos);
    *ref = t; *off = (int32_t)(best_pos - lib->offsets[t]); *mm = best;
    return 1;
}

/* Align one (already selected) read against one library.  indexed=0: brute force. */
int oracle_align_one(const char *read, int32_t L, const oracle_lib *lib, const oracle_policy *p,
                     int32_t *ref, int32_t *off, int32_t *mm) {
    const char *r; int l;
    oracle_policy q = *p; q.len_lt = 0; q.len_gt = 0;
    if (!effective_read(read, L, &q, &r, &l)) return 0;
    return align_brute(r, l, lib, p, ref, off, mm);
}

/*
 * The cascade (mirge/libs/manifoldAlign.py:68-146): passes in order; pass p looks only at
 * the rows its subset rule selects; a hit writes column p and sets annotFlag (here:
 * out_pass[i]=p).  Exactly one column can ever be set per row: passes 0 and 1 are disjoint
 * by length (:93,:104) and every later pass requires annotFlag==0 (:120,:129).
 * out_pass[i] = -1 for rows that stay unannotated.
 */
int oracle_cascade(const char *reads, const int64_t *roff, int64_t n,
                   const oracle_lib *libs, const oracle_policy *pol, int32_t n_pass,
                   int32_t indexed, int32_t threads,
                   int32_t *out_pass, int32_t *out_ref, int32_t *out_off, int32_t *out_mm) {
    for (int64_t i = 0; i < n; i++) { out_pass[i] = -1; out_ref[i] = -1; out_off[i] = -1; out_mm[i] = -1; }
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    for (int p = 0; p < n_pass; p++) {
        if (libs[p].n <= 0) continue;
        lib_index li; memset(&li, 0, sizeof li); li.lib = &libs[p];
        const oracle_policy *q = &pol[p];
        if (indexed) { /* which k-mer tables does this pass need? */
            for (int64_t i = 0; i < n; i++) {
                if (q->need_unannotated && out_pass[i] >= 0) continue;
                const char *r; int l;
                if (!effective_read(reads + roff[i], (int)(roff[i + 1] - roff[i]), q, &r, &l)) continue;
                int seed = q->mode == 0 ? (l < q->seedlen ? l : q->seedlen) : l;
                int h = seed / (q->mm + 1);
                int k = h < OR_KMAX ? h : OR_KMAX;
                if (k >= 4 && !li.idx[k].start) { double t0 = now_s(); build_kindex(&libs[p], k, &li.idx[k]); g_build_seconds += now_s() - t0; }
            }
        }
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t i = 0; i < n; i++) {
            /* passes 0/1 do not test annotFlag (manifoldAlign.py:93,104) but their length
             * subsets are disjoint, so "skip rows already annotated" is the same thing */
            if (out_pass[i] >= 0) continue;
            const char *r; int l;
            const char *read = reads + roff[i];
            int L = (int)(roff[i + 1] - roff[i]);
            if (!effective_read(read, L, q, &r, &l)) continue;
            int32_t ref, off, mm;
            int hit = indexed ? align_indexed(r, l, &li, q, &ref, &off, &mm)
                              : align_brute(r, l, &libs[p], q, &ref, &off, &mm);
            if (hit) { out_pass[i] = p; out_ref[i] = ref; out_off[i] = off; out_mm[i] = mm; }
        }
        for (int k = 0; k <= OR_KMAX; k++) { free(li.idx[k].start); free(li.idx[k].pos); }
    }
    return 0;
}

/*
 * Collapse (mirge/libs/digest.py:141-163): count identical sequences; uniques come out in
 * order of first appearance (dict insertion order when the chunks are consumed in file order).
 * uniq_first[u] = index of the first raw read of unique u, uniq_count[u] = its multiplicity,
 * inverse[i] = unique index of raw read i.  Returns U.
 */
typedef struct { const char *s; int32_t len; int64_t idx; } sref;
static int cmp_sref(const void *a, const void *b) {
    const sref *x = (const sref *)a, *y = (const sref *)b;
    int m = x->len < y->len ? x->len : y->len;
    int c = memcmp(x->s, y->s, (size_t)m);
    if (c) return c;
    if (x->len != y->len) return x->len < y->len ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b; return x < y ? -1 : (x > y);
}
int64_t oracle_collapse(const char *reads, const int64_t *roff, int64_t n,
                        int64_t *uniq_first, int64_t *uniq_count, int64_t *inverse) {
    if (n == 0) return 0;
    sref *v = (sref *)malloc(sizeof(sref) * (size_t)n);
    for (int64_t i = 0; i < n; i++) { v[i].s = reads + roff[i]; v[i].len = (int32_t)(roff[i + 1] - roff[i]); v[i].idx = i; }
    qsort(v, (size_t)n, sizeof(sref), cmp_sref);
    /* run heads are first appearances because idx is the last sort key */
    int64_t U = 0;
    int64_t *firsts = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; i++)
        if (i == 0 || v[i].len != v[i - 1].len || memcmp(v[i].s, v[i - 1].s, (size_t)v[i].len)) firsts[U++] = v[i].idx;
    qsort(firsts, (size_t)U, sizeof(int64_t), cmp_i64);
    /* rank of a run = position of its first index among the sorted firsts */
    int64_t run_first = -1, run_rank = -1;
    for (int64_t u = 0; u < U; u++) { uniq_first[u] = firsts[u]; uniq_count[u] = 0; }
    for (int64_t i = 0; i < n; i++) {
        if (i == 0 || v[i].len != v[i - 1].len || memcmp(v[i].s, v[i - 1].s, (size_t)v[i].len)) {
            run_first = v[i].idx;
            int64_t lo = 0, hi = U - 1;
            while (lo < hi) { int64_t mid = (lo + hi) / 2; if (firsts[mid] < run_first) lo = mid + 1; else hi = mid; }
            run_rank = lo;
        }
        uniq_count[run_rank]++;
        inverse[v[i].idx] = run_rank;
    }
    free(firsts); free(v);
    return U;
}
