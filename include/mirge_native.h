/*
 * mirge_native.h -- C ABI of libmirge_native.so: the MI355X (gfx950) implementation of the
 * miRge3.0 hot path (collapse -> annotation cascade -> count join).
 *
 * The reference (mhalushka/miRge3.0) has no FFI for this path: it is three Python call sites
 * and one process boundary.  Each entry point below names what it replaces;
 * INTEGRATION.md shows the ctypes stub a miRge3.0 maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative code on failure; mirge_last_error()
 *     returns the message of the calling thread's last failure (the reference either raises
 *     CalledProcessError from subprocess.run(check=True), manifoldAlign.py:19, or prints and
 *     exits; the Python wrapper turns a non-zero code into RuntimeError);
 *   - plain pointers and sizes only; "host" pointers are ordinary process memory (numpy
 *     buffers), all device memory is owned by the opaque handles;
 *   - sequences cross the boundary as one ASCII byte array plus int64 offsets[n+1];
 *   - handles are created/destroyed by the library, the caller allocates every output array;
 *   - one mirge_ctx = one GPU + one HIP stream; calls on one ctx are serialised by the caller,
 *     different ctx (different GPUs) may be driven from different threads/processes.
 */
#ifndef MIRGE_NATIVE_H
#define MIRGE_NATIVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mirge_ctx mirge_ctx;       /* device + stream + workspace pool              */
typedef struct mirge_lib mirge_lib;       /* one reference library, packed + indexed in HBM */
typedef struct mirge_reads mirge_reads;   /* device-resident packed read set (+ counts)     */
typedef struct mirge_result mirge_result; /* device-resident per-read annotation            */

/* One cascade pass = one of the ten bowtie argument strings of manifoldAlign.py:85, restated
 * (SURVEY.md 8 table a8-P).  Same fields as oracle_policy minus the annotFlag test, which the
 * cascade applies itself (manifoldAlign.py:120,129). */
typedef struct mirge_policy {
    int32_t mode;     /* 0: -n N (seeded), 1: -v V (end to end)                          */
    int32_t mm;       /* N or V                                                          */
    int32_t seedlen;  /* -l, bowtie default 28                                           */
    int32_t maxtotal; /* -n: 2 (FASTA Q40 -> Maq-rounded 30, -e 70); -v: V               */
    int32_t trim5;    /* -5                                                              */
    int32_t trim3;    /* -3                                                              */
    int32_t ttail;    /* 1: only reads matching T{3,}$, aligned without that run (:118-126) */
    int32_t len_lt;   /* >0: only reads shorter than this (pass 0: 26, :93)              */
    int32_t len_gt;   /* >0: only reads longer than this (pass 1: 25, :104)              */
    int32_t reserved;
} mirge_policy;

#define MIRGE_MAX_PASSES 16
#define MIRGE_NO_PASS (-1)

const char* mirge_last_error(void);

/* replaces: miRgeEssential.check_dependencies probing `bowtie --version` (miRgeEssential.py:6-34) */
int mirge_device_count(void);
/* hip_stream: a hipStream_t to run on (e.g. torch's current stream) or NULL for a private one */
int mirge_ctx_create(int device, void* hip_stream, mirge_ctx** out);
/* Ownership: libraries, read sets and results hold device blocks of the context they were made in -- destroy them BEFORE the
   context (the ctypes binding does that by itself: closing a Context closes what still lives in it).  A cascade may be handed
   a library of ANOTHER context of the same device (its probe tables are shared under the library's lock); the whole-read tables
   of the exact passes are built, named and dropped only through the context the library was made in -- a borrower's exact
   passes take the probe path, with the same results. */
void mirge_ctx_destroy(mirge_ctx* ctx);
int mirge_ctx_sync(mirge_ctx* ctx);

/* ---- libraries: replaces `bowtie <index>` opening an .ebwt index (manifoldAlign.py:97-99)
 * and `bowtie-build`.  seq_ascii/offsets: the reference sequences in library order.          */
int mirge_lib_create(mirge_ctx* ctx, const char* seq_ascii, const int64_t* offsets, int64_t n_refs,
                     mirge_lib** out);
/* The same from the packed image mirge_lib_create derives (2-bit text, invalid-base bitmap, reference starts): what a
 * library cache next to the index holds, so that a process neither parses the FASTA / decodes the .ebwt nor packs it
 * again.  mirge_lib_packed_sizes / _copy hand the image of a library out (sizes[4] = words of T, words of inv, total
 * bases incl. separators, valid positions).  Probe tables are built on the device either way. */
int mirge_lib_create_packed(mirge_ctx* ctx, const uint64_t* T, int64_t n_T, const uint64_t* inv, int64_t n_inv,
                            const uint32_t* ref_start, int64_t n_refs, uint64_t total, int32_t kmax,
                            uint64_t valid_positions, mirge_lib** out);
int mirge_lib_packed_sizes(const mirge_lib* lib, int64_t* sizes, int32_t* kmax);
int mirge_lib_packed_copy(const mirge_lib* lib, uint64_t* T, uint64_t* inv, uint32_t* ref_start);
void mirge_lib_destroy(mirge_lib* lib);
int64_t mirge_lib_n_refs(const mirge_lib* lib);
int64_t mirge_lib_device_bytes(const mirge_lib* lib);
/* build the k-mer table for probe length k now (otherwise built on first need) */
int mirge_lib_prepare(mirge_lib* lib, int32_t k);

/* ---- input: a whole .gz file's bytes -> its text on `threads` host threads (0: all).  What xopen's pigz / igzip threads are to the
 * reference's reader (digest.py:136-140): one zlib stream inflates at ~0.4 GB/s of text, twenty times slower than everything
 * behind it here.  An ordinary member is cut at deflate block starts found by search and decoded in parallel (csrc/native_gz.hpp);
 * a BGZF file member by member.  out[cap]; 0 = done and verified against the file's own CRC-32 and length; negative = not
 * taken by this route (several ordinary members, too small, not text, damaged, cap too small): inflate it serially. */
int mirge_gz_inflate(const uint8_t* gz, int64_t n_gz, uint8_t* out, int64_t cap, int64_t* n_out, int32_t threads);
/* the same; while it runs `*progress` (may be NULL) is advanced to the number of leading bytes of `out` that are final, so that a
 * second thread can upload and parse whole records of that prefix beside the inflation (the reference's reader hands chunks to its
 * workers while xopen's threads inflate, digest.py:136-140).  The CRC-32 is verified at the end: on a negative return whatever was
 * read ahead must be dropped. */
int mirge_gz_inflate_progress(const uint8_t* gz, int64_t n_gz, uint8_t* out, int64_t cap, int64_t* n_out, int32_t threads, int64_t* progress);

/* ---- reads: replaces writing bwtInput.fasta (manifoldAlign.py:92-95) / dnaio parsing ---- */
int mirge_reads_pack(mirge_ctx* ctx, const char* ascii, const int64_t* offsets, int64_t n,
                     mirge_reads** out);
/* The same from the FILE's text, parsed on the device: replaces dnaio's record parsing and the length filter of the
 * per-chunk worker (digest.py:320-375, --minimum-length :348,368).  format: 1 FASTQ (4-line records), 2 FASTA (one
 * sequence line per record), 3 one sequence per line, 0 = by the first byte.  *n_records = records seen before the
 * filter (`count`, digest.py:326).  Reads come out in file order.  A read may be up to 65 535 nt (the reference has no upper
 * bound -- parse.py:102 `-M` is never read, digest.py:348,368 test the minimum only --; reads beyond 255 nt take kernels written
 * for any length, csrc/kernels_long.hpp); a longer one is refused (-6). */
int mirge_reads_parse(mirge_ctx* ctx, const char* text, int64_t nbytes, int32_t format, int32_t min_len,
                      mirge_reads** out, int64_t* n_records);
/* 1 when some read of the set held an IUPAC ambiguity code other than N: such a base is packed -- and later printed --
 * as N (bowtie aligns it as N too; the reference's dictionary would have kept the letter). */
int32_t mirge_reads_iupac_seen(const mirge_reads* reads);
/* The same with the read modifiers the reference runs through cutadapt before it counts a read (digest.py:59-101; SURVEY.md
 * 8f row N4), applied on the device between record finding and the length filter: NextSeq quality trimming, quality
 * trimming, adapter removal (one or two adapters, 3' or 5', regular or anchored, or one linked adapter), N trimming,
 * unconditional cuts -- in that order, each present when its field says so.
 * count_per_modifier = 1 reproduces the reference's worker at HEAD, which tests the length and counts the read after
 * EVERY modifier (digest.py:354-373); 0 counts the fully trimmed read once.  *n_records stays the number of records of
 * the text.  trim == NULL: mirge_reads_parse.  Restated from cutadapt's published algorithms: parity unpinned. */
typedef struct mirge_trim {
    int32_t nextseq_cutoff;   /* --nextseq-trim, -1 = off                                   */
    int32_t quality_front;    /* -q 5'CUTOFF (0 when only one value is given)                */
    int32_t quality_back;     /* -q 3'CUTOFF (the reference's default: 10), -1 = off         */
    int32_t phred_base;       /* 33 (or 64)                                                  */
    const char* adapter;      /* -a: 3' adapter, A/C/G/T/N (or -g, see adapter_front), or NULL */
    int32_t adapter_len;
    int32_t min_overlap;      /* --overlap (3)                                               */
    double error_rate;        /* --error-rate (0.12), of the aligned adapter length          */
    int32_t trim_n;           /* --trim-n                                                    */
    int32_t n_cut;            /* -u, up to two values: > 0 from the 5' end, < 0 from the 3'  */
    int32_t cut[2];
    int32_t count_per_modifier;
    int32_t adapter_front;    /* 1: `adapter` is a 5' adapter (-g): the read keeps what follows it; no N */
    const char* adapter2;     /* a second adapter or NULL: per read the better match of the two is removed (cutadapt's   */
    int32_t adapter2_len;     /* AdapterCutter with times = 1: most matches, then fewest errors, then the first given)  */
    int32_t adapter2_front;
    int32_t times;            /* -n COUNT: remove adapters up to COUNT times (0 / 1: once)                               */
    int32_t no_indels;        /* --no-indels: substitutions only in the adapter alignment                                 */
    int32_t match_read_wildcards;  /* --match-read-wildcards: an N in the read matches every adapter base                  */
    int32_t no_adapter_wildcards;  /* -N: an N in the adapter is not a wildcard                                            */
    int32_t action_none;      /* --action none: adapters are searched, the read is left as it is                          */
    /* anchored and linked adapters, as cutadapt's parser makes them of the specification strings the reference hands through
     * unchanged (digest.py:66-84; docs/source/quick_start.md:213-220 shows `-g "TTAGGC...TGGAATTCTCGGGTGCCAAGGAACTCCAGT"`):  */
    int32_t adapter_anchored; /* `-g ^ADAPTER`: the whole adapter at the read's first base / `-a ADAPTER$`: up to its last base */
    int32_t adapter2_anchored;
    int32_t linked;           /* 1: `adapter` (adapter_front = 1) and `adapter2` (3') are the two parts of ONE linked adapter
                               * ADAPTER1...ADAPTER2: the 5' part is searched first, the 3' part in what follows it             */
    int32_t linked_required;  /* bit 0: no match without the 5' part, bit 1: without the 3' part (`-a A...B`: 1, `-g A...B`: 3)  */
} mirge_trim;
int mirge_reads_parse_trim(mirge_ctx* ctx, const char* text, int64_t nbytes, int32_t format, int32_t min_len,
                           const mirge_trim* trim, mirge_reads** out, int64_t* n_records);
/* The same with the reference's UMI handling: replaces the UMI branches of the per-chunk worker
 * (digest.py:334-365), `UMIParser` (:305-315) and the UMI stage of baking (:164-205) for one file.
 *   front, back  -umi f,b: read[f : len - b] is the insert (Python's s[f:-b]; s[f:] when b == 0), the rest the UMI
 *   qiagen       --qiagenumi (:334-352): the dictionary key is the trimmed read + the `back` bases that follow the 3'
 *                adapter in the untrimmed line (`currentSeq.split(trimmed)[1][:len(adapter)+b][-b:]`, first occurrence,
 *                "" when the trimmed read is empty); counted once, after the last modifier; needs trim->adapter (3')
 *   dedup        -udd (:183-205): a count is a number of distinct UMI-tagged reads
 * The worker's length test applies to the insert (`len(pureSeq) >= min_len`, :360), with qiagen to the trimmed read (:349),
 * and baking's to the insert again (:173,192).
 * *out: RAW inserts in file order -- without dedup one per counted read; with dedup one per DISTINCT tagged read whose
 * insert passes, in the order the tagged reads first appeared, so that mirge_collapse of *out gives molecule counts and
 * first indices that are ranks in the reference's dictionary order.  mirge_reads_count(*out) = 'Trimmed Reads (all)'.
 * *tagged_out (dedup only; may be NULL): the distinct tagged reads with their counts (a collapse result): the rows of
 * <sample>_umiCounts.csv (:183-197) are those whose insert passes the length test.  umi == NULL: mirge_reads_parse_trim. */
typedef struct mirge_umi {
    int32_t front;
    int32_t back;
    int32_t qiagen;
    int32_t dedup;
} mirge_umi;
int mirge_reads_parse_umi(mirge_ctx* ctx, const char* text, int64_t nbytes, int32_t format, int32_t min_len,
                          const mirge_trim* trim, const mirge_umi* umi, mirge_reads** out, int64_t* n_records,
                          mirge_reads** tagged_out);
/* Several raw read sets as one, in the order given (the samples of a run, one file each, before the joint collapse
 * that replaces the per-file dicts and their outer join, digest.py:133-163,243).  The parts stay valid. */
int mirge_reads_concat(mirge_ctx* ctx, const mirge_reads* const* parts, int32_t n_parts, mirge_reads** out);
void mirge_reads_destroy(mirge_reads* reads);
int64_t mirge_reads_count(const mirge_reads* reads);
int64_t mirge_reads_total_bases(const mirge_reads* reads);
int32_t mirge_reads_n_samples(const mirge_reads* reads);
/* reads per storage group (width class x {no ambiguous call, has an N}): out[0 .. min(cap, groups)), returns the
 * number of groups.  Diagnostics and bench.py's units: "collapsed reads of the bulk group" is the largest entry. */
int32_t mirge_reads_group_counts(const mirge_reads* reads, int64_t* out, int32_t cap);
/* sequences back as ASCII: ascii_out[total_bases], offsets_out[n+1], in handle order */
int mirge_reads_unpack(mirge_ctx* ctx, const mirge_reads* reads, char* ascii_out, int64_t* offsets_out);

/* ---- collapse: replaces the dict merge of digest.py:141-163 and the sample matrix of
 * digest.py:237-245.  sample_ids[n] (host, may be NULL when n_samples == 1) gives the sample
 * of every raw read.  The result holds the U distinct sequences (grouped by width class; their order
 * inside a group is unspecified -- sort by first_index for the reference's dict order) and a
 * U x n_samples count matrix.                                                              */
int mirge_collapse(mirge_ctx* ctx, const mirge_reads* raw, const int32_t* sample_ids,
                   int32_t n_samples, mirge_reads** uniq, int64_t* n_uniq);
/* The same where raw read i stands for weights[i] (host, >= 1) copies: merges already collapsed dictionaries -- the
 * per-sample results a sharded run gathers on rank 0 -- into the sample matrix (the outer join of digest.py:243)
 * without expanding them.  first indices refer to the raw set as given. */
int mirge_collapse_weighted(mirge_ctx* ctx, const mirge_reads* raw, const int32_t* sample_ids, int32_t n_samples,
                            const uint32_t* weights, mirge_reads** uniq, int64_t* n_uniq);
/* The sample matrix of several samples from their per-sample dictionaries, entirely on the device (round 6; what the CLI's route for
 * several files calls): parts[s] = the collapse result of sample s alone (one count column); -> the unique reads of the union with an
 * n_parts-column count matrix, as mirge_collapse with sample ids over the samples' raw reads would give -- the outer join of
 * mirge/libs/digest.py:243 -- without putting the raw reads of all samples through one table.  The parts stay valid. */
int mirge_collapse_merge(mirge_ctx* ctx, const mirge_reads* const* parts, int32_t n_parts, mirge_reads** uniq, int64_t* n_uniq);
/* counts_out[U * n_samples] (row-major), first_index_out[U] (index of the first raw read, may be NULL) */
int mirge_collapse_fetch(mirge_ctx* ctx, const mirge_reads* uniq, uint32_t* counts_out,
                         int64_t* first_index_out);
/* order_out[k] (k < U) = index of the unique read that appeared k-th in the raw reads: the row order of the reference's
 * per-sample dictionary (insertion order, digest.py:158-163), from one device sort of the first indices. */
int mirge_collapse_order(mirge_ctx* ctx, const mirge_reads* uniq, int64_t* order_out);
/* order_out[k] (k < U) = index of the unique read in row k of the SORTED union of the sequences (Python string order): the
 * row order of the sample matrix of several samples (pandas `join(how='outer')` sorts its index, digest.py:243), from a
 * radix sort on the device instead of a host-side sort of U strings. */
int mirge_collapse_order_sorted(mirge_ctx* ctx, const mirge_reads* uniq, int64_t* order_out);
/* nonzero_out[n_samples] = unique reads with a count in each sample ('Trimmed Reads (unique)', digest.py:214) */
int mirge_collapse_nonzero(mirge_ctx* ctx, const mirge_reads* uniq, int64_t* nonzero_out);
/* attach a caller-made count matrix (U x n_samples, host) to a packed read set, e.g. after -rr */
int mirge_reads_set_counts(mirge_ctx* ctx, mirge_reads* reads, const uint32_t* counts, int32_t n_samples);

/* ---- cascade: replaces bwtAlign + alignPlusParse (manifoldAlign.py:12-146): the n_pass
 * bowtie runs, their FASTA/SAM round trips and the DataFrame updates.  libs[p] == NULL skips
 * pass p.  Every pass sees the rows no earlier pass annotated (the reference tests annotFlag only from
 * pass 2 on, :120,129, but its passes 0 and 1 are disjoint by length, :93,104, so this is the same).  Result per read: pass index (or MIRGE_NO_PASS), reference index in libs[pass],
 * 0-based offset in that reference, mismatches.
 * The call returns with the small read groups' streams not yet joined into the ctx stream: the library's next entry point on
 * this ctx makes the join (mirge_count_join puts the bulk read group's part of its work in front of it).  Work the CALLER
 * enqueues on that stream is ordered behind a cascade only after such a call (mirge_ctx_sync, for one).  */
int mirge_cascade_run(mirge_ctx* ctx, const mirge_reads* reads, const mirge_lib* const* libs,
                      const mirge_policy* policies, int32_t n_pass, mirge_result** out);
/* optional: build NOW what a cascade over `reads` (raw or collapsed: only the read lengths present matter) builds on first
 * use -- merged libraries, probe tables, plan tables -- and wait for it.  The reference pays this as bowtie-build, once per
 * library release; here it is device work of a process's first sample (timed separately by bench.py's cli_path). */
int mirge_cascade_prepare(mirge_ctx* ctx, const mirge_reads* reads, const mirge_lib* const* libs, const mirge_policy* policies,
                          int32_t n_pass);
/* How the ctx's current cascade configuration (the last mirge_cascade_prepare / _run / mirge_collapse_cascade) walks the bulk
 * group of one-word reads: walks[0] = walks over its survivor list, walks[1] = passes answered by one whole-read lookup
 * ("-v 0", or "-n 0" inside the seed, over a small library: manifoldAlign.py:85 passes 0 and 3) that ride in another pass's
 * walk, walks[2] = passes in all.  Nothing in the reference corresponds to it: diagnostics for tests and bench.py. */
int mirge_cascade_walks(mirge_ctx* ctx, int32_t* walks);
/* Test / measurement aid: the constant-rate clock ticks every workgroup of the last PROFILED bulk-cascade launch took (its
 * segment of the reads through all passes), ticks_out[0 .. min(cap, *grid_out)), and the clock's rate in kHz: the longest
 * workgroup's share of the launch says whether one read's candidate list is what the launch waited for. */
int mirge_cascade_wg_times(mirge_ctx* ctx, uint32_t* ticks_out, int32_t cap, int32_t* grid_out, int32_t* khz_out);
/* mirge_collapse followed by mirge_cascade_run for ONE sample, as one call: the bulk read group's passes are queued on
 * the GPU behind the collapse kernels before the host has read the unique counts back (the kernels take the count
 * from device memory), so the GPU does not idle across the collapse's host synchronisation.  Same results; falls
 * back to the two calls when the bulk group is not on the partitioned key path.  Replaces the baking -> bwtAlign
 * hand-over of mirge/__main__.py:140-157 for a single sample. */
int mirge_collapse_cascade(mirge_ctx* ctx, const mirge_reads* raw, const mirge_lib* const* libs, const mirge_policy* policies,
                           int32_t n_pass, mirge_reads** uniq, int64_t* n_uniq, mirge_result** out);
int mirge_result_fetch(mirge_ctx* ctx, const mirge_result* res, int8_t* pass_out, int32_t* ref_out,
                       int32_t* off_out, int8_t* mm_out);
void mirge_result_destroy(mirge_result* res);

/* ---- count join: replaces the pandas sums of summary.py:686-698 (per-class), :749-752
 * (exact / isomiR group-by) and :769-771.  class_sums[n_pass * S]; exact/iso[n_mirna * S]
 * accumulate the counts of reads annotated in exact_pass / iso_pass per reference.          */
int mirge_count_join(mirge_ctx* ctx, const mirge_reads* uniq, const mirge_result* res,
                     int32_t exact_pass, int32_t iso_pass, int64_t n_mirna,
                     int64_t* class_sums, int64_t* exact, int64_t* iso);

/* same sums from host arrays (the reference's summarize() is handed a DataFrame, summary.py:677):
 * pass[n] (MIRGE_NO_PASS = unannotated), ref[n], counts[n * S] row-major                   */
int mirge_count_join_host(mirge_ctx* ctx, const int8_t* pass, const int32_t* ref, const uint32_t* counts,
                          int64_t n, int32_t n_samples, int32_t n_pass, int32_t exact_pass, int32_t iso_pass,
                          int64_t n_mirna, int64_t* class_sums, int64_t* exact, int64_t* iso);

/* ---- the per-read tables: replaces pdMapped.to_csv / pdUnmapped.to_csv (mirge/__main__.py:164-173) and the U-row
 * DataFrame of Python strings they print.  Host arrays in (what mirge_reads_unpack / mirge_result_fetch /
 * mirge_collapse_fetch returned), the reference's bytes out, formatted on the host's cores.  rows[k] = read printed in
 * row k; a read with pass >= 0 goes to mapped_path, the others to unmapped_path (either may be NULL).  Columns:
 * Sequence, annotFlag, n_name_cols name columns (pass p writes its reference name into column col_of_pass[p]),
 * S counts.  name_data[p] / name_off[p] / name_n[p]: the reference names of pass p's library.  No GPU involved. */
int mirge_annotation_csv(const char* mapped_path, const char* unmapped_path, const char* header,
                         const char* seq_ascii, const int64_t* seq_off, const int8_t* pass, const int32_t* ref,
                         const uint32_t* counts, int32_t n_samples, const int64_t* rows, int64_t n_rows,
                         int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                         const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n);

/* The same two files straight from the device-resident run (what the CLI's route calls): the rows are formatted by kernels
 * from the packed unique reads, their count matrix and the cascade's annotation, only the files' text crosses PCIe.
 * rows / header / columns / names as above.  Returns -4 when a reference name would need CSV quoting: format on the host. */
int mirge_annotation_csv_device(mirge_ctx* ctx, const mirge_reads* uniq, const mirge_result* res, const char* mapped_path,
                                const char* unmapped_path, const char* header, const int64_t* rows, int64_t n_rows,
                                int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                                const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n);

/* ---- the sharded run's parallel tail (round 6).  The reference's ONE mapped.csv / unmapped.csv hold the sorted union of all
 * samples' sequences with one count column per sample (the outer join of mirge/libs/digest.py:243, written at
 * mirge/__main__.py:164-173).  With one sample per GPU the union's key space is cut into one range per rank by the first 21
 * bases of that order; every rank merges, annotates, orders and formats ITS stretch of the two files.
 *   mirge_reads_range_sample   k evenly spaced quantiles of a dictionary's first-word sort keys (3 bits per base: end 0, A 1, C 2,
 *                              G 3, N 4, T 5; 21 bases, most significant first -- Python's string order).  The ranks pool them.
 *   mirge_reads_range_split    the dictionary as host arrays ordered by (owner range, handle index): range q =
 *                              [splitters[q-1], splitters[q]) of that key; bounds_out[q] .. bounds_out[q+1] are its rows.
 *                              ascii_out / off_out as mirge_reads_unpack, counts_out [n][n_samples].  n_parts <= 256.
 *   mirge_annotation_csv_device_sizes   bytes_out[0] / [1] = bytes the listed rows take in mapped.csv / unmapped.csv.
 *   mirge_annotation_csv_device_at      the rows' text at the given offsets of the two EXISTING files (no header, no
 *                                       truncation): the caller created them at their final size from every rank's sizes. */
int mirge_reads_range_sample(mirge_ctx* ctx, const mirge_reads* uniq, int32_t k, uint64_t* keys_out);
int mirge_reads_range_split(mirge_ctx* ctx, const mirge_reads* uniq, const uint64_t* splitters, int32_t n_parts, char* ascii_out,
                            int64_t* offsets_out, uint32_t* counts_out, int64_t* bounds_out);
int mirge_annotation_csv_device_sizes(mirge_ctx* ctx, const mirge_reads* uniq, const mirge_result* res, const int64_t* rows,
                                      int64_t n_rows, int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                                      const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n,
                                      int64_t* bytes_out);
int mirge_annotation_csv_device_at(mirge_ctx* ctx, const mirge_reads* uniq, const mirge_result* res, const char* mapped_path,
                                   const char* unmapped_path, int64_t mapped_off, int64_t unmapped_off, const int64_t* rows,
                                   int64_t n_rows, int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                                   const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n);

/* ---- per-position variant tally / A-to-I counting core (BASELINE config 5): replaces
 * align2TargetSeq / judgeAllign / A2IEditing / mismatchCountAnalysis (mirge2_tRF_a2i.py:246-518) and the membership
 * rules of a2i_editing (:988-1016) for the reads the cascade annotated to a miRNA in exact_pass / iso_pass.
 *   fam_of_ref[n_mirna]   family (merged miRNA name) of every reference of the miRNA library, -1 = none
 *   target_ascii/off      canonical sequence of each family (<org>_mirna_SNP_pseudo_<db>.fa), <= 32 nt
 *   retained[n reads]     the genome filter's answer per read in handle order (:1085-1096), NULL = every read
 *   freq[S]               1e6 / Filtered miRNA Reads per sample (an isomiR read is a member iff count*freq >= 1 somewhere)
 * out: fam_tables[5][n_fam][S] = n_seqs, seq_true, count_true, canon, kept_exact;
 *      census[n_fam][32][16][3][S] (position, target base * 4 + read base, variant raw / accepted / accepted+retained),
 *      read base != target base and position < len - 5 only;  A=0 C=1 G=2 T=3;
 *      diag_out/state_out[n reads] (may be NULL): diagonal of the alignment; 1 accepted, 0 rejected, -1 not a member. */
#define MIRGE_TALLY_POSITIONS 32
int mirge_variant_tally(mirge_ctx* ctx, const mirge_reads* uniq, const mirge_result* res, int32_t exact_pass,
                        int32_t iso_pass, const int32_t* fam_of_ref, int64_t n_mirna, const char* target_ascii,
                        const int64_t* target_off, int64_t n_fam, const uint8_t* retained, const double* freq,
                        int64_t* fam_tables, int64_t* census, int8_t* diag_out, int8_t* state_out);

/* ---- isomiR typing for the miRTop GFF3: replaces the per-read difflib.Differ diff and list
 * rewriting of create_gff (mirge/libs/summary.py:204-470) with one kernel.  For every read the cascade annotated in
 * exact_pass / iso_pass: the canonical sequence of its miRNA name (master_of_ref -> master tables), the precursor and
 * the canonical's position in it (what summary.py:147-186 looks up), and from those the record the reference prints:
 * type, precursor coordinates, Variant string, Cigar string.  slot_of_read[read] = row of records_out (-1: skip);
 * a record is MIRGE_ISO_RECORD_BYTES: int32 start, int32 end, uint8 kind (0 none, 1 ref_miRNA, 2 isomiR), uint8 pad,
 * uint16 variant length, uint16 cigar length, char text[320] (variant then cigar).                                 */
#define MIRGE_ISO_RECORD_BYTES 336
int mirge_isomir_type(mirge_ctx* ctx, const mirge_reads* uniq, const mirge_result* res, int32_t exact_pass, int32_t iso_pass,
                      const int32_t* master_of_ref, int64_t n_mirna, const char* master_ascii, const int32_t* master_off,
                      const int32_t* pre_of_master, const int32_t* start0, int64_t n_master, const char* pre_ascii,
                      const int32_t* pre_off, int64_t n_pre, const int32_t* slot_of_read, int64_t n_rows, void* records_out);
/* the GFF3 file from those records (summary.py:60-64 header, :204 / :465 lines; UID = miRgeEssential.UID), formatted on
 * the host's cores; rows with kind 0 print nothing.  read_ascii/read_off, counts[.. * S] are per ROW when read_of_row is NULL;
 * otherwise they are the caller's whole table of n_reads unique reads and row k prints entry read_of_row[k] of it. */
int mirge_gff_write(const char* path, const char* head, const char* source, const void* records, int64_t n_rows,
                    const char* read_ascii, const int64_t* read_off, const uint32_t* counts, int32_t n_samples,
                    const int32_t* name_of_row, const char* name_data, const int64_t* name_off, int64_t n_names,
                    const int32_t* parent_of_row, const char* parent_data, const int64_t* parent_off, int64_t n_parents,
                    const int64_t* read_of_row, int64_t n_reads);
/* The same file from the DEVICE-resident run (round 6; what the CLI's -gff route calls): the rows are chosen (the exact-miRNA rows of
 * the mapped frame in frame order, then its isomiR rows: summary.py:50-60), typed, measured and formatted by kernels where the reads,
 * counts and annotation lie; only the file's text crosses PCIe.  order[k] = handle index of the read in row k of the run's frame
 * (mirge_collapse_order / mirge_collapse_order_sorted); the typing tables as mirge_isomir_type; name_of_ref / parent_of_ref
 * [n_mirna] index the two string tables (-1: the reference prints no line for reads of that name).  *n_lines_out = lines below
 * the head.  Replaces create_gff's per-read loop and its writes (mirge/libs/summary.py:204-470). */
int mirge_gff_write_device(mirge_ctx* ctx, const mirge_reads* uniq, const mirge_result* res, int32_t exact_pass, int32_t iso_pass,
                           const int32_t* master_of_ref, int64_t n_mirna, const char* master_ascii, const int32_t* master_off,
                           const int32_t* pre_of_master, const int32_t* start0, int64_t n_master, const char* pre_ascii,
                           const int32_t* pre_off, int64_t n_pre, const int32_t* name_of_ref, const char* name_data,
                           const int64_t* name_off, int64_t n_names, const int32_t* parent_of_ref, const char* parent_data,
                           const int64_t* parent_off, int64_t n_parents, const int64_t* order, const char* path, const char* head,
                           const char* source, int64_t* n_lines_out);

/* ---- measurement (bench.py): HIP events on the ctx stream ---- */
int mirge_ctx_timer_start(mirge_ctx* ctx);
int mirge_ctx_timer_stop(mirge_ctx* ctx, double* ms_out);
int mirge_ctx_profile_enable(mirge_ctx* ctx, int32_t on);
/* record only launches whose name contains `substr` (NULL or "" = every launch) */
int mirge_ctx_profile_only(mirge_ctx* ctx, const char* substr);
/* on = 0: the bracketed cascade launches are timed but the reads handed to each pass (their "units") are no longer read
 * back -- a device-to-host copy per call on the critical path (bench.py's timed region: the units of the same batch are
 * known from its profiled warm-up steps) */
int mirge_ctx_profile_units(mirge_ctx* ctx, int32_t on);
int mirge_ctx_profile_reset(mirge_ctx* ctx);
int32_t mirge_ctx_profile_count(mirge_ctx* ctx);
int mirge_ctx_profile_get(mirge_ctx* ctx, int32_t i, char* name_out, int32_t name_cap,
                          int64_t* launches, double* total_ms, double* units);

#ifdef __cplusplus
}
#endif
#endif
