// native_host.hpp -- the library's PURE-HOST functions: no HIP call, no device pointer.  Included by mirge_native.hip like the
// other native_*.hpp, and -- on its own, with g++ -fsanitize=address,undefined -- by tests/hostsim/host_only_main.cpp: the GPU pool
// has no device sanitizer, and these ~400 lines of raw-cursor formatting, index checks and text rebuilding are where a host-side
// memory error would live ("race detection / sanitizers").  What is here: the error state (fail / mirge_last_error),
// the host timing hook, the mapped.csv / unmapped.csv formatter (mirge_annotation_csv), the GFF3 writer (mirge_gff_write), the
// text of a merged library, the argument checks of mirge_lib_create_packed.
#pragma once
// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define CHECK(expr)            \
    do {                       \
        int _c = (expr);       \
        if (_c != 0) return _c; \
    } while (0)


extern "C" const char* mirge_last_error(void) { return g_err.c_str(); }

// MIRGE_HOST_TIMING=1: host microseconds spent in the stages of a call, to stderr (enqueue-bound phases)
static std::chrono::steady_clock::time_point g_last_exit = std::chrono::steady_clock::now();
struct HostClock {
    const char* what;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit HostClock(const char* w) : what(w), t0(std::chrono::steady_clock::now()) {
        static const bool e = std::getenv("MIRGE_HOST_TIMING") != nullptr;
        on = e;
        if (on) std::fprintf(stderr, "[host] %s entered %.1f us after the previous call returned\n", what,
                             std::chrono::duration<double, std::micro>(t0 - g_last_exit).count());
    }
    ~HostClock() { if (on) g_last_exit = std::chrono::steady_clock::now(); }
    void lap(const char* stage) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[host] %s/%s %.1f us\n", what, stage, std::chrono::duration<double, std::micro>(t - t0).count());
        t0 = t;
    }
};


namespace {
struct CsvNames {  // reference names of one pass: one ASCII blob + offsets (n + 1); data == nullptr: pass has no library
    const char* data = nullptr;
    const int64_t* off = nullptr;
    int64_t n = 0;
};

// output of one formatting thread: a flat buffer written through a raw cursor (std::string::push_back per comma was a
// third of the formatting time)
struct CsvBuf {
    char* mem = nullptr;   // malloc'ed: growing must not zero-fill 200 MB that are about to be overwritten
    size_t cap = 0, n = 0;
    CsvBuf() = default;
    CsvBuf(const CsvBuf&) = delete;
    CsvBuf& operator=(const CsvBuf&) = delete;
    ~CsvBuf() { std::free(mem); }
    bool room(size_t extra) {
        if (n + extra <= cap) return true;
        const size_t want = std::max(cap * 2, n + extra + (1u << 16));
        char* m = (char*)std::realloc(mem, want);
        if (!m) return false;
        mem = m; cap = want;
        return true;
    }
    void put(char c) { mem[n++] = c; }
    void put(const char* s, size_t len) { std::memcpy(mem + n, s, len); n += len; }
    const char* data() const { return mem; }
    size_t size() const { return n; }
};

// pandas.to_csv quoting (csv.QUOTE_MINIMAL): quote a field that holds the delimiter, a quote or a line break.
// The caller has made room for 2 * len + 2 bytes.
inline void csv_field(CsvBuf& out, const char* s, size_t len) {
    bool q = false;
    for (size_t i = 0; i < len; i++) q |= s[i] == ',' || s[i] == '"' || s[i] == '\n' || s[i] == '\r';
    if (!q) { out.put(s, len); return; }
    out.put('"');
    for (size_t i = 0; i < len; i++) { if (s[i] == '"') out.put('"'); out.put(s[i]); }
    out.put('"');
}
inline void csv_uint(CsvBuf& out, uint64_t v) {  // room for 20 digits
    char buf[24];
    int k = 24;
    do { buf[--k] = (char)('0' + v % 10); v /= 10; } while (v);
    out.put(buf + k, (size_t)(24 - k));
}
inline void csv_uint(std::string& out, uint64_t v) {  // (the GFF writer's small tables)
    char buf[24];
    int k = 24;
    do { buf[--k] = (char)('0' + v % 10); v /= 10; } while (v);
    out.append(buf + k, (size_t)(24 - k));
}
}  // namespace

// rows[k] (k < n_rows) = index of the read printed in row k (the caller's row order: first appearance for one sample,
// sorted sequences for several).  A read goes to `mapped_path` when pass[i] >= 0, else to `unmapped_path` (either may
// be NULL).  Columns: Sequence, annotFlag, one name column per pass column (col_of_pass[p] = which column pass p
// writes, -1 = none; n_name_cols columns in all), then the S counts.  `header` is the first line, written as given.
extern "C" int mirge_annotation_csv(const char* mapped_path, const char* unmapped_path, const char* header,
                                    const char* seq_ascii, const int64_t* seq_off, const int8_t* pass, const int32_t* ref,
                                    const uint32_t* counts, int32_t S, const int64_t* rows, int64_t n_rows,
                                    int32_t n_pass, const int32_t* col_of_pass, int32_t n_name_cols,
                                    const char* const* name_data, const int64_t* const* name_off, const int64_t* name_n) {
    if (!header || !seq_off || !pass || !ref || !counts || !rows || S < 1 || n_rows < 0 || n_pass < 1 || n_pass > MIRGE_MAX_PASSES ||
        !col_of_pass || n_name_cols < 0 || !name_data || !name_off || !name_n || (n_rows > 0 && !seq_ascii))
        return fail(-1, "mirge_annotation_csv: bad argument");
    CsvNames nm[MIRGE_MAX_PASSES];
    for (int p = 0; p < n_pass; p++) {
        nm[p].data = name_data[p]; nm[p].off = name_off[p]; nm[p].n = name_n[p];
        if (col_of_pass[p] >= n_name_cols) return fail(-1, "mirge_annotation_csv: column index out of range");
    }
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const int T = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n_rows / 16384));
    std::vector<CsvBuf> bufm((size_t)T), bufu((size_t)T);
    std::vector<int> bad((size_t)T, 0);
    auto work = [&](int t) {
        const int64_t lo = n_rows * t / T, hi = n_rows * (t + 1) / T;
        CsvBuf &M = bufm[(size_t)t], &U = bufu[(size_t)t];
        if (!M.room((size_t)(hi - lo) * 56)) { bad[(size_t)t] = 2; return; }
        const size_t fixed = 2 + (size_t)n_name_cols + (size_t)S * 21 + 1 + 4;  // flag, commas, counts, newline, the quotes of two quoted fields
        for (int64_t k = lo; k < hi; k++) {
            // the rows come in the order of first appearance, the arrays in the order the device emitted the reads: every row
            // is four cache misses unless they are asked for ahead
            if (k + 16 < hi) {
                const int64_t j = rows[k + 16];
                __builtin_prefetch(&pass[j]); __builtin_prefetch(&ref[j]); __builtin_prefetch(&counts[(size_t)j * S]);
                __builtin_prefetch(&seq_off[j]);
            }
            if (k + 8 < hi) __builtin_prefetch(seq_ascii + seq_off[rows[k + 8]]);
            const int64_t i = rows[k];
            const int p = pass[i];
            if (p >= n_pass) { bad[(size_t)t] = 1; continue; }
            CsvBuf& out = p >= 0 ? M : U;
            if ((p >= 0 ? mapped_path : unmapped_path) == nullptr) continue;
            const size_t slen = (size_t)(seq_off[i + 1] - seq_off[i]);
            const int col = p >= 0 ? col_of_pass[p] : -1;
            const char* name = nullptr;
            size_t nlen = 0;
            if (col >= 0) {
                const int32_t r = ref[i];
                if (!nm[p].data || r < 0 || r >= nm[p].n) { bad[(size_t)t] = 1; continue; }
                name = nm[p].data + nm[p].off[r];
                nlen = (size_t)(nm[p].off[r + 1] - nm[p].off[r]);
            }
            if (!out.room(2 * slen + 2 * nlen + fixed)) { bad[(size_t)t] = 2; return; }
            csv_field(out, seq_ascii + seq_off[i], slen);
            out.put(',');
            out.put(p >= 0 ? '1' : '0');
            for (int cidx = 0; cidx < n_name_cols; cidx++) {
                out.put(',');
                if (cidx == col) csv_field(out, name, nlen);
            }
            for (int s = 0; s < S; s++) { out.put(','); csv_uint(out, counts[(size_t)i * S + s]); }
            out.put('\n');
        }
    };
    HostClock hc("annotation_csv");
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    hc.lap("format");
    for (int t = 0; t < T; t++) {
        if (bad[(size_t)t] == 2) return fail(-3, "mirge_annotation_csv: out of host memory");
        if (bad[(size_t)t]) return fail(-1, "mirge_annotation_csv: pass or reference index out of range");
    }
    for (int which = 0; which < 2; which++) {
        const char* path = which == 0 ? mapped_path : unmapped_path;
        if (!path) continue;
        // every thread writes its own chunk at its own offset (pwrite): a 200 MB table is bound by the copy into the
        // page cache, which one thread does at a fraction of the machine's memory bandwidth.  (A shared mapping of the
        // sized file with one memcpy per thread was 3.5 x slower on the 256-thread host of the GPU box: page faults on
        // one mapping contend more than positional writes do.)
        const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return fail(-8, std::string("cannot write ") + path);
        const size_t hl = std::strlen(header);
        std::vector<size_t> at((size_t)T + 1, hl);
        for (int t = 0; t < T; t++) at[(size_t)t + 1] = at[(size_t)t] + (which == 0 ? bufm[(size_t)t] : bufu[(size_t)t]).size();
        bool ok = ::pwrite(fd, header, hl, 0) == (ssize_t)hl;
        std::vector<int> werr((size_t)T, 0);
        auto put = [&](int t) {
            const CsvBuf& b = which == 0 ? bufm[(size_t)t] : bufu[(size_t)t];
            size_t done = 0;
            while (done < b.size()) {
                const ssize_t w = ::pwrite(fd, b.data() + done, b.size() - done, (off_t)(at[(size_t)t] + done));
                if (w <= 0) { werr[(size_t)t] = 1; return; }
                done += (size_t)w;
            }
        };
        std::vector<std::thread> wt;
        for (int t = 1; t < T; t++) wt.emplace_back(put, t);
        put(0);
        for (auto& x : wt) x.join();
        for (int t = 0; t < T; t++) ok = ok && !werr[(size_t)t];
        ok = (::close(fd) == 0) && ok;
        if (!ok) return fail(-8, std::string("write error on ") + path);
        hc.lap("write");
    }
    return 0;
}



namespace {
// miRgeEssential.UID (:364-370): prefix-length-, then two symbols of a 32-letter alphabet per 5-mer (value / 32,
// value % 32, A C G T = 0..3, first base most significant); a last chunk of k < 5 bases is numbered after all shorter
// k-mers (offsets 0, 4, 20, 84) and printed as one symbol below 32, two from there on
inline void uid_append(std::string& out, const char* s, size_t n) {
    static const char AL[] = "BD0EF1HI2JK3LM4NO5PQ6RS7UV8WX9YZ";
    static const int OFFS[5] = {0, 0, 4, 20, 84};
    for (size_t at = 0; at < n; at += 5) {
        const size_t k = std::min<size_t>(5, n - at);
        int v = 0;
        for (size_t t = 0; t < k; t++) v = v * 4 + (s[at + t] == 'A' ? 0 : s[at + t] == 'C' ? 1 : s[at + t] == 'G' ? 2 : 3);
        if (k == 5) { out.push_back(AL[v / 32]); out.push_back(AL[v % 32]); }
        else {
            v += OFFS[k];
            if (v < 32) out.push_back(AL[v]);
            else { out.push_back(AL[v / 32]); out.push_back(AL[v % 32]); }
        }
    }
}
}  // namespace

// The GFF3 body: one line per row with kind != 0, in row order (summary.py:204, :465).  name_of_row / parent_of_row
// index two string tables (the miRNA name as printed, its precursor's name); `head` = the four '#' lines.
extern "C" int mirge_gff_write(const char* path, const char* head, const char* source, const void* records, int64_t n_rows,
                               const char* read_ascii, const int64_t* read_off, const uint32_t* counts, int32_t S,
                               const int32_t* name_of_row, const char* name_data, const int64_t* name_off, int64_t n_names,
                               const int32_t* parent_of_row, const char* parent_data, const int64_t* parent_off, int64_t n_parents,
                               const int64_t* read_of_row, int64_t n_reads) {
    // read_of_row (may be NULL): row k prints read and counts number read_of_row[k] of read_off / counts (n_reads of them) --
    // the caller hands over its whole table of unique reads instead of gathering the rows' reads first; NULL: number k
    if (!path || !head || !source || n_rows < 0 || S < 1 || !read_off || !name_off || !parent_off ||
        (n_rows > 0 && (!records || !read_ascii || !counts || !name_of_row || !parent_of_row)))
        return fail(-1, "mirge_gff_write: bad argument");
    const MirgeIsoRec* rec = static_cast<const MirgeIsoRec*>(records);
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const int T = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n_rows / 8192));
    std::vector<std::string> buf((size_t)T);
    std::vector<int> bad((size_t)T, 0);
    const std::string src(source);
    auto work = [&](int t) {
        std::string& out = buf[(size_t)t];
        for (int64_t k = n_rows * t / T; k < n_rows * (t + 1) / T; k++) {
            const MirgeIsoRec& r = rec[k];
            if (r.kind == 0) continue;
            const int32_t ni = name_of_row[k], pi = parent_of_row[k];
            if (ni < 0 || ni >= n_names || pi < 0 || pi >= n_parents || (size_t)r.vlen + r.clen > MIRGE_ISO_TEXT) { bad[(size_t)t] = 1; continue; }
            const char* nm = name_data + name_off[ni];
            const size_t nl = (size_t)(name_off[ni + 1] - name_off[ni]);
            const int64_t ri = read_of_row ? read_of_row[k] : k;
            if (read_of_row && (ri < 0 || ri >= n_reads)) { bad[(size_t)t] = 1; continue; }
            const char* rd = read_ascii + read_off[ri];
            const size_t rl = (size_t)(read_off[ri + 1] - read_off[ri]);
            out.append(nm, nl); out.push_back('\t'); out += src; out.push_back('\t');
            out += r.kind == 1 ? "ref_miRNA" : "isomiR";
            out.push_back('\t'); out += std::to_string(r.start); out.push_back('\t'); out += std::to_string(r.end);
            out += "\t.\t+\t.\tRead="; out.append(rd, rl); out += "; UID=";
            bool has_n = false;
            for (size_t q = 0; q < rl; q++) has_n |= rd[q] == 'N';
            if (has_n) out.push_back('.');
            else { out += r.kind == 1 ? "ref-" : "iso-"; out += std::to_string(rl); out.push_back('-'); uid_append(out, rd, rl); }
            out += "; Name="; out.append(nm, nl);
            out += "; Parent="; out.append(parent_data + parent_off[pi], (size_t)(parent_off[pi + 1] - parent_off[pi]));
            out += "; Variant="; out.append(r.text, r.vlen);
            out += "; Cigar="; out.append(r.text + r.vlen, r.clen);
            std::string ex;
            for (int s = 0; s < S; s++) { if (s) ex.push_back(','); csv_uint(ex, counts[(size_t)ri * S + s]); }
            out += "; Expression="; out += ex; out += "; Filter=Pass; Hits="; out += ex; out.push_back('\n');
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < T; t++) if (bad[(size_t)t]) return fail(-1, "mirge_gff_write: name index or record out of range");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(-8, std::string("cannot write ") + path);
    bool ok = std::fputs(head, f) >= 0;
    for (int t = 0; t < T && ok; t++) ok = buf[(size_t)t].empty() || std::fwrite(buf[(size_t)t].data(), 1, buf[(size_t)t].size(), f) == buf[(size_t)t].size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) return fail(-8, std::string("write error on ") + path);
    return 0;
}

// One library out of the members of a run of same-policy passes: their references in order (same bases, same separators), so
// that a position in the merged text minus the member's start is the position in the member's own text.  seq / off: what
// mirge_lib_create takes.
static void merged_library_text(const MirgeHostLib* const* members, int n, std::string& seq, std::vector<int64_t>& off) {
    seq.clear();
    off.assign(1, 0);
    for (int i = 0; i < n; i++) {
        const MirgeHostLib& h = *members[i];
        for (int64_t r = 0; r < h.n_refs; r++) {
            for (uint64_t g = h.ref_start[(size_t)r]; g + 1 < h.ref_start[(size_t)r + 1]; g++) {
                const bool bad = (h.inv[g >> 6] >> (g & 63)) & 1ull;
                seq.push_back(bad ? 'N' : "ACGT"[(h.T[g >> 5] >> (2 * (g & 31))) & 3ull]);
            }
            off.push_back((int64_t)seq.size());
        }
    }
}

// what mirge_lib_create_packed requires of a packed image before anything is uploaded (a cache of another layout, a damaged file)
static int lib_packed_args_check(const uint64_t* T, int64_t n_T, const uint64_t* inv, int64_t n_inv, const uint32_t* ref_start, int64_t n_refs,
                                 uint64_t total, int32_t kmax) {
    if (!T || !inv || !ref_start || n_refs < 0 || total >= 0xFFFFFFF0ull || kmax < 8 || kmax > MIRGE_KMAX ||
        n_T != (int64_t)((total + 31) / 32) + 8 || n_inv != (int64_t)((total + 63) / 64) + 4 || ref_start[n_refs] != (uint32_t)total)
        return fail(-1, "mirge_lib_create_packed: bad argument (a cache of another layout?)");
    // reference t occupies [ref_start[t], ref_start[t + 1] - 1) and its separator: starts ascend by at least one, from 0
    if (n_refs > 0 && ref_start[0] != 0) return fail(-1, "mirge_lib_create_packed: ref_start[0] is not 0 (a damaged cache?)");
    for (int64_t t = 0; t < n_refs; t++)
        if (ref_start[t + 1] <= ref_start[t])
            return fail(-1, "mirge_lib_create_packed: ref_start is not ascending at reference " + std::to_string(t) + " (a damaged cache?)");
    return 0;
}
