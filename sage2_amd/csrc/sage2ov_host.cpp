// sage2_amd/csrc/sage2ov_host.cpp -- host side of the C ABI (include/sage2ov.h).
//
// Step 1 (FASTA/FASTQ parsing, filter, canonical orientation, 2-bit packing, sort, dedupe) runs on the
// host for now and hands the packed unique reads to HBM; steps 2-3 are HIP kernels
// (sage2ov_device.hip).  The exact serial BFS of the reduce phase (economyGraph.cpp:513-564) is
// replayed here from device-computed hit lists.  There is no CPU fallback for the device work.
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <parallel/algorithm>
#include <string>
#include <thread>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <unordered_map>
#include <vector>
#include <omp.h>
#include <sched.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "sage2ov.h"
#include "sage2ov_internal.h"

using namespace s2;

namespace {

thread_local std::string g_create_error;

inline uint64_t rev2_host(uint64_t x) {
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
}
inline uint64_t bits64_host(const uint64_t* w, int nw, int bitpos) {
    int q = bitpos >> 6, r = bitpos & 63;
    uint64_t a = q < nw ? w[q] : 0;
    if (r == 0) return a;
    uint64_t b = q + 1 < nw ? w[q + 1] : 0;
    return (a << r) | (b >> (64 - r));
}
inline uint64_t mask_top_host(int nb) { return nb >= 32 ? ~0ull : (nb <= 0 ? 0ull : (~0ull << (64 - 2 * nb))); }
// reverse complement of an L-base word-packed read (left aligned), nw words in and out
inline void revcomp_words(const uint64_t* in, int nw, int L, uint64_t* out) {
    for (int c = 0; c < nw; c++) {
        int rem = L - 32 * c; uint64_t r;
        if (rem <= 0) r = 0;
        else if (rem >= 32) r = ~rev2_host(bits64_host(in, nw, 2 * (rem - 32)));
        else r = (~rev2_host(in[0] >> (64 - 2 * rem))) & mask_top_host(rem);
        out[c] = r;
    }
}
inline int flip_type_host(int t) { return t == 0 ? 3 : (t == 3 ? 0 : t); }   // utils.cpp:212

struct AdjEdge { uint32_t to; uint8_t type; uint8_t mark; uint32_t len; };

// SAGE2OV_TIMING: wall-clock laps of the host-side stages (file input, step 1, the writers) on stderr
struct HostLap {
    const bool on; const char* stage; std::chrono::steady_clock::time_point tp = std::chrono::steady_clock::now();
    HostLap(const sage2ov_ctx* c, const char* s);
    void operator()(const char* what) { if (!on) return; auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[%s] %-36s %8.1f ms\n", stage, what, std::chrono::duration<double, std::milli>(t - tp).count()); tp = t; }
};

}  // namespace

extern char** environ;
namespace s2 {
Options Options::from_env() {
    Options o;
    for (char** e = environ; e && *e; e++) {
        const char* v = *e;
        if (strncmp(v, "SAGE2OV_", 8) != 0 && strncmp(v, "LOCAL_WORLD_SIZE=", 17) != 0) continue;
        const char* eq = strchr(v, '='); if (!eq) continue;
        o.kv.emplace_back(std::string(v, eq - v), std::string(eq + 1));
    }
    std::sort(o.kv.begin(), o.kv.end());
    return o;
}
}  // namespace s2

struct sage2ov_ctx {
    sage2ov_config cfg{};
    std::string err;
    s2::Options opt = s2::Options::from_env();         // the SAGE2OV_* switches as they were when the context was created (sage2ov_options_reload re-reads them)
    // the device: opened by sage2ov_ctx_create, or (SAGE2OV_FLAG_ASYNC_DEVICE) on a helper thread that the first device() call joins
    mutable Device* dev_ = nullptr; mutable std::thread devOpen; mutable std::string devErr;
    bool gpu() const { return cfg.device != SAGE2OV_DEVICE_NONE; }
    Device* device() const { if (devOpen.joinable()) devOpen.join(); return dev_; }
    // ---- step 1 staging: variable-length word-packed canonical reads
    RawU64 pool, poolOff; RawU16 poolLen;
    // ... or, for ASCII handed over through sage2ov_reads_add_ascii on a GPU context, the raw bases: filter, 2-bit pack and canonical orientation
    // then run on the device (utils.cpp:144-166, :96-119; readLoader.cpp:195) when the reads are organised
    std::vector<char> ascii; std::vector<uint64_t> asciiOff;
    uint64_t totalReads = 0, goodReads = 0, totalBP = 0, smallReads = 0;
    std::vector<uint32_t> replayDense;                 // scratch of the host replay (read id -> dense index), all zero between calls
    // ---- organised reads (host copy, ids 1..N)
    uint64_t N = 0; int S = 0, maxL = 0; bool organized = false;
    RawU64 words; std::vector<uint16_t> len; RawU16 freq;
    // ---- step 2/3 state
    bool indexBuilt = false, probed = false, reciprocalDone = false, reduced = false, converted = false;
    sage2ov_index_stats istats{};
    sage2ov_overlap_stats ostats{};
    std::vector<FinalEdge> edges; bool edgesOnHost = false;
    SimplifiedGraph g4; bool g4Valid = false;
    double reduce_ms = 0, total_ms = 0;
    // multi-rank contexts: the survivors this rank's share of the reduce phase re-emitted = candidates [survBase, survBase + survCount) of the device list
    uint64_t survBase = 0, survCount = 0, removedPartial = 0; bool survivorsExchanged = true;

    int fail(int code, const std::string& m) { err = m; return code; }
};
namespace { HostLap::HostLap(const sage2ov_ctx* c, const char* s) : on(c->opt.flag("SAGE2OV_TIMING")), stage(s) {} }

namespace {

// ------------------------------------------------------------------------------------------ step 1
// filter + canonical + pack one read (readLoader.cpp:145-158, utils.cpp:144, readLoader.cpp:179-213)
// codes: 0..3 per base or 255 for anything that is not ACGTacgt
template <class VP, class VL>
inline void stage_codes(sage2ov_ctx* c, const uint8_t* codes, int L, VP& pool, VP& off, VL& lens, uint64_t& good, uint64_t& bp, uint64_t& small) {
    if (L <= (int)c->cfg.min_overlap) { small++; return; }
    const int nw = (L + 31) / 32;
    uint64_t f[34], r[34];
    if (nw > 32) { lens.push_back(0xFFFF); off.push_back(pool.size()); good++; bp += L; return; }   // too long: organise reports the limit
    for (int w = 0; w < nw; w++) f[w] = 0;
    for (int i = 0; i < L; i++) { if (codes[i] > 3) return; f[i >> 5] |= (uint64_t)codes[i] << (62 - 2 * (i & 31)); }
    // the forward strand is staged; the canonical orientation (readLoader.cpp:195) is chosen by the organiser (device or host)
    (void)r;
    off.push_back(pool.size()); lens.push_back((uint16_t)L);
    pool.insert(pool.end(), f, f + nw);
    good++; bp += L;
}
static uint8_t g_code[256];
struct CodeInit { CodeInit() { memset(g_code, 255, 256); g_code['A'] = g_code['a'] = 0; g_code['C'] = g_code['c'] = 1; g_code['G'] = g_code['g'] = 2; g_code['T'] = g_code['t'] = 3; } } g_code_init;
// 2-bit codes of 8 ASCII bases (first base in the lowest byte of x) as 16 bits, first base highest; false when a byte is not one of ACGTacgt
inline bool pack8(uint64_t x, uint64_t& out) {
    const uint64_t K1 = 0x0101010101010101ull;
    const uint64_t y = x & 0xDFDFDFDFDFDFDFDFull;                              // upper case
    const uint64_t cd = ((y >> 1) ^ (y >> 2)) & (3 * K1);                      // A 0, C 1, G 2, T 3 (bits 1 and 2 of the letters)
    const uint64_t b0 = cd & K1, b1 = (cd >> 1) & K1, both = b0 & b1;
    const uint64_t letter = (0x40 * K1) | (both << 4) | (b1 << 2) | ((b0 ^ b1) << 1) | (both ^ K1);      // the letter each code stands for: 41 43 47 54
    uint64_t t = __builtin_bswap64(cd);
    t = (t | (t >> 6)) & 0x000F000F000F000Full; t = (t | (t >> 12)) & 0x000000FF000000FFull; t = (t | (t >> 24)) & 0xFFFFull;
    out = t; return letter == y;
}
// stage_codes straight from the text of a sequence line; -1: the line holds white space (the general reader drops it: utils of fastAQReader.cpp:16-45), nothing staged
template <class VP, class VL>
inline int stage_ascii(sage2ov_ctx* c, const char* b, size_t Ls, VP& pool, VP& off, VL& lens, uint64_t& good, uint64_t& bp, uint64_t& small) {
    if (Ls > 1024) return -1;                                                    // (beyond the 32-word layout: the general reader's business)
    bool valid = true; uint64_t f[34]; const int L = (int)Ls, nw = (L + 31) / 32;
    for (int w = 0; w < nw; w++) f[w] = 0;
    int i = 0;
    for (; i + 8 <= L; i += 8) { uint64_t x, o; memcpy(&x, b + i, 8); valid &= pack8(x, o); f[i >> 5] |= o << (48 - 2 * (i & 31)); }
    for (; i < L; i++) { const uint8_t cd = g_code[(unsigned char)b[i]]; valid &= cd <= 3; f[i >> 5] |= (uint64_t)(cd & 3) << (62 - 2 * (i & 31)); }
    if (!valid) { for (size_t q = 0; q < Ls; q++) if ((unsigned char)b[q] <= ' ') return -1; }
    if (L <= (int)c->cfg.min_overlap) { small++; return 0; }
    if (!valid) return 0;
    off.push_back(pool.size()); lens.push_back((uint16_t)L);
    pool.insert(pool.end(), f, f + nw);
    good++; bp += L;
    return 0;
}
inline void canonicalise_words(uint64_t* f, int L) {        // readLoader.cpp:195: read < revcomp ? read : revcomp (tie: revcomp, same bytes)
    const int nw = (L + 31) / 32; uint64_t r[34];
    revcomp_words(f, nw, L, r);
    bool useF = false;
    for (int w = 0; w < nw; w++) { if (f[w] != r[w]) { useF = f[w] < r[w]; break; } }
    if (!useF) for (int w = 0; w < nw; w++) f[w] = r[w];
}
// worker threads of the host-side I/O helpers: the configured number, else what OpenMP and the CPU affinity allow, at most 16
// (many short parallel regions on an oversubscribed CPU share cost more than they give)
static int io_threads(const sage2ov_ctx* c) {
    if (c->cfg.host_threads) return (int)c->cfg.host_threads;
    int nt = omp_get_max_threads();
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) nt = std::min(nt, CPU_COUNT(&set));
    return std::max(1, std::min(nt, 16));
}

// minimal FASTA/FASTQ(.gz) record reader with kseq-like rules (fastAQReader.cpp:16-45)
struct SeqFile {
    gzFile fp = nullptr; std::vector<char> buf; size_t pos = 0, end = 0; bool eof = false, ioError = false; const char* win = nullptr;   // win: the current window (buf, or a mapped range)
    // gzip inflates at a few hundred MB/s on one thread: a reader thread fills the next 8 MB window while the caller splits the current one
    std::vector<char> nextBuf; std::thread reader; std::mutex mu; std::condition_variable cv; int nextN = 0; bool nextReady = false, wantNext = false, quit = false;
    void reader_loop() {
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return wantNext || quit; }); if (quit) return; wantNext = false; }
            const int n = gzread(fp, nextBuf.data(), (unsigned)nextBuf.size());
            { std::lock_guard<std::mutex> lk(mu); nextN = n; nextReady = true; }
            cv.notify_all();
            if (n <= 0) return;
        }
    }
    bool open(const char* p) {
        fp = gzopen(p, "r"); if (!fp) return false;
        gzbuffer(fp, 1 << 20); buf.resize(8 << 20); nextBuf.resize(8 << 20); win = buf.data();
        wantNext = true; reader = std::thread([this] { reader_loop(); });
        return true;
    }
    void open_range(const char* b, size_t n) { win = b; pos = 0; end = n; eof = true; }        // parse a range of a mapped file: one window, never refilled
    ~SeqFile() {
        if (reader.joinable()) { { std::lock_guard<std::mutex> lk(mu); quit = true; } cv.notify_all(); reader.join(); }
        if (fp) gzclose(fp);
    }
    bool refill() {
        if (eof) return false;
        int n;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return nextReady; }); n = nextN; nextReady = false; if (n > 0) { buf.swap(nextBuf); wantNext = true; } }
        if (n <= 0) { eof = true; if (n < 0) ioError = true; return false; }      // n < 0: corrupt or truncated gzip stream (gzerror), not an end of file
        cv.notify_all();
        win = buf.data(); end = (size_t)n; pos = 0; return true;
    }
    // next line (without the line end) appended to `s`; lines are found with memchr on the 8 MB window
    bool line(std::string& s) {
        s.clear(); bool any = false;
        for (;;) {
            if (pos >= end && !refill()) break;
            any = true;
            const char* b = win + pos; const char* nl = (const char*)memchr(b, '\n', end - pos);
            if (nl) { s.append(b, (size_t)(nl - b)); pos = (size_t)(nl - win) + 1; break; }
            s.append(b, end - pos); pos = end;
        }
        if (!s.empty() && s.back() == '\r') s.pop_back();
        return any;
    }
    std::string held, l; bool haveHeld = false;
    // The common shapes -- a FASTA record with one sequence line, a four-line FASTQ record -- when the whole record and the first
    // character behind it lie inside the current window: found with memchr, the sequence copied once, no per-line strings.
    // Anything else (multi-line sequences, white space inside a line, a record that crosses the window) is left to the general path.
    bool fast_next(std::string& out, size_t& len) {
        if (haveHeld || pos >= end) return false;
        const char* const b = win + pos; const char* const e = win + end;
        const char kind = *b; if (kind != '>' && kind != '@') return false;
        const char* nl1 = (const char*)memchr(b, '\n', (size_t)(e - b)); if (!nl1) return false;
        const char* sq = nl1 + 1; const char* nl2 = (const char*)memchr(sq, '\n', (size_t)(e - sq)); if (!nl2 || nl2 + 1 >= e) return false;
        size_t n = (size_t)(nl2 - sq); if (n && sq[n - 1] == '\r') n--;
        if (n == 0) return false;
        for (size_t i = 0; i < n; i++) if ((unsigned char)sq[i] <= ' ') return false;
        const char* after = nl2 + 1;
        if (kind == '>') { if (*after != '>') return false; }
        else {
            if (*after != '+') return false;
            const char* nl3 = (const char*)memchr(after, '\n', (size_t)(e - after)); if (!nl3) return false;
            const char* q = nl3 + 1; const char* nl4 = (const char*)memchr(q, '\n', (size_t)(e - q)); if (!nl4 || nl4 + 1 >= e) return false;
            size_t nq = (size_t)(nl4 - q); if (nq && q[nq - 1] == '\r') nq--;
            if (nq != n || nl4[1] != '@') return false;
            after = nl4 + 1;
        }
        out.append(sq, n); len = n; pos = (size_t)(after - win);
        return true;
    }
    // bases of the next record appended to `out` (white space dropped); false at end of file
    bool next(std::string& out, size_t& len) {
        if (fast_next(out, len)) return true;
        const size_t start = out.size();
        for (;;) { if (haveHeld) { l.swap(held); haveHeld = false; } else if (!line(l)) return false; if (!l.empty() && (l[0] == '>' || l[0] == '@')) break; }
        for (;;) {
            if (!line(l)) { len = out.size() - start; return true; }
            if (!l.empty() && (l[0] == '>' || l[0] == '@')) { held.swap(l); haveHeld = true; len = out.size() - start; return true; }
            if (!l.empty() && l[0] == '+') break;
            const size_t at = out.size(); out.append(l);                      // (white space inside a sequence line is rare: compact only then)
            bool ws = false; for (size_t x = at; x < out.size(); x++) ws |= (unsigned char)out[x] <= ' ';
            if (ws) { size_t w = at; for (size_t x = at; x < out.size(); x++) if ((unsigned char)out[x] > ' ') out[w++] = out[x]; out.resize(w); }
        }
        len = out.size() - start;
        size_t q = 0; while (q < len) { if (!line(l)) break; q += l.size(); }          // quality: as many characters as bases
        return true;
    }
};

// A plain (uncompressed) single file is mapped and cut at record starts into chunks that the threads parse, filter and pack
// independently -- the read ids do not depend on the input order (they are ranks after the sort).  FASTA: a '>' at a line start is
// always a header.  FASTQ: a quality line may start with '@' as well; in the four-line form a line starting with '@' is a header iff
// the line two below starts with '+', and every chunk is parsed strictly as four-line records (equal sequence and quality lengths).
// Anything else -- multi-line FASTQ, a file that is neither, a chunk that does not parse -- returns false and the sequential reader
// below takes the whole file.
struct StagePart { RawU64 pool, off; RawU16 lens; uint64_t good = 0, bp = 0, small = 0, records = 0; };      // what one thread staged
static bool parse_plain_file_parallel(sage2ov_ctx* c, const char* path, std::vector<StagePart>& parts) {
    const int fd = ::open(path, O_RDONLY); if (fd < 0) return false;
    struct stat sb; if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < (1 << 20)) { ::close(fd); return false; }
    const size_t size = (size_t)sb.st_size;
    const char* m = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0); ::close(fd);
    if (m == MAP_FAILED) return false;
    struct Unmap { const char* p; size_t n; ~Unmap() { munmap((void*)p, n); } } um{m, size};
    madvise((void*)m, size, MADV_SEQUENTIAL);
    if ((unsigned char)m[0] == 0x1f && (unsigned char)m[1] == 0x8b) return false;                          // gzip
    const char kind = m[0]; if (kind != '>' && kind != '@') return false;
    auto line_start_after = [&](size_t p) -> size_t { const char* nl = (const char*)memchr(m + p, '\n', size - p); return nl ? (size_t)(nl - m) + 1 : size; };
    auto boundary = [&](size_t from) -> size_t {                                                        // first record start at or after `from` (a line start)
        size_t p = from == 0 ? 0 : line_start_after(from - 1);
        while (p < size) {
            if (m[p] == kind) {
                if (kind == '>') return p;
                const size_t l1 = line_start_after(p), l2 = l1 < size ? line_start_after(l1) : size;
                if (l2 < size && m[l2] == '+') return p;
            }
            p = line_start_after(p);
        }
        return size;
    };
    const int nt = io_threads(c); HostLap lap(c, "input");
    const size_t nchunks = std::max<size_t>((size_t)nt, std::min<size_t>(4096, size >> 24));
    std::vector<size_t> cut(nchunks + 1); cut[0] = 0; cut[nchunks] = size;
    for (size_t x = 1; x < nchunks; x++) cut[x] = std::max(cut[x - 1], boundary((size * x) / nchunks));
    using Part = StagePart;
    parts.clear(); parts.resize(nt); bool bad = false;
    #pragma omp parallel num_threads(nt)
    {
        Part& P = parts[omp_get_thread_num()]; std::string seq; std::vector<uint8_t> codes;
        #pragma omp for schedule(dynamic, 1)
        for (int64_t x = 0; x < (int64_t)nchunks; x++) {
            if (cut[x] >= cut[x + 1]) continue;
            auto stage = [&](const char* b, size_t L) { codes.resize(L); for (size_t i = 0; i < L; i++) codes[i] = g_code[(unsigned char)b[i]]; stage_codes(c, codes.data(), (int)L, P.pool, P.off, P.lens, P.good, P.bp, P.small); P.records++; };
            if (kind == '>') {
                // records of one header line + one sequence line, packed straight from the mapping (eight bases per step); the first record that
                // is anything else -- a second sequence line, white space inside the line, the last record of the chunk -- hands the rest of the chunk to the general reader
                size_t p = cut[x]; const size_t e = cut[x + 1];
                while (p < e) {
                    const char* nl1 = (const char*)memchr(m + p, '\n', e - p); if (!nl1) break;
                    const char* sq = nl1 + 1; const char* nl2 = (const char*)memchr(sq, '\n', (size_t)(m + e - sq)); if (!nl2 || nl2 + 1 >= m + e || nl2[1] != '>') break;
                    size_t n = (size_t)(nl2 - sq); if (n && sq[n - 1] == '\r') n--;
                    if (n == 0) break;
                    const int st = stage_ascii(c, sq, n, P.pool, P.off, P.lens, P.good, P.bp, P.small);
                    if (st < 0) break;                                                                   // white space inside the line
                    P.records++; p = (size_t)(nl2 + 1 - m);
                }
                if (p < e) {
                    SeqFile f; f.open_range(m + p, e - p); size_t len = 0;
                    for (;;) { seq.clear(); if (!f.next(seq, len)) break; stage(seq.data(), len); }
                }
            } else {                                                                                    // strict four-line FASTQ
                size_t p = cut[x]; const size_t e = cut[x + 1];
                auto trimmed = [&](size_t a, size_t b) { size_t n = b - a; if (n && m[b - 1] == '\n') n--; if (n && m[a + n - 1] == '\r') n--; return n; };
                while (p < e) {
                    const size_t l1 = line_start_after(p), l2 = l1 < e ? line_start_after(l1) : e, l3 = l2 < e ? line_start_after(l2) : e, l4 = l3 < e ? line_start_after(l3) : e;
                    const size_t ns = trimmed(l1, l2), nq = trimmed(l3, l4);
                    bool ws = false; for (size_t i = 0; i < ns; i++) ws |= (unsigned char)m[l1 + i] <= ' ';
                    if (m[p] != '@' || l2 >= e || m[l2] != '+' || ns != nq || ns == 0 || ws) {
                        #pragma omp atomic write
                        bad = true;
                        break;
                    }
                    if (stage_ascii(c, m + l1, ns, P.pool, P.off, P.lens, P.good, P.bp, P.small) < 0) stage(m + l1, ns); else P.records++;     // (eight bases per step; longer than the 32-word layout: the table form)
                    p = l4;
                }
            }
        }
    }
    if (bad) return false;
    lap("split + filter + pack (threads)");
    return true;
}
// the threads' pools, one behind the other at the end of the context's staging arrays (each thread copies its own: the arrays are sized without being written)
static void commit_parts(sage2ov_ctx* c, std::vector<StagePart>& parts) {
    using Part = StagePart; const int nt = (int)parts.size(); HostLap lap(c, "input");
    std::vector<uint64_t> wordBase(nt + 1), readBase(nt + 1); wordBase[0] = c->pool.size(); readBase[0] = c->poolOff.size();
    for (int t = 0; t < nt; t++) { wordBase[t + 1] = wordBase[t] + parts[t].pool.size(); readBase[t + 1] = readBase[t] + parts[t].off.size(); }
    c->pool.resize(wordBase[nt]); c->poolOff.resize(readBase[nt]); c->poolLen.resize(readBase[nt]);
    #pragma omp parallel for num_threads(nt) schedule(static, 1)
    for (int t = 0; t < nt; t++) {
        Part& P = parts[t];
        if (!P.pool.empty()) memcpy(c->pool.data() + wordBase[t], P.pool.data(), P.pool.size() * sizeof(uint64_t));
        uint64_t* po = c->poolOff.data() + readBase[t]; const uint64_t base = wordBase[t];
        for (size_t i = 0; i < P.off.size(); i++) po[i] = base + P.off[i];
        if (!P.lens.empty()) memcpy(c->poolLen.data() + readBase[t], P.lens.data(), P.lens.size() * sizeof(uint16_t));
        RawU64().swap(P.pool); RawU64().swap(P.off); RawU16().swap(P.lens);
    }
    for (Part& P : parts) { c->goodReads += P.good; c->totalBP += P.bp; c->smallReads += P.small; c->totalReads += P.records; }
    lap("concatenation of the threads' pools");
}
static bool add_plain_file_parallel(sage2ov_ctx* c, const char* path) {
    std::vector<StagePart> parts; if (!parse_plain_file_parallel(c, path, parts)) return false;
    commit_parts(c, parts); return true;
}
// Two mate files are read alternately, file 1 first, until the file whose turn it is has no record left (inputReader.cpp:26-49): with n1 = n2 or n1 = n2 + 1 records
// that is every record of both files, and since the ids are ranks after the sort the order they are staged in is free -- both files then go through the parallel
// reader one after the other.  Any other pair of counts (or a file the parallel reader does not take) is the sequential reader's business.
static bool add_mate_files_parallel(sage2ov_ctx* c, const char* p1, const char* p2) {
    std::vector<StagePart> a, b; if (!parse_plain_file_parallel(c, p1, a) || !parse_plain_file_parallel(c, p2, b)) return false;
    uint64_t n1 = 0, n2 = 0; for (auto& P : a) n1 += P.records; for (auto& P : b) n2 += P.records;
    if (n1 != n2 && n1 != n2 + 1) return false;
    commit_parts(c, a); commit_parts(c, b); return true;
}

// records are split sequentially (cheap: memchr), filtered and packed by all threads in batches
int add_files(sage2ov_ctx* c, const char* p1, const char* p2) {
    if (!c->opt.get("SAGE2OV_SEQUENTIAL_READER") && ((p2 && *p2) ? add_mate_files_parallel(c, p1, p2) : add_plain_file_parallel(c, p1))) return SAGE2OV_OK;
    SeqFile f1, f2; if (!f1.open(p1)) return c->fail(SAGE2OV_ERR_IO, std::string("cannot open ") + p1);
    const bool two = p2 && *p2; if (two && !f2.open(p2)) return c->fail(SAGE2OV_ERR_IO, std::string("cannot open ") + p2);
    const int nt = io_threads(c);
    const size_t BATCH = 1 << 19;
    std::string flat; std::vector<uint64_t> boff; uint64_t inFile = 0; bool more = true;
    std::vector<std::vector<uint64_t>> pools(nt), offs(nt); std::vector<std::vector<uint16_t>> lens(nt);
    while (more) {
        flat.clear(); boff.clear(); boff.push_back(0);
        while (boff.size() <= BATCH) {                             // inputReader.cpp:26-49: alternate by parity
            SeqFile& f = (two && (inFile & 1)) ? f2 : f1;
            size_t len = 0;
            if (!f.next(flat, len)) { more = false; break; }
            boff.push_back(flat.size()); inFile++;
        }
        const int64_t nb = (int64_t)boff.size() - 1; if (nb == 0) break;
        std::vector<uint64_t> good(nt, 0), bp(nt, 0), small(nt, 0);
        for (int t = 0; t < nt; t++) { pools[t].clear(); offs[t].clear(); lens[t].clear(); }
        #pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num(); const int64_t chunk = (nb + nt - 1) / nt, a = t * chunk, b = std::min<int64_t>(nb, a + chunk);
            std::vector<uint8_t> codes;
            for (int64_t r = a; r < b; r++) {
                const size_t L = boff[r + 1] - boff[r]; codes.resize(L);
                const unsigned char* sq = (const unsigned char*)flat.data() + boff[r];
                for (size_t i = 0; i < L; i++) codes[i] = g_code[sq[i]];
                stage_codes(c, codes.data(), (int)L, pools[t], offs[t], lens[t], good[t], bp[t], small[t]);
            }
        }
        for (int t = 0; t < nt; t++) {                              // concatenate in file order (thread t holds a contiguous slice)
            const uint64_t base = c->pool.size();
            c->pool.insert(c->pool.end(), pools[t].begin(), pools[t].end());
            for (uint64_t o : offs[t]) c->poolOff.push_back(base + o);
            c->poolLen.insert(c->poolLen.end(), lens[t].begin(), lens[t].end());
            c->goodReads += good[t]; c->totalBP += bp[t]; c->smallReads += small[t];
        }
        c->totalReads += (uint64_t)nb;
    }
    for (SeqFile* f : {&f1, &f2}) if (f->ioError) {
        int en = 0; const char* m = f->fp ? gzerror(f->fp, &en) : nullptr;
        return c->fail(SAGE2OV_ERR_IO, std::string("read error in ") + (f == &f1 ? p1 : p2) + ": " + (m && *m ? m : "corrupt or truncated input"));
    }
    return SAGE2OV_OK;
}
std::string trim(const std::string& s) { size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n"); return a == std::string::npos ? "" : s.substr(a, b - a + 1); }

// ------------------------------------------------------------------------------------------ reduce replay
// exact restatement of economyGraph.cpp:495-574 over device-computed hit lists
// Serial replay of the reduce phase (economyGraph.cpp:513-707).  All state is in flat arrays over a dense numbering of the reads
// it can touch (unresolved reads and the ends of the candidates near them): no hashing on the hot path.
struct Replay {
    sage2ov_ctx* c;
    std::vector<uint32_t>& dense;                      // read id -> dense index + 1 (0: not part of the replay); lives in the context, all zero between replays
    explicit Replay(std::vector<uint32_t>& d) : dense(d) {}
    ~Replay() { for (uint32_t id : idOf) dense[id] = 0; }
    std::vector<uint32_t> idOf;                        // dense index -> read id
    std::vector<std::vector<AdjEdge>> adj;             // per dense index
    std::vector<uint8_t> st;                           // per dense index: 0 unexplored, 1 explored, 2 marked, 4 not an unresolved read
    std::vector<std::pair<uint64_t, uint64_t>> hitRange;
    const std::vector<Hit>* hits = nullptr;
    uint64_t inserted = 0, removed = 0;
    uint32_t add(uint32_t id) { uint32_t& d = dense[id]; if (!d) { idOf.push_back(id); d = (uint32_t)idOf.size(); } return d - 1; }
    void finish_setup() { adj.resize(idOf.size()); st.assign(idOf.size(), 4); hitRange.assign(idOf.size(), {0, 0}); }
    uint8_t status_id(uint32_t id) const { const uint32_t d = dense[id]; return d ? st[d - 1] : 4; }
    int insert_edge(uint32_t u, uint32_t v, uint32_t delta, int type) {              // economyGraph.cpp:813-849 (u, v: read ids)
        if (u == v) return 0;
        int d2 = (int)c->len[u] - ((int)c->len[v] - (int)delta);
        adj[dense[u] - 1].push_back(AdjEdge{v, (uint8_t)type, 0, delta & 0xFFFFFu});
        adj[dense[v] - 1].push_back(AdjEdge{u, (uint8_t)flip_type_host(type), 0, (uint32_t)d2 & 0xFFFFFu});
        return 1;
    }
    uint64_t explore(uint32_t r1) {                                                  // economyGraph.cpp:580-638
        const uint32_t d1 = dense[r1]; if (!d1 || st[d1 - 1] != 0) return 0;
        st[d1 - 1] = 1; uint64_t ins = 0;
        for (uint64_t x = hitRange[d1 - 1].first; x < hitRange[d1 - 1].second; x++) {
            const Hit& h = (*hits)[x];
            if (status_id(h.to) != 0) continue;                                      // :605 current status
            if (h.len == -1) continue;                                               // :627 sentinel quirk
            ins += insert_edge(r1, h.to, (uint32_t)h.len, h.type);
        }
        auto& a = adj[d1 - 1];
        if (a.size() > 1) std::sort(a.begin(), a.end(), [](const AdjEdge& x, const AdjEdge& y) {   // :853-871
            if (x.len != y.len) return x.len > y.len; if (x.to != y.to) return x.to > y.to; return x.type > y.type; });
        return ins * 2;
    }
    // markTransitiveEdge (economyGraph.cpp:643-679).  When it runs in the serial BFS the lists of `from` and of all its neighbours are
    // complete and nothing has been removed from them yet (every neighbour is explored by then, nobody appends to the list of an
    // explored read, and a read's removal waits until all its neighbours are marked): the marks are a function of the final lists, so
    // the BFS below only keeps the statuses that steer the traversal and the marks are computed afterwards, reads in parallel.
    void mark_edges(uint32_t from, std::vector<uint8_t>& mkb) {
        auto& a = adj[dense[from] - 1];
        for (auto& e : a) mkb[dense[e.to] - 1] = 1;
        for (auto& e : a) {
            if (mkb[dense[e.to] - 1] != 1) continue;
            for (auto& f : adj[dense[e.to] - 1]) {
                const uint32_t df = dense[f.to]; if (!df || mkb[df - 1] != 1) continue;
                int t1 = e.type, t2 = f.type;
                if ((t1 == 0 || t1 == 2) && (t2 == 0 || t2 == 1)) mkb[df - 1] = 2;
                else if ((t1 == 1 || t1 == 3) && (t2 == 2 || t2 == 3)) mkb[df - 1] = 2;
            }
        }
        for (auto& e : a) if (mkb[dense[e.to] - 1] == 2) e.mark = 1;
        for (auto& e : a) mkb[dense[e.to] - 1] = 0;
    }
    void mark(uint32_t from) { st[dense[from] - 1] = 2; }                            // traversal only: status 1 -> 2 (:679)
    uint64_t remove_marked(uint32_t r) {                                             // economyGraph.cpp:681-707
        auto& a = adj[dense[r] - 1]; size_t before = a.size();
        a.erase(std::remove_if(a.begin(), a.end(), [](const AdjEdge& e) { return e.mark != 0; }), a.end());
        return before - a.size();
    }
    bool timing = false;
    void run(const std::vector<uint32_t>& ids) {
        auto tp = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) { if (!timing) return; auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[reduce/host]   %-26s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tp).count()); tp = t; };
        std::vector<uint32_t> queue;
        for (uint32_t i : ids) {                                                     // the serial part: exploration order (:513-564)
            if (status_id(i) != 0) continue;
            queue.clear(); size_t start = 0; queue.push_back(i);
            while (start < queue.size()) {
                uint32_t r1 = queue[start++];
                if (status_id(r1) == 0) inserted += explore(r1);
                const uint32_t d1 = dense[r1] - 1;
                if (adj[d1].empty()) continue;
                if (st[d1] == 1) {
                    for (size_t x = 0; x < adj[d1].size(); x++) { uint32_t r2 = adj[d1][x].to; if (status_id(r2) == 0) { queue.push_back(r2); inserted += explore(r2); } }
                    mark(r1);
                }
                if (st[d1] == 2) {
                    for (size_t x = 0; x < adj[d1].size(); x++) {
                        uint32_t r2 = adj[d1][x].to; if (status_id(r2) != 1) continue;
                        const uint32_t d2 = dense[r2] - 1;
                        for (size_t y = 0; y < adj[d2].size(); y++) { uint32_t r3 = adj[d2][y].to; if (status_id(r3) == 0) { queue.push_back(r3); inserted += explore(r3); } }
                        mark(r2);
                    }
                }
            }
        }
        lap("serial exploration");
        // marks on the final lists (all reads first: they read their neighbours' unreduced lists), then the removals (:681-707)
        const int64_t n = (int64_t)ids.size();
        const int nthr = io_threads(c);
        #pragma omp parallel num_threads(nthr) if (n > 4096)
        {
            std::vector<uint8_t> mkb(idOf.size(), 0);
            #pragma omp for schedule(dynamic, 1024)
            for (int64_t x = 0; x < n; x++) { const uint32_t d = dense[ids[x]] - 1; if (st[d] == 2 && !adj[d].empty()) mark_edges(ids[x], mkb); }
        }
        lap("marks (parallel)");
        uint64_t rem = 0;
        #pragma omp parallel for num_threads(nthr) schedule(dynamic, 1024) reduction(+ : rem) if (n > 4096)
        for (int64_t x = 0; x < n; x++) { const uint32_t d = dense[ids[x]] - 1; if (st[d] == 2) rem += remove_marked(ids[x]); }
        removed += rem;
        lap("removals (parallel)");
    }
};

}  // namespace

static inline char* put_u(char* p, unsigned long long v) {          // decimal, no sign, no padding (what %u / %llu print)
    char t[24]; int n = 0; do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v); while (n) *p++ = t[--n]; return p;
}
// text of items [0, n) produced by `fmt(i, out)` (appends to a std::string): chunks of about 4 MB of text, formatted by whichever thread is free into its own
// (cache-resident, reused) buffer and copied into the page cache at the chunk's offset.  The offset of chunk j + 1 is published as soon as chunk j is
// FORMATTED, so nobody waits for somebody else's write: formatting and the copies into the page cache overlap (in the batch form this replaces --
// all format, barrier, all write -- thread 0 spent 0.26 of 0.46 s working on the 2.8 GB .reads file of 10 M reads).
template <class F>
static int write_formatted(sage2ov_ctx* c, FILE* f, uint64_t n, size_t bytes_per_item, F fmt) {
    const int nt = io_threads(c); HostLap lap(c, "writer");
    if (fflush(f) != 0) return c->fail(SAGE2OV_ERR_IO, "write failed");
    const off_t pos0 = ftello(f); const int fd = fileno(f);
    // (items of very uneven size -- the edges of P.graph4 carry read lists -- still give every thread several chunks)
    const uint64_t per = std::max<uint64_t>(64, std::min<uint64_t>((4u << 20) / std::max<size_t>(bytes_per_item, 1), (n + (uint64_t)nt * 8 - 1) / ((uint64_t)nt * 8))), nchunks = (n + per - 1) / per;
    std::unique_ptr<std::atomic<int64_t>[]> at(new std::atomic<int64_t>[nchunks + 1]);
    for (uint64_t j = 0; j <= nchunks; j++) at[j].store(-1, std::memory_order_relaxed);
    at[0].store((int64_t)pos0);
    std::atomic<uint64_t> next{0}; std::atomic<bool> failed{false};
    #pragma omp parallel num_threads(nt)
    {
        std::string out; out.reserve((size_t)per * bytes_per_item + 4096);
        for (;;) {
            const uint64_t j = next.fetch_add(1); if (j >= nchunks || failed.load()) break;
            out.clear(); const uint64_t a = j * per, e = std::min(n, a + per);
            for (uint64_t i = a; i < e; i++) fmt(i, out);
            int64_t o; while ((o = at[j].load(std::memory_order_acquire)) < 0 && !failed.load()) sched_yield();
            if (o < 0) break;
            at[j + 1].store(o + (int64_t)out.size(), std::memory_order_release);
            const char* p = out.data(); size_t left = out.size();
            while (left) { const ssize_t w = pwrite(fd, p, left, (off_t)o); if (w <= 0) { failed.store(true); break; } p += w; left -= (size_t)w; o += w; }
        }
    }
    lap("format + pwrite (threads)");
    if (failed.load() || fseeko(f, (off_t)at[nchunks].load(), SEEK_SET) != 0) return c->fail(SAGE2OV_ERR_IO, "write failed");
    return SAGE2OV_OK;
}

// ============================================================================================ C ABI
extern "C" {

const char* sage2ov_version(void) { return "sage2ov 0.1 (gfx950)"; }
const char* sage2ov_last_error(const sage2ov_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }
void* sage2ov_stream(sage2ov_ctx* c) { return c && c->device() ? dev_stream(c->device()) : nullptr; }

int sage2ov_ctx_create(const sage2ov_config* cfg, sage2ov_ctx** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return SAGE2OV_ERR_ARG; }
    if (cfg->min_overlap == 0) { g_create_error = "min_overlap (-k) is required"; return SAGE2OV_ERR_ARG; }   // main.cpp:506-510
    if (cfg->rank >= std::max<uint32_t>(1, cfg->world)) { g_create_error = "rank >= world"; return SAGE2OV_ERR_ARG; }
    auto* c = new sage2ov_ctx(); c->cfg = *cfg;
    if (c->cfg.world == 0) c->cfg.world = 1;
    if (cfg->device != SAGE2OV_DEVICE_NONE) {
        const int ordinal = cfg->device; const double share = 1.0 / (double)c->cfg.world;
        auto open = [c, ordinal, share] { c->dev_ = dev_create(ordinal, c->opt, c->devErr); if (c->dev_) dev_set_probe_share(c->dev_, share); };
        if (cfg->flags & SAGE2OV_FLAG_ASYNC_DEVICE) c->devOpen = std::thread(open);      // the caller stages its input meanwhile; device() joins
        else { open(); if (!c->dev_) { g_create_error = c->devErr; delete c; return SAGE2OV_ERR_DEVICE; } }
    }
    *out = c; return SAGE2OV_OK;
}
void sage2ov_ctx_destroy(sage2ov_ctx* c) { if (!c) return; if (Device* d = c->device()) dev_destroy(d); delete c; }
int sage2ov_options_reload(sage2ov_ctx* c) { if (!c) return SAGE2OV_ERR_ARG; c->opt = s2::Options::from_env(); if (Device* d = c->device()) dev_set_options(d, c->opt); return SAGE2OV_OK; }

int sage2ov_reads_add_ascii(sage2ov_ctx* c, const char* bases, const uint64_t* off, uint64_t n) {
    if (!c || !bases || !off) return SAGE2OV_ERR_ARG;
    if (c->organized) return c->fail(SAGE2OV_ERR_ARG, "reads already organised");
    if (c->gpu() && !c->opt.get("SAGE2OV_HOST_PACK") && !c->opt.get("SAGE2OV_HOST_ORGANIZE")) {        // staged as they come: the device filters and packs them
        if (c->asciiOff.empty()) c->asciiOff.push_back(0);
        const uint64_t base = c->ascii.size(), first = off[0], bytes = off[n] - first;
        c->ascii.insert(c->ascii.end(), bases + first, bases + first + bytes);
        for (uint64_t r = 1; r <= n; r++) c->asciiOff.push_back(base + (off[r] - first));
        c->totalReads += n;
        return SAGE2OV_OK;
    }
    std::vector<uint8_t> codes;
    for (uint64_t r = 0; r < n; r++) {
        const int L = (int)(off[r + 1] - off[r]); codes.resize(L);
        for (int i = 0; i < L; i++) codes[i] = g_code[(unsigned char)bases[off[r] + i]];
        stage_codes(c, codes.data(), L, c->pool, c->poolOff, c->poolLen, c->goodReads, c->totalBP, c->smallReads);
        c->totalReads++;
    }
    return SAGE2OV_OK;
}
int sage2ov_reads_add_file(sage2ov_ctx* c, const char* p1, const char* p2) {
    if (!c || !p1) return SAGE2OV_ERR_ARG;
    if (c->organized) return c->fail(SAGE2OV_ERR_ARG, "reads already organised");
    return add_files(c, p1, p2);
}
int sage2ov_reads_add_list(sage2ov_ctx* c, const char* lp) {                         // readLoader.cpp:73-131
    if (!c || !lp) return SAGE2OV_ERR_ARG;
    FILE* f = fopen(lp, "r"); if (!f) return c->fail(SAGE2OV_ERR_IO, std::string("cannot open ") + lp);
    char buf[8192]; std::string val1; unsigned mate = 0; int rc = SAGE2OV_OK;
    while (fgets(buf, sizeof buf, f)) {
        std::string line = buf; while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        size_t p = line.find('=');
        if (p == std::string::npos) { rc = c->fail(SAGE2OV_ERR_IO, "list of input files in a wrong format"); break; }
        std::string var = trim(line.substr(0, p)), val = trim(line.substr(p + 1));
        if (mate % 2 == 0 && var == "f1") val1 = val;
        else if (mate % 2 == 0 && var == "f") { rc = add_files(c, val.c_str(), nullptr); mate++; }
        else if (mate % 2 == 1 && var == "f2") rc = add_files(c, val1.c_str(), val.c_str());
        else { rc = c->fail(SAGE2OV_ERR_IO, "list of input files in a wrong format"); }
        if (rc) break;
        mate++;
    }
    fclose(f); return rc;
}

int sage2ov_reads_add_synth(sage2ov_ctx* c, const sage2ov_synth_params* p, const uint8_t* genome, uint64_t first, uint64_t n) {
    if (!c || !p || !genome) return SAGE2OV_ERR_ARG;
    if (c->organized) return c->fail(SAGE2OV_ERR_ARG, "reads already organised");
    const int nt = io_threads(c);
    std::vector<std::vector<uint64_t>> pools(nt), offs(nt); std::vector<std::vector<uint16_t>> lens(nt);
    std::vector<uint64_t> good(nt, 0), bp(nt, 0), small(nt, 0);
    int rcAll = 0;
    #pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num(); const uint64_t chunk = (n + nt - 1) / nt, a = first + t * chunk, b = std::min(first + n, a + chunk);
        std::vector<char> bases(p->read_len + 1); std::vector<uint8_t> codes(p->read_len + 1); uint64_t o[2];
        for (uint64_t r = a; r < b; r++) {
            int rc = sage2ov_synth_reads_ascii(p, genome, r, 1, bases.data(), o);
            if (rc) { rcAll = rc; break; }
            const int L = (int)o[1];
            for (int i = 0; i < L; i++) codes[i] = g_code[(unsigned char)bases[i]];
            stage_codes(c, codes.data(), L, pools[t], offs[t], lens[t], good[t], bp[t], small[t]);
        }
    }
    if (rcAll) return c->fail(rcAll, "synthetic generator failed");
    for (int t = 0; t < nt; t++) {                                  // concatenate in read order (thread t holds a contiguous slice)
        const uint64_t base = c->pool.size();
        c->pool.insert(c->pool.end(), pools[t].begin(), pools[t].end());
        for (uint64_t o : offs[t]) c->poolOff.push_back(base + o);
        c->poolLen.insert(c->poolLen.end(), lens[t].begin(), lens[t].end());
        c->goodReads += good[t]; c->totalBP += bp[t]; c->smallReads += small[t];
    }
    c->totalReads += n;
    return SAGE2OV_OK;
}

static int upload(sage2ov_ctx* c) {
    if (c->gpu() && !c->device()) return c->fail(SAGE2OV_ERR_DEVICE, c->devErr);
    if (!c->device()) { c->organized = true; return SAGE2OV_OK; }      // step-1-only context (SAGE2OV_DEVICE_NONE)
    int minL = c->N ? 0xFFFF : 0; for (uint64_t i = 1; i <= c->N; i++) minL = std::min<int>(minL, c->len[i]);
    int rc = dev_upload_reads(c->device(), c->words.data(), c->N, c->S, minL, c->maxL, (int)c->cfg.min_overlap, c->err);
    if (rc) return rc;
    c->organized = true; c->indexBuilt = c->probed = c->reciprocalDone = c->reduced = c->converted = false;
    return SAGE2OV_OK;
}
// a slot of S words holds the bases and, in the low 9 bits of its last word, the length: 123 / 251 / 507 bases for S = 4 / 8 / 16
static constexpr uint64_t slot_len_mask(int S) { return S > 16 ? 0x7FFull : 0x1FFull; }     // (kernels_common.inc: the 32-word layout keeps 11 bits)
#define SLOT_LEN_MASK slot_len_mask(c->S)
// words per slot: 2 bits per base + the length field (9 bits up to 16 words = 504 bases, 11 bits in the 32-word layout = 1018 bases)
static int choose_S(int maxL) { int need = (2 * maxL + 9 + 63) / 64; int S = 4; while (S < need) S *= 2; if (S > 16) { need = (2 * maxL + 11 + 63) / 64; S = need <= 32 ? 32 : 64; } return S; }

int sage2ov_reads_organize(sage2ov_ctx* c) {                                          // readLoader.cpp:215-260
    if (!c) return SAGE2OV_ERR_ARG;
    if (c->organized) return c->fail(SAGE2OV_ERR_ARG, "reads already organised");
    if (c->gpu() && !c->device()) return c->fail(SAGE2OV_ERR_DEVICE, c->devErr);      // (a device opened in the background: this is where a failure surfaces)
    const int nthr = io_threads(c);                                                   // (clauses, not omp_set_num_threads: a library call must not change the host's OpenMP state)
    if (!c->asciiOff.empty() && !c->poolLen.empty()) {                                // both kinds of input staged: pack the ASCII here and organise one pool
        std::vector<uint8_t> codes; const uint64_t na = c->asciiOff.size() - 1;
        for (uint64_t r = 0; r < na; r++) { const int L = (int)(c->asciiOff[r + 1] - c->asciiOff[r]); codes.resize(L); for (int i = 0; i < L; i++) codes[i] = g_code[(unsigned char)c->ascii[c->asciiOff[r] + i]];
            stage_codes(c, codes.data(), L, c->pool, c->poolOff, c->poolLen, c->goodReads, c->totalBP, c->smallReads); }
        std::vector<char>().swap(c->ascii); std::vector<uint64_t>().swap(c->asciiOff);
    }
    if (!c->asciiOff.empty()) {                                                       // ASCII only: step 1 entirely on the device
        OrgAscii A{c->ascii.data(), c->ascii.size(), c->asciiOff.data(), c->asciiOff.size() - 1};
        uint64_t N = 0;
        int rc = dev_organize_reads(c->device(), nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, (int)c->cfg.min_overlap, &N, c->words, c->freq, c->err, &A);
        if (rc) return rc;
        c->goodReads += A.good; c->totalBP += A.total_bp; c->smallReads += A.small; c->maxL = A.maxL; c->S = A.S;
        c->N = N; c->len.assign(N + 1, 0);
        #pragma omp parallel for num_threads(nthr)
        for (uint64_t i = 1; i <= N; i++) c->len[i] = (uint16_t)(c->words[i * c->S + c->S - 1] & SLOT_LEN_MASK);
        std::vector<char>().swap(c->ascii); std::vector<uint64_t>().swap(c->asciiOff);
        c->organized = true; c->indexBuilt = c->probed = c->reciprocalDone = c->reduced = c->converted = false;
        return SAGE2OV_OK;
    }
    const uint64_t n = c->poolLen.size(); HostLap lap(c, "step 1");
    int maxL = 0; for (uint64_t i = 0; i < n; i++) maxL = std::max<int>(maxL, c->poolLen[i]);
    c->maxL = maxL; c->S = choose_S(std::max(maxL, 1));
    if (c->S > 32 || maxL > 1018) return c->fail(SAGE2OV_ERR_LIMIT, "reads longer than 1018 bases are not supported");
    if (n >= (1ull << 32)) return c->fail(SAGE2OV_ERR_LIMIT, "too many reads for the host organiser");
    for (uint64_t i = 0; i < n; i++) if (c->poolLen[i] == 0xFFFF) return c->fail(SAGE2OV_ERR_LIMIT, "reads longer than 1018 bases are not supported");
    if (c->device() && !c->opt.get("SAGE2OV_HOST_ORGANIZE")) {                                 // step 1 on the device: canonical orientation, sort, unique, ids
        uint64_t N = 0;
        int minL = n ? 0xFFFF : 0; for (uint64_t i = 0; i < n; i++) minL = std::min<int>(minL, c->poolLen[i]);
        lap("length scans");
        int rc = dev_organize_reads(c->device(), c->pool.data(), c->pool.size(), c->poolOff.data(), c->poolLen.data(), n, c->S, minL, c->maxL, (int)c->cfg.min_overlap,
                                    &N, c->words, c->freq, c->err);
        if (rc) return rc;
        lap("device organiser (incl. transfers)");
        c->N = N; c->len.assign(N + 1, 0);
        #pragma omp parallel for num_threads(nthr)
        for (uint64_t i = 1; i <= N; i++) c->len[i] = (uint16_t)(c->words[i * c->S + c->S - 1] & SLOT_LEN_MASK);
        RawU64().swap(c->pool); RawU64().swap(c->poolOff); RawU16().swap(c->poolLen);
        lap("lengths + release of the staging");
        c->organized = true; c->indexBuilt = c->probed = c->reciprocalDone = c->reduced = c->converted = false;
        return SAGE2OV_OK;
    }
    #pragma omp parallel for num_threads(nthr)
    for (uint64_t i = 0; i < n; i++) canonicalise_words(&c->pool[c->poolOff[i]], c->poolLen[i]);
    std::vector<uint32_t> ord(n);
    #pragma omp parallel for num_threads(nthr)
    for (uint64_t i = 0; i < n; i++) ord[i] = (uint32_t)i;
    const uint64_t* pool = c->pool.data(); const uint64_t* off = c->poolOff.data(); const uint16_t* pl = c->poolLen.data();
    auto cmp = [&](uint32_t a, uint32_t b) -> int {                                   // utils.cpp:224-242 on big-endian words
        const int la = pl[a], lb = pl[b], wa = (la + 31) / 32, wb = (lb + 31) / 32, wm = std::min(wa, wb);
        const uint64_t *pa = pool + off[a], *pb = pool + off[b];
        for (int w = 0; w < wm; w++) { if (pa[w] != pb[w]) return pa[w] < pb[w] ? -1 : 1; }
        // bytes beyond the shorter read's last WORD may still be inside its last byte range: the reference compares
        // ceil(len/4) bytes, all of which lie in the first wm words except when the longer read has more words;
        // those extra bytes only matter while they are within min(bytes): they are not (min bytes <= 8*wm).
        if (la != lb) {
            // the shorter read is zero padded inside its last word, so a difference within min(bytes) was caught above
            return la < lb ? -1 : 1;
        }
        return 0;
    };
    __gnu_parallel::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return cmp(a, b) < 0; }, __gnu_parallel::default_parallel_tag(nthr));   // readLoader.cpp:221
    // unique + frequency (readLoader.cpp:225-235)
    std::vector<uint32_t> firstOf; firstOf.reserve(n); std::vector<uint16_t> fr; fr.reserve(n);
    for (uint64_t x = 0; x < n; x++) {
        if (x == 0 || cmp(ord[x - 1], ord[x]) != 0) { firstOf.push_back(ord[x]); fr.push_back(0); }
        fr.back()++;                                                                  // u16 wrap like the reference
    }
    const uint64_t N = firstOf.size(); const int S = c->S;
    c->N = N; c->words.assign((N + 1) * S, 0); c->len.assign(N + 1, 0); c->freq.assign(N + 1, 0);
    #pragma omp parallel for num_threads(nthr)
    for (uint64_t i = 1; i <= N; i++) {
        const uint32_t a = firstOf[i - 1]; const int L = pl[a], nw = (L + 31) / 32;
        uint64_t* w = &c->words[i * S];
        for (int q = 0; q < nw; q++) w[q] = pool[off[a] + q];
        w[S - 1] |= (uint64_t)L;                                                       // length in the low 9 bits of the last word
        c->len[i] = (uint16_t)L; c->freq[i] = fr[i - 1];
    }
    RawU64().swap(c->pool); RawU64().swap(c->poolOff); RawU16().swap(c->poolLen);
    return upload(c);
}

int sage2ov_reads_stats(const sage2ov_ctx* c, sage2ov_read_stats* o) {
    if (!c || !o) return SAGE2OV_ERR_ARG;
    o->total_reads = c->totalReads; o->good_reads = c->goodReads; o->unique_reads = c->N; o->total_bp = c->totalBP;
    o->average_read_length = c->goodReads ? c->totalBP / c->goodReads : 0; o->max_read_length = (uint32_t)c->maxL; o->words_per_read = (uint32_t)c->S;
    return SAGE2OV_OK;
}
static void unpack_bytes(const uint64_t* w, int L, uint8_t* out, uint64_t stride) {    // word image -> utils.cpp:96 byte image
    const int nb = (L + 3) / 4; for (uint64_t b = 0; b < stride; b++) out[b] = 0;
    for (int b = 0; b < nb && (uint64_t)b < stride; b++) out[b] = (uint8_t)(w[b >> 3] >> (56 - 8 * (b & 7)));
    if (L & 3) out[nb - 1] &= (uint8_t)(0xFF << (8 - 2 * (L & 3)));
}
int sage2ov_reads_export(const sage2ov_ctx* c, uint8_t* packed, uint64_t stride, uint16_t* length, uint16_t* frequency) {
    if (!c || !c->organized) return SAGE2OV_ERR_ARG;
    for (uint64_t i = 0; i <= c->N; i++) {
        if (packed) { if (i == 0) memset(packed, 0, stride); else unpack_bytes(&c->words[i * c->S], c->len[i], packed + i * stride, stride); }
        if (length) length[i] = c->len[i];
        if (frequency) frequency[i] = c->freq[i];
    }
    return SAGE2OV_OK;
}
static void words_to_ascii(const uint64_t* w, int L, char* out) { static const char B[4] = {'A', 'C', 'G', 'T'}; for (int i = 0; i < L; i++) out[i] = B[(w[i >> 5] >> (62 - 2 * (i & 31))) & 3]; }
int sage2ov_reads_save(sage2ov_ctx* c, const char* path) {                            // readLoader.cpp:270-287, :29-36
    if (!c || !path || !c->organized) return SAGE2OV_ERR_ARG;
    FILE* f = fopen(path, "w"); if (!f) return c->fail(SAGE2OV_ERR_IO, std::string("cannot open ") + path);
    std::vector<char> io(1 << 22); setvbuf(f, io.data(), _IOFBF, io.size());
    fprintf(f, "%llu\n", (unsigned long long)c->N);
    const int S = c->S;
    int rc = write_formatted(c, f, c->N, (size_t)(2 * c->maxL + 24), [&](uint64_t x, std::string& o) {
        const uint64_t i = x + 1; const int L = c->len[i]; const uint64_t* w = &c->words[i * S];
        uint64_t tmp[34], r[34]; const int nw = (L + 31) / 32; for (int q = 0; q < nw; q++) tmp[q] = w[q]; if (nw == S) tmp[nw - 1] &= ~SLOT_LEN_MASK;
        revcomp_words(tmp, nw, L, r);
        char hd[48]; char* p = put_u(hd, c->freq[i]); *p++ = '\t'; p = put_u(p, (unsigned)L); *p++ = '\t';
        const size_t at = o.size(); o.resize(at + (size_t)(p - hd) + 2 * (size_t)L + 2);
        char* d = &o[at]; memcpy(d, hd, (size_t)(p - hd)); d += p - hd;
        words_to_ascii(tmp, L, d); d += L; *d++ = '\t'; words_to_ascii(r, L, d); d += L; *d++ = '\n';
    });
    fclose(f); return rc;
}
// P.reads as the writers produce it (readLoader.cpp:29-36: "frequency TAB length TAB read TAB reverse complement NL" per read, N on the first line), mapped and
// parsed by all I/O threads: a first pass over line-aligned chunks counts the lines and finds the longest read, a second packs every read straight into its slot
// (eight bases per step, pack8).  1: loaded.  0: the file is not strictly of that shape (other separators, CRLF, a sequence whose length differs from its length
// field, a wrong line count, ...) -- the fscanf reader below then takes it, with the reference's token semantics.  < 0: error.
static int reads_load_parallel(sage2ov_ctx* c, const char* path) {
    const int fd = ::open(path, O_RDONLY); if (fd < 0) return 0;
    struct stat sb; if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 4) { ::close(fd); return 0; }
    const size_t size = (size_t)sb.st_size;
    const char* m = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0); ::close(fd);
    if (m == MAP_FAILED) return 0;
    struct Unmap { const char* p; size_t n; ~Unmap() { munmap((void*)p, n); } } um{m, size};
    madvise((void*)m, size, MADV_SEQUENTIAL);
    size_t h = 0; uint64_t N = 0; while (h < size && m[h] >= '0' && m[h] <= '9' && h < 19) N = N * 10 + (uint64_t)(m[h++] - '0');
    if (h == 0 || h >= size || m[h] != '\n') return 0;
    h++;
    if (N >= (1ull << 32) || m[size - 1] != '\n') return 0;
    const int nt = io_threads(c); const size_t nchunks = (size_t)nt * 4;
    std::vector<size_t> cut(nchunks + 1); cut[0] = h; cut[nchunks] = size;
    for (size_t x = 1; x < nchunks; x++) { size_t p = h + (size - h) / nchunks * x; if (p < cut[x - 1]) p = cut[x - 1]; const char* nl = (const char*)memchr(m + p, '\n', size - p); cut[x] = nl ? (size_t)(nl - m) + 1 : size; }
    // one line: digits TAB digits TAB <len> characters TAB ... NL; returns the position behind the line, 0 when the line is not of that shape
    auto line = [&](size_t p, size_t e, uint64_t& fr, uint64_t& L, size_t& sq) -> size_t {
        size_t q = p; fr = 0; while (q < e && m[q] >= '0' && m[q] <= '9' && q - p < 10) fr = fr * 10 + (uint64_t)(m[q++] - '0');
        if (q == p || q >= e || m[q] != '\t') return 0;
        const size_t p2 = ++q; L = 0; while (q < e && m[q] >= '0' && m[q] <= '9' && q - p2 < 7) L = L * 10 + (uint64_t)(m[q++] - '0');
        if (q == p2 || q >= e || m[q] != '\t') return 0;
        sq = ++q; if (sq + L >= e || m[sq + L] != '\t') return 0;
        const char* nl = (const char*)memchr(m + sq + L + 1, '\n', e - (sq + L + 1)); if (!nl) return 0;
        return (size_t)(nl - m) + 1;
    };
    std::vector<uint64_t> cnt(nchunks + 1, 0); std::vector<int> mx(nchunks, 0); bool bad = false;
    #pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int64_t x = 0; x < (int64_t)nchunks; x++) {
        size_t p = cut[x]; const size_t e = cut[x + 1]; uint64_t n = 0, fr, L; size_t sq; int ml = 0;
        while (p < e) {
            const size_t nx = line(p, e, fr, L, sq);
            bool ws = false; if (nx) for (size_t i = 0; i < L; i++) ws |= (unsigned char)m[sq + i] <= ' ';
            if (!nx || ws || L > 1018) {
                #pragma omp atomic write
                bad = true;
                break;
            }
            ml = std::max(ml, (int)L); n++; p = nx;
        }
        cnt[x + 1] = n; mx[x] = ml;
    }
    if (bad) return 0;
    int maxL = 0; for (size_t x = 0; x < nchunks; x++) { cnt[x + 1] += cnt[x]; maxL = std::max(maxL, mx[x]); }
    if (cnt[nchunks] != N) return 0;
    c->maxL = maxL; c->S = choose_S(std::max(maxL, 1)); const int S = c->S;
    c->N = N; c->words.resize((N + 1) * S); c->len.resize(N + 1); c->freq.resize(N + 1);
    for (int q = 0; q < S; q++) c->words[q] = 0;
    c->len[0] = 0; c->freq[0] = 0;
    #pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int64_t x = 0; x < (int64_t)nchunks; x++) {
        size_t p = cut[x]; const size_t e = cut[x + 1]; uint64_t i = cnt[x] + 1, fr, L; size_t sq;
        while (p < e) {
            p = line(p, e, fr, L, sq);
            uint64_t* w = &c->words[i * S]; for (int q = 0; q < S; q++) w[q] = 0;
            const char* b = m + sq; int y = 0;
            for (; y + 8 <= (int)L; y += 8) {
                uint64_t v, o; memcpy(&v, b + y, 8);
                if (!pack8(v, o)) { o = 0; for (int z = 0; z < 8; z++) { uint8_t cd = g_code[(unsigned char)b[y + z]]; if (cd > 3) cd = 0; o |= (uint64_t)cd << (14 - 2 * z); } }   // (anything but ACGT reads as A, like the loop below)
                w[y >> 5] |= o << (48 - 2 * (y & 31));
            }
            for (; y < (int)L; y++) { uint8_t cd = g_code[(unsigned char)b[y]]; if (cd > 3) cd = 0; w[y >> 5] |= (uint64_t)cd << (62 - 2 * (y & 31)); }
            w[S - 1] |= L; c->len[i] = (uint16_t)L; c->freq[i] = (uint16_t)(unsigned)fr;
            i++;
        }
    }
    return 1;
}
int sage2ov_reads_load(sage2ov_ctx* c, const char* path) {                            // readLoader.cpp:289-307, :38-48
    if (!c || !path) return SAGE2OV_ERR_ARG;
    if (!c->opt.get("SAGE2OV_SEQUENTIAL_READER")) { const int pr = reads_load_parallel(c, path); if (pr < 0) return pr; if (pr == 1) { c->organized = false; return upload(c); } }
    FILE* f = fopen(path, "r"); if (!f) return c->fail(SAGE2OV_ERR_IO, std::string("cannot open ") + path);
    unsigned long long N = 0; if (fscanf(f, "%llu", &N) != 1) { fclose(f); return c->fail(SAGE2OV_ERR_IO, "bad .reads header"); }
    std::vector<std::string> seqs(N + 1); std::vector<unsigned> fr(N + 1), ln(N + 1); int maxL = 0;
    std::vector<char> a(70000), b(70000);
    for (unsigned long long i = 1; i <= N; i++) {
        if (fscanf(f, "%u %u %69999s %69999s", &fr[i], &ln[i], a.data(), b.data()) != 4) { fclose(f); return c->fail(SAGE2OV_ERR_IO, "bad .reads record"); }
        seqs[i] = a.data(); maxL = std::max<int>(maxL, (int)ln[i]);
    }
    fclose(f);
    c->maxL = maxL; c->S = choose_S(std::max(maxL, 1)); if (c->S > 32) return c->fail(SAGE2OV_ERR_LIMIT, "reads longer than 1018 bases are not supported");
    c->N = N; const int S = c->S; c->words.assign((N + 1) * S, 0); c->len.assign(N + 1, 0); c->freq.assign(N + 1, 0);
    for (unsigned long long i = 1; i <= N; i++) {
        const int L = (int)ln[i]; uint64_t* w = &c->words[i * S];
        for (int p = 0; p < L && p < (int)seqs[i].size(); p++) { uint8_t x = g_code[(unsigned char)seqs[i][p]]; if (x > 3) x = 0; w[p >> 5] |= (uint64_t)x << (62 - 2 * (p & 31)); }
        w[S - 1] |= (uint64_t)L; c->len[i] = (uint16_t)L; c->freq[i] = (uint16_t)fr[i];
    }
    c->organized = false;
    return upload(c);
}
// word-layout image of the organised reads (what lives in HBM): lets one rank organise and the others import
int sage2ov_reads_export_words(const sage2ov_ctx* c, uint64_t* words, uint64_t cap_words, uint16_t* frequency) {
    if (!c || !c->organized || !words) return SAGE2OV_ERR_ARG;
    if (cap_words < c->words.size()) return SAGE2OV_ERR_ARG;
    memcpy(words, c->words.data(), c->words.size() * sizeof(uint64_t));
    if (frequency) memcpy(frequency, c->freq.data(), (c->N + 1) * sizeof(uint16_t));
    return SAGE2OV_OK;
}
int sage2ov_reads_import_words(sage2ov_ctx* c, const uint64_t* words, uint64_t n_unique, uint32_t words_per_read, uint32_t max_read_length,
                               const uint16_t* frequency, uint64_t good_reads, uint64_t total_bp) {
    if (!c || !words) return SAGE2OV_ERR_ARG;
    if (words_per_read != 4 && words_per_read != 8 && words_per_read != 16 && words_per_read != 32) return c->fail(SAGE2OV_ERR_ARG, "words_per_read must be 4, 8, 16 or 32");
    c->N = n_unique; c->S = (int)words_per_read; c->maxL = (int)max_read_length;
    c->words.assign(words, words + (n_unique + 1) * words_per_read);
    c->len.assign(n_unique + 1, 0); c->freq.assign(n_unique + 1, 0);
    for (uint64_t i = 1; i <= n_unique; i++) { c->len[i] = (uint16_t)(c->words[i * c->S + c->S - 1] & SLOT_LEN_MASK); c->freq[i] = frequency ? frequency[i] : 1; }
    c->goodReads = good_reads; c->totalBP = total_bp; c->totalReads = good_reads; c->organized = false;
    return upload(c);
}
int sage2ov_reads_set_totals(sage2ov_ctx* c, uint64_t good, uint64_t bp) { if (!c) return SAGE2OV_ERR_ARG; c->goodReads = good; c->totalBP = bp; return SAGE2OV_OK; }

// ------------------------------------------------------------------------------------------ step 2
int sage2ov_index_build(sage2ov_ctx* c) {
    if (!c) return SAGE2OV_ERR_ARG;
    if (!c->device()) return c->fail(SAGE2OV_ERR_DEVICE, "this context has no GPU (SAGE2OV_DEVICE_NONE): steps 2-3 are device-only");
    if (!c->organized) return c->fail(SAGE2OV_ERR_ARG, "organise (or load) the reads first");
    uint64_t slots, keys, csr, nlong; uint32_t reb;
    int rc = dev_build_index(c->device(), &slots, &keys, &csr, &nlong, &reb, c->err); if (rc) return rc;
    c->istats.slots = slots; c->istats.keys = keys; c->istats.csr_entries = csr; c->istats.long_buckets = nlong;
    c->istats.hash_string_length = c->cfg.min_overlap > 64 ? 64 : c->cfg.min_overlap; c->istats.rebuilds = reb; c->istats.minimiser_groups = dev_has_minimiser_groups(c->device()) ? 1u : 0u; c->istats.reserved = 0;
    c->indexBuilt = true; c->probed = c->reciprocalDone = c->reduced = c->converted = false;
    return SAGE2OV_OK;
}
int sage2ov_index_stats_get(const sage2ov_ctx* c, sage2ov_index_stats* o) { if (!c || !o) return SAGE2OV_ERR_ARG; *o = c->istats; return SAGE2OV_OK; }
int sage2ov_index_lookup(sage2ov_ctx* c, const uint64_t key[2], uint64_t* entries, uint32_t cap, uint32_t* count) {
    if (!c || !key || !count) return SAGE2OV_ERR_ARG;
    if (!c->indexBuilt) return c->fail(SAGE2OV_ERR_ARG, "index not built");
    return dev_lookup(c->device(), key[0], key[1], entries, cap, count, c->err);
}

// P.hashTable (hashTable.cpp:256-273, :12-20): the text dump of the REFERENCE's own table -- one line per slot of its double-hashed table, in
// slot order -- which SAGE2 writes with -s or -M 2 and reads back with -m 3.  Our index has another shape, so the file
// is produced by replaying the reference's serial insertion (hashTable.cpp:94-109, :133-187) on the host: same table size (:243-254), same
// start slot (:233), same probe step (:140), same cap of 101 entries per bucket (:178), same flag for buckets of >= 100 entries (:111-123).
// Only the file needs this; look-ups never do (the placement is invisible in P.graph3).
namespace {
// The reference picks its table size from a literal list of 450 primes (hashTable.cpp:246).  The list follows a rule, which is what is
// implemented here instead of the list: k * 100000 < p prime, smallest, for k = 1..10; then the smallest SAFE prime (p and (p-1)/2 prime)
// above m * 65536 for m = mant * 2^e, mant = 17..31, e = 0, 1, 2, ... starting at m = 27.  (Checked against the reference's numbers when the
// golden fixtures were made: oracle/make_golden.py records the table size the reference printed; tests/test_hashtable_file.py.)
inline uint64_t mulmod64(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)((unsigned __int128)a * b % m); }
inline uint64_t powmod64(uint64_t a, uint64_t e, uint64_t m) { uint64_t r = 1; a %= m; while (e) { if (e & 1) r = mulmod64(r, a, m); a = mulmod64(a, a, m); e >>= 1; } return r; }
bool is_prime64(uint64_t n) {
    if (n < 2) return false;
    for (uint64_t p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) { if (n % p == 0) return n == p; }
    uint64_t d = n - 1; int r = 0; while ((d & 1) == 0) { d >>= 1; r++; }
    for (uint64_t a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {          // deterministic for 64-bit n
        uint64_t x = powmod64(a, d, n); if (x == 1 || x == n - 1) continue;
        bool comp = true; for (int i = 1; i < r; i++) { x = mulmod64(x, x, n); if (x == n - 1) { comp = false; break; } }
        if (comp) return false;
    }
    return true;
}
// n-th table size of the reference (0-based), 0 beyond the list's 450 entries
uint64_t ref_table_size(int n) {
    if (n < 0 || n >= 450) return 0;
    if (n < 10) { uint64_t p = (uint64_t)(n + 1) * 100000; do p++; while (!is_prime64(p)); return p; }
    const int x = n - 10 + 10;                                       // position counted from mantissa 17 of exponent 0: the list starts at mantissa 27
    const int e = x / 15, mant = 17 + x % 15;
    uint64_t p = ((uint64_t)mant << e) * 65536ull;
    for (;;) { p++; if ((p & 3) == 3 && is_prime64(p) && is_prime64((p - 1) / 2)) return p; }
}
void ref_table_sizes(uint64_t value, uint64_t* next, uint64_t* prev) {      // findNextPrime / findPreviousPrime (hashTable.cpp:243-254, :303-314)
    uint64_t last = 0, before = 0;
    for (int n = 0; n < 450; n++) { before = last; last = ref_table_size(n); if (last > value || n == 449) break; }
    *next = last; *prev = before;                                    // (value <= 100003: the reference indexes [-1]; the caller refuses)
}
}  // namespace
int sage2ov_hashtable_save(sage2ov_ctx* c, const char* path) {
    if (!c || !path) return SAGE2OV_ERR_ARG;
    if (!c->organized) return c->fail(SAGE2OV_ERR_ARG, "organise (or load) the reads first");
    const uint64_t N = c->N; const int S = c->S; const int h = c->cfg.min_overlap > 64 ? 64 : (int)c->cfg.min_overlap;
    if (8 * N <= 100003) return c->fail(SAGE2OV_ERR_LIMIT, "P.hashTable: fewer than 12501 unique reads (the reference's own table-size look-up is undefined there, hashTable.cpp:309-313)");
    uint64_t M, Mprev; ref_table_sizes(8 * N, &M, &Mprev);
    if (M <= 8 * N) return c->fail(SAGE2OV_ERR_LIMIT, "P.hashTable: more reads than the reference's largest table holds");
    const uint64_t pre = ((0xFFFFFFFFFFFFFFFFull) % M + 1) % M;                        // hashTable.cpp:85
    // the four keys of a read (hashTable.cpp:96-104) as (v0, v1) of utils.cpp:171-187
    auto take = [](const uint64_t* w, int nw, int a, int n) -> uint64_t { return n <= 0 ? 0ull : (bits64_host(w, nw, 2 * a) >> (64 - 2 * n)); };
    auto key_of = [&](const uint64_t* w, int nw, int a, uint64_t& v0, uint64_t& v1) { if (h <= 32) { v0 = 0; v1 = take(w, nw, a, h); } else { v0 = take(w, nw, a, h - 32); v1 = take(w, nw, a + h - 32, 32); } };
    struct Slot { uint64_t v0, v1; int64_t head, tail; uint32_t count; };
    std::vector<Slot> slot(M + 1, Slot{0, 0, -1, -1, 0});                              // (index M is reachable: `while (p > M)`, hashTable.cpp:163)
    std::vector<uint64_t> entVal; std::vector<int64_t> entNext; entVal.reserve(4 * N); entNext.reserve(4 * N);
    for (uint64_t i = 1; i <= N; i++) {                                                // hashTable.cpp:94-109: serial, ids ascending, types 0..3
        const int L = c->len[i], nw = (L + 31) / 32; uint64_t f[34], r[34];
        for (int q = 0; q < nw; q++) f[q] = c->words[i * S + q];
        if (nw == S) f[nw - 1] &= ~SLOT_LEN_MASK;
        f[nw] = 0; revcomp_words(f, nw, L, r); r[nw] = 0;
        uint64_t kv[4][2];
        key_of(f, nw + 1, 0, kv[0][0], kv[0][1]); key_of(f, nw + 1, L - h, kv[1][0], kv[1][1]);
        key_of(r, nw + 1, 0, kv[2][0], kv[2][1]); key_of(r, nw + 1, L - h, kv[3][0], kv[3][1]);
        for (int t = 0; t < 4; t++) {
            const uint64_t v0 = kv[t][0], v1 = kv[t][1];
            const uint64_t probe = ((v1 % M) + (v0 % M) * pre) % M, inc = 1 + ((v0 + v1) % Mprev);   // :233, :140
            uint64_t pp = probe, miss = 0;
            while (slot[pp].head >= 0 && !(slot[pp].v1 == v1 && slot[pp].v0 == v0)) { miss++; pp = probe + miss * inc; while (pp > M) pp -= M; }
            Slot& sl = slot[pp];
            if (sl.head < 0 || sl.count <= HASH_THRESHOLD) {                           // :168-186: at most 101 entries
                const int64_t e = (int64_t)entVal.size(); entVal.push_back(i * 4 + (uint64_t)t); entNext.push_back(-1);
                if (sl.head < 0) { sl.head = e; sl.v0 = v0; sl.v1 = v1; } else entNext[sl.tail] = e;
                sl.tail = e; sl.count++;
            }
        }
    }
    FILE* fo = fopen(path, "w"); if (!fo) return c->fail(SAGE2OV_ERR_IO, std::string("cannot open ") + path);
    std::vector<char> io(1 << 22); setvbuf(fo, io.data(), _IOFBF, io.size());
    fprintf(fo, "%llu\n", (unsigned long long)M);
    const uint64_t longHash = N + 100;                                                 // :77, :111-123
    int rc = write_formatted(c, fo, M, 24, [&](uint64_t x, std::string& o) {            // hashTable.cpp:12-20 per slot + the "\n" of :268
        const Slot& sl = slot[x];
        if (sl.head < 0) { o.append("0\n"); return; }
        char buf[64]; char* q = put_u(buf, sl.count); *q++ = '\n'; o.append(buf, (size_t)(q - buf));
        bool first = true;
        for (int64_t e = sl.head; e >= 0; e = entNext[e]) {
            const uint64_t id = (first && sl.count >= HASH_THRESHOLD) ? longHash : (entVal[e] >> 2); first = false;
            q = put_u(buf, id); *q++ = '\t'; q = put_u(q, entVal[e] & 3); *q++ = '\t'; o.append(buf, (size_t)(q - buf));
        }
        o.push_back('\n');
    });
    fclose(fo); return rc;
}

// ------------------------------------------------------------------------------------------ step 3
int sage2ov_shard_range(const sage2ov_ctx* c, uint64_t* lo, uint64_t* hi) {
    if (!c || !lo || !hi) return SAGE2OV_ERR_ARG;
    const uint64_t N = c->N, w = c->cfg.world, r = c->cfg.rank;
    *lo = 1 + (N * r) / w; *hi = 1 + (N * (r + 1)) / w; return SAGE2OV_OK;
}
int sage2ov_shard_record_bytes(const sage2ov_ctx* c, uint64_t* b) { if (!c || !b) return SAGE2OV_ERR_ARG; *b = 16; return SAGE2OV_OK; }
int sage2ov_overlap_probe_shard(sage2ov_ctx* c) {
    if (!c) return SAGE2OV_ERR_ARG;
    if (!c->indexBuilt) return c->fail(SAGE2OV_ERR_ARG, "build the index first");
    uint64_t lo, hi; sage2ov_shard_range(c, &lo, &hi);
    int rc = dev_probe(c->device(), lo, hi, c->err); if (rc) return rc;
    c->probed = true; c->reciprocalDone = c->reduced = c->converted = false; return SAGE2OV_OK;
}
int sage2ov_shard_export_records(sage2ov_ctx* c, void* dst, uint64_t max_reads) {
    if (!c || !dst) return SAGE2OV_ERR_ARG; if (!c->probed) return c->fail(SAGE2OV_ERR_ARG, "probe first");
    uint64_t lo, hi; sage2ov_shard_range(c, &lo, &hi); if (hi - lo > max_reads) return c->fail(SAGE2OV_ERR_ARG, "destination too small");
    return dev_export_records(c->device(), dst, lo, hi, c->err);
}
int sage2ov_shard_import_records(sage2ov_ctx* c, const void* src, uint64_t first, uint64_t n) {
    if (!c || !src) return SAGE2OV_ERR_ARG; if (first < 1 || first + n > c->N + 1) return c->fail(SAGE2OV_ERR_ARG, "id range out of bounds");
    return dev_import_records(c->device(), src, first, n, c->err);
}
int sage2ov_overlap_reciprocal(sage2ov_ctx* c) {
    if (!c) return SAGE2OV_ERR_ARG; if (!c->probed) return c->fail(SAGE2OV_ERR_ARG, "probe first");
    uint64_t lo, hi; sage2ov_shard_range(c, &lo, &hi);
    uint64_t nov, cont, csize; int rc = dev_reciprocal(c->device(), lo, hi, &nov, &cont, &csize, c->err); if (rc) return rc;
    c->ostats = sage2ov_overlap_stats{}; c->ostats.verified_overlaps = nov; c->ostats.contained_extension = cont; c->ostats.contained_size = csize;
    c->ostats.left_to_explore = c->N - cont - csize;
    c->reciprocalDone = true; c->reduced = c->converted = false; return SAGE2OV_OK;
}
int sage2ov_shard_flags_bytes(const sage2ov_ctx* c, uint64_t* b) { if (!c || !b) return SAGE2OV_ERR_ARG; *b = 2 * (c->N + 1); return SAGE2OV_OK; }
int sage2ov_shard_export_flags(sage2ov_ctx* c, void* dst) { if (!c || !dst) return SAGE2OV_ERR_ARG; if (!c->probed) return c->fail(SAGE2OV_ERR_ARG, "probe first"); return dev_export_flags(c->device(), dst, c->err); }
int sage2ov_shard_import_flags(sage2ov_ctx* c, const void* src) { if (!c || !src) return SAGE2OV_ERR_ARG; if (!c->probed) return c->fail(SAGE2OV_ERR_ARG, "probe first"); return dev_import_flags(c->device(), src, c->err); }
int sage2ov_shard_edges_count(const sage2ov_ctx* c, uint64_t* n) { if (!c || !n || !c->reciprocalDone) return SAGE2OV_ERR_ARG; *n = dev_cand_count(c->device()); return SAGE2OV_OK; }
int sage2ov_shard_edges_export(sage2ov_ctx* c, void* dst, uint64_t cap) { if (!c || !dst) return SAGE2OV_ERR_ARG; if (!c->reciprocalDone) return c->fail(SAGE2OV_ERR_ARG, "reciprocal pass first"); return dev_export_cands(c->device(), dst, cap, c->err); }
int sage2ov_shard_edges_set(sage2ov_ctx* c, const void* src, uint64_t n) { if (!c || (!src && n)) return SAGE2OV_ERR_ARG; if (!c->reciprocalDone) return c->fail(SAGE2OV_ERR_ARG, "reciprocal pass first"); return dev_set_cands(c->device(), src, n, c->err); }
int sage2ov_overlap_initial(sage2ov_ctx* c) {
    if (!c) return SAGE2OV_ERR_ARG;
    if (c->cfg.world > 1) return c->fail(SAGE2OV_ERR_ARG, "multi-GPU contexts use probe_shard / export / import / reciprocal");
    int rc = sage2ov_overlap_probe_shard(c); if (rc) return rc;
    return sage2ov_overlap_reciprocal(c);
}

int sage2ov_overlap_reduce(sage2ov_ctx* c) {
    if (!c) return SAGE2OV_ERR_ARG; if (!c->reciprocalDone) return c->fail(SAGE2OV_ERR_ARG, "run the initial pass first");
    auto t0 = std::chrono::steady_clock::now();
    // Many unresolved reads and no long bucket: the order-independent form runs on the device; the serial
    // replay below stays the path for long-bucket indexes (A.7) and for a handful of reads.  Both are exact.
    {
        const char* ev = c->opt.get("SAGE2OV_DEVICE_REDUCE_MIN");
        const uint64_t minUn = ev ? strtoull(ev, nullptr, 10) : 4096;
        uint64_t nun0 = 0, nh0 = 0, ins = 0, rem = 0; int done = 0;
        const bool multi = c->cfg.world > 1;
        c->survBase = dev_cand_count(c->device()); c->survCount = 0; c->removedPartial = 0; c->survivorsExchanged = !multi;
        if (!c->opt.get("SAGE2OV_HOST_REDUCE")) {
            int rc0 = dev_reduce_device(c->device(), minUn, &nun0, &nh0, &ins, &rem, &done, c->err, c->cfg.rank, multi ? c->cfg.world : 1); if (rc0) return rc0;
        }
        if (done) {
            c->survCount = dev_cand_count(c->device()) - c->survBase; c->removedPartial = rem;
            c->ostats.unresolved_hits = nh0; c->ostats.edges_inserted = ins; c->ostats.transitive_removed = rem;
            c->reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            c->reduced = true; c->converted = false; return SAGE2OV_OK;
        }
    }
    std::vector<Hit> hits; uint64_t nun = 0;
    const bool timing = c->opt.get("SAGE2OV_TIMING") != nullptr; auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[reduce/host] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tp).count()); tp = t; };
    std::vector<uint32_t> ids;
    int rc = dev_unresolved_hits(c->device(), hits, &nun, c->err, &ids); if (rc) return rc;
    lap("hit lists (device + download)");
    c->ostats.unresolved_hits = hits.size(); c->ostats.edges_inserted = 0; c->ostats.transitive_removed = 0;
    if (nun) {
        std::vector<EdgeCand> near; rc = dev_collect_reduce_edges(c->device(), ids, near, c->err); if (rc) return rc;
        lap("ids + nearby candidates");
        __gnu_parallel::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.from != b.from ? a.from < b.from : a.seq < b.seq; }, __gnu_parallel::default_parallel_tag(io_threads(c)));
        lap("sort hits");
        if (c->replayDense.size() != c->N + 2) c->replayDense.assign(c->N + 2, 0);
        Replay R(c->replayDense); R.c = c; R.hits = &hits;
        for (uint32_t i : ids) R.add(i);
        for (auto& e : near) { R.add(e.from); R.add(e.to); }
        for (auto& h : hits) R.add(h.to);                                              // (status-0 reads: already in, kept for safety)
        R.finish_setup();
        for (uint32_t i : ids) R.st[R.dense[i] - 1] = 0;
        for (uint64_t x = 0; x < hits.size();) { uint64_t y = x; while (y < hits.size() && hits[y].from == hits[x].from) y++; R.hitRange[R.dense[hits[x].from] - 1] = {x, y}; x = y; }
        for (auto& e : near) {                                                        // both directed entries of every stored edge
            R.adj[R.dense[e.from] - 1].push_back(AdjEdge{e.to, (uint8_t)e.type, 0, e.len});
            // e is the entry of list[from]; it was created either directly (u=from) or as the twin of (u=to): either way
            // list[to] holds the involutive twin, length L_from - (L_to - len)  (economyGraph.cpp:821)
            const int d2 = (int)c->len[e.from] - ((int)c->len[e.to] - (int)e.len);
            R.adj[R.dense[e.to] - 1].push_back(AdjEdge{e.from, (uint8_t)flip_type_host(e.type), 0, (uint32_t)d2 & 0xFFFFFu});
        }
        lap("replay set-up");
        R.timing = timing; R.run(ids);
        lap("replay (total)");
        c->ostats.edges_inserted = R.inserted; c->ostats.transitive_removed = R.removed;
        // surviving list entries of unresolved reads with to > from replace what the device dropped
        std::vector<EdgeCand> survivors;
        for (uint32_t i : ids) for (auto& e : R.adj[R.dense[i] - 1]) if (e.to > i) survivors.push_back(EdgeCand{i, e.to, e.len, e.type});
        // (multi-rank contexts: the replay is replicated -- every rank computes the same survivors; rank 0's are the ones that are exchanged)
        if (c->cfg.world > 1 && c->cfg.rank != 0) { survivors.clear(); c->removedPartial = 0; } else c->removedPartial = R.removed;
        rc = dev_append_edges(c->device(), survivors.data(), survivors.size(), c->err); if (rc) return rc;
        c->survCount = survivors.size();
        lap("survivors -> device");
    }
    c->reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->reduced = true; c->converted = false; return SAGE2OV_OK;
}

int sage2ov_debug_table(sage2ov_ctx* c, uint64_t* out5) { if (!c || !out5 || !c->device() || !c->indexBuilt) return SAGE2OV_ERR_ARG; return dev_debug_table(c->device(), out5, c->err); }
int sage2ov_debug_meminfo(sage2ov_ctx* c, uint64_t* out4) { if (!c || !out4 || !c->device()) return SAGE2OV_ERR_ARG; return dev_meminfo(c->device(), out4, c->err); }
int sage2ov_debug_keys(sage2ov_ctx* c, uint64_t* out) { if (!c || !out || !c->device() || !c->organized) return SAGE2OV_ERR_ARG; return dev_debug_keys(c->device(), out, c->err); }
// diagnostic (tests/tools): every read's verified hits as 5 x u32 rows {from, to, type, len, seq}, sorted by (from, seq)
int sage2ov_debug_all_hits(sage2ov_ctx* c, uint32_t* out, uint64_t cap_rows, uint64_t* n_rows) {
    if (!c || !n_rows) return SAGE2OV_ERR_ARG; if (!c->reciprocalDone) return c->fail(SAGE2OV_ERR_ARG, "run the initial pass first");
    std::vector<Hit> hits; int rc = dev_debug_all_hits(c->device(), hits, c->err); if (rc) return rc;
    std::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.from != b.from ? a.from < b.from : a.seq < b.seq; });
    *n_rows = hits.size();
    if (out) { if (cap_rows < hits.size()) return c->fail(SAGE2OV_ERR_ARG, "buffer too small");
        for (size_t x = 0; x < hits.size(); x++) { out[5 * x] = hits[x].from; out[5 * x + 1] = hits[x].to; out[5 * x + 2] = hits[x].type; out[5 * x + 3] = (uint32_t)hits[x].len; out[5 * x + 4] = hits[x].seq; } }
    return SAGE2OV_OK;
}
// ---- sharded reduce phase: survivor buckets (multi-rank contexts)
int sage2ov_shard_survivors_count(const sage2ov_ctx* c, uint64_t* n, uint64_t* removed_partial) {
    if (!c || !n || !c->reduced) return SAGE2OV_ERR_ARG;
    *n = c->survCount; if (removed_partial) *removed_partial = c->removedPartial; return SAGE2OV_OK;
}
int sage2ov_shard_survivors_export(sage2ov_ctx* c, void* dst, uint64_t cap) {
    if (!c || (!dst && c->survCount)) return SAGE2OV_ERR_ARG; if (!c->reduced) return c->fail(SAGE2OV_ERR_ARG, "run the reduce phase first");
    if (c->survCount > cap) return c->fail(SAGE2OV_ERR_ARG, "destination too small");
    return dev_export_cand_range(c->device(), dst, c->survBase, c->survCount, c->err);
}
int sage2ov_shard_survivors_set(sage2ov_ctx* c, const void* src, uint64_t n_total, uint64_t removed_total) {
    if (!c || (!src && n_total)) return SAGE2OV_ERR_ARG; if (!c->reduced) return c->fail(SAGE2OV_ERR_ARG, "run the reduce phase first");
    int rc = dev_replace_cand_tail(c->device(), c->survBase, src, n_total, c->err); if (rc) return rc;
    c->survCount = n_total; c->removedPartial = removed_total; c->ostats.transitive_removed = removed_total; c->survivorsExchanged = true; c->converted = false;
    return SAGE2OV_OK;
}
int sage2ov_overlap_convert(sage2ov_ctx* c) {
    if (!c) return SAGE2OV_ERR_ARG; if (!c->reduced) return c->fail(SAGE2OV_ERR_ARG, "run the reduce phase first");
    if (!c->survivorsExchanged) return c->fail(SAGE2OV_ERR_ARG, "multi-rank context: exchange the survivor buckets of the reduce phase first (sage2ov_shard_survivors_*)");
    uint64_t nf = 0; int rc = dev_convert(c->device(), &nf, c->err); if (rc) return rc;
    c->ostats.edges = nf; c->edgesOnHost = false; c->converted = true; return SAGE2OV_OK;
}
int sage2ov_overlap_stats_get(const sage2ov_ctx* c, sage2ov_overlap_stats* o) { if (!c || !o) return SAGE2OV_ERR_ARG; *o = c->ostats; return SAGE2OV_OK; }
int sage2ov_overlap_export_initial(sage2ov_ctx* c, uint64_t* r, uint64_t* l, uint8_t* st, uint32_t* cn) {
    if (!c) return SAGE2OV_ERR_ARG; if (!c->reciprocalDone) return c->fail(SAGE2OV_ERR_ARG, "run the initial pass first");
    return dev_download_initial(c->device(), r, l, st, cn, c->err);
}
static int fetch_edges(sage2ov_ctx* c) {
    if (!c->converted) return c->fail(SAGE2OV_ERR_ARG, "run convert first");
    if (c->edgesOnHost) return SAGE2OV_OK;
    int rc = dev_download_edges(c->device(), c->edges, c->err); if (rc) return rc;
    c->edgesOnHost = true; return SAGE2OV_OK;
}
int sage2ov_edges_count(const sage2ov_ctx* c, uint64_t* n) { if (!c || !n || !c->converted) return SAGE2OV_ERR_ARG; *n = c->ostats.edges; return SAGE2OV_OK; }
int sage2ov_edges_export(sage2ov_ctx* c, sage2ov_edge* out, uint64_t cap) {
    if (!c || !out) return SAGE2OV_ERR_ARG; int rc = fetch_edges(c); if (rc) return rc;
    if (cap < c->edges.size()) return c->fail(SAGE2OV_ERR_ARG, "edge buffer too small");
    for (size_t x = 0; x < c->edges.size(); x++) { const FinalEdge& e = c->edges[x]; sage2ov_edge o{}; o.from = e.from; o.to = e.to; o.length = e.len; o.length_twin = e.len_twin; o.type = (uint8_t)e.type; out[x] = o; }
    return SAGE2OV_OK;
}
// the counterpart of loadOverlapGraphFromFile for a graph of simple edges (overlapGraph.cpp:371-442): take the canonical edge list instead
// of computing it (`-m 4` on existing files; synthetic graphs in the step-4 tests).  Pairs keep the order given = the order they are pushed.
int sage2ov_edges_import(sage2ov_ctx* c, const sage2ov_edge* e, uint64_t n) {
    if (!c || (!e && n)) return SAGE2OV_ERR_ARG;
    if (!c->device()) return c->fail(SAGE2OV_ERR_DEVICE, "no GPU context");
    if (!c->organized) return c->fail(SAGE2OV_ERR_ARG, "sage2ov_edges_import: the read set comes first (reads_organize / reads_load)");
    c->edges.resize(n);
    for (uint64_t x = 0; x < n; x++) {
        if (e[x].from == 0 || e[x].to == 0 || e[x].from > c->N || e[x].to > c->N || e[x].type > 3) return c->fail(SAGE2OV_ERR_ARG, "sage2ov_edges_import: read id or edge type out of range");
        c->edges[x] = FinalEdge{(uint32_t)e[x].from, (uint32_t)e[x].to, e[x].length, e[x].length_twin, e[x].type};
    }
    int rc = dev_upload_edges(c->device(), c->edges, c->err); if (rc) return rc;
    c->ostats.edges = n; c->edgesOnHost = true; c->converted = true; c->g4Valid = false;
    return SAGE2OV_OK;
}
// loadOverlapGraphFromFile (overlapGraph.cpp:371-442) for the graph step 3 writes: simple edges only (list size 0)
// P.graph3 as the writers produce it (overlapGraph.cpp:338-369, :12-20: three header lines, then per edge the record "from TAB to TAB type TAB 1 TAB length TAB 0 TAB 0",
// an empty line, the twin's record, an empty line), mapped and parsed by all I/O threads: a first pass over line-aligned chunks counts the record lines, which tells
// every chunk whether it starts at an edge or at a twin; the second parses edge + twin pairs (the owner of an edge reads its twin past the chunk's end).
// 1: edges filled.  0: not strictly that shape (the parser below then takes the file, with its own diagnostics).
static int graph_load_parallel(sage2ov_ctx* c, const char* path, std::vector<sage2ov_edge>& edges, unsigned long long& nr, unsigned long long& avg) {
    const int fd = ::open(path, O_RDONLY); if (fd < 0) return 0;
    struct stat sb; if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 8) { ::close(fd); return 0; }
    const size_t size = (size_t)sb.st_size;
    const char* m = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0); ::close(fd);
    if (m == MAP_FAILED) return 0;
    struct Unmap { const char* p; size_t n; ~Unmap() { munmap((void*)p, n); } } um{m, size};
    if (m[size - 1] != '\n') return 0;
    auto num = [&](size_t& p, size_t e, unsigned long long& v) -> bool { const size_t p0 = p; v = 0; while (p < e && m[p] >= '0' && m[p] <= '9' && p - p0 < 19) v = v * 10 + (unsigned long long)(m[p++] - '0'); return p > p0; };
    size_t h = 0; unsigned long long hd[3];
    for (int x = 0; x < 3; x++) { if (!num(h, size, hd[x]) || h >= size || m[h] != '\n') return 0; h++; }
    nr = hd[1]; avg = hd[2];
    // a record line + its empty line; v: from, to, type, 1, length, flow, list size.  Returns the position behind the empty line, 0 when the lines are not of that shape
    auto record = [&](size_t p, size_t e, unsigned long long* v) -> size_t {
        for (int x = 0; x < 7; x++) { if (!num(p, e, v[x]) || p >= e || m[p] != (x < 6 ? '\t' : '\n')) return 0; p++; }
        if (p >= e || m[p] != '\n') return 0;
        return p + 1;
    };
    const int nt = io_threads(c); const size_t nchunks = (size_t)nt * 4;
    std::vector<size_t> cut(nchunks + 1); cut[0] = h; cut[nchunks] = size;
    for (size_t x = 1; x < nchunks; x++) {            // cut behind an empty line (i.e. at the start of a record line)
        size_t p = std::max(cut[x - 1], h + (size - h) / nchunks * x);
        for (;;) { const char* nl = p < size ? (const char*)memchr(m + p, '\n', size - p) : nullptr; if (!nl || (size_t)(nl - m) + 1 >= size) { p = size; break; } p = (size_t)(nl - m) + 1; if (m[p] == '\n') { p++; break; } }
        cut[x] = std::max(p, cut[x - 1]);
    }
    std::vector<uint64_t> cnt(nchunks + 1, 0); bool bad = false;
    #pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int64_t x = 0; x < (int64_t)nchunks; x++) {
        size_t p = cut[x]; const size_t e = cut[x + 1]; uint64_t n = 0; unsigned long long v[7];
        while (p < e) { p = record(p, e, v); if (!p || v[6] != 0) { bad = true; break; } n++; }      // (benign race: only ever set)
        cnt[x + 1] = n;
    }
    if (bad) return 0;
    for (size_t x = 0; x < nchunks; x++) cnt[x + 1] += cnt[x];
    if (cnt[nchunks] & 1) return 0;
    edges.resize(cnt[nchunks] / 2);
    #pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int64_t x = 0; x < (int64_t)nchunks; x++) {
        size_t p = cut[x]; const size_t e = cut[x + 1]; uint64_t r = cnt[x]; unsigned long long a[7], b[7];
        if (p < e && (r & 1)) { p = record(p, size, a); r++; }                                       // a twin: its edge belongs to the chunk before
        while (p < e) {
            p = record(p, size, a); const size_t q = record(p, size, b);
            if (!q || a[0] != b[1] || a[1] != b[0]) { bad = true; break; }
            sage2ov_edge o{}; o.from = a[0]; o.to = a[1]; o.type = (uint8_t)a[2]; o.length = (uint32_t)a[4]; o.length_twin = (uint32_t)b[4];
            edges[r / 2] = o; r += 2; p = q;
        }
    }
    return bad ? 0 : 1;
}
int sage2ov_graph_load(sage2ov_ctx* c, const char* path) {
    if (!c || !path) return SAGE2OV_ERR_ARG;
    if (!c->opt.get("SAGE2OV_SEQUENTIAL_READER")) {
        std::vector<sage2ov_edge> pe; unsigned long long pnr = 0, pavg = 0;
        if (graph_load_parallel(c, path, pe, pnr, pavg) == 1) { c->goodReads = pnr; c->totalBP = pavg * pnr; return sage2ov_edges_import(c, pe.data(), pe.size()); }
    }
    FILE* f = fopen(path, "rb"); if (!f) return c->fail(SAGE2OV_ERR_IO, std::string("cannot open ") + path);
    std::string txt; { char buf[1 << 16]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) txt.append(buf, n); } fclose(f);
    const char* p = txt.c_str(); const char* end = p + txt.size();
    auto num = [&](unsigned long long& v) -> bool {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
        if (p >= end || *p < '0' || *p > '9') return false;
        v = 0; while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (unsigned long long)(*p++ - '0');
        if (p < end && *p == '.') { p++; while (p < end && *p >= '0' && *p <= '9') p++; }          // the flow column is a float ("0")
        return true;
    };
    unsigned long long gs, nr, avg;
    if (!num(gs) || !num(nr) || !num(avg)) return c->fail(SAGE2OV_ERR_IO, "bad .graph3 header");
    std::vector<sage2ov_edge> e;
    for (;;) {
        unsigned long long a[7], b[7];
        if (!num(a[0])) break;
        for (int x = 1; x < 7; x++) if (!num(a[x])) return c->fail(SAGE2OV_ERR_IO, "bad .graph3 record");
        for (int x = 0; x < 7; x++) if (!num(b[x])) return c->fail(SAGE2OV_ERR_IO, "bad .graph3 record (twin)");
        if (a[6] != 0 || b[6] != 0) return c->fail(SAGE2OV_ERR_LIMIT, "sage2ov_graph_load: composite edges (a graph already simplified) are not supported");
        if (a[0] != b[1] || a[1] != b[0]) return c->fail(SAGE2OV_ERR_IO, "bad .graph3: a record is not followed by its twin");
        sage2ov_edge o{}; o.from = a[0]; o.to = a[1]; o.type = (uint8_t)a[2]; o.length = (uint32_t)a[4]; o.length_twin = (uint32_t)b[4];
        e.push_back(o);
    }
    c->goodReads = nr; c->totalBP = avg * nr;
    return sage2ov_edges_import(c, e.data(), e.size());
}
int sage2ov_graph_save(sage2ov_ctx* c, const char* path) {                            // overlapGraph.cpp:338-369, :12-20
    if (!c || !path) return SAGE2OV_ERR_ARG; int rc = fetch_edges(c); if (rc) return rc;
    FILE* f = fopen(path, "w"); if (!f) return c->fail(SAGE2OV_ERR_IO, std::string("cannot open ") + path);
    std::vector<char> io(1 << 22); setvbuf(f, io.data(), _IOFBF, io.size());
    fprintf(f, "0\n%llu\n%llu\n", (unsigned long long)c->goodReads, (unsigned long long)(c->goodReads ? c->totalBP / c->goodReads : 0));
    auto rec = [](char* p, unsigned a, unsigned b, unsigned t, unsigned len) {           // "%u\t%u\t%u\t1\t%u\t0\t0\n\n"
        p = put_u(p, a); *p++ = '\t'; p = put_u(p, b); *p++ = '\t'; p = put_u(p, t); memcpy(p, "\t1\t", 3); p += 3; p = put_u(p, len); memcpy(p, "\t0\t0\n\n", 6); return p + 6;
    };
    rc = write_formatted(c, f, c->edges.size(), 96, [&](uint64_t x, std::string& o) {
        const FinalEdge& e = c->edges[x]; char buf[128];
        char* p = rec(buf, e.from, e.to, e.type, e.len);
        p = rec(p, e.to, e.from, (unsigned)flip_type_host((int)e.type), e.len_twin);
        o.append(buf, (size_t)(p - buf));
    });
    fclose(f); return rc;
}

// ---- step 4
int sage2ov_graph_simplify(sage2ov_ctx* c) {                                          // main.cpp:139-172
    if (!c) return SAGE2OV_ERR_ARG;
    if (!c->device()) return c->fail(SAGE2OV_ERR_DEVICE, "no GPU context: step 4 runs on the device only");
    if (!c->converted) return c->fail(SAGE2OV_ERR_ARG, "sage2ov_graph_simplify: call sage2ov_overlap_convert first");
    c->g4Valid = false; c->g4 = SimplifiedGraph();
    int rc = dev_simplify(c->device(), c->g4, c->err); if (rc) return rc;
    c->g4Valid = true; return SAGE2OV_OK;
}
int sage2ov_simplify_stats_get(const sage2ov_ctx* c, sage2ov_simplify_stats* o) {
    if (!c || !o || !c->g4Valid) return SAGE2OV_ERR_ARG;
    const SimplifiedGraph& g = c->g4; const uint64_t e = g.pairs_alive, r = g.reads_on_edges;
    o->nodes_contracted = g.contracted; o->removed = g.removed; o->loop_iterations = g.iterations; o->edges = e; o->reads_on_edges = r; o->device_ms = g.device_ms;
    return SAGE2OV_OK;
}
int sage2ov_graph4_save(sage2ov_ctx* c, const char* path) {                           // overlapGraph.cpp:338-369, :12-20
    if (!c || !path) return SAGE2OV_ERR_ARG;
    if (!c->g4Valid) return c->fail(SAGE2OV_ERR_ARG, "sage2ov_graph4_save: call sage2ov_graph_simplify first");
    HostLap lap(c, "graph4");
    { int rc = dev_simplify_download(c->device(), c->g4, c->err); if (rc) return rc; }
    lap("download of the simplified graph");
    const SimplifiedGraph& g = c->g4; const uint64_t N = g.N, nh = g.n_half_edges;
    // a node's list, oldest first = its alive half-edges by ascending index (the writer walks each list from its tail, :356-358)
    std::vector<uint32_t> offs(N + 2, 0), order;
    for (uint64_t h = 0; h < nh; h++) if (g.alive[h] && g.from[h] <= g.to[h]) offs[g.from[h] + 1]++;
    for (uint64_t i = 0; i <= N; i++) offs[i + 1] += offs[i];
    order.resize(offs[N + 1]);
    { std::vector<uint32_t> cur(offs.begin(), offs.end() - 1); for (uint64_t h = 0; h < nh; h++) if (g.alive[h] && g.from[h] <= g.to[h]) order[cur[g.from[h]]++] = (uint32_t)h; }
    FILE* f = fopen(path, "w"); if (!f) return c->fail(SAGE2OV_ERR_IO, std::string("cannot open ") + path);
    std::vector<char> io(1 << 22); setvbuf(f, io.data(), _IOFBF, io.size());
    fprintf(f, "0\n%llu\n%llu\n", (unsigned long long)c->goodReads, (unsigned long long)(c->goodReads ? c->totalBP / c->goodReads : 0));
    // items: a header line, a block of up to 4096 list entries, or the blank line that ends a record -- a contracted chromosome is ONE
    // edge with millions of entries, so the unit of parallel formatting cannot be the edge
    struct Item { uint32_t h, first, count; };                                          // count == 0: header; first == ~0u: blank line
    std::vector<Item> items; items.reserve(order.size() * 4);
    for (uint32_t h0 : order)
        for (uint32_t h : {h0, h0 ^ 1u}) {
            items.push_back(Item{h, 0, 0});
            for (uint32_t x = 0; x < g.cnt[h]; x += 4096) items.push_back(Item{h, x, std::min<uint32_t>(4096, g.cnt[h] - x)});
            items.push_back(Item{h, ~0u, 1});
        }
    lap("edge order + items");
    int rc = write_formatted(c, f, items.size(), 64, [&](uint64_t x, std::string& o) {
        const Item it = items[x]; const uint32_t h = it.h; char buf[96]; char* p;
        if (it.first == ~0u) { o.push_back('\n'); return; }
        if (it.count == 0) {
            p = put_u(buf, g.from[h]); *p++ = '\t'; p = put_u(p, g.to[h]); *p++ = '\t'; p = put_u(p, g.type[h]); memcpy(p, "\t1\t", 3); p += 3;
            p = put_u(p, g.len[h]); memcpy(p, "\t0\t", 3); p += 3; p = put_u(p, g.cnt[h]); *p++ = '\n'; o.append(buf, (size_t)(p - buf));
            return;
        }
        for (uint32_t y = it.first; y < it.first + it.count; y++) {
            const uint64_t e = g.lists[(uint64_t)g.off[h] + y];
            p = put_u(buf, (unsigned long long)(e & ((1ull << 40) - 1))); *p++ = '\t'; *p++ = (char)('0' + ((e >> 40) & 1)); *p++ = '\t'; *p++ = (char)('0' + ((e >> 41) & 1)); *p++ = '\t';
            p = put_u(p, (unsigned)((e >> 42) & 0x7FF)); *p++ = '\t'; p = put_u(p, (unsigned)((e >> 53) & 0x7FF)); *p++ = '\n'; o.append(buf, (size_t)(p - buf));
        }
    });
    fclose(f); return rc;
}

int sage2ov_run_steps23(sage2ov_ctx* c) {
    if (!c) return SAGE2OV_ERR_ARG;
    auto t0 = std::chrono::steady_clock::now();
    if (c->device()) dev_reset_timings(c->device());
    int rc = sage2ov_index_build(c); if (rc) return rc;
    rc = sage2ov_overlap_initial(c); if (rc) return rc;
    rc = sage2ov_overlap_reduce(c); if (rc) return rc;
    rc = sage2ov_overlap_convert(c); if (rc) return rc;
    c->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SAGE2OV_OK;
}
int sage2ov_timings_reset(sage2ov_ctx* c) { if (!c) return SAGE2OV_ERR_ARG; if (c->device()) dev_reset_timings(c->device()); c->reduce_ms = 0; c->total_ms = 0; return SAGE2OV_OK; }
int sage2ov_timings_get(const sage2ov_ctx* c, sage2ov_timings* o) {
    if (!c || !o) return SAGE2OV_ERR_ARG;
    DevTimings t; if (c->device()) dev_timings(c->device(), &t);
    o->index_ms = t.index_ms; o->probe_ms = t.probe_ms; o->reciprocal_ms = t.reciprocal_ms; o->reduce_ms = c->reduce_ms; o->convert_ms = t.convert_ms;
    o->total_ms = c->total_ms; o->probe_kernel_ms = t.probe_kernel_ms; o->probe_kernel_launches = t.probe_launches; o->sequential_reads = t.slow_reads; o->organize_ms = t.organize_ms; o->probe_fast_launches = t.probe_fast_launches;
    o->reciprocal_cond_ms = t.recip_cond_ms; o->reduce_marks_ms = t.marks_ms;
    return SAGE2OV_OK;
}

}  // extern "C"
