// sage2_amd/csrc/sage2ov_internal.h -- interface between the host side (sage2ov_host.cpp, g++)
// and the device side (sage2ov_device.hip, hipcc).  Plain C++ structs, no HIP types leak out.
#pragma once
#include <cstdint>
#include <sys/mman.h>
#include <algorithm>
#include <cstdlib>
#include <new>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace s2 {

// ----------------------------------------------------------------------------------------------
// HBM layout of the read store
//   reads: (N+1) slots of S u64 words (S = 4, 8, 16 or 32: a power of two so a read never straddles a
//   64-byte sector boundary more than its size requires).  Base p of a read lives in word p/32 at
//   bits 63-2(p%32)..62-2(p%32) (A0 C1 G2 T3): the byte-swapped image of the reference's MSB-first
//   byte packing (utils.cpp:96), so unsigned word compare == the reference's byte compare.
//   The read length sits in the low 9 bits of the slot's last word (SLOT_LEN_MASK 0x1FF; bases never reach
//   there: S is chosen with 2*maxL + 9 <= 64*S; the 32-word layout, 505..1018 bases, keeps 11 bits: slot_len_mask(S)).
//   Slot 0 is all zero (ids are 1-based).
// Index slot (8 bytes):  tag:24 | cnt:7 | payload:33
//   0 = empty; cnt 1 -> payload is the single entry id*4+type; 2..99 -> payload = offset into csr[];
//   127 -> long bucket (>= 100 entries, hashTable.cpp:111-123): never matches a lookup.
// ----------------------------------------------------------------------------------------------
constexpr uint64_t SLOT_TAG_SHIFT = 40;
constexpr uint64_t SLOT_CNT_SHIFT = 33;
constexpr uint64_t SLOT_PAY_MASK = (1ull << 33) - 1;
constexpr uint32_t SLOT_CNT_LONG = 127;
constexpr uint32_t HASH_THRESHOLD = 100;   // hashTable.cpp:76
constexpr uint32_t CONN_LIMIT = 300;       // economyGraph.cpp:43

struct EdgeCand { uint32_t from, to; uint32_t len; uint32_t type; };   // from < to, len already 20-bit masked
// a verified directional hit of an unresolved read (economyGraph.cpp:591-633).  16 bytes, one aligned 16-byte access (20 with a byte-sized type until round 4: the hit
// lists of noisy data are 450 M entries that the probe kernel writes and two kernels of the reduce phase stream); seq: the hit's number among its read's hits
struct alignas(16) Hit { uint32_t from; uint32_t to; int32_t len; uint32_t seq : 30; uint32_t type : 2; };
static_assert(sizeof(Hit) == 16, "Hit: 16 bytes");
struct FinalEdge { uint32_t from, to, len, len_twin; uint32_t type; };

struct DevTimings { double index_ms = 0, probe_ms = 0, reciprocal_ms = 0, hits_ms = 0, convert_ms = 0, probe_kernel_ms = 0, organize_ms = 0, recip_cond_ms = 0, marks_ms = 0; uint64_t probe_launches = 0, slow_reads = 0, probe_fast_launches = 0; };

// ----------------------------------------------------------------------------------------------
// Every switch the library takes from the environment (SAGE2OV_*; INTEGRATION.md lists them with their class: T = test-only override of a decision the library
// otherwise makes by itself -- every route is exact, the tests force each --, G = grid / tuning sweep, D = diagnostic), read ONCE when a context is created:
// Options::from_env snapshots the SAGE2OV_* variables that are set, the context and its device carry the snapshot, and nothing on the timed path calls getenv
// (round 3 had ~60 getenv calls, several per step).  sage2ov_options_reload(ctx) takes a new snapshot (tests that change a switch between two steps of one context).
struct Options {
    std::vector<std::pair<std::string, std::string>> kv;                 // the variables that are set, sorted by name (a handful at most)
    static Options from_env();
    const char* get(const char* name) const {                            // like getenv: the value, or nullptr
        for (const auto& e : kv) if (e.first == name) return e.second.c_str();
        return nullptr;
    }
    bool flag(const char* name) const { return get(name) != nullptr; }
    long long num(const char* name, long long dflt) const { const char* v = get(name); return v ? strtoll(v, nullptr, 10) : dflt; }
};
// the exploration order of the ranked reduce (sage2ov_walk.cpp, host code): rank[position] for the unresolved reads whose potential lists are `lists` / `lenp`
void explore_order(const Options& O, const std::vector<uint32_t>& pos, const std::vector<const uint32_t*>& lists, const std::vector<uint32_t>& lenp, const std::vector<uint8_t>& hasCand,
                   unsigned long long N, const std::vector<uint32_t>& startOrder, std::vector<uint32_t>& rank);


struct Device;   // opaque, lives in sage2ov_device.hip

// every function returns 0 on success, else a negative SAGE2OV_ERR_* and fills err
Device* dev_create(int device_ordinal, const Options& opt, std::string& err);
void dev_set_options(Device* d, const Options& opt);
void dev_destroy(Device* d);
void* dev_stream(Device* d);

int dev_upload_reads(Device* d, const uint64_t* words, uint64_t N, int S, int minL, int maxL, int k, std::string& err);
int dev_build_index(Device* d, uint64_t* slots, uint64_t* keys, uint64_t* csr, uint64_t* nlong, uint32_t* rebuilds, std::string& err);
int dev_lookup(Device* d, uint64_t hi, uint64_t lo, uint64_t* entries, uint32_t cap, uint32_t* count, std::string& err);
// probe+verify+extension kernel over ids [lo, hi)
int dev_probe(Device* d, uint64_t lo, uint64_t hi, std::string& err);
// multi-GPU record exchange: 24 bytes per read {right u64, left u64, conn u32, cflag u32}
int dev_export_records(Device* d, void* dev_dst, uint64_t lo, uint64_t hi, std::string& err);
int dev_import_records(Device* d, const void* dev_src, uint64_t first, uint64_t n, std::string& err);
// reciprocal pass (economyGraph.cpp:455-480) over all reads -> status[], edge candidates on device
// cond() for every read; edge candidates are emitted only for reads in [emit_lo, emit_hi)
int dev_reciprocal(Device* d, uint64_t emit_lo, uint64_t emit_hi, uint64_t* n_ov, uint64_t* contained, uint64_t* contained_size, std::string& err);
int dev_export_flags(Device* d, void* dev_dst, std::string& err);      // 2*(N+1) bytes
int dev_import_flags(Device* d, const void* dev_src, std::string& err);
uint64_t dev_cand_count(Device* d);
int dev_export_cands(Device* d, void* dev_dst, uint64_t cap, std::string& err);   // 16 bytes per candidate
int dev_set_cands(Device* d, const void* dev_src, uint64_t n, std::string& err);
bool dev_has_minimiser_groups(Device* d);
void dev_set_probe_share(Device* d, double share);    // share of the reads this context probes (1 / world): decides whether the minimiser groups pay
int dev_download_initial(Device* d, uint64_t* right, uint64_t* left, uint8_t* status, uint32_t* conn, std::string& err);
// directional hit lists of status-0 reads (economyGraph.cpp:591-633), sorted by (from, seq)
int dev_unresolved_hits(Device* d, std::vector<Hit>& hits, uint64_t* n_unresolved, std::string& err, std::vector<uint32_t>* ids_out = nullptr);   // ids_out: the unresolved ids, ascending
int dev_download_status(Device* d, std::vector<uint8_t>& status, std::string& err);
int dev_unresolved_ids(Device* d, std::vector<uint32_t>& ids, std::string& err);          // ascending
int dev_collect_reduce_edges(Device* d, const std::vector<uint32_t>& unresolved, std::vector<EdgeCand>& out, std::string& err);
// host arrays of hundreds of MB that are overwritten right after they are sized: resize() must not write zeros through them on one thread
// -- and from 32 MB on they ask for 2 MB pages (transparent huge pages in `madvise` mode): 512 times fewer page faults on first touch and on release
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default; template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < (32u << 20)) { void* p = malloc(bytes ? bytes : 1); if (!p) throw std::bad_alloc(); return (T*)p; }
        void* p = nullptr; if (posix_memalign(&p, 2u << 20, (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1)) != 0) throw std::bad_alloc();
        madvise(p, bytes, MADV_HUGEPAGE);
        return (T*)p;
    }
    void deallocate(T* p, size_t) { free(p); }
    template <class U, class... A> void construct(U* p, A&&... a) { if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...); }
};
using RawU64 = std::vector<uint64_t, NoInitAlloc<uint64_t>>;
using RawU16 = std::vector<uint16_t, NoInitAlloc<uint16_t>>;
// first touch of a freshly sized array by `nt` threads (a copy into it on one thread would take its page faults one by one); the content is about to be overwritten
inline void touch_pages(void* p, size_t bytes, int nt) {
    if (bytes < (64u << 20) || nt < 2) return;
    std::vector<std::thread> th; const size_t per = (bytes / nt + 4095) & ~(size_t)4095;
    for (int t = 0; t < nt; t++) th.emplace_back([=] { char* b = (char*)p; const size_t a = per * t, e = std::min(bytes, a + per); for (size_t x = a; x < e; x += 4096) b[x] = 0; });
    for (auto& x : th) x.join();
}
// ASCII input of step 1 (sage2ov_reads_add_ascii): raw bases + offsets in, filter / pack / canonical orientation on the device; the counters come back
struct OrgAscii { const char* bases; uint64_t nbytes; const uint64_t* off; uint64_t n_in; uint64_t good = 0, total_bp = 0, small = 0; int maxL = 0, minL = 0, S = 0; };
int dev_organize_reads(Device* d, const uint64_t* pool, uint64_t pool_words, const uint64_t* off, const uint16_t* len, uint64_t n, int S, int minL, int maxL, int k,
                       uint64_t* N_out, RawU64& words_out, RawU16& freq_out, std::string& err, OrgAscii* ascii = nullptr);
int dev_reduce_device(Device* d, uint64_t min_unresolved, uint64_t* n_unresolved, uint64_t* n_hits, uint64_t* inserted, uint64_t* removed, int* done, std::string& err,
                      uint32_t shareRank = 0, uint32_t shareWorld = 1);
int dev_export_cand_range(Device* d, void* dev_dst, uint64_t first, uint64_t n, std::string& err);
int dev_replace_cand_tail(Device* d, uint64_t keep, const void* dev_src, uint64_t n, std::string& err);
// append host-computed edge candidates (from the reduce replay) to the device candidate list
int dev_debug_table(Device* d, uint64_t* out5, std::string& err);
int dev_meminfo(Device* d, uint64_t* out4, std::string& err);
int dev_debug_keys(Device* d, uint64_t* out, std::string& err);
int dev_debug_all_hits(Device* d, std::vector<Hit>& hits, std::string& err);
int dev_append_edges(Device* d, const EdgeCand* e, uint64_t n, std::string& err);
// sortEconomyGraph + convertGraph: canonical list
int dev_convert(Device* d, uint64_t* n_final, std::string& err);     // result stays in HBM
int dev_download_edges(Device* d, std::vector<FinalEdge>& out, std::string& err);
int dev_upload_edges(Device* d, const std::vector<FinalEdge>& in, std::string& err);
// step 4 (graph simplification) on the device: the surviving half-edges (pair p = 2p, 2p+1; index = age) and their read lists
struct SimplifiedGraph {
    uint64_t N = 0, n_half_edges = 0, contracted = 0, removed = 0, iterations = 0, pairs_alive = 0, reads_on_edges = 0; double device_ms = 0; bool downloaded = false;
    std::vector<uint32_t> from, to, len, cnt, off; std::vector<uint8_t> type, alive; std::vector<uint64_t> lists;
};
int dev_simplify(Device* d, SimplifiedGraph& out, std::string& err);             // result stays in HBM
int dev_simplify_download(Device* d, SimplifiedGraph& out, std::string& err);    // fills the arrays (once)
void dev_simplify_release(Device* d);
void dev_timings(Device* d, DevTimings* t);
void dev_reset_timings(Device* d);
int dev_sync(Device* d, std::string& err);

}  // namespace s2
