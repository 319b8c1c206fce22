// sage2_amd/csrc/sage2ov_walk.cpp -- the serial part of the reduce phase when some bucket is long (economyGraph.cpp:513-564): the ORDER in which the
// unresolved reads are explored, walked on the host over the potential lists the device built (sage2ov_device.hip: dev_reduce_device, the ranked form).
// Host code only (until round 4 it sat inside the device translation unit): g++, AVX2 gathers behind __x86_64__ + a run-time check, a scalar form everywhere else.
// plist[offp[w] .. offp[w+1]) = potential list of the w-th unresolved read, sorted like the reference sorts a list when the read is explored (:853-871); an entry is
// `to | twin << 31`.  An own hit is in the read's list iff the target was still unexplored when the read was explored, a twin iff its source had been explored before;
// candidates of the reciprocal pass (hasCand) are always there but their far ends are never explorable.  Returns rank[id] (1-based exploration order; 0: not an
// unresolved read).  Both inner loops look for RARE entries (a still unexplored target; an explored but unmarked neighbour) among ~100 per list, so they are written as
// "find the next entry that satisfies the test": eight entries per step with AVX2 gathers of rank[] where the host has them.  After every event the search restarts
// behind it with fresh values, so a batch never acts on state that an event of the same batch has changed.  Tables of the walk live on 2 MB pages where the kernel hands
// them out (transparent huge pages, madvise mode): the walk's accesses are spread over ~2 GB (lists) + 170 MB (tables), far beyond what a TLB of 4 KB pages covers.
#include "sage2ov_internal.h"
#include <sys/mman.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace s2 {
typedef unsigned int u32; typedef unsigned long long u64;
struct HostHuge {                                        // anonymous memory on 2 MB pages where the kernel hands them out
    void* p = nullptr; size_t bytes = 0;
    void* get(size_t n) {
        bytes = (std::max<size_t>(n, 1) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) { p = nullptr; return nullptr; }
        madvise(p, bytes, MADV_HUGEPAGE);
        return p;
    }
    ~HostHuge() { if (p) munmap(p, bytes); }
};
constexpr u32 XO_MARK = 0x80000000u, XO_RK = 0x7FFFFFFFu, XO_IDM = 0x3FFFFFFFu;
#if defined(__x86_64__)
static inline void xo_relax() { __builtin_ia32_pause(); }
static inline u64 xo_ticks() { return __builtin_ia32_rdtsc(); }
#else
static inline void xo_relax() { std::this_thread::yield(); }
static inline u64 xo_ticks() { return (u64)std::chrono::steady_clock::now().time_since_epoch().count(); }
#endif
struct XoScalar {
    static inline u32 next_unexplored(const u32* plist, u32 x, u32 n, const u32* rank) {      // kind != 2 and rank[to] == 0
        for (; x < n; x++) { const u32 e = plist[x]; if ((e >> 30) != 2u && rank[e & XO_IDM] == 0) return x; }
        return n;
    }
    static inline u32 next_unmarked(const u32* plist, u32 x, u32 n, const u32* rank) {        // rank[to] != 0 and not marked
        for (; x < n; x++) { const u32 v = rank[plist[x] & XO_IDM]; if (v != 0 && !(v & XO_MARK)) return x; }
        return n;
    }
};
#if defined(__x86_64__)
#include <immintrin.h>
struct XoAvx2 {
    __attribute__((target("avx2"))) static inline u32 next_unexplored(const u32* plist, u32 x, u32 n, const u32* rank) {
        const __m256i idm = _mm256_set1_epi32((int)XO_IDM), two = _mm256_set1_epi32(2), zero = _mm256_setzero_si256();
        for (; x + 8 <= n; x += 8) {
            const __m256i e = _mm256_loadu_si256((const __m256i*)(plist + x));
            const __m256i v = _mm256_i32gather_epi32((const int*)rank, _mm256_and_si256(e, idm), 4);
            const __m256i hit = _mm256_andnot_si256(_mm256_cmpeq_epi32(_mm256_srli_epi32(e, 30), two), _mm256_cmpeq_epi32(v, zero));
            const int m = _mm256_movemask_ps(_mm256_castsi256_ps(hit));
            if (m) return x + (u32)__builtin_ctz((unsigned)m);
        }
        return XoScalar::next_unexplored(plist, x, n, rank);
    }
    __attribute__((target("avx2"))) static inline u32 next_unmarked(const u32* plist, u32 x, u32 n, const u32* rank) {
        const __m256i idm = _mm256_set1_epi32((int)XO_IDM), zero = _mm256_setzero_si256();
        for (; x + 8 <= n; x += 8) {
            const __m256i e = _mm256_loadu_si256((const __m256i*)(plist + x));
            const __m256i v = _mm256_i32gather_epi32((const int*)rank, _mm256_and_si256(e, idm), 4);
            // explored (v != 0) and not marked (sign bit clear): v > 0 as a signed number
            const int m = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(v, zero)));
            if (m) return x + (u32)__builtin_ctz((unsigned)m);
        }
        return XoScalar::next_unmarked(plist, x, n, rank);
    }
};
#endif
// Everything here is indexed by a read's POSITION IN THE LOCALITY ORDER (1-based; the order the probe kernel uses: reads bucketed by their
// global minimiser), not by its id: ids are ranks in lexicographic order, i.e. random with respect to the genome, and the walk touches
// rank[] once per list entry -- with positions the ~100 neighbours of a read sit in a handful of cache lines.  `startOrder` lists the
// positions of the unresolved reads in ASCENDING ID order (the order in which the serial loop starts its searches, :513).
template <class F>
static void explore_order_impl(const Options& O, const std::vector<u32>& pos, const std::vector<const u32*>& lists, const std::vector<u32>& lenp, const std::vector<uint8_t>& hasCand,
                               u64 N, const std::vector<u32>& startOrder, std::vector<u32>& rankv) {
    const size_t n = pos.size();
    // per read: where its list is, how long, whether candidates of the reciprocal pass hang on it -- one 16-byte record, one cache line per visit
    struct PL { const u32* p; u32 n; u32 cand; };
    HostHuge plBuf, rankBuf;
    PL* const pl = (PL*)plBuf.get((N + 2) * sizeof(PL)); u32* const rank = (u32*)rankBuf.get((N + 2) * sizeof(u32));      // (fresh anonymous pages: zero)
    if (!pl || !rank) { rankv.clear(); return; }
    for (size_t w = 0; w < n; w++) pl[pos[w]] = PL{lists[w], lenp[w], (u32)hasCand[w] | 2u};     // bit 1: an unresolved read (a start of the outer loop, :513)
    // rank by position: 0 = unexplored (status 0), else the 1-based exploration order, bit 31 = marked (status 2, :679) -- one table, one look-up
    u32 ctr = 0;
    // entry = to | kind << 30: 0 both sides see each other, 1 own hit only, 2 twin only.
    // an own-only hit is in the list iff its target was explored later (or not yet), a twin-only one iff its source was explored earlier
    auto present = [&](u32 rw, u32 e) -> bool { const u32 k = e >> 30; if (k == 0) return true; const u32 rt = rank[e & XO_IDM] & XO_RK; return k == 1 ? (rt == 0 || rt > rw) : (rt != 0 && rt < rw); };
    // The queue is one array for the whole walk (every read enters it once), its fill level and the pop position are published for the
    // run-ahead helper thread below.
    struct Queue { u32* d; size_t n = 0; std::atomic<size_t> pub{0};
                   void push_back(u32 v) { d[n++] = v; pub.store(n, std::memory_order_release); } size_t size() const { return n; } u32 operator[](size_t i) const { return d[i]; } } queue;
    HostHuge qBuf; queue.d = (u32*)qBuf.get((N + 2) * sizeof(u32)); if (!queue.d) { rankv.clear(); return; }
    std::atomic<size_t> popPos{0}; std::atomic<bool> walkDone{false};
    // Run-ahead helper: the critical path of a pop is the first touch of the list of the neighbour it marks (explored long ago, its list
    // long evicted) and of the rank[] lines around it -- a pointer chase along the genome, one DRAM latency per pop.  A second thread, on
    // a neighbouring core of this thread's core complex (shared L3) where it can be placed, replays the second loop READ-ONLY for the reads a few pops ahead of the
    // walk (racy reads of rank[]: only hints) and touches the lists it would scan, so that they are in the shared caches when the walk
    // arrives.  It changes nothing the walk reads; SAGE2OV_WALK_HELPER=0 turns it off.
    std::thread helper; cpu_set_t savedMask; CPU_ZERO(&savedMask); bool pinnedMain = false;
    { const char* ev = O.get("SAGE2OV_WALK_HELPER"); const bool want = ev ? atoi(ev) != 0 : std::thread::hardware_concurrency() > 1;
      if (want) {
        int sib = -1; const int me = sched_getcpu();
        if (me >= 0) { char path[128]; snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", me);
            if (FILE* f = fopen(path, "r")) { int a = -1, b = -1; char sep = 0; if (fscanf(f, "%d%c%d", &a, &sep, &b) >= 3) sib = a == me ? b : a; fclose(f); }
            // measured (EPYC 9575F, 10 M reads): helper on the next core of the same 8-core complex (shared L3) 1.9 s, on the SMT sibling 2.3 s
            // (it shares the walk's issue slots), no helper 3.3 s -- so the neighbour core is tried first
            if (!O.get("SAGE2OV_WALK_SMT_SIBLING")) { const int nb = (me & ~7) | ((me + 1) & 7); cpu_set_t al; CPU_ZERO(&al); if (sched_getaffinity(0, sizeof al, &al) == 0 && CPU_ISSET(nb, &al)) sib = nb; } }
        cpu_set_t allowed; CPU_ZERO(&allowed); if (sib >= 0 && (sched_getaffinity(0, sizeof allowed, &allowed) != 0 || !CPU_ISSET(sib, &allowed))) sib = -1;
        // Pinning (this thread to its current core for the duration of the walk, the helper to a neighbour) is what the 1.9 s were measured with, but a library
        // call should not fight over cores with other ranks of the same job: off by default when a launcher started several local ranks
        // (LOCAL_WORLD_SIZE > 1), SAGE2OV_WALK_PIN=0/1 decides otherwise.  Without it the helper still runs, wherever the scheduler puts it.
        { const char* pe = O.get("SAGE2OV_WALK_PIN"); const char* lw = O.get("LOCAL_WORLD_SIZE"); const bool pin = pe ? atoi(pe) != 0 : !(lw && atoi(lw) > 1); if (!pin) sib = -1; }
        if (sib >= 0 && pthread_getaffinity_np(pthread_self(), sizeof savedMask, &savedMask) == 0) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(me, &one); pinnedMain = pthread_setaffinity_np(pthread_self(), sizeof one, &one) == 0; }   // (restored when the walk ends)
        const int AHEAD = O.get("SAGE2OV_WALK_AHEAD") ? atoi(O.get("SAGE2OV_WALK_AHEAD")) : 4, WINDOW = 24;
        helper = std::thread([&, sib]() {
            if (sib >= 0) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(sib, &one); pthread_setaffinity_np(pthread_self(), sizeof one, &one); }
            size_t done = 0; u32 sink = 0;
            while (!walkDone.load(std::memory_order_acquire)) {
                const size_t s0 = popPos.load(std::memory_order_relaxed), e0 = queue.pub.load(std::memory_order_acquire);
                size_t a = std::max(done, s0 + (size_t)AHEAD), b = std::min(e0, s0 + (size_t)AHEAD + WINDOW);
                if (a >= b) { xo_relax(); continue; }
                for (size_t i = a; i < b; i++) {
                    const PL nx = pl[queue.d[i]]; const u32* q = nx.p; if (!q) continue;
                    for (u32 x = 0; x < nx.n; x++) {
                        const u32 to = q[x] & XO_IDM; const u32 v = __atomic_load_n(&rank[to], __ATOMIC_RELAXED);
                        if (v != 0 && !(v & XO_MARK)) { const PL t = pl[to]; if (t.p) for (u32 o = 0; o < t.n; o += 16) sink += __atomic_load_n(t.p + o, __ATOMIC_RELAXED); }
                    }
                }
                done = b;
            }
            if (sink == 0x9E3779B9u) fprintf(stderr, " ");                        // (keeps the loads alive)
        });
      } }
    struct HelperJoin { std::thread& t; std::atomic<bool>& d; cpu_set_t& m; bool& pinned;
                        ~HelperJoin() { d.store(true, std::memory_order_release); if (t.joinable()) t.join(); if (pinned) pthread_setaffinity_np(pthread_self(), sizeof m, &m); } } helperJoin{helper, walkDone, savedMask, pinnedMain};
    u64 tcA = 0, tcB = 0, tcC = 0, tcD = 0;
    u64 stPops = 0, stMarks = 0, stScanA = 0, stScanB = 0, stAnyFalse = 0, stEvB = 0, stStarts = 0; const bool stats = O.get("SAGE2OV_TIMING") != nullptr;
    auto explore_neighbours = [&](u32 r) {                                       // every still unexplored neighbour this read sees, in list order (:531-541)
        const u32* plist = pl[r].p; const u32 en = pl[r].n; stMarks++; stScanA += en;
        const u64 t0_ = stats ? xo_ticks() : 0;
        for (u32 x = F::next_unexplored(plist, 0, en, rank); x < en; x = F::next_unexplored(plist, x + 1, en, rank)) {
            const u32 to = plist[x] & XO_IDM; __atomic_store_n(&rank[to], ++ctr, __ATOMIC_RELAXED); queue.push_back(to);
            const int PFE = O.get("SAGE2OV_WALK_PFE") ? atoi(O.get("SAGE2OV_WALK_PFE")) : 1;
            if (PFE) { const PL& t = pl[to]; if (t.p) { __builtin_prefetch(t.p); __builtin_prefetch(t.p + 16); __builtin_prefetch(t.p + 32); __builtin_prefetch(t.p + 48); } }   // it is marked (its list scanned) within a few pops
        }
        if (stats) tcA += xo_ticks() - t0_;
    };
    for (u32 p0 : startOrder) {                                                  // ascending ids, as the serial loop starts its searches
        if (!(pl[p0].cand & 2u) || rank[p0] != 0) continue;
        size_t start = queue.size(); queue.push_back(p0); stStarts++;              // (the queue is never cleared: a search starts where the last one ended)
        while (start < queue.size()) {
            const u32 r1 = queue[start++]; stPops++; popPos.store(start, std::memory_order_relaxed);
            if (start + 8 < queue.size()) { const PL& nx = pl[queue[start + 8]]; if (nx.p) for (u32 o = 0; o < nx.n; o += 16) __builtin_prefetch(nx.p + o); }   // the popped read's own list, 8 pops ahead
            const u64 t1_ = stats ? xo_ticks() : 0;
            if (rank[r1] == 0) __atomic_store_n(&rank[r1], ++ctr, __ATOMIC_RELAXED);
            const u32 rw = rank[r1] & XO_RK, e1 = pl[r1].n; const u32* plist = pl[r1].p;
            bool any = (pl[r1].cand & 1u) != 0;
            for (u32 x = 0; !any && x < e1; x++) any = present(rw, plist[x]);
            if (stats) tcB += xo_ticks() - t1_;
            if (!any) { stAnyFalse++; continue; }                                // an empty list (:527)
            if (!(rank[r1] & XO_MARK)) { explore_neighbours(r1); __atomic_store_n(&rank[r1], rank[r1] | XO_MARK, __ATOMIC_RELAXED); }
            stScanB += e1;
            const u64 t2_ = stats ? xo_ticks() : 0; const u64 a0_ = tcA;
            // (:543-561) neighbours that are explored but not yet marked
            for (u32 x = F::next_unmarked(plist, 0, e1, rank); x < e1; x = F::next_unmarked(plist, x + 1, e1, rank)) {
                const u32 e = plist[x], r2 = e & XO_IDM, k = e >> 30, rt = rank[r2] & XO_RK;
                if (k == 1 ? !(rt > rw) : (k == 2 ? !(rt < rw) : false)) continue;  // not in this read's list
                explore_neighbours(r2); __atomic_store_n(&rank[r2], rank[r2] | XO_MARK, __ATOMIC_RELAXED); stEvB++;
            }
            if (stats) tcC += (xo_ticks() - t2_) - (tcA - a0_);
        }
    }
    if (stats) fprintf(stderr, "[walk] Mcycles: explore_neighbours %llu, pop head + any %llu, second loop (without the explores) %llu\n", (unsigned long long)(tcA >> 20), (unsigned long long)(tcB >> 20), (unsigned long long)(tcC >> 20));
    (void)tcD;
    if (stats) fprintf(stderr, "[walk] starts %llu pops %llu (empty %llu) marks %llu (by a neighbour %llu) entries scanned: explore %llu, second loop %llu\n", (unsigned long long)stStarts, (unsigned long long)stPops,
                       (unsigned long long)stAnyFalse, (unsigned long long)stMarks, (unsigned long long)stEvB, (unsigned long long)stScanA, (unsigned long long)stScanB);
    for (u32 q : pos) rank[q] &= XO_RK;
    rankv.assign(rank, rank + N + 2);
}
// (Round 3, measured and dropped: the same walk on bit maps -- lists kept as {64-position word of the order, member mask} segments, 5.6 per list, "explored" / "marked" one
// bit per position, a scan = `mask & ~explored[word]` resp. `mask & explored[word] & ~marked[word]` per segment, set bits resolved against the list only when order or
// kind matter.  Exact (0 of 8.7 M ranks differ on the 10 M-read repeat data set) and 1.28 -> 0.9-1.0 s for the walk itself, but the segments take 115 ms to build on 16
// host threads and the reduce phase as a whole came out level (1.80-1.84 s against 1.81-1.99 s): the walk is a chain of ~1 event per pop with ~100 cycles of scattered work
// each, not a scan-bound loop.  DESIGN 5.5.)
void explore_order(const Options& O, const std::vector<u32>& pos, const std::vector<const u32*>& lists, const std::vector<u32>& lenp, const std::vector<uint8_t>& hasCand,
                          u64 N, const std::vector<u32>& startOrder, std::vector<u32>& rank) {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2") && !O.get("SAGE2OV_WALK_SCALAR")) { explore_order_impl<XoAvx2>(O, pos, lists, lenp, hasCand, N, startOrder, rank); return; }
#endif
    explore_order_impl<XoScalar>(O, pos, lists, lenp, hasCand, N, startOrder, rank);
}
}  // namespace s2
