// ============================================================================
// oracle/sage2_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the SAGE2 read-overlap path (CLI steps 1-3).  It exists
// to CHECK the HIP implementation; nothing in the product path (sage2_amd/)
// links, imports or executes it.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use it.
//
// Parity status: PINNED.  tests/test_oracle_golden.py compares the files this
// restatement writes (.reads, .graph3) byte-for-byte with fixtures produced by
// the reference binary itself (oracle/_ref/SAGE2, built from /root/reference by
// oracle/Makefile; fixtures made by oracle/make_golden.py).
//
// Every function cites the reference file:line it restates (paths relative to
// the reference checkout).  The code is a restatement with flat arrays, not a
// copy: no per-k-mer malloc, no linked buckets, no globals.
// ============================================================================
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <parallel/algorithm>
#include <omp.h>

namespace {

// ---------------------------------------------------------------- bit utils
// utils.cpp:96-119 charsToBytes: 2 bits/base, MSB first, A0 C1 G2 T3, pad 0.
static void pack_bases(const char* s, int len, uint8_t* out, int nbytes) {
    memset(out, 0, nbytes);
    for (int i = 0; i < len; i++) {
        uint8_t x = 0;
        switch (s[i]) { case 'C': case 'c': x = 1; break; case 'G': case 'g': x = 2; break;
                        case 'T': case 't': x = 3; break; default: x = 0; }
        out[i >> 2] |= (uint8_t)(x << (6 - 2 * (i & 3)));
    }
}
// utils.cpp:124-137 bytesToChars
static void unpack_bases(const uint8_t* b, int len, char* out) {
    static const char L[4] = {'A', 'C', 'G', 'T'};
    for (int i = 0; i < len; i++) out[i] = L[(b[i >> 2] >> (6 - 2 * (i & 3))) & 3];
}
// utils.cpp:73-91 reverseComplement (on ASCII)
static void revcomp_ascii(const char* s, int len, char* out) {
    for (int i = 0; i < len; i++) {
        char c = s[len - 1 - i], r = 'N';
        if (c == 'A') r = 'T'; else if (c == 'C') r = 'G'; else if (c == 'G') r = 'C'; else if (c == 'T') r = 'A';
        out[i] = r;
    }
}
// utils.cpp:189-207 get64BitInt: `length` (<=32) bases from `start`, right aligned.
static inline uint64_t get64(const uint8_t* read, int start, int length) {
    uint64_t number = 0;
    int f1 = (start & 3) << 1, f2 = ((start + length) & 3) << 1;
    if ((start >> 2) == ((start + length) >> 2))
        return (uint64_t)((read[start >> 2] & (0xFF >> f1)) >> (8 - f2));
    int byte;
    for (byte = start >> 2; byte < ((start + length) >> 2); byte++) {
        if (byte == (start >> 2)) number = read[byte] & (0xFF >> f1);
        else number = (number << 8) | read[byte];
    }
    // utils.cpp:205 reads read[byte] even when f2==0 (one past the end at the read's
    // tail); every packed read here carries one pad byte so that is defined.
    number = (number << f2) | (uint64_t)(read[byte] >> (8 - f2));
    return number;
}
struct Key { uint64_t v0, v1; };
// utils.cpp:171-187 get64Bit2Int: v1 = last <=32 bases, v0 = leading length-32 bases.
static inline Key get128(const uint8_t* read, int start, int length) {
    Key k{0, 0};
    if (length <= 32) k.v1 = get64(read, start, length);
    else { k.v0 = get64(read, start, length - 32); k.v1 = get64(read, start + length - 32, 32); }
    return k;
}
// utils.cpp:224-242 stringCompareInBytes
static inline int cmp_packed(const uint8_t* a, int la, const uint8_t* b, int lb) {
    int na = (la + 3) / 4, nb = (lb + 3) / 4;
    for (int i = 0; i < na && i < nb; i++) { if (a[i] < b[i]) return -1; if (a[i] > b[i]) return 1; }
    if (la < lb) return -1; if (la > lb) return 1; return 0;
}
// utils.cpp:212-219 reverseEdgeType
static inline int flip_type(int t) { return t == 0 ? 3 : (t == 3 ? 0 : t); }

// hashTable.cpp:243-254 / :303-314 pick the table size from a fixed list of 450 primes
// (first listed prime > 8N, and the one before it for the probe step).  The list only
// decides WHERE a key lands, never which entries share a bucket or their order, so it does
// not influence .reads/.graph3.  This restatement does not carry the list: it uses the
// smallest prime > 8N and the largest prime < 8N computed directly.
static bool is_prime(uint64_t n) {
    if (n < 2) return false; if (n % 2 == 0) return n == 2;
    for (uint64_t d = 3; d * d <= n; d += 2) if (n % d == 0) return false;
    return true;
}

struct EdgeE { uint64_t to; uint8_t type; uint8_t mark; uint32_t len; };  // economyGraph.h:14-22
struct Ext { uint64_t id; uint8_t type; uint32_t len; };                  // economyGraph.h:24-30

struct Oracle {
    int k = 0;                       // minOverlap
    int threads = 0;
    // ---- step 1 state (readLoader.h:44-55)
    uint64_t totalReads = 0, numberOfReads = 0, totalBP = 0, smallReads = 0;
    uint64_t N = 0;                  // numberOfUniqueReads
    int stride = 0;                  // bytes per packed read slot (max bytes + 1 pad)
    std::vector<std::string> raw;    // canonical ASCII before organise
    std::vector<uint8_t> fwd, rc;    // (N+1)*stride, id 1-based
    std::vector<uint16_t> len, freq;
    // ---- step 2 state (hashTable.h:20-33)
    int h = 0; uint64_t M = 0, Mprev = 0, pre = 0, longHash = 0, hashMissBuild = 0, nLong = 0;
    std::vector<int64_t> slotHead, slotTail; std::vector<uint32_t> slotCount; std::vector<uint8_t> slotLong;
    std::vector<uint64_t> entVal; std::vector<int64_t> entNext;
    // ---- step 3 state
    std::vector<Ext> rightE, leftE; std::vector<uint8_t> status; std::vector<uint32_t> conn;
    std::vector<std::vector<EdgeE>> adj;
    uint64_t nOv = 0, contained = 0, containedSize = 0, edgesInserted = 0, transRemoved = 0, hashMissSearch = 0;
    // final canonical edge list (overlapGraph.cpp:84-111)
    struct OutEdge { uint64_t from, to; uint32_t len, lenTwin; uint8_t type; };
    std::vector<OutEdge> out;
    double tIndex = 0, tInitial = 0, tReduce = 0, tConvert = 0;

    const uint8_t* F(uint64_t id) const { return &fwd[id * stride]; }
    const uint8_t* R(uint64_t id) const { return &rc[id * stride]; }

    // ------------------------------------------------------------ step 1
    // readLoader.cpp:145-158 + utils.cpp:144-166 (filter) + readLoader.cpp:179-213 (canonical)
    void add_ascii(const char* s, int L) {
        totalReads++;
        if (L <= k) { smallReads++; return; }
        std::string r(s, L);
        for (int i = 0; i < L; i++) {
            char c = r[i];
            if (c == 'A' || c == 'C' || c == 'G' || c == 'T') continue;
            if (c == 'a') r[i] = 'A'; else if (c == 'c') r[i] = 'C'; else if (c == 'g') r[i] = 'G';
            else if (c == 't') r[i] = 'T'; else return;       // bad read
        }
        std::string q(L, 'N'); revcomp_ascii(r.data(), L, &q[0]);
        numberOfReads++; totalBP += L;
        raw.push_back(r.compare(q) < 0 ? r : q);              // readLoader.cpp:195
    }
    // readLoader.cpp:215-260 organizeReads
    void organize() {
        int maxL = 0; for (auto& s : raw) maxL = std::max<int>(maxL, s.size());
        stride = (maxL + 3) / 4 + 1;
        size_t n = raw.size();
        std::vector<uint8_t> tmp(n * stride); std::vector<uint16_t> tl(n);
        #pragma omp parallel for
        for (size_t i = 0; i < n; i++) { tl[i] = raw[i].size(); pack_bases(raw[i].data(), tl[i], &tmp[i * stride], stride); }
        std::vector<uint32_t> ord(n); for (size_t i = 0; i < n; i++) ord[i] = i;
        __gnu_parallel::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {   // readLoader.cpp:221
            return cmp_packed(&tmp[(size_t)a * stride], tl[a], &tmp[(size_t)b * stride], tl[b]) < 0; });
        fwd.assign(stride, 0); len.assign(1, 0); freq.assign(1, 0);
        N = 0;
        for (size_t x = 0; x < n; x++) {                                             // readLoader.cpp:225-235
            uint32_t a = ord[x];
            if (N == 0 || cmp_packed(&fwd[N * stride], len[N], &tmp[(size_t)a * stride], tl[a]) != 0) {
                N++; fwd.insert(fwd.end(), &tmp[(size_t)a * stride], &tmp[(size_t)a * stride] + stride);
                len.push_back(tl[a]); freq.push_back(0);
            }
            freq[N]++;                                                               // u16 wrap as in the reference
        }
        rc.assign((N + 1) * stride, 0);
        #pragma omp parallel for
        for (uint64_t i = 1; i <= N; i++) {                                          // readLoader.cpp:249-255
            std::string a(len[i], 'N'), b(len[i], 'N');
            unpack_bases(F(i), len[i], &a[0]); revcomp_ascii(a.data(), len[i], &b[0]);
            pack_bases(b.data(), len[i], &rc[i * stride], stride);
        }
        raw.clear(); raw.shrink_to_fit();
    }
    uint64_t avg_len() const { return numberOfReads ? totalBP / numberOfReads : 0; }  // readLoader.cpp:161

    // ------------------------------------------------------------ step 2
    uint64_t hash_value(const Key& v) const {                                        // hashTable.cpp:233-237
        return ((v.v1 % M) + (v.v0 % M) * pre) % M;
    }
    Key key_of(uint64_t val) const {                                                 // hashTable.cpp:147-154
        uint64_t id = val >> 2; int t = val & 3;
        if (t == 0) return get128(F(id), 0, h);
        if (t == 1) return get128(F(id), len[id] - h, h);
        if (t == 2) return get128(R(id), 0, h);
        return get128(R(id), len[id] - h, h);
    }
    uint64_t insert(const Key& v, uint64_t id, int type) {                           // hashTable.cpp:133-188
        uint64_t miss = 0, probe = hash_value(v), p = probe;
        uint64_t inc = 1 + ((v.v0 + v.v1) % Mprev);
        while (slotHead[p] >= 0) {
            Key r = key_of(entVal[slotHead[p]]);
            if (r.v1 == v.v1 && r.v0 == v.v0) break;
            miss++; p = probe + miss * inc;
            while (p > M) p -= M;                                                    // :163 (index M is reachable)
        }
        if (slotHead[p] < 0 || slotCount[p] <= 100) {                                // :168-186 (cap 101)
            int64_t e = entVal.size(); entVal.push_back(id * 4 + type); entNext.push_back(-1);
            if (slotHead[p] < 0) slotHead[p] = e; else entNext[slotTail[p]] = e;
            slotTail[p] = e; slotCount[p]++;
        }
        return miss;
    }
    int64_t search(const Key& v, uint64_t& missCounter) const {                      // hashTable.cpp:193-231
        uint64_t probe = hash_value(v), p = probe, miss = 0;
        uint64_t inc = 1 + ((v.v0 + v.v1) % Mprev);
        while (slotHead[p] >= 0) {
            if (!slotLong[p]) {                                                      // :203 long buckets never match
                Key r = key_of(entVal[slotHead[p]]);
                if (r.v1 == v.v1 && r.v0 == v.v0) return (int64_t)p;
            }
            missCounter++; miss++; p = probe + miss * inc;
            while (p > M) p -= M;
        }
        return -1;
    }
    void build_index() {                                                             // hashTable.cpp:70-128
        auto t0 = std::chrono::steady_clock::now();
        h = k > 64 ? 64 : k;                                                         // :78-81
        longHash = N + 100;
        M = std::max<uint64_t>(8 * N, 100003) + 1; while (!is_prime(M)) M++;          // :243-254 (see is_prime note)
        Mprev = std::max<uint64_t>(8 * N, 100003) - 1; while (!is_prime(Mprev)) Mprev--; // :303-314
        pre = ((0xFFFFFFFFFFFFFFFFULL) % M + 1) % M;                                 // :85
        slotHead.assign(M + 1, -1); slotTail.assign(M + 1, -1); slotCount.assign(M + 1, 0); slotLong.assign(M + 1, 0);
        entVal.clear(); entNext.clear(); entVal.reserve(4 * N); entNext.reserve(4 * N);
        hashMissBuild = 0;
        for (uint64_t i = 1; i <= N; i++) {                                          // :94-109 (serial)
            hashMissBuild += insert(get128(F(i), 0, h), i, 0);
            hashMissBuild += insert(get128(F(i), len[i] - h, h), i, 1);
            hashMissBuild += insert(get128(R(i), 0, h), i, 2);
            hashMissBuild += insert(get128(R(i), len[i] - h, h), i, 3);
        }
        nLong = 0;
        for (uint64_t s = 0; s <= M; s++) if (slotHead[s] >= 0 && slotCount[s] >= 100) { slotLong[s] = 1; nLong++; }  // :111-123
        tIndex = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

    // ------------------------------------------------------------ step 3
    // economyGraph.cpp:712-758 (sideEffect=true) and :763-808 (sideEffect=false)
    // returns: 1 match / 0 no; *containedEq set when read 2 ends inside read 1 and all compared equal.
    int compare_ext(const uint8_t* a, const uint8_t* b, int start, int L1, int L2, bool previous, bool* containedEq) const {
        int s1 = start + h, s2 = h;
        if (L2 - s2 <= L1 - s1) {
            while (s2 < L2) {
                int l = std::min(64, L2 - s2);
                Key x = get128(a, s1, l), y = get128(b, s2, l);
                if (x.v0 != y.v0 || x.v1 != y.v1) return 0;
                s1 += l; s2 += l;
            }
            if (previous) return 1;                                                  // :786
            *containedEq = true; return 0;                                           // :735-736
        }
        while (s1 < L1) {
            int l = std::min(64, L1 - s1);
            Key x = get128(a, s1, l), y = get128(b, s2, l);
            if (x.v0 != y.v0 || x.v1 != y.v1) return 0;
            s1 += l; s2 += l;
        }
        return 1;
    }

    // economyGraph.cpp:37-490 buildInitialOverlapGraph
    void initial_pass() {
        auto t0 = std::chrono::steady_clock::now();
        rightE.assign(N + 1, Ext{0, 0, 0}); leftE.assign(N + 1, Ext{0, 0, 0});
        status.assign(N + 1, 0); conn.assign(N + 1, 0); adj.assign(N + 1, {});
        // The reference writes exploredReads[r2]=6 from any thread (:735) and
        // exploredReads[i]=5 from i's own thread (:444): a race.  Its single-thread
        // outcome is: 6 wins iff some container has a larger id than i.  Keep the largest
        // container id per read and resolve after the loop -> thread-count independent.
        std::vector<uint64_t> maxContainer(N + 1, 0);
        uint64_t miss = 0, ov = 0;
        #pragma omp parallel for schedule(dynamic, 256) reduction(+ : miss, ov)
        for (uint64_t i = 1; i <= N; i++) {
            const int L1 = len[i];
            Ext right{0, 0, 0}, left{0, 0, 0};
            uint64_t prevR = 0, prevL = 0; int prevTR = 0, prevTL = 0, prevLenR = 0, prevLenL = 0;
            int ambR = 0, ambL = 0; uint64_t connections = 0;
            for (int j = 0; j <= L1 - h; j++) {                                      // :79
                int64_t idx = search(get128(F(i), j, h), miss);                      // :81-82
                if (idx < 0) continue;
                int mAR = 0, mAL = 0, mFR = 0;                                       // :86-88
                for (int64_t e = slotHead[idx]; e >= 0; e = entNext[e]) {            // :89
                    uint64_t r2 = entVal[e] >> 2; int type = entVal[e] & 3; const int L2 = len[r2];
                    if (r2 == i) continue;
                    bool cont = false;
                    if ((type == 0 || type == 2) && j <= L1 - k) {                   // :94 / :187
                        const uint8_t* two = type == 0 ? F(r2) : R(r2);
                        int ok = compare_ext(F(i), two, j, L1, L2, false, &cont);
                        if (cont) { uint64_t* mc = &maxContainer[r2]; uint64_t old = __atomic_load_n(mc, __ATOMIC_RELAXED);
                                    while (old < i && !__atomic_compare_exchange_n(mc, &old, i, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }
                        if (!ok) continue;
                        const int o = type == 0 ? 0 : 1;
                        connections++;
                        if (right.id == 0) {                                         // :97-107
                            right = Ext{r2, (uint8_t)o, (uint32_t)((L2 - (L1 - j)) & 0x3FFFFF)};
                            prevR = r2; prevTR = o; prevLenR = j; mAR = 1; mFR = 1;
                        } else {
                            const uint8_t* P = prevTR == 0 ? F(prevR) : R(prevR); const int LP = len[prevR];
                            bool dummy;
                            if (compare_ext(P, two, j - prevLenR, LP, L2, true, &dummy)) {   // :114 etc.
                                if (mAR) { if (L2 > LP) { if (mFR) right = Ext{r2, (uint8_t)o, (uint32_t)((L2 - (L1 - j)) & 0x3FFFFF)};
                                                          prevR = r2; prevTR = o; prevLenR = j; } }
                                else { prevR = r2; prevTR = o; prevLenR = j; mAR = 1; }
                            } else ambR = 1;
                        }
                    } else if ((type == 1 || type == 3) && j >= k - h) {             // :279 / :359
                        const uint8_t* two = type == 1 ? R(r2) : F(r2);
                        const int off = L1 - j - h;
                        int ok = compare_ext(R(i), two, off, L1, L2, false, &cont);
                        if (cont) { uint64_t* mc = &maxContainer[r2]; uint64_t old = __atomic_load_n(mc, __ATOMIC_RELAXED);
                                    while (old < i && !__atomic_compare_exchange_n(mc, &old, i, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }
                        if (!ok) continue;
                        const int o = type == 1 ? 0 : 1;
                        connections++;
                        const uint32_t ovh = (uint32_t)((L2 - j - h) & 0x3FFFFF);
                        if (left.id == 0) {                                          // :282-291
                            left = Ext{r2, (uint8_t)o, ovh}; prevL = r2; prevTL = o; prevLenL = off; mAL = 1;
                        } else {
                            const uint8_t* P = prevTL == 0 ? R(prevL) : F(prevL); const int LP = len[prevL];
                            bool dummy;
                            if (compare_ext(two, P, prevLenL - off, L2, LP, true, &dummy)) { // :298 etc. (argument order!)
                                if (mAL) { if (L2 > LP) { left = Ext{r2, (uint8_t)o, ovh}; prevL = r2; prevTL = o; prevLenL = off; } }
                                else { left = Ext{r2, (uint8_t)o, ovh}; prevL = r2; prevTL = o; prevLenL = off; mAL = 1; }
                            } else ambL = 1;
                        }
                    }
                }
            }
            conn[i] = (uint32_t)connections; ov += connections;
            if (connections > 300) status[i] = 5;                                    // :443-444
            if (ambR || ambL) { left.len = 0; right.len = 0; }                       // :446-450 (ids stay)
            rightE[i] = right; leftE[i] = left;
        }
        hashMissSearch += miss; nOv = ov;
        for (uint64_t i = 1; i <= N; i++)                                            // single-thread outcome of the :444/:735 race
            if (maxContainer[i] && (status[i] != 5 || maxContainer[i] > i)) status[i] = 6;
        // :455-480 serial reciprocal pass
        contained = containedSize = 0;
        for (uint64_t i = 1; i <= N; i++) {
            if (status[i] == 6) { containedSize++; continue; }
            const Ext &l = leftE[i], &r = rightE[i];
            if (l.len != 0 && (rightE[l.id].id == i || leftE[l.id].id == i) &&
                r.len != 0 && (rightE[r.id].id == i || leftE[r.id].id == i)) {
                if (status[l.id] != 4) insert_edge(i, l.id, l.len, l.type == 0 ? 0 : 1);
                if (status[r.id] != 4) insert_edge(i, r.id, r.len, r.type == 0 ? 3 : 2);
                contained++; status[i] = 4;
            }
        }
        tInitial = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // economyGraph.cpp:813-849 insertEdgeEconomy (lengths are 20-bit fields, economyGraph.h:20)
    int insert_edge(uint64_t u, uint64_t v, uint32_t delta, int type) {
        if (u == v) return 0;
        int delta2 = (int)len[u] - ((int)len[v] - (int)delta);
        adj[u].push_back(EdgeE{v, (uint8_t)type, 0, delta & 0xFFFFF});
        adj[v].push_back(EdgeE{u, (uint8_t)flip_type(type), 0, (uint32_t)delta2 & 0xFFFFF});
        return 1;
    }
    static bool length_based(const EdgeE& a, const EdgeE& b) {                       // economyGraph.cpp:853-871
        if (a.len != b.len) return a.len > b.len;
        if (a.to != b.to) return a.to > b.to;
        return a.type > b.type;
    }
    static bool id_based(const EdgeE& a, const EdgeE& b) {                           // economyGraph.cpp:875-893
        if (a.to != b.to) return a.to < b.to;
        if (a.type != b.type) return a.type < b.type;
        return a.len < b.len;
    }
    // economyGraph.cpp:580-638 insertAllEdgesOfRead
    uint64_t explore(uint64_t r1) {
        if (status[r1] != 0) return 0;
        status[r1] = 1;
        const int L1 = len[r1]; uint64_t ins = 0; bool dummy;
        for (int j = 0; j <= L1 - h; j++) {
            int64_t idx = search(get128(F(r1), j, h), hashMissSearch);
            if (idx < 0) continue;
            for (int64_t e = slotHead[idx]; e >= 0; e = entNext[e]) {
                uint64_t r2 = entVal[e] >> 2; int type = entVal[e] & 3; const int L2 = len[r2];
                int32_t ovl = -1; int t = -1;
                if (status[r2]) continue;                                            // :605
                if (type == 0 && j <= L1 - k && compare_ext(F(r1), F(r2), j, L1, L2, true, &dummy)) { ovl = L2 - (L1 - j); t = 3; }
                else if (type == 1 && j >= k - h && compare_ext(R(r1), R(r2), L1 - j - h, L1, L2, true, &dummy)) { ovl = L2 - j - h; t = 0; }
                else if (type == 2 && j <= L1 - k && compare_ext(F(r1), R(r2), j, L1, L2, true, &dummy)) { ovl = L2 - (L1 - j); t = 2; }
                else if (type == 3 && j >= k - h && compare_ext(R(r1), F(r2), L1 - j - h, L1, L2, true, &dummy)) { ovl = L2 - j - h; t = 1; }
                if (ovl != -1) ins += insert_edge(r1, r2, (uint32_t)ovl, t);          // :627 (-1 is the "none" sentinel)
            }
        }
        if (adj[r1].size() > 1) std::sort(adj[r1].begin(), adj[r1].end(), length_based);  // :635-636
        return ins * 2;
    }
    // economyGraph.cpp:643-679 markTransitiveEdge
    void mark_transitive(uint64_t from, std::vector<uint8_t>& mk) {
        for (auto& e : adj[from]) mk[e.to] = 1;
        for (auto& e : adj[from]) {
            if (mk[e.to] != 1) continue;
            for (auto& f : adj[e.to]) {
                if (mk[f.to] != 1) continue;
                int t1 = e.type, t2 = f.type;
                if ((t1 == 0 || t1 == 2) && (t2 == 0 || t2 == 1)) mk[f.to] = 2;
                else if ((t1 == 1 || t1 == 3) && (t2 == 2 || t2 == 3)) mk[f.to] = 2;
            }
        }
        for (auto& e : adj[from]) if (mk[e.to] == 2) e.mark = 1;
        for (auto& e : adj[from]) mk[e.to] = 0;
        mk[from] = 0; status[from] = 2;
    }
    // economyGraph.cpp:681-707 removeTransitiveEdges
    uint64_t remove_transitive(uint64_t r) {
        size_t before = adj[r].size();
        adj[r].erase(std::remove_if(adj[r].begin(), adj[r].end(), [](const EdgeE& e) { return e.mark != 0; }), adj[r].end());
        return before - adj[r].size();
    }
    // economyGraph.cpp:495-574 buildOverlapGraphEconomy (serial BFS)
    void reduce_pass() {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<uint8_t> mk(N + 1, 0); std::vector<uint64_t> queue; queue.reserve(1024);
        edgesInserted = transRemoved = 0;
        for (uint64_t i = 1; i <= N; i++) {
            if (status[i] != 0) continue;
            queue.clear(); size_t start = 0; queue.push_back(i);
            while (start < queue.size()) {
                uint64_t r1 = queue[start++];
                if (status[r1] == 0) edgesInserted += explore(r1);
                if (adj[r1].empty()) continue;
                if (status[r1] == 1) {
                    for (size_t x = 0; x < adj[r1].size(); x++) {
                        uint64_t r2 = adj[r1][x].to;
                        if (status[r2] == 0) { queue.push_back(r2); edgesInserted += explore(r2); }
                    }
                    mark_transitive(r1, mk);
                }
                if (status[r1] == 2) {
                    for (size_t x = 0; x < adj[r1].size(); x++) {
                        uint64_t r2 = adj[r1][x].to;
                        if (status[r2] != 1) continue;
                        for (size_t y = 0; y < adj[r2].size(); y++) {
                            uint64_t r3 = adj[r2][y].to;
                            if (status[r3] == 0) { queue.push_back(r3); edgesInserted += explore(r3); }
                        }
                        mark_transitive(r2, mk);
                    }
                    transRemoved += remove_transitive(r1);
                }
            }
        }
        tReduce = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // economyGraph.cpp:896-913 sortEconomyGraph + overlapGraph.cpp:84-111 convertGraph (+ :136-163 twin length)
    void convert() {
        auto t0 = std::chrono::steady_clock::now();
        #pragma omp parallel for schedule(dynamic, 1024)
        for (uint64_t i = 1; i <= N; i++) if (adj[i].size() > 1) std::sort(adj[i].begin(), adj[i].end(), id_based);
        out.clear();
        for (uint64_t i = 1; i <= N; i++) {
            auto& a = adj[i];
            for (size_t j = 0; j < a.size(); j++) {
                if (j > 0 && a[j - 1].to == a[j].to && a[j - 1].type == a[j].type) continue;    // :101
                if (i < a[j].to) {
                    uint32_t ud = a[j].len, vd = (uint32_t)len[i] - ((uint32_t)len[a[j].to] - ud);
                    out.push_back(OutEdge{i, a[j].to, ud, vd, a[j].type});
                }
            }
        }
        tConvert = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // readLoader.cpp:270-287 + :29-36
    int write_reads(const char* path) const {
        FILE* f = fopen(path, "w"); if (!f) return -1;
        fprintf(f, "%llu\n", (unsigned long long)N);
        std::string a, b;
        for (uint64_t i = 1; i <= N; i++) {
            a.assign(len[i], 'N'); b.assign(len[i], 'N');
            unpack_bases(F(i), len[i], &a[0]); unpack_bases(R(i), len[i], &b[0]);
            fprintf(f, "%u\t%u\t%s\t%s\n", (unsigned)freq[i], (unsigned)len[i], a.c_str(), b.c_str());
        }
        fclose(f); return 0;
    }
    // overlapGraph.cpp:338-369 + :12-20 (genomeSize is 0 at step 3)
    int write_graph3(const char* path) const {
        FILE* f = fopen(path, "w"); if (!f) return -1;
        fprintf(f, "0\n%llu\n%llu\n", (unsigned long long)numberOfReads, (unsigned long long)avg_len());
        for (auto& e : out) {
            fprintf(f, "%llu\t%llu\t%u\t1\t%u\t0\t0\n\n", (unsigned long long)e.from, (unsigned long long)e.to, (unsigned)e.type, e.len);
            fprintf(f, "%llu\t%llu\t%u\t1\t%u\t0\t0\n\n", (unsigned long long)e.to, (unsigned long long)e.from, (unsigned)flip_type(e.type), e.lenTwin);
        }
        fclose(f); return 0;
    }
};
}  // namespace

// ------------------------------------------------------------------ C ABI (ctypes)
extern "C" {
void* orc_create(int k, int threads) { auto* o = new Oracle(); o->k = k; o->threads = threads; if (threads > 0) omp_set_num_threads(threads); return o; }
void orc_destroy(void* p) { delete (Oracle*)p; }
// reads packed back to back in `bases`, read r = bases[offsets[r] .. offsets[r+1])
void orc_add_reads_ascii(void* p, const char* bases, const uint64_t* offsets, uint64_t n) {
    auto* o = (Oracle*)p; for (uint64_t r = 0; r < n; r++) o->add_ascii(bases + offsets[r], (int)(offsets[r + 1] - offsets[r]));
}
void orc_organize(void* p) { ((Oracle*)p)->organize(); }
void orc_build_index(void* p) { ((Oracle*)p)->build_index(); }
void orc_initial(void* p) { ((Oracle*)p)->initial_pass(); }
void orc_reduce(void* p) { ((Oracle*)p)->reduce_pass(); }
void orc_convert(void* p) { ((Oracle*)p)->convert(); }
void orc_run_all(void* p) { auto* o = (Oracle*)p; o->build_index(); o->initial_pass(); o->reduce_pass(); o->convert(); }
int orc_write_reads(void* p, const char* path) { return ((Oracle*)p)->write_reads(path); }
int orc_write_graph3(void* p, const char* path) { return ((Oracle*)p)->write_graph3(path); }
// counters: 0 N, 1 numberOfReads, 2 totalReads, 3 totalBP, 4 avgLen, 5 nOv, 6 contained, 7 containedSize,
// 8 edgesInserted, 9 transRemoved, 10 nEdgesOut, 11 nLongBuckets, 12 M, 13 stride, 14 h, 15 distinct keys
uint64_t orc_counter(void* p, int which) {
    auto* o = (Oracle*)p;
    switch (which) { case 0: return o->N; case 1: return o->numberOfReads; case 2: return o->totalReads; case 3: return o->totalBP;
        case 4: return o->avg_len(); case 5: return o->nOv; case 6: return o->contained; case 7: return o->containedSize;
        case 8: return o->edgesInserted; case 9: return o->transRemoved; case 10: return o->out.size(); case 11: return o->nLong;
        case 12: return o->M; case 13: return (uint64_t)o->stride; case 14: return (uint64_t)o->h;
        case 15: { uint64_t n = 0; for (auto hd : o->slotHead) n += hd >= 0; return n; } }
    return 0;
}
double orc_time(void* p, int which) { auto* o = (Oracle*)p; double t[4] = {o->tIndex, o->tInitial, o->tReduce, o->tConvert}; return which >= 0 && which < 4 ? t[which] : 0; }
// per-read exports (arrays sized N+1, index 0 unused)
void orc_export_reads(void* p, uint8_t* fwd, uint16_t* len, uint16_t* freq) {
    auto* o = (Oracle*)p; memcpy(fwd, o->fwd.data(), (o->N + 1) * o->stride);
    memcpy(len, o->len.data(), (o->N + 1) * 2); memcpy(freq, o->freq.data(), (o->N + 1) * 2);
}
// ext records packed id | type<<40 | len<<42 (the reference's ExtensionTable bit layout)
void orc_export_initial(void* p, uint64_t* right, uint64_t* left, uint8_t* status, uint32_t* conn) {
    auto* o = (Oracle*)p;
    for (uint64_t i = 0; i <= o->N; i++) {
        right[i] = o->rightE[i].id | ((uint64_t)o->rightE[i].type << 40) | ((uint64_t)o->rightE[i].len << 42);
        left[i] = o->leftE[i].id | ((uint64_t)o->leftE[i].type << 40) | ((uint64_t)o->leftE[i].len << 42);
        status[i] = o->status[i]; conn[i] = o->conn[i];
    }
}
// canonical edges: 5 u64 per edge (from, to, type, len, lenTwin)
void orc_export_edges(void* p, uint64_t* e) {
    auto* o = (Oracle*)p; size_t x = 0;
    for (auto& d : o->out) { e[x++] = d.from; e[x++] = d.to; e[x++] = d.type; e[x++] = d.len; e[x++] = d.lenTwin; }
}
// index probe for unit tests: returns bucket size (0 = absent or long), fills up to cap entries (id*4+type)
int orc_lookup(void* p, uint64_t v0, uint64_t v1, uint64_t* entries, int cap) {
    auto* o = (Oracle*)p; uint64_t miss = 0; int64_t s = o->search(Key{v0, v1}, miss); if (s < 0) return 0;
    int n = 0; for (int64_t e = o->slotHead[s]; e >= 0; e = o->entNext[e]) { if (n < cap) entries[n] = o->entVal[e]; n++; }
    return n;
}
// diagnostic: hits of read r1 with the semantics of economyGraph.cpp:591-633 ignoring the explored filter;
// rows {to, type, len}; returns the count
int orc_debug_hits(void* p, uint64_t r1, int64_t* out, int cap) {
    auto* o = (Oracle*)p; int n = 0; bool dummy; uint64_t miss = 0;
    const int L1 = o->len[r1], h = o->h, k = o->k;
    for (int j = 0; j <= L1 - h; j++) {
        int64_t idx = o->search(get128(o->F(r1), j, h), miss); if (idx < 0) continue;
        for (int64_t e = o->slotHead[idx]; e >= 0; e = o->entNext[e]) {
            uint64_t r2 = o->entVal[e] >> 2; int type = o->entVal[e] & 3; const int L2 = o->len[r2]; int ovl = 0, t = -1;
            if (r2 == r1) continue;
            if (type == 0 && j <= L1 - k && o->compare_ext(o->F(r1), o->F(r2), j, L1, L2, true, &dummy)) { ovl = L2 - (L1 - j); t = 3; }
            else if (type == 1 && j >= k - h && o->compare_ext(o->R(r1), o->R(r2), L1 - j - h, L1, L2, true, &dummy)) { ovl = L2 - j - h; t = 0; }
            else if (type == 2 && j <= L1 - k && o->compare_ext(o->F(r1), o->R(r2), j, L1, L2, true, &dummy)) { ovl = L2 - (L1 - j); t = 2; }
            else if (type == 3 && j >= k - h && o->compare_ext(o->R(r1), o->F(r2), L1 - j - h, L1, L2, true, &dummy)) { ovl = L2 - j - h; t = 1; }
            if (t >= 0) { if (n < cap) { out[3 * n] = (int64_t)r2; out[3 * n + 1] = t; out[3 * n + 2] = ovl; } n++; }
        }
    }
    return n;
}
// bit-utility known-answer hooks
void orc_pack(const char* s, int len, uint8_t* out) { pack_bases(s, len, out, (len + 3) / 4); }
uint64_t orc_get64(const uint8_t* read, int start, int length) { return get64(read, start, length); }
}
