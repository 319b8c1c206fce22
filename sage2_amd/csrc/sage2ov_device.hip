// sage2_amd/csrc/sage2ov_device.hip -- hand-written HIP kernels for gfx950 (MI355X) and their launchers.
//
// Integer / bit / index work only (no MFMA): everything here is bound by HBM gathers.  Design notes
// (layouts, algorithmic bytes, rooflines) are in DESIGN.md; reference citations are relative to the
// SAGE2 tree.  Wavefront = 64 lanes everywhere; one wavefront owns one read in the probe kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "sage2ov.h"
#include "sage2ov_internal.h"

namespace s2 {

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            char b_[512];                                                                              \
            snprintf(b_, sizeof b_, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            err = b_;                                                                                  \
            return SAGE2OV_ERR_DEVICE;                                                                 \
        }                                                                                              \
    } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

struct Device {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    // reads
    u64 N = 0; int S = 0, maxL = 0, k = 0, h = 0;
    u64* reads = nullptr;        // (N+1)*S words
    // index
    u64 T = 0; u64* slots = nullptr; u32* csr = nullptr; u64 n_csr = 0; u64 seed = 0x5A6E2D0Full;
    u64 n_keys = 0, n_long = 0;
    u64* mi1 = nullptr; u64 TL = 0; u64* krec = nullptr; u64 n_groups = 0;      // minimiser index (fast kernel)
    int uniL = 0;                                                                // the common read length when all reads have one (else 0): no length gathers
    // per-read results
    u64* right = nullptr; u64* left = nullptr; u32* conn = nullptr; u32* cflag = nullptr; uint8_t* status = nullptr;
    // edge candidates
    EdgeCand* cand = nullptr; u64 cand_cap = 0; u64* d_counters = nullptr;  // [0]=n_cand [1]=n_ov [2]=contained [3]=containedSize [4]=n_hits [5]=flag
    u64 n_cand = 0;
    // final edges (device resident)
    FinalEdge* final_edges = nullptr; u64 n_final = 0;
    DevTimings tm;
    // workspace arena: buffers of the timed path are allocated once and only ever grow (no hipMalloc/hipFree per step)
    struct Buf { void* p = nullptr; size_t cap = 0; };
    Buf ws[48];
};
enum { WS_SLOTS, WS_CNT, WS_WHERE, WS_BIG, WS_CSR, WS_SLOW, WS_NEED, WS_DEG, WS_OFFS, WS_CURSOR, WS_KEYS, WS_KEEP, WS_POS, WS_OWNER, WS_FINAL,
       WS_PARTIAL, WS_IDS, WS_NEAR, WS_HITS, WS_MINH, WS_OCNT, WS_OOFF, WS_OCUR, WS_ORDER, WS_MI1, WS_MICNT, WS_MICUR, WS_KREC, WS_SLOTMH, WS_RA_DEG, WS_RA_OFF, WS_RA_CUR, WS_RA_ENT, WS_RA_RM, WS_ORG_POOL, WS_ORG_OFF, WS_ORG_LEN, WS_ORG_IMG, WS_ORG_K0, WS_ORG_K1, WS_ORG_V0, WS_ORG_V1, WS_ORG_HIST, WS_ORG_HSCAN, WS_ORG_FLAG, WS_ORG_UID, WS_ORG_HEAD };
static void* ws_get(Device* d, int id, size_t bytes) {
    Device::Buf& b = d->ws[id];
    if (b.cap < bytes || !b.p) {
        if (b.p) hipFree(b.p);
        b.p = nullptr; b.cap = 0;
        size_t want = bytes + bytes / 16 + 256;
        if (hipMalloc(&b.p, want) != hipSuccess) { b.p = nullptr; return nullptr; }
        b.cap = want;
    }
    return b.p;
}
#define WS(var, type, id, count)                                                                     \
    type* var = (type*)ws_get(d, id, (size_t)(count) * sizeof(type));                                \
    if (!var) { err = std::string("workspace allocation failed: ") + #id; return SAGE2OV_ERR_NOMEM; }

// =============================================================================================
// device helpers
// =============================================================================================
__device__ __forceinline__ u64 rev2(u64 x) {   // reverse the order of the 32 two-bit groups
    x = __brevll(x);
    return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}
__device__ __forceinline__ u64 mask_top(int nb) {   // top 2*nb bits set, nb in [0,32]
    return nb >= 32 ? ~0ull : (nb <= 0 ? 0ull : (~0ull << (64 - 2 * nb)));
}
// 64 bits of a big-endian bit string starting at bit `bitpos`; words beyond `nw` read as 0
// (branch free: both words are always loaded, the second from a clamped index)
__device__ __forceinline__ u64 bits64(const u64* w, int nw, int bitpos) {
    const int q = bitpos >> 6, r = bitpos & 63;
    const int q1 = (q + 1 < nw) ? q + 1 : q;
    const u64 a = w[q];
    u64 b = w[q1];
    b = (q + 1 < nw) ? b : 0ull;
    return (a << r) | ((b >> 1) >> (63 - r));
}
// h-base key starting at base j, right aligned in (hi,lo): the integer (v0<<64|v1) of utils.cpp:171-187
__device__ __forceinline__ void key_at(const u64* w, int nw, int j, int h, u64& hi, u64& lo) {
    if (h <= 32) { hi = 0; lo = bits64(w, nw, 2 * j) >> (64 - 2 * h); }
    else { hi = bits64(w, nw, 2 * j) >> (128 - 2 * h); lo = bits64(w, nw, 2 * j + 2 * h - 64); }
}
// reverse complement of an h-base key
__device__ __forceinline__ void rc_key(u64 hi, u64 lo, int h, u64& rhi, u64& rlo) {
    u64 a = rev2(lo), b = rev2(hi);          // (a:b) = 128-bit group-reversed value, key now in the top 2h bits
    int sh = 128 - 2 * h;
    u64 nh, nl;
    if (sh >= 64) { nh = 0; nl = (sh == 64) ? a : (a >> (sh - 64)); }
    else if (sh == 0) { nh = a; nl = b; }
    else { nh = a >> sh; nl = (b >> sh) | (a << (64 - sh)); }
    nh = ~nh; nl = ~nl;
    if (2 * h <= 64) { nh = 0; if (2 * h < 64) nl &= (1ull << (2 * h)) - 1; }
    else if (2 * h < 128) nh &= (1ull << (2 * h - 64)) - 1;
    rhi = nh; rlo = nl;
}
// ---- hashing: the key is taken LEFT aligned as four big-endian dwords k0..k3 (2h bits, rest zero) and mixed with
// rotate/add/xor only (Bob Jenkins' lookup3 final mix): full-rate 32-bit VALU ops, no 64-bit multiplies.
// The returned pair is (h1 -> home slot by multiply-high with T, h2 -> 24-bit tag).
__device__ __forceinline__ u32 rotl32(u32 x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }
#define S2_FINAL(a, b, c) { c ^= b; c -= rotl32(b, 14); a ^= c; a -= rotl32(c, 11); b ^= a; b -= rotl32(a, 25); c ^= b; c -= rotl32(b, 16); \
                            a ^= c; a -= rotl32(c, 4); b ^= a; b -= rotl32(a, 14); c ^= b; c -= rotl32(b, 24); }
__device__ __forceinline__ u64 hash4(u32 k0, u32 k1, u32 k2, u32 k3, u32 seed, bool four) {
    u32 a = 0xdeadbeefu + seed + k0, b = 0x9e3779b9u + k1, c = 0x7f4a7c15u + k2;
    S2_FINAL(a, b, c);
    if (four) { a += k3; S2_FINAL(a, b, c); }          // `four` is uniform: keys longer than 48 bases
    return ((u64)c << 32) | b;
}
// (hi,lo) = right-aligned 2h-bit key (the integer of utils.cpp:171-187) -> same hash as the left-aligned dwords
__device__ __forceinline__ u64 hash_key(u64 hi, u64 lo, int h, u64 seed) {
    const int sh = 128 - 2 * h; u64 nh, nl;
    if (sh >= 64) { nh = sh == 64 ? lo : (lo << (sh - 64)); nl = 0; }
    else if (sh == 0) { nh = hi; nl = lo; }
    else { nh = (hi << sh) | (lo >> (64 - sh)); nl = lo << sh; }
    return hash4((u32)(nh >> 32), (u32)nh, (u32)(nl >> 32), (u32)nl, (u32)seed, 2 * h > 96);
}
// 24-bit tag of a key.  g_tag_mask is 0xFFFFFF except in tests, which shrink it (SAGE2OV_TEST_TAG_BITS) so that different keys share
// tags often: everything that depends on "equal tag and same probe chain => one bucket" then runs thousands of times per data set
// instead of about once per ten million reads.
__device__ u32 g_tag_mask = 0xFFFFFFu;
__device__ __forceinline__ u32 tag_of(u64 hv) { u32 t = (u32)hv & g_tag_mask; return t ? t : 1u; }
// ---- minimiser index (second access path used by the fast kernel): distinct keys grouped by their minimiser
// (smallest hashed w-mer of the key, w = min(16,h)).  The ~13 consecutive windows of a read that share a minimiser
// find their keys in ONE contiguous group instead of 13 random sectors of the uniform table.
__device__ __forceinline__ u32 mix32(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ u32 wmer_hash(u32 wm) { wm *= 0x9E3779B1u; return wm ^ (wm >> 15); }        // order of w-mers (one multiply)
__device__ __forceinline__ u32 minim_hash(u32 minh, u32 seed) { return mix32(minh ^ (seed * 0x85EBCA77u + 0x165667B1u)); }   // uniform 32 bits
__device__ u32 g_mtag_mask = 0xFFFFFFu;          // tests shrink it (SAGE2OV_TEST_MTAG_BITS): groups of different minimisers then merge often
__device__ __forceinline__ u32 minim_tag(u32 mh) { u32 t = mix32(mh + 0x2545F491u) & g_mtag_mask; return t ? t : 1u; }
__device__ __forceinline__ u32 funnel32k(u32 a, u32 b, int r) { return (u32)(((((u64)a) << 32) | b) >> (32 - r)); }
// smallest w-mer hash of a left-aligned h-base key (dwords k0..k3)
__device__ __forceinline__ u32 key_min_hash(u32 k0, u32 k1, u32 k2, u32 k3, int h) {
    const int w = h < 16 ? h : 16, m = h - w + 1;
    u32 best = ~0u;
    for (int p = 0; p < m; p++) {
        const int q = (2 * p) >> 5, r = (2 * p) & 31;
        const u32 a = q == 0 ? k0 : (q == 1 ? k1 : (q == 2 ? k2 : k3));
        const u32 b = q == 0 ? k1 : (q == 1 ? k2 : (q == 2 ? k3 : 0u));
        const u32 wm = funnel32k(a, b, r) >> (32 - 2 * w);
        const u32 hh = wmer_hash(wm);
        best = hh < best ? hh : best;
    }
    return best;
}
// left-aligned dwords of a right-aligned (hi,lo) key
__device__ __forceinline__ void key_left_align(u64 hi, u64 lo, int h, u32& k0, u32& k1, u32& k2, u32& k3) {
    const int sh = 128 - 2 * h; u64 nh, nl;
    if (sh >= 64) { nh = sh == 64 ? lo : (lo << (sh - 64)); nl = 0; }
    else if (sh == 0) { nh = hi; nl = lo; }
    else { nh = (hi << sh) | (lo >> (64 - sh)); nl = lo << sh; }
    k0 = (u32)(nh >> 32); k1 = (u32)nh; k2 = (u32)(nl >> 32); k3 = (u32)nl;
}
constexpr u32 MI_BIG = 255;      // group too large or ambiguous: its windows use the uniform table
// home slot: always even, so that a 16-byte load covers two consecutive slots of the (linear) probe sequence; T is even
__device__ __forceinline__ u64 home_of(u64 hv, u64 T) { return 2ull * (u64)__umulhi((u32)(hv >> 32), (u32)(T >> 1)); }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
    u32 lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { u32 t = __shfl_up(v, d); if (lane >= (u32)d) v += t; }
    return v;
}

// look a key up: returns slot word (0 = not found / long bucket handled by caller)
__device__ __forceinline__ u64 table_find(const u64* __restrict__ slots, u64 T, u64 hv) {
    u64 idx = home_of(hv, T);
    const u64 tag = tag_of(hv);
    for (;;) {
        u64 s = slots[idx];
        if (s == 0) return 0;
        if ((s >> SLOT_TAG_SHIFT) == tag) return s;
        if (++idx == T) idx = 0;
    }
}

// =============================================================================================
// locality order.  Read ids follow the reference's lexicographic order, i.e. they are random with respect to
// the genome, so consecutive waves would probe unrelated keys and gather unrelated reads (every access an HBM
// sector).  Overlapping reads share their windows and their neighbours: processing reads grouped by a
// locality-sensitive key turns most slot probes and read gathers into L2 hits.  Key = the smallest hashed
// canonical 16-mer of the read (its "global minimiser"): reads sharing it come from the same ~300 bp of genome.
// Only the PROCESSING order changes (a permutation of the ids); results do not depend on it.
// =============================================================================================
__global__ void k_minimizer(const u64* __restrict__ reads, u64 lo, u64 hi, int S, u32* minh) {
    const u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    const u64* w = reads + i * S;
    const int L = (int)(w[S - 1] & 0xFFFF);
    u32 f = 0, r = 0, best = ~0u;
    u64 cur = 0;
    for (int p = 0; p < L; p++) {
        if ((p & 31) == 0) cur = w[p >> 5];
        const u32 b = (u32)(cur >> 62); cur <<= 2;
        f = (f << 2) | b; r = (r >> 2) | ((3u - b) << 30);
        if (p >= 15) { const u32 c = f < r ? f : r; const u32 hsh = mix32(c); best = hsh < best ? hsh : best; }
    }
    minh[i - lo] = best;
}
__global__ void k_order_count(const u32* __restrict__ minh, u64 n, int shift, u32* cnt) {
    const u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (x >= n) return;
    atomicAdd(&cnt[minh[x] >> shift], 1u);
}
// (ordering the reads of a bucket by their position relative to the minimiser was tried: no effect on the probe kernel)
__global__ void k_order_fill(const u32* __restrict__ minh, u64 n, u64 lo, int shift, const u32* __restrict__ offs, u32* cursor, u32* order) {
    const u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (x >= n) return;
    const u32 b = minh[x] >> shift; const u32 p = atomicAdd(&cursor[b], 1u);
    order[offs[b] + p] = (u32)(lo + x);
}

// =============================================================================================
// Step 1 on the device (readLoader.cpp:179-260): canonical orientation, sort, unique + frequency, read ids.
// Input: the good reads as the host staged them (forward strand, 2-bit big-endian words, variable length).
//   k_org_canon   read < revcomp ? read : revcomp (:195), written as an S-word slot with the length in the low 16 bits of
//                 the last word: comparing slots word by word IS stringCompareInBytes (utils.cpp:224: bytes, then length)
//   k_rs_*        stable LSD radix sort of (first word, read index), 8 passes of 8 bits, one wave per 2048-element tile
//   k_org_ties    runs of equal first words ordered by the remaining words (insertion sort; duplicates cost one compare each)
//   k_org_heads   first read of every run of equal slots = a unique read; exclusive scan = id - 1; run length = frequency (u16 wrap)
//   k_org_gather  the HBM read store in id order (slot 0 = zeros)
// =============================================================================================
constexpr int RS_TILE = 2048;
__global__ void k_org_canon(const u64* __restrict__ pool, const u64* __restrict__ off, const unsigned short* __restrict__ len, u64 n, int S, u64* img, u64* key0, u32* val) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    const int L = len[i], nw = (L + 31) / 32; const u64* f = pool + off[i];
    u64 fw[17], rw[16];
#pragma unroll 1
    for (int c = 0; c < 17; c++) fw[c] = c < nw ? f[c] : 0ull;
    bool useF = false, decided = false;                                  // readLoader.cpp:195 (tie: the reverse complement, same bytes)
#pragma unroll 1
    for (int c = 0; c < nw; c++) {
        const int rem = L - 32 * c; u64 r;
        if (rem >= 32) r = ~rev2(bits64(fw, 17, 2 * (rem - 32)));
        else r = (~rev2(fw[0] >> (64 - 2 * rem))) & mask_top(rem);
        rw[c] = r;
        if (!decided && fw[c] != r) { useF = fw[c] < r; decided = true; }
    }
    u64* o = img + i * S;
#pragma unroll 1
    for (int c = 0; c < S; c++) { u64 v = c < nw ? (useF ? fw[c] : rw[c]) : 0ull; if (c == S - 1) v |= (u64)L; o[c] = v; if (c == 0) key0[i] = v; }
    val[i] = (u32)i;
}
__global__ __launch_bounds__(64) void k_rs_hist(const u64* __restrict__ keys, u64 n, int shift, u32* hist, u32 nb) {
    __shared__ u32 h[256];
    for (int x = threadIdx.x; x < 256; x += 64) h[x] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * RS_TILE;
    for (int c = 0; c < RS_TILE; c += 64) { const u64 x = base + c + threadIdx.x; if (x < n) atomicAdd(&h[(u32)(keys[x] >> shift) & 255u], 1u); }
    __syncthreads();
    for (int x = threadIdx.x; x < 256; x += 64) hist[(u64)x * nb + blockIdx.x] = h[x];
}
__global__ __launch_bounds__(64) void k_rs_scatter(const u64* __restrict__ keys, const u32* __restrict__ vals, u64 n, int shift, const u32* __restrict__ hscan, u32 nb,
                                                  u64* keysOut, u32* valsOut) {
    __shared__ u32 cnt[256];
    const u32 lane = threadIdx.x;
    for (int x = lane; x < 256; x += 64) cnt[x] = hscan[(u64)x * nb + blockIdx.x];
    wave_sync();
    const u64 base = (u64)blockIdx.x * RS_TILE;
    for (int c = 0; c < RS_TILE; c += 64) {
        const u64 x = base + c + lane; const bool valid = x < n;
        const u64 k = valid ? keys[x] : 0ull; const u32 v = valid ? vals[x] : 0u;
        const u32 dgt = (u32)(k >> shift) & 255u;
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) { const u64 m = __ballot((dgt >> b) & 1u); peers &= ((dgt >> b) & 1u) ? m : ~m; }
        const u32 rank = (u32)__popcll(peers & ((1ull << lane) - 1ull));
        const u32 old = cnt[dgt];
        wave_sync();
        if (valid && rank == 0) cnt[dgt] = old + (u32)__popcll(peers);
        wave_sync();
        if (valid) { keysOut[old + rank] = k; valsOut[old + rank] = v; }
    }
}
__device__ __forceinline__ int org_cmp(const u64* __restrict__ img, int S, u32 a, u32 b) {          // words 1.. (word 0 is known equal)
    const u64 *pa = img + (u64)a * S, *pb = img + (u64)b * S;
    for (int c = 1; c < S; c++) { const u64 x = pa[c], y = pb[c]; if (x != y) return x < y ? -1 : 1; }
    return 0;
}
__global__ void k_org_ties(const u64* __restrict__ keys, u32* vals, u64 n, const u64* __restrict__ img, int S) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    const u64 k = keys[i];
    if (i > 0 && keys[i - 1] == k) return;                               // not the first of its run
    u64 e = i + 1; while (e < n && keys[e] == k) e++;
    for (u64 x = i + 1; x < e; x++) {
        const u32 v = vals[x]; u64 j = x;
        while (j > i && org_cmp(img, S, vals[j - 1], v) > 0) { vals[j] = vals[j - 1]; j--; }
        vals[j] = v;
    }
}
__global__ void k_org_heads(const u64* __restrict__ keys, const u32* __restrict__ vals, u64 n, const u64* __restrict__ img, int S, u32* flag) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    flag[i] = (i == 0 || keys[i - 1] != keys[i] || org_cmp(img, S, vals[i - 1], vals[i]) != 0) ? 1u : 0u;
}
__global__ void k_org_headpos(const u32* __restrict__ flag, const u32* __restrict__ uid, u64 n, u32* headPos, u64 N) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) headPos[uid[i]] = (u32)i;
    if (i == 0) headPos[N] = (u32)n;
}
__global__ void k_org_gather(const u32* __restrict__ vals, const u32* __restrict__ headPos, u64 N, const u64* __restrict__ img, int S, u64* reads, unsigned short* freq) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 id = t / S + 1; const int c = (int)(t % S);
    if (id > N) return;
    const u32 hp = headPos[id - 1];
    reads[id * S + c] = img[(u64)vals[hp] * S + c];
    if (c == 0) freq[id] = (unsigned short)(headPos[id] - hp);          // u16 wrap like the reference (readLoader.cpp:232)
}

// =============================================================================================
// index build: count -> alloc -> fill -> sort (+ purity check of long buckets)
// hashTable.cpp:70-128 semantics: multimap key -> entries in (id,type) order; >= 100 entries = long.
// =============================================================================================
__device__ __forceinline__ void entry_key(const u64* __restrict__ reads, int S, int h, u64 id, int t, u64& hi, u64& lo) {
    const u64* w = reads + id * S;
    int L = (int)(w[S - 1] & 0xFFFF);
    u64 phi, plo;
    if (t == 0 || t == 3) key_at(w, S, 0, h, phi, plo); else key_at(w, S, L - h, h, phi, plo);
    if (t >= 2) rc_key(phi, plo, h, hi, lo); else { hi = phi; lo = plo; }   // hashTable.cpp:96-104
}

// During the build a slot is tag:24 | number of entries seen so far:40 and a group word is mtag:24 | number of keys so far:40:
// claiming and counting are one atomic on one word, and the value the atomic returns is the entry's rank inside its bucket
// (the key's rank inside its group), so the fill kernel needs no cursors.  k_index_alloc / k_mi_alloc rewrite the words
// into their final form.
constexpr u64 BUILD_CNT_MASK = (1ull << SLOT_TAG_SHIFT) - 1;       // group words: mtag:24 | keys so far:40
// slot words during the build: tag:24 | fingerprint:16 | entries so far:24.  The fingerprint (16 more hash bits) tells a later
// arrival whether the bucket it joins was claimed by ANOTHER key with the same tag on the same probe chain.  The uniform table
// wants exactly that merge (one tag, one bucket: look-ups verify every candidate), but the minimiser groups are per key: the
// arrival then files a second record of the bucket under its own key's minimiser (WHERE_REC), or its windows would never find it.
constexpr u64 SLOT_BUILD_CNT = (1ull << 24) - 1;
constexpr int SLOT_FP_SHIFT = 24;
constexpr u64 WHERE_REC = 1ull << 63;                                // where[e]: this entry writes a group record in the fill kernel
__device__ __forceinline__ u64 fp_of(u64 hv) { return ((hv >> 24) ^ (hv >> 47)) & 0xFFFFull; }
__device__ __forceinline__ u64 mi_claim_count(u64* mi1, u64 TL, u32 mh, u32& rank) {
    const u64 mt = minim_tag(mh); u64 idx = __umulhi(mh, (u32)TL);
    for (u32 step = 0; step < 2048u; step++) {
        u64 v = __hip_atomic_load(&mi1[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 0) {
            v = atomicCAS((u64*)&mi1[idx], 0ull, (mt << SLOT_TAG_SHIFT) | 1ull);
            if (v == 0) { rank = 0; return idx; }
        }
        if ((v >> SLOT_TAG_SHIFT) == mt) { rank = (u32)(atomicAdd((u64*)&mi1[idx], 1ull) & BUILD_CNT_MASK); return idx; }
        if (++idx == TL) idx = 0;
    }
    return ~0ull;                                                      // table too crowded: the caller gives the minimiser index up
}
__global__ void k_index_count(const u64* __restrict__ reads, u64 N, int S, int h, u64 seed, u64* slots, u64 T, u64* where,
                              u64* whereG, u64* mi1, u64 TL, u64* micounters) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (; e < 4 * N; e += stride) {
        u64 hi, lo; entry_key(reads, S, h, (e >> 2) + 1, (int)(e & 3), hi, lo);
        u64 hv = hash_key(hi, lo, h, seed); const u64 tag = tag_of(hv); u64 idx = home_of(hv, T);
        u64 rank = 0; bool files = false; const u64 fp = fp_of(hv);
        for (;;) {
            u64 s = __hip_atomic_load(&slots[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s == 0) {
                s = atomicCAS((u64*)&slots[idx], 0ull, (tag << SLOT_TAG_SHIFT) | (fp << SLOT_FP_SHIFT) | 1ull);
                if (s == 0) { files = true; break; }                   // claimed: rank 0
            }
            if ((s >> SLOT_TAG_SHIFT) == tag) {
                rank = atomicAdd((u64*)&slots[idx], 1ull) & SLOT_BUILD_CNT;
                if (rank >= SLOT_BUILD_CNT - 8) atomicAdd(&micounters[8], 1ull);          // more than 16 M entries under one key: not supported
                files = ((s >> SLOT_FP_SHIFT) & 0xFFFFull) != fp;      // another key's bucket (same tag, same chain)
                break;
            }
            if (++idx == T) idx = 0;
        }
        if (whereG && files) {                                         // (at least) one thread per key, key still in registers: its minimiser group (stage B)
            u32 k0, k1, k2, k3; key_left_align(hi, lo, h, k0, k1, k2, k3);
            u32 grank = 0;
            const u64 g = mi_claim_count(mi1, TL, minim_hash(key_min_hash(k0, k1, k2, k3, h), (u32)seed), grank);
            if (g == ~0ull) { atomicAdd(&micounters[7], 1ull); whereG[e] = ~0ull; }
            else whereG[e] = g | ((u64)grank << 32);                   // read back by the same entry in the fill kernel: a stream, not a gather
        }
        where[e] = idx | (rank << 32) | (files ? WHERE_REC : 0ull);    // (ranks stay below 2^24)
    }
}
__global__ void k_debug_table(const u64* __restrict__ slots, u64 T, u64* out) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (idx >= T) return;
    u64 s = slots[idx]; if (s == 0) return;
    u32 c7 = (u32)(s >> SLOT_CNT_SHIFT) & 127u;
    atomicAdd(&out[0], 1ull);
    if (c7 == 1) atomicAdd(&out[1], 1ull);
    if (c7 == 0) atomicAdd(&out[2], 1ull);                 // claimed, never filled
    if ((s >> SLOT_TAG_SHIFT) == 0) atomicAdd(&out[3], 1ull);
    if (c7 >= 2 && c7 < 127) atomicAdd(&out[4], (u64)c7);
}
__global__ void k_debug_keys(const u64* __restrict__ reads, u64 N, int S, int h, u64* out) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 4 * N) return;
    u64 hi, lo; entry_key(reads, S, h, (e >> 2) + 1, (int)(e & 3), hi, lo);
    out[2 * e] = hi; out[2 * e + 1] = lo;
}
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* sh, u32& total);
// counters: [0] csr total, [1] occupied slots, [2] number of big buckets, [3] impurity flag, [4] pure long buckets
// One block owns ALLOC_ITEMS consecutive slots and draws its CSR space with ONE atomic (a per-wave atomic on
// the same word serialises: 25 ms for 68 M slots; this form streams at HBM speed).
constexpr int ALLOC_PER_THREAD = 64, ALLOC_ITEMS = 256 * ALLOC_PER_THREAD;   // 16384 slots per block: its two atomics stay off the critical path
__global__ __launch_bounds__(256) void k_index_alloc(u64* slots, u64 T, u64* counters, u64* big, u32 big_cap) {
    __shared__ u32 sh[4]; __shared__ u64 shBase;
    const u64 base0 = (u64)blockIdx.x * ALLOC_ITEMS;
    u32 c[ALLOC_PER_THREAD]; u32 need = 0, occ = 0;
#pragma unroll
    for (int x = 0; x < ALLOC_PER_THREAD; x++) {
        const u64 idx = base0 + (u64)x * 256 + threadIdx.x;
        c[x] = idx < T ? (u32)(slots[idx] & SLOT_BUILD_CNT) : 0u;
        need += c[x] >= 2 ? c[x] : 0u; occ += c[x] != 0;
    }
    u32 total; u32 excl = block_excl_scan(need, sh, total);
    u32 occTotal; block_excl_scan(occ, sh, occTotal);
    if (threadIdx.x == 0) { shBase = total ? atomicAdd(&counters[0], (u64)total) : 0ull; if (occTotal) atomicAdd(&counters[1], (u64)occTotal); }
    __syncthreads();
    u64 start = shBase + excl;
#pragma unroll
    for (int x = 0; x < ALLOC_PER_THREAD; x++) {
        if (c[x] == 1) { const u64 idx = base0 + (u64)x * 256 + threadIdx.x; slots[idx] &= ~0ull << SLOT_TAG_SHIFT; }   // single entry: the fill kernel writes it inline
        if (c[x] >= 2) {
            const u64 idx = base0 + (u64)x * 256 + threadIdx.x;
            const u32 c7 = c[x] >= HASH_THRESHOLD ? SLOT_CNT_LONG : c[x];
            slots[idx] = (slots[idx] & (~0ull << SLOT_TAG_SHIFT)) | ((u64)c7 << SLOT_CNT_SHIFT) | start;
            if (c[x] >= HASH_THRESHOLD) {
                u64 b = atomicAdd(&counters[2], 1ull);
                if (b < big_cap) { big[3 * b] = idx; big[3 * b + 1] = start; big[3 * b + 2] = c[x]; }
            }
            start += c[x];
        }
    }
}
__global__ void k_index_fill(u64 N, u64* slots, const u64* __restrict__ where, u32* csr,
                             const u64* __restrict__ whereG, const u64* __restrict__ mi1, u64* krec) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (; e < 4 * N; e += stride) {
        const u64 wv = where[e]; const u64 idx = wv & 0xFFFFFFFFull; const u32 rank = (u32)(wv >> 32) & 0x7FFFFFFFu;
        u64 s = slots[idx];
        const u32 c7 = (u32)(s >> SLOT_CNT_SHIFT) & 127u;
        const u32 entry = (u32)(((e >> 2) + 1) * 4 + (e & 3));
        if (c7 == 0) { s = (s & (~0ull << SLOT_TAG_SHIFT)) | (1ull << SLOT_CNT_SHIFT) | entry; slots[idx] = s; }   // the only entry: inline
        else csr[(s & SLOT_PAY_MASK) + rank] = entry;
        if (whereG && (wv & WHERE_REC)) {                              // one entry per key: the bucket's record goes into the key's minimiser group (stage B)
            const u64 sg = whereG[e];
            if (sg != ~0ull) {
                const u64 v = mi1[(u32)sg];
                if (((v >> 32) & 255u) != MI_BIG) krec[(u32)v + (u32)(sg >> 32)] = s;      // (oversized groups are never scanned)
            }
        }
    }
}
__global__ void k_index_sort(const u64* __restrict__ slots, u64 T, u32* csr) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T) return;
    u64 s = slots[idx]; u32 c7 = (u32)(s >> SLOT_CNT_SHIFT) & 127u;
    if (c7 < 2 || c7 == SLOT_CNT_LONG) return;
    u32* a = csr + (s & SLOT_PAY_MASK);
    for (u32 i = 1; i < c7; i++) { u32 v = a[i]; int j = (int)i - 1; while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; j--; } a[j + 1] = v; }
}
// one wave per big bucket: every entry must carry the same true key, else two keys were merged by
// their 24-bit tags and the ">= 100 entries" verdict is not trustworthy -> ask for a reseed.
__global__ void k_index_purity(const u64* __restrict__ reads, int S, int h, const u64* __restrict__ big, u64 nbig, const u32* __restrict__ csr, u64* counters) {
    u64 b = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= nbig) return;
    u64 start = big[3 * b + 1], c = big[3 * b + 2];
    u32 e0 = csr[start]; u64 hi0, lo0; entry_key(reads, S, h, e0 >> 2, e0 & 3, hi0, lo0);
    bool bad = false;
    for (u64 x = lane_id(); x < c; x += 64) { u32 e = csr[start + x]; u64 hi, lo; entry_key(reads, S, h, e >> 2, e & 3, hi, lo); if (hi != hi0 || lo != lo0) bad = true; }
    if (__ballot(bad)) { if (lane_id() == 0) atomicAdd(&counters[3], 1ull); }
    else if (lane_id() == 0) atomicAdd(&counters[4], 1ull);
}
// ---- stage B of the index build: group the distinct-key records of the uniform table by minimiser
// bounded: a full group table (more distinct minimisers than expected) must never spin forever; ~0 = gave up
__global__ __launch_bounds__(256) void k_mi_alloc(u64* mi1, u64 TL, u64* counters) {   // counters[5]: records placed, [6]: groups
    __shared__ u32 sh[4]; __shared__ u64 shBase;
    const u64 base0 = (u64)blockIdx.x * ALLOC_ITEMS;
    u32 c[ALLOC_PER_THREAD]; u32 need = 0, occ = 0;
#pragma unroll
    for (int x = 0; x < ALLOC_PER_THREAD; x++) { const u64 idx = base0 + (u64)x * 256 + threadIdx.x; c[x] = idx < TL ? (u32)(mi1[idx] & BUILD_CNT_MASK) : 0u; need += c[x]; occ += c[x] != 0; }
    u32 total; const u32 excl = block_excl_scan(need, sh, total);
    u32 occTotal; block_excl_scan(occ, sh, occTotal);
    if (threadIdx.x == 0) { shBase = total ? atomicAdd(&counters[5], (u64)total) : 0ull; if (occTotal) atomicAdd(&counters[6], (u64)occTotal); }
    __syncthreads();
    u64 start = shBase + excl;
#pragma unroll
    for (int x = 0; x < ALLOC_PER_THREAD; x++) {
        if (c[x]) {
            const u64 idx = base0 + (u64)x * 256 + threadIdx.x;
            mi1[idx] = (mi1[idx] & (~0ull << SLOT_TAG_SHIFT)) | ((u64)(c[x] >= MI_BIG ? MI_BIG : c[x]) << 32) | start;
            start += c[x];
        }
    }
}
// self-check of the minimiser groups (SAGE2OV_VERIFY_MI): every entry's key must find, in the group of its own minimiser, a record that
// is the word of the slot the entry went to.  counters: [0] entries without such a record, [1] entries checked
__global__ void k_mi_verify(const u64* __restrict__ reads, u64 N, int S, int h, u64 seed, const u64* __restrict__ slots, const u64* __restrict__ where,
                            const u64* __restrict__ mi1, u64 TL, const u64* __restrict__ krec, u64* out) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (e >= 4 * N) return;
    u64 hi, lo; entry_key(reads, S, h, (e >> 2) + 1, (int)(e & 3), hi, lo);
    u32 k0, k1, k2, k3; key_left_align(hi, lo, h, k0, k1, k2, k3);
    const u32 mh = minim_hash(key_min_hash(k0, k1, k2, k3, h), (u32)seed); const u64 mt = minim_tag(mh);
    u64 idx = __umulhi(mh, (u32)TL); u64 gv = 0;
    for (u32 step = 0; step < 4096u; step++) { gv = mi1[idx]; if (gv == 0 || (gv >> SLOT_TAG_SHIFT) == mt) break; if (++idx == TL) idx = 0; }
    const u64 want = slots[where[e] & 0xFFFFFFFFull];
    bool ok = false;
    if (gv != 0) { const u32 n = (u32)(gv >> 32) & 255u, st = (u32)gv; if (n == MI_BIG) ok = true; else for (u32 x = 0; x < n; x++) if (krec[st + x] == want) ok = true; }
    atomicAdd(&out[1], 1ull);
    if (!ok) { if (atomicAdd(&out[0], 1ull) < 8) printf("[verify-mi] e=%llu where=%llx want=%llx gv=%llx fp=%llx\n", (unsigned long long)e, (unsigned long long)where[e], (unsigned long long)want, (unsigned long long)gv, (unsigned long long)fp_of(hash_key(hi, lo, h, seed))); }
}
__global__ void k_lookup(const u64* __restrict__ slots, u64 T, const u32* __restrict__ csr, u64 seed, int h, u64 hi, u64 lo, u64* out, u32 cap) {
    u64 s = table_find(slots, T, hash_key(hi, lo, h, seed));
    u32 c7 = (u32)(s >> SLOT_CNT_SHIFT) & 127u;
    if (s == 0 || c7 == SLOT_CNT_LONG) { out[0] = 0; return; }
    out[0] = c7;
    if (c7 == 1) { if (cap) out[1] = s & SLOT_PAY_MASK; return; }
    for (u32 x = 0; x < c7 && x < cap; x++) out[1 + x] = csr[(s & SLOT_PAY_MASK) + x];
}

// =============================================================================================
// exclusive scan of u32 (3 kernels, 2048 items per block)
// =============================================================================================
constexpr int SCAN_ITEMS = 8, SCAN_THREADS = 256, SCAN_BLOCK = SCAN_ITEMS * SCAN_THREADS;
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* sh, u32& total) {   // sh: 4 words (one per wave)
    u32 incl = wave_incl_scan(v); u32 w = threadIdx.x >> 6;
    if (lane_id() == 63) sh[w] = incl;
    __syncthreads();
    u32 add = 0, t = 0;
    for (u32 x = 0; x < SCAN_THREADS / 64; x++) { u32 s = sh[x]; if (x < w) add += s; t += s; }
    __syncthreads();
    total = t;
    return add + incl - v;
}
__global__ void k_scan_reduce(const u32* __restrict__ in, u64 n, u64* partial) {
    __shared__ u32 sh[4];
    u64 base = (u64)blockIdx.x * SCAN_BLOCK + (u64)threadIdx.x * SCAN_ITEMS; u32 s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) s += in[base + i];
    u32 total; block_excl_scan(s, sh, total);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}
__global__ void k_scan_partials(u64* partial, u64 nb, u64* total_out) {   // single block
    __shared__ u64 carry; __shared__ u64 shw[16];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u64 base = 0; base < nb; base += blockDim.x) {
        u64 i = base + threadIdx.x; u64 v = i < nb ? partial[i] : 0;
        // wave inclusive scan (u64)
        u64 incl = v; u32 lane = lane_id();
        for (int d = 1; d < 64; d <<= 1) { u64 t = __shfl_up(incl, d); if (lane >= (u32)d) incl += t; }
        u32 w = threadIdx.x >> 6;
        if (lane == 63) shw[w] = incl;
        __syncthreads();
        u64 add = 0, tot = 0;
        for (u32 x = 0; x < blockDim.x / 64; x++) { u64 s = shw[x]; if (x < w) add += s; tot += s; }
        u64 c = carry;
        if (i < nb) partial[i] = c + add + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}
__global__ void k_scan_final(const u32* __restrict__ in, u64 n, const u64* __restrict__ partial, u32* out) {
    __shared__ u32 sh[4];
    u64 base = (u64)blockIdx.x * SCAN_BLOCK + (u64)threadIdx.x * SCAN_ITEMS; u32 v[SCAN_ITEMS]; u32 s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    u32 total; u32 ex = block_excl_scan(s, sh, total) + (u32)partial[blockIdx.x];
    for (int i = 0; i < SCAN_ITEMS; i++) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
}

// =============================================================================================
// probe + verify + extension state machine  (economyGraph.cpp:72-451), one wavefront per read.
// MODE 0: initial pass.  MODE 1: directional hit lists of status-0 reads (economyGraph.cpp:591-633).
// =============================================================================================
template <int S>
struct WaveLds {
    u64 x[2][S + 1];        // this read: forward strand, reverse complement (+1 zero pad word)
    u64 y[S + 1][64];       // candidate reads, word-major so lane-consecutive (row S = zero pad)
    u32 candJ[64], candE[64];
    u32 hitR2[64], hitA[64], hitB[64];   // A: side | o<<1 | L2<<2 | j<<18 ; B: overhang length
    u64 hitOv[S][64];       // overhang strings, left aligned
};

template <int S>
__device__ __forceinline__ u64 ybits(const u64 (*y)[64], u32 lane, int bitpos) {
    const int q = bitpos >> 6, r = bitpos & 63;
    const u64 a = y[q][lane], b = y[q + 1][lane];          // row S is a zero pad
    return (a << r) | ((b >> 1) >> (63 - r));
}
// n bases of X from xa equal n bases of candidate (lane column of y) from ya
template <int S>
__device__ __forceinline__ bool eq_range(const u64* x, int xa, const u64 (*y)[64], u32 lane, int ya, int n) {
    for (int c = 0; c * 32 < n; c++) {
        u64 a = bits64(x, S + 1, 2 * xa + 64 * c), b = ybits<S>(y, lane, 2 * ya + 64 * c);
        if ((a ^ b) & mask_top(n - 32 * c)) return false;
    }
    return true;
}

struct ProbeArgs {
    const u64* reads; u64 N; int S, k, h;
    const u64* slots; u64 T; const u32* csr; u64 seed;
    u64 lo, hi;                    // read id range [lo, hi)
    u64* right; u64* left; u32* conn; u32* cflag;         // MODE 0 outputs
    const uint8_t* status; Hit* hits; u64 hits_cap; u64* counters;   // MODE 1
    u32* hitcount;                                         // MODE 1, optional: number of hits of every read (written for status-0 reads)
    const u32* ids; u64 n_ids;                             // optional explicit read list (replaces [lo,hi))
    const u64* mi1; u64 TL; const u64* krec;               // minimiser index (may be null)
    u32* slow; u64 slow_cap;                               // fast kernel: reads handed to the sequential kernel (count in counters[6])
    u64* stamps;                                           // diagnostic build (-DSAGE2OV_STAMPS): wave cycles per phase
};
#ifdef SAGE2OV_STAMPS
#define STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const u64 t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

template <int S, int MODE, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_probe(ProbeArgs A) {
    __shared__ WaveLds<S> lds_all[WPB];
    WaveLds<S>& W = lds_all[threadIdx.x >> 6];
    const u32 lane = lane_id();
    const u64 wave0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    const int k = A.k, h = A.h;
    if (lane <= 64 - 1) W.y[S][lane] = 0;

    const u64 first = A.ids ? 0 : A.lo, last = A.ids ? A.n_ids : A.hi;
    for (u64 it = first + wave0; it < last; it += nwaves) {
        const u64 i = A.ids ? (u64)A.ids[it] : it;
        if (MODE == 1) { if (A.status[i] != 0) continue; }
        // ---- stage the read and its reverse complement in LDS
        wave_sync();
        if (lane < S) W.x[0][lane] = A.reads[i * S + lane];
        if (lane == S) { W.x[0][S] = 0; W.x[1][S] = 0; }
        wave_sync();
        const int L1 = (int)(W.x[0][S - 1] & 0xFFFF);
        if (lane < S) {
            int rem = L1 - 32 * (int)lane; u64 r;
            if (rem <= 0) r = 0;
            else if (rem >= 32) r = ~rev2(bits64(W.x[0], S + 1, 2 * (rem - 32)));
            else r = (~rev2(W.x[0][0] >> (64 - 2 * rem))) & mask_top(rem);
            W.x[1][lane] = r;
        }
        wave_sync();

        // ---- extension state (uniform across the wave)
        u64 rightId = 0, leftId = 0; u32 rightO = 0, leftO = 0, rightLen = 0, leftLen = 0;
        u32 pRL = 0, pRov = 0, pLL = 0, pLov = 0; u64 pR[S], pL[S];
#pragma unroll
        for (int c = 0; c < S; c++) { pR[c] = 0; pL[c] = 0; }
        int curJ = -1; bool mAR = false, mAL = false, mFR = false, ambR = false, ambL = false;
        u32 connections = 0; u32 seq = 0;

        const int nwin = L1 - h + 1;
        for (int jb = 0; jb < nwin; jb += 64) {
            // ---- probe: one window per lane
            const int j = jb + (int)lane;
            u32 cnt = 0; u32 pay = 0; bool inl = false;
            if (j < nwin) {
                u64 khi, klo; key_at(W.x[0], S + 1, j, h, khi, klo);
                u64 s = table_find(A.slots, A.T, hash_key(khi, klo, h, A.seed));
                u32 c7 = (u32)(s >> SLOT_CNT_SHIFT) & 127u;
                if (s != 0 && c7 != SLOT_CNT_LONG) { cnt = c7; pay = (u32)(s & SLOT_PAY_MASK); inl = (c7 == 1); }
            }
            // ---- expand buckets into candidates, 64 at a time, in (window, bucket) order
            u32 rem = cnt, cur = 0;
            for (;;) {
                u32 incl = wave_incl_scan(rem); u32 total = __shfl(incl, 63);
                if (total == 0) break;
                u32 excl = incl - rem, take = 0;
                if (excl < 64 && rem) {
                    take = min(rem, 64u - excl);
                    for (u32 e = 0; e < take; e++) { W.candJ[excl + e] = (u32)j | (inl ? 0u : 0x80000000u); W.candE[excl + e] = inl ? pay : pay + cur + e; }
                }
                rem -= take; cur += take;
                const u32 nb = min(total, 64u);
                wave_sync();
                // ---- verify: one candidate per lane
                bool isHit = false; u32 hr2 = 0, hA = 0, hov = 0; u64 ovw[S];
#pragma unroll
                for (int c = 0; c < S; c++) ovw[c] = 0;
                int hlen = 0; u32 htype = 0;
                if (lane < nb) {
                    u32 cj = W.candJ[lane], ce = W.candE[lane];
                    u32 entry = (cj & 0x80000000u) ? A.csr[ce] : ce;
                    const int jj = (int)(cj & 0x7FFFFFFFu);
                    const u64 r2 = entry >> 2; const int t = entry & 3;
                    const bool rightSide = (t == 0 || t == 2);
                    bool gate = (r2 != i) && (rightSide ? (jj <= L1 - k) : (jj >= k - h));
                    if (MODE == 1 && gate) gate = (A.status[r2] == 0);
                    if (gate) {
                        const u64* yp = A.reads + r2 * S;
#pragma unroll
                        for (int c = 0; c < S; c++) W.y[c][lane] = yp[c];
                        const int L2 = (int)(W.y[S - 1][lane] & 0xFFFF);
                        // overlap geometry (see DESIGN.md "mirrored compares"): region length n on both reads
                        const int span = rightSide ? (L1 - jj) : (jj + h);      // bases of read i from the window to its end
                        const bool cont = (L2 <= span);                          // read 2 ends inside read i
                        const int n = cont ? L2 : span;
                        bool eq;
                        if (t == 0)      eq = eq_range<S>(W.x[0], jj, W.y, lane, 0, n);
                        else if (t == 3) eq = eq_range<S>(W.x[1], L1 - jj - h, W.y, lane, 0, n);
                        else if (t == 2) eq = eq_range<S>(W.x[1], L1 - jj - n, W.y, lane, L2 - n, n);
                        else             eq = eq_range<S>(W.x[0], jj + h - n, W.y, lane, L2 - n, n);
                        if (MODE == 0) {
                            if (eq && cont) atomicOr(&A.cflag[r2], i > r2 ? 1u : 2u);     // economyGraph.cpp:735
                            if (eq && !cont) {
                                isHit = true; hr2 = (u32)r2; hov = (u32)(L2 - n);
                                hA = (rightSide ? 0u : 1u) | ((u32)((t == 2 || t == 3) ? 1 : 0) << 1) | ((u32)L2 << 2) | ((u32)jj << 18);
                                const int ov = L2 - n;
                                if (t == 0 || t == 3) {   // overhang = tail of read 2
#pragma unroll
                                    for (int c = 0; c < S; c++) if (32 * c < ov) ovw[c] = ybits<S>(W.y, lane, 2 * n + 64 * c) & mask_top(ov - 32 * c);
                                } else {                  // overhang = reverse complement of the head of read 2
#pragma unroll
                                    for (int c = 0; c < S; c++) {
                                        int rm = ov - 32 * c;
                                        if (rm >= 32) ovw[c] = ~rev2(ybits<S>(W.y, lane, 2 * (rm - 32)));
                                        else if (rm > 0) ovw[c] = (~rev2(W.y[0][lane] >> (64 - 2 * rm))) & mask_top(rm);
                                    }
                                }
                            }
                        } else {
                            if (eq) {   // economyGraph.cpp:607-626: contained-and-equal counts as a hit here
                                isHit = true; hr2 = (u32)r2;
                                hlen = rightSide ? (L2 - (L1 - jj)) : (L2 - jj - h);
                                htype = t == 0 ? 3u : (t == 1 ? 0u : (t == 2 ? 2u : 1u));
                            }
                        }
                    }
                }
                const u64 hb = __ballot(isHit);
                const u32 nh = (u32)__popcll(hb);
                const u32 hidx = (u32)__popcll(hb & ((1ull << lane) - 1ull));
                if (MODE == 1) {
                    if (nh) {
                        u64 base = 0; if (lane == 0) base = atomicAdd(&A.counters[4], (u64)nh); base = __shfl(base, 0);
                        if (isHit && base + hidx < A.hits_cap) {
                            Hit hh; hh.from = (u32)i; hh.to = hr2; hh.len = hlen; hh.seq_hi = 0; hh.type = (uint8_t)htype; hh.pad = 0; hh.seq = seq + hidx;
                            A.hits[base + hidx] = hh;
                        }
                        seq += nh;
                    }
                    wave_sync();
                    continue;
                }
                if (isHit) {
                    W.hitR2[hidx] = hr2; W.hitA[hidx] = hA; W.hitB[hidx] = hov;
#pragma unroll
                    for (int c = 0; c < S; c++) W.hitOv[c][hidx] = ovw[c];
                }
                wave_sync();
                // ---- sequential extension state machine over the verified hits (economyGraph.cpp:95-438)
                for (u32 x = 0; x < nh; x++) {
                    const u32 a = W.hitA[x], ov = W.hitB[x]; const u64 r2 = W.hitR2[x];
                    const bool isLeft = a & 1; const u32 o = (a >> 1) & 1, L2 = (a >> 2) & 0xFFFF; const int jj = (int)(a >> 18);
                    u64 q[S];
#pragma unroll
                    for (int c = 0; c < S; c++) q[c] = W.hitOv[c][x];
                    if (jj != curJ) { curJ = jj; mAR = mAL = mFR = false; }
                    connections++;
                    if (!isLeft) {
                        if (rightId == 0) { rightId = r2; rightO = o; rightLen = ov; pRL = L2; pRov = ov; mAR = true; mFR = true;
#pragma unroll
                            for (int c = 0; c < S; c++) pR[c] = q[c];
                        } else {
                            const int m = (int)min(pRov, ov); bool cons = true;
#pragma unroll
                            for (int c = 0; c < S; c++) if (32 * c < m && ((pR[c] ^ q[c]) & mask_top(m - 32 * c))) cons = false;
                            if (cons) {
                                bool upd = false;
                                if (mAR) { if (L2 > pRL) { if (mFR) { rightId = r2; rightO = o; rightLen = ov; } upd = true; } }
                                else { upd = true; mAR = true; }
                                if (upd) { pRL = L2; pRov = ov;
#pragma unroll
                                    for (int c = 0; c < S; c++) pR[c] = q[c];
                                }
                            } else ambR = true;
                        }
                    } else {
                        if (leftId == 0) { leftId = r2; leftO = o; leftLen = ov; pLL = L2; pLov = ov; mAL = true;
#pragma unroll
                            for (int c = 0; c < S; c++) pL[c] = q[c];
                        } else {
                            const int m = (int)min(pLov, ov); bool cons = true;
#pragma unroll
                            for (int c = 0; c < S; c++) if (32 * c < m && ((pL[c] ^ q[c]) & mask_top(m - 32 * c))) cons = false;
                            if (cons) {
                                bool upd = false;
                                if (mAL) { if (L2 > pLL) upd = true; } else { upd = true; mAL = true; }
                                if (upd) { leftId = r2; leftO = o; leftLen = ov; pLL = L2; pLov = ov;
#pragma unroll
                                    for (int c = 0; c < S; c++) pL[c] = q[c];
                                }
                            } else ambL = true;
                        }
                    }
                }
                wave_sync();
            }
        }
        if (MODE == 1 && A.hitcount && lane == 0) A.hitcount[i] = seq;
        if (MODE == 0 && lane == 0) {
            if (ambR || ambL) { rightLen = 0; leftLen = 0; }                                        // economyGraph.cpp:446-450
            A.right[i] = rightId | ((u64)rightO << 40) | ((u64)(rightLen & 0x3FFFFFu) << 42);
            A.left[i] = leftId | ((u64)leftO << 40) | ((u64)(leftLen & 0x3FFFFFu) << 42);
            A.conn[i] = connections;
        }
    }
}


// =============================================================================================
// FAST probe + verify kernel (the hot kernel), all-32-bit arithmetic.
// Same results as k_probe<S,0> for every read whose verified hits are mutually consistent (every right
// overhang is a prefix of the longest one, same on the left): then the extension state machine of
// economyGraph.cpp:95-438 never raises an ambiguity flag and reduces to
//     right = in the FIRST window with a right hit, the hit with the largest L2 (first in bucket order on ties)
//     left  = in the LAST  window with a left  hit, the hit with the largest L2 (first in bucket order on ties)
//     connections = number of verified hits
// which needs no sequential pass.  Anything else (an inconsistent hit, more than 128 candidates, a failed
// speculation) sends the read to the sequential kernel.  One wavefront per read:
//   1. WPL windows per lane: key (funnel shifts out of LDS) -> lookup3 hash -> open-addressed probes, all
//      chains of a lane in flight together
//   2. bucket sizes -> DPP wave scan -> <= 128 candidates, two per lane, both 64-byte gathers in flight
//   3. speculation: the candidate reaching furthest right (left) is the longest-overhang hit if it verifies;
//      the read extended by its overhang is written to LDS in both orientations (lanes 0-31 right, 32-63 left)
//   4. every candidate compared ONCE, over its whole length, with that extended string: the first mismatch
//      position classifies it as hit+consistent / not a hit / hit but inconsistent
//   5. DPP wave reductions pick the extension records
// =============================================================================================
constexpr u64 MI_SCAN_PAD = 272;   // records behind krec[]: the scan of a small group runs up to the wave's largest group (< 255) rounded up to 16
constexpr int FAST_CAP = 128;
constexpr int TAIL_OVW = 8;        // dwords per stored overhang (128 bases); longer overhangs go to the sequential kernel
constexpr int FAST_CHUNK = 64;     // reads per block visit
template <int S, int NW, bool TAILED>
struct FastLds {                   // every string has one zero dword in front (index 0) so that bit positions down to -32 are readable
    u32 xf[2][1 + 2 * S + 2];      // forward, reverse complement as big-endian dwords (+ zero pad behind)
    u32 e[4][1 + 6 * S + 2];       // XR0 = fwd ++ right overhang, XR1 = rc(XR0), XL0 = rc ++ left overhang, XL1 = rc(XL0)
    u32 m[2][1 + 2 * S + 2];       // the two speculated longest-reach reads
    u32 candJ[FAST_CAP], candSrc[FAST_CAP];   // (after the entries are resolved the two arrays hold the packed geometry / read id of the verified hits)
    // inconsistent reads only (the in-kernel state machine): one padded row per lane to cut overhangs out of a candidate, and the
    // overhang of every verified hit (<= 128 bases) in this read's orientation
    uint8_t tslotIdx[TAILED ? FAST_CAP : 4];          // slot number of the x-th verified hit (visiting order)
    u32 tslot[TAILED ? FAST_CAP : 1][TAILED ? NW + 3 : 1];   // per candidate slot: the candidate's dwords (zero dword in front, two behind), then its overhang in place
};
__device__ __forceinline__ u32 rev2_32(u32 x) { x = __brev(x); return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); }
__device__ __forceinline__ u32 mask_top32(int nb) { return nb >= 16 ? ~0u : (nb <= 0 ? 0u : (~0u << (32 - 2 * nb))); }
__device__ __forceinline__ u32 range_mask32(int lo, int hi) { return mask_top32(hi) & ~mask_top32(lo); }
__device__ __forceinline__ u32 funnel32(u32 a, u32 b, int r) { return (u32)(((((u64)a) << 32) | b) >> (32 - r)); }   // r in [0,31]
// 32 bits at bit position p of a big-endian dword string whose dword 0 sits at D[0]; p >= -32 when D[-1] is the zero pad
// One v_alignbit_b32 ({hi,lo} >> s, s in [0,31]): with q = (p-1)>>5 and s = (-p)&31 the aligned case (p%32 == 0) reads
// the wanted dword as `lo` with s = 0, every other case is the usual funnel.  (D[q] is touched but unused when aligned.)
__device__ __forceinline__ u32 get32(const u32* D, int p) { const int q = (p - 1) >> 5; return __builtin_amdgcn_alignbit(D[q], D[q + 1], (u32)(-p) & 31u); }
// The masked overlap compares use get32(X, max(p, -32)): wherever the mask of a dword is non-zero its window starts at p >= -32 (the
// overlap region maps to non-negative positions of this read and a dword reaches at most 16 bases in front of it), and it ends
// inside the padded string (2*(16*9 + L1 - h) + 32 <= 32*(D+2)); fully masked dwords may read anything.
// same with bounds: anything outside [0, 32*n) reads as zero, p may be any negative number (slow, rare paths only)
__device__ __forceinline__ u32 get32z(const u32* D, int n, int p) {
    const int pp = p < 0 ? 0 : p, sh = pp - p, q = pp >> 5;
    const int q0 = q < n ? q : n - 1, q1 = q + 1 < n ? q + 1 : n - 1;
    u32 a = D[q0], b = D[q1]; a = q < n ? a : 0u; b = q + 1 < n ? b : 0u;
    const u32 v = funnel32(a, b, pp & 31);
    return sh >= 32 ? 0u : (v >> sh);
}
// ---- DPP wave primitives (gfx9 row_shr / row_bcast)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ u32 dpp_mov(u32 old, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWMASK, 0xF, false); }
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 v) {
    v += dpp_mov<0x111, 0xF>(0, v); v += dpp_mov<0x112, 0xF>(0, v); v += dpp_mov<0x114, 0xF>(0, v); v += dpp_mov<0x118, 0xF>(0, v);
    v += dpp_mov<0x142, 0xA>(0, v); v += dpp_mov<0x143, 0xC>(0, v);
    return v;
}
__device__ __forceinline__ u32 wave_min_dpp(u32 v) {                 // result valid in lane 63, returned broadcast
    v = min(v, dpp_mov<0x111, 0xF>(~0u, v)); v = min(v, dpp_mov<0x112, 0xF>(~0u, v)); v = min(v, dpp_mov<0x114, 0xF>(~0u, v)); v = min(v, dpp_mov<0x118, 0xF>(~0u, v));
    v = min(v, dpp_mov<0x142, 0xA>(~0u, v)); v = min(v, dpp_mov<0x143, 0xC>(~0u, v));
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// OR of all differences between Y[0..L2) and E at base offset d (>= 0).  cl / tailMask describe L2: dword cl is the
// partial one, tailMask its valid bits.  When they are wave-uniform the masks cost nothing per dword.
template <int NW>
__device__ __forceinline__ u32 any_mismatch(const u32 (&Y)[NW], const u32* E, int d, int cl, u32 tailMask) {
    const int q = (2 * d - 1) >> 5; const u32 sh = (u32)(-2 * d) & 31u;       // see get32
    u32 acc = 0, nxt = E[q + NW];
#pragma unroll
    for (int c = NW - 1; c >= 0; c--) {
        const u32 cur = E[q + c];
        const u32 diff = Y[c] ^ __builtin_amdgcn_alignbit(cur, nxt, sh);
        acc |= (c < cl) ? diff : (c == cl ? (diff & tailMask) : 0u);
        nxt = cur;
    }
    return acc;
}

#ifndef SAGE2OV_FAST_WPB
#define SAGE2OV_FAST_WPB 8
#endif
// HITS = 1: the same look-up and gather machinery emits the directional hit lists of the status-0 reads for the reduce phase
// (economyGraph.cpp:591-633) instead of extension records: every candidate is compared directly with this read.
constexpr u32 HITS_CHUNK = 2048;   // hit slots a wave reserves at a time (one atomic per ~30 reads instead of one per read)
template <int S, int NW, int WPL, int WPB, int HITS>
#ifndef SAGE2OV_FAST_WAVES
#define SAGE2OV_FAST_WAVES 4
#endif
__global__ __launch_bounds__(64 * WPB, SAGE2OV_FAST_WAVES) void k_probe_fast(ProbeArgs A) {
    // the in-kernel state machine for inconsistent reads needs 6.7 KB of LDS per wave at NW = 10; the long-read layouts keep their
    // occupancy instead and hand such reads to the sequential kernel
    constexpr bool TAILED = (HITS == 0) && (NW <= 10);
    __shared__ FastLds<S, NW, TAILED> lds_all[WPB];
    FastLds<S, NW, TAILED>& L = lds_all[threadIdx.x >> 6];
    const u32 lane = lane_id();
    const int k = A.k, h = A.h;
    constexpr int D = 2 * S;                       // dwords per read slot
    u32* const X0 = L.xf[0] + 1; u32* const X1 = L.xf[1] + 1;
    const u32* reads32 = (const u32*)A.reads;
    const int nk = (2 * h + 31) >> 5;              // key dwords
    const bool four = nk > 3;
    const u32 lastKeyMask = (2 * h) & 31 ? (~0u << (32 - ((2 * h) & 31))) : ~0u;
    const u32 Th = (u32)(A.T >> 1), seed32 = (u32)A.seed;      // T/2 slot pairs
    const uint4* pairs = (const uint4*)A.slots;

    // a block owns FAST_CHUNK consecutive positions of the (locality ordered) id list at a time, so reads that share
    // keys and neighbours run on one CU, back to back
    const u64 nItems = A.ids ? A.n_ids : (A.hi > A.lo ? A.hi - A.lo : 0);
    const u32 wib = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform: loop bookkeeping stays on the scalar unit
#ifdef SAGE2OV_STAMPS
    u64 st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; u64 st_prev = __builtin_amdgcn_s_memtime();
#endif
    if (lane == 0) { L.xf[0][0] = 0; L.xf[1][0] = 0; L.m[0][0] = 0; L.m[1][0] = 0; L.e[0][0] = 0; L.e[1][0] = 0; L.e[2][0] = 0; L.e[3][0] = 0; }
    // the wave's n-th item: chunk blockIdx.x + (n / PER) * gridDim.x, position (n % PER) * WPB + wave.  Two-deep software pipeline:
    // the id of item n+2 and the bases of item n+1 are fetched while item n is processed (both chains off the critical path)
    constexpr u32 PER = FAST_CHUNK / WPB;
    auto id_of = [&](u32 n) -> u32 {
        const u64 chunk = blockIdx.x + (u64)(n / PER) * gridDim.x, it = chunk * FAST_CHUNK + (n % PER) * WPB + wib;
        const bool ok = it < nItems;
        const u64 itc = ok ? it : 0;
        const u32 v = A.ids ? A.ids[itc] : (u32)(A.lo + itc);
        return ok ? v : 0u;                                    // 0: nothing to do (slot 0 of the read store is all zero)
    };
    u64 hBase = 0; u32 hLeft = 0; u64 hTotal = 0;              // HITS: this wave's current chunk of the hit buffer
    const u32 ldw = (lane < (u32)D ? lane : 0u) ^ 1u;
    u32 idCur = id_of(0), idNext = id_of(1);
    u32 wNext = reads32[(u64)idCur * D + ldw];
    for (u32 n = 0; (blockIdx.x + (u64)(n / PER) * gridDim.x) * FAST_CHUNK < nItems; n++) {
        const u64 i = idCur; const u32 wCur = wNext;
        idCur = idNext; idNext = id_of(n + 2);
        wNext = reads32[(u64)idCur * D + ldw];
        if (i == 0) continue;
        if (HITS) { if (A.status[i] != 0) continue; }
        STAMP(0);
        // ---------------------------------------------------------------- stage the read (big-endian dwords) + its reverse complement
        wave_sync();
        if (lane < D) X0[lane] = wCur;
        else if (lane < D + 2) { X0[lane] = 0; X1[lane] = 0; }
        wave_sync();
        const int L1 = (int)(X0[D - 1] & 0xFFFFu);
        if (lane == 0) X0[D - 1] &= 0xFFFF0000u;
        wave_sync();
        if (lane < D) { const int rem = L1 - 16 * (int)lane; X1[lane] = rem > 0 ? (~rev2_32(get32(X0, 2 * (rem - 16))) & mask_top32(rem)) : 0u; }
        wave_sync();
        const int nwin = L1 - h + 1;
        bool slowpath = nwin > 64 * WPL;

        STAMP(1);
        // ---------------------------------------------------------------- 1. lookups
        int jj[WPL]; u32 pidx[WPL], tg[WPL]; u64 sl[WPL]; bool pend[WPL];
#pragma unroll
        for (int q = 0; q < WPL; q++) {
            jj[q] = 64 * q + (int)lane; pend[q] = jj[q] < nwin; sl[q] = 0;
            const int p = 2 * (pend[q] ? jj[q] : 0);
            u32 k0 = get32(X0, p), k1 = nk > 1 ? get32(X0, p + 32) : 0u, k2 = nk > 2 ? get32(X0, p + 64) : 0u, k3 = nk > 3 ? get32(X0, p + 96) : 0u;
            if (nk == 1) k0 &= lastKeyMask; else if (nk == 2) k1 &= lastKeyMask; else if (nk == 3) k2 &= lastKeyMask; else k3 &= lastKeyMask;
            const u64 hv = hash4(k0, k1, k2, k3, seed32, four);
            pidx[q] = __umulhi((u32)(hv >> 32), Th); tg[q] = tag_of(hv);
        }
        STAMP(2);
        if (A.mi1) {
            // 1a. minimiser of every window: hash all w-mers once (position 64c+lane in register c), then a sliding minimum of
            //     width m = h-w+1 by doubling across lanes (ds_bpermute; no LDS storage, no barriers)
            u32 winMin[WPL];
            {
                const int w = h < 16 ? h : 16, m = h - w + 1, npos = L1 - w + 1;
                constexpr int NC = (16 * NW - 15 + 63) / 64;                      // 64-position chunks that can hold a w-mer start
                u32 a[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) { const int p0 = 64 * c + (int)lane; a[c] = p0 < npos ? wmer_hash(get32(X0, 2 * p0) >> (32 - 2 * w)) : ~0u; }
                int width = 1;
                for (; 2 * width <= m; width *= 2) {                              // a[c][l] = min over positions [p, p + 2*width)
                    const u32 src = (lane + (u32)width) & 63u; const bool wrap = lane + (u32)width >= 64u;
                    u32 t[NC];
#pragma unroll
                    for (int c = 0; c < NC; c++) t[c] = (u32)__shfl((int)a[c], (int)src);
#pragma unroll
                    for (int c = 0; c < NC; c++) a[c] = min(a[c], wrap ? (c + 1 < NC ? t[c + 1] : ~0u) : t[c]);
                }
                if (m > width) {                                                  // [p, p+m) = [p, p+width) U [p+m-width, p+m)
                    const u32 sh = (u32)(m - width), src = (lane + sh) & 63u; const bool wrap = lane + sh >= 64u;
                    u32 t[NC];
#pragma unroll
                    for (int c = 0; c < NC; c++) t[c] = (u32)__shfl((int)a[c], (int)src);
#pragma unroll
                    for (int q = 0; q < WPL; q++) winMin[q] = q < NC ? min(a[q < NC ? q : 0], wrap ? (q + 1 < NC ? t[q + 1 < NC ? q + 1 : 0] : ~0u) : t[q < NC ? q : 0]) : ~0u;   // (windows past the last chunk do not exist)
                } else {
#pragma unroll
                    for (int q = 0; q < WPL; q++) winMin[q] = q < NC ? a[q < NC ? q : 0] : ~0u;
                }
            }
            STAMP(3);
            // 1b. group of the minimiser (lanes that share a minimiser read the same words), then a scan of the group
            u32 goff[WPL], gn[WPL]; const u32 TL32 = (u32)A.TL;
            {
                u32 gi[WPL], mt[WPL]; bool gp[WPL];
#pragma unroll
                for (int q = 0; q < WPL; q++) { const u32 mh = minim_hash(winMin[q], seed32); gi[q] = __umulhi(mh, TL32); mt[q] = minim_tag(mh); gp[q] = pend[q]; goff[q] = 0; gn[q] = 0; }
                for (;;) {
                    bool any = false;
#pragma unroll
                    for (int q = 0; q < WPL; q++) any |= gp[q];
                    if (!__any(any)) break;
                    u64 gv[WPL];
#pragma unroll
                    for (int q = 0; q < WPL; q++) gv[q] = A.mi1[gi[q]];
#pragma unroll
                    for (int q = 0; q < WPL; q++) {
                        if (gp[q]) {
                            if (gv[q] == 0) { gp[q] = false; pend[q] = false; }                              // no key has this minimiser: a miss
                            else if ((u32)(gv[q] >> SLOT_TAG_SHIFT) == mt[q]) { gp[q] = false; gn[q] = (u32)(gv[q] >> 32) & 255u; goff[q] = (u32)gv[q]; }
                            else if (++gi[q] == TL32) gi[q] = 0;
                        }
                    }
                }
            }
            STAMP(4);
            u32 nmax = 0;
#pragma unroll
            for (int q = 0; q < WPL; q++) { if (gn[q] == MI_BIG) gn[q] = 0; else if (pend[q]) { pend[q] = false; nmax = max(nmax, gn[q]); } else gn[q] = 0; }
            // (lanes of an oversized / ambiguous group keep pend = true and fall through to the uniform table below)
            u32 wmax = nmax;
            wmax = max(wmax, dpp_mov<0x111, 0xF>(0, wmax)); wmax = max(wmax, dpp_mov<0x112, 0xF>(0, wmax)); wmax = max(wmax, dpp_mov<0x114, 0xF>(0, wmax)); wmax = max(wmax, dpp_mov<0x118, 0xF>(0, wmax));
            wmax = max(wmax, dpp_mov<0x142, 0xA>(0, wmax)); wmax = max(wmax, dpp_mov<0x143, 0xC>(0, wmax));
            wmax = (u32)__builtin_amdgcn_readlane((int)wmax, 63);
            // The scan reads only the high dword of a record (tag | count | top payload bit), 16 records per window in flight
            // off one base pointer (immediate offsets, no address arithmetic).  It does not stop at the end of the group: a record
            // of a neighbouring group that happens to carry the tag (2^-24) only adds candidates that fail verification, like
            // any merged tag, or makes the window ambiguous (-> sequential kernel).  krec[] is padded by 16 records for this.
            u32 nmatch[WPL], mpos[WPL];
#pragma unroll
            for (int q = 0; q < WPL; q++) { nmatch[q] = 0; mpos[q] = 0; }
            {
                const u32* kp[WPL]; u32 tgs[WPL];
#pragma unroll
                for (int q = 0; q < WPL; q++) { kp[q] = (const u32*)(A.krec + goff[q]) + 1; tgs[q] = tg[q] << (SLOT_TAG_SHIFT - 32); }
                for (u32 x = 0; x < wmax; x += 16) {
                    u32 r[WPL][16];
#pragma unroll
                    for (int q = 0; q < WPL; q++)
#pragma unroll
                        for (int u = 0; u < 16; u++) r[q][u] = kp[q][2 * u];
#pragma unroll
                    for (int q = 0; q < WPL; q++) {
#pragma unroll
                        for (int u = 15; u >= 0; u--) { const bool hit = (r[q][u] ^ tgs[q]) < (1u << (SLOT_TAG_SHIFT - 32)); mpos[q] = hit ? x + (u32)u : mpos[q]; nmatch[q] += hit ? 1u : 0u; }
                        kp[q] += 32;
                    }
                }
            }
            {
                u64 rec[WPL];
#pragma unroll
                for (int q = 0; q < WPL; q++) rec[q] = A.krec[goff[q] + mpos[q]];
#pragma unroll
                for (int q = 0; q < WPL; q++) if (nmatch[q] && gn[q]) sl[q] = rec[q];
            }
            // two records of one group with the same tag (two keys, ~2^-24 per pair): the scan cannot tell them apart
            bool amb = false;
#pragma unroll
            for (int q = 0; q < WPL; q++) amb |= nmatch[q] > 1;
            if (__any(amb)) slowpath = true;
        }
        // 1c. uniform table (everything when there is no minimiser index; otherwise only windows of oversized groups)
        for (;;) {
            bool any = false;
#pragma unroll
            for (int q = 0; q < WPL; q++) any |= pend[q];
            if (!__any(any)) break;
            uint4 pv[WPL];
#pragma unroll
            for (int q = 0; q < WPL; q++) pv[q] = pairs[pidx[q]];               // unconditional: idle lanes re-read their last pair
#pragma unroll
            for (int q = 0; q < WPL; q++) {
                if (pend[q]) {
                    const u64 s0 = ((u64)pv[q].y << 32) | pv[q].x, s1 = ((u64)pv[q].w << 32) | pv[q].z;
                    if (s0 == 0 || (pv[q].y >> (SLOT_TAG_SHIFT - 32)) == tg[q]) { sl[q] = s0; pend[q] = false; }
                    else if (s1 == 0 || (pv[q].w >> (SLOT_TAG_SHIFT - 32)) == tg[q]) { sl[q] = s1; pend[q] = false; }
                    else if (++pidx[q] == Th) pidx[q] = 0;
                }
            }
        }
        STAMP(5);
        // ---------------------------------------------------------------- 2. candidates
        u32 cnt[WPL], pay[WPL], mine = 0;
#pragma unroll
        for (int q = 0; q < WPL; q++) {
            const u32 c7 = (u32)(sl[q] >> SLOT_CNT_SHIFT) & 127u;
            cnt[q] = (sl[q] != 0 && c7 != SLOT_CNT_LONG) ? c7 : 0u; pay[q] = (u32)(sl[q] & SLOT_PAY_MASK); mine += cnt[q];
        }
        // candidate slots in window order (j ascending, bucket order inside a window = the order the reference visits them):
        // window q of lane l is j = 64q + l, so all q = 0 windows come first; two 16-bit prefix sums ride in one DPP scan
        u32 cbase[WPL]; u32 total = 0;
#pragma unroll
        for (int q0 = 0; q0 < WPL; q0 += 2) {
            const u32 pk = cnt[q0] | ((q0 + 1 < WPL ? cnt[q0 + 1 < WPL ? q0 + 1 : q0] : 0u) << 16);
            const u32 inc = wave_incl_scan_dpp(pk);
            const u32 tot = (u32)__builtin_amdgcn_readlane((int)inc, 63);
            cbase[q0] = total + (inc & 0xFFFFu) - cnt[q0]; total += tot & 0xFFFFu;
            if (q0 + 1 < WPL) { cbase[q0 + 1 < WPL ? q0 + 1 : q0] = total + (inc >> 16) - cnt[q0 + 1 < WPL ? q0 + 1 : q0]; total += tot >> 16; }
        }
        (void)mine;
        if (total > (u32)FAST_CAP) slowpath = true;
        u32 nhits = 0; u32 selR = ~0u, selL = ~0u;
        u32 myEnt[2] = {0, 0}; int myJ[2] = {0, 0}, myL2[2] = {0, 0};
        if (!slowpath) {
#pragma unroll
            for (int q = 0; q < WPL; q++)
                for (u32 e = 0; e < cnt[q]; e++) { L.candJ[cbase[q] + e] = (u32)jj[q] | (cnt[q] == 1 ? 0u : 0x80000000u); L.candSrc[cbase[q] + e] = cnt[q] == 1 ? pay[q] : pay[q] + e; }
            wave_sync();
            // ---- gather: entry (CSR for multi-entry buckets), then the 64-byte read slot, two candidates per lane
            u32 Y[2][NW]; bool gate[2];
            const bool two = total > 64u;                // the second slot of every lane is empty otherwise: skip its work (wave-uniform)
            gate[1] = false;
#pragma unroll
            for (int c = 0; c < NW; c++) Y[1][c] = 0;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (q == 1 && !two) continue;
                const u32 ci = lane + 64 * q; const bool have = ci < total;
                const u32 cjj = L.candJ[have ? ci : 0], src = L.candSrc[have ? ci : 0];
                const bool isCsr = (cjj & 0x80000000u) != 0;
                const u32 ce = A.csr[(have && isCsr) ? src : 0u];
                myEnt[q] = isCsr ? ce : src; myJ[q] = (int)(cjj & 0x7FFFFFFFu);
                const u32 r2 = myEnt[q] >> 2; const int t = myEnt[q] & 3;
                gate[q] = have && (r2 != (u32)i) && ((t == 0 || t == 2) ? (myJ[q] <= L1 - k) : (myJ[q] >= k - h));
                if (HITS) { const uint8_t st2 = A.status[gate[q] ? r2 : 0u]; gate[q] = gate[q] && st2 == 0; }     // economyGraph.cpp:605 (status[0] is never 0)
#ifdef SAGE2OV_TRAFFIC_PROBE      // diagnostic build: no candidate read is fetched (results are meaningless; the PMC traffic of the rest is what is measured)
                gate[q] = false;
#endif
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (q == 1 && !two) continue;
                // 16-byte loads of the (at least 32-byte aligned) slot; gated-off lanes read slot 0 (zeros, always cached)
                const uint4* yp = (const uint4*)(A.reads + (gate[q] ? (u64)(myEnt[q] >> 2) : 0ull) * S);
#pragma unroll
                for (int c = 0; c < (NW + 3) / 4; c++) {
                    const uint4 v = yp[c];                       // memory dwords 4c..4c+3 = (lo,hi) of words 2c, 2c+1
                    if (4 * c + 0 < NW) Y[q][4 * c + 0] = v.y;
                    if (4 * c + 1 < NW) Y[q][4 * c + 1] = v.x;
                    if (4 * c + 2 < NW) Y[q][4 * c + 2] = v.w;
                    if (4 * c + 3 < NW) Y[q][4 * c + 3] = v.z;
                }
                myL2[q] = (int)(((const u32*)yp)[2 * S - 2] & 0xFFFFu);     // low dword of the last word
                if (NW == D) Y[q][NW - 1] &= 0xFFFF0000u;
            }
            STAMP(6);
            if constexpr (HITS != 0) {
                // ---------------------------------------------------------------- hit lists: direct masked compare of the overlap region
                bool hit[2] = {false, false}; int hlen[2] = {0, 0}; u32 htype[2] = {0, 0};
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if (q == 1 && !two) continue;
                    if (gate[q]) {
                        const int t = myEnt[q] & 3, j = myJ[q], L2 = myL2[q];
                        const bool rightSide = (t == 0 || t == 2);
                        const int span = rightSide ? (L1 - j) : (j + h);
                        const int n = L2 <= span ? L2 : span;
                        const int off = L1 - j - h;
                        const u32* X = (t == 0 || t == 1) ? X0 : X1;
                        int dd, lo, hi;
                        if (t == 0) { dd = j; lo = 0; hi = n; } else if (t == 3) { dd = off; lo = 0; hi = n; }
                        else if (t == 2) { dd = L1 - j - L2; lo = L2 - n; hi = L2; } else { dd = j + h - L2; lo = L2 - n; hi = L2; }
                        u32 diff = 0;
#pragma unroll
                        for (int c = 0; c < NW; c++) diff |= (Y[q][c] ^ get32(X, min(max(2 * (16 * c + dd), -32), 32 * D))) & range_mask32(lo - 16 * c, hi - 16 * c);
                        hit[q] = diff == 0;                                         // :607-626 (contained-and-equal counts here)
                        hlen[q] = rightSide ? (L2 - (L1 - j)) : (L2 - j - h);
                        htype[q] = t == 0 ? 3u : (t == 1 ? 0u : (t == 2 ? 2u : 1u));
                    }
                }
                const u64 b0 = __ballot(hit[0]), b1 = __ballot(hit[1]);
                const u32 n0 = (u32)__popcll(b0), nh = n0 + (u32)__popcll(b1);
                if (nh) {
                    if (nh > hLeft) {                                               // next chunk; the rest of the old one is marked empty
                        for (u32 x = lane; x < hLeft; x += 64) A.hits[hBase + x].from = 0;
                        u64 nb_ = 0; if (lane == 0) nb_ = atomicAdd(&A.counters[4], (u64)HITS_CHUNK);
                        hBase = ((u64)__builtin_amdgcn_readfirstlane((int)(nb_ >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)nb_);
                        hLeft = hBase + HITS_CHUNK <= A.hits_cap ? HITS_CHUNK : 0u;     // over capacity: dropped, the host retries with a larger buffer
                    }
                    if (nh <= hLeft) {
                        const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            if (hit[q]) {
                                const u32 sq = q == 0 ? (u32)__popcll(b0 & lt) : n0 + (u32)__popcll(b1 & lt);
                                Hit hh; hh.from = (u32)i; hh.to = myEnt[q] >> 2; hh.len = hlen[q]; hh.seq_hi = 0; hh.type = (uint8_t)htype[q]; hh.pad = 0; hh.seq = sq;
                                A.hits[hBase + sq] = hh;
                            }
                        }
                        hBase += nh; hLeft -= nh;
                    }
                    hTotal += nh;
                }
                if (lane == 0) A.hitcount[i] = nh;
                continue;
            }
            // ---------------------------------------------------------------- 3. speculation: furthest reach per side
            u32 reachR = ~0u, reachL = ~0u; bool sameLen = true;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (q == 1 && !two) continue;
                if (gate[q]) {
                    const int t = myEnt[q] & 3; const u32 ci = lane + 64 * q;
                    if (t == 0 || t == 2) reachR = min(reachR, ((u32)(0x7FF - (myJ[q] + myL2[q])) << 7) | ci);
                    else reachL = min(reachL, ((u32)(0x7FF - (L1 - myJ[q] - h + myL2[q])) << 7) | ci);
                    sameLen &= (myL2[q] == L1);
                }
            }
            reachR = wave_min_dpp(reachR); reachL = wave_min_dpp(reachL);
            const bool uniformLen = __all(sameLen);
            // (t, j, L2) of the two speculated reads, broadcast with readlane (EXEC independent); they publish their bases
            u32 mEnt[2] = {0, 0}; int mJ[2] = {0, 0}, mL2[2] = {0, 0};
#pragma unroll
            for (int sd = 0; sd < 2; sd++) {
                const u32 rk = sd ? reachL : reachR;
                if (rk != ~0u) {
                    const int ol = (int)(rk & 63u); const bool oq = ((rk >> 6) & 1u) != 0;
                    mEnt[sd] = (u32)__builtin_amdgcn_readlane((int)(oq ? myEnt[1] : myEnt[0]), ol);
                    mJ[sd] = __builtin_amdgcn_readlane(oq ? myJ[1] : myJ[0], ol);
                    mL2[sd] = __builtin_amdgcn_readlane(oq ? myL2[1] : myL2[0], ol);
                    if ((int)lane == ol) {
                        u32* M = L.m[sd] + 1;
#pragma unroll
                        for (int c = 0; c < NW; c++) M[c] = oq ? Y[1][c] : Y[0][c];
#pragma unroll
                        for (int c = NW; c < D + 2; c++) M[c] = 0;
                    }
                }
            }
            const int posR = mJ[0], posL = L1 - mJ[1] - h;                                  // offsets in the side's own coordinates
            const int LR = reachR != ~0u ? posR + mL2[0] : 0, LL = reachL != ~0u ? posL + mL2[1] : 0;
            // a speculated read that ends inside this read (containment) or that does not reach its end is left to the sequential kernel
            bool tail = false;                      // exact state machine over the verified hits, inside this kernel (see the tail below)
            if ((reachR != ~0u && LR <= L1) || (reachL != ~0u && LL <= L1)) tail = true;
            wave_sync();
            if (!tail) {
                // ---- extended strings: lanes 0-31 build the right side, lanes 32-63 the left side, one dword per lane
                {
                    const int side = lane >> 5, c = (int)(lane & 31);
                    const u32 rk = side ? reachL : reachR;
                    bool mbad = false;
                    if (rk != ~0u) {
                        const int t = (side ? mEnt[1] : mEnt[0]) & 3, L2M = side ? mL2[1] : mL2[0], posM = side ? posL : posR, tot = posM + L2M;
                        const bool straight = side ? (t == 3) : (t == 0);
                        const u32* own = side ? X1 : X0; const u32* oth = side ? X0 : X1; const u32* M = L.m[side] + 1;
                        u32* dstA = L.e[2 * side + (straight ? 0 : 1)] + 1; u32* dstB = L.e[2 * side + (straight ? 1 : 0)] + 1;
                        // straight: own[0,posM) ++ M ; mirrored: M ++ other strand [L1-posM, L1).  B is shifted right by sB >= 0 bases.
                        const u32* Aa = straight ? own : M; const int lenA = straight ? posM : L2M;
                        const u32* Bb = straight ? M : oth; const int sB = straight ? posM : (tot - L1);
                        const int cb = 16 * c;
                        const int pB = 2 * (cb - sB);                                       // bit position in B; >= -32 is readable (front pad)
                        const u32 va = Aa[c];
                        const u32 vb = pB > -32 ? get32(Bb, pB) : 0u;
                        const u32 mA = mask_top32(lenA - cb);
                        dstA[c] = ((va & mA) | (vb & ~mA)) & mask_top32(tot - cb);
                        wave_sync();
                        const int rem = tot - cb;
                        dstB[c] = rem > 0 ? (~rev2_32(get32(dstA, 2 * (rem - 16))) & mask_top32(rem)) : 0u;
                        wave_sync();
                        // the speculated read must really overlap: the straight string equals the own strand on [posM, L1)
                        const u32* Es = L.e[2 * side] + 1;
                        mbad = ((Es[c] ^ own[c]) & range_mask32(posM - cb, L1 - cb)) != 0 && c < D + 2;
                    } else { wave_sync(); wave_sync(); }
                    if (__any(mbad)) tail = true;
                }
                wave_sync();
            }
            STAMP(7);
            if (!tail) {
                // ---------------------------------------------------------------- 4. one whole-length compare per candidate
                bool bad = false;
                const int clU = L1 >> 4; const u32 tailU = mask_top32(L1 & 15);
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if (q == 1 && !two) continue;
                    bool hit = false;
                    if (gate[q]) {
                        const int t = myEnt[q] & 3, j = myJ[q], L2 = myL2[q];
                        const bool rightSide = (t == 0 || t == 2);
                        const int span = rightSide ? (L1 - j) : (j + h);
                        const bool cont = L2 <= span; const int n = cont ? L2 : span;
                        const int off = L1 - j - h;
                        const u32* E = L.e[t == 0 ? 0 : (t == 2 ? 1 : (t == 3 ? 2 : 3))] + 1;
                        const int d = t == 0 ? j : (t == 2 ? LR - j - L2 : (t == 3 ? off : LL - off - L2));
                        const bool usable = !cont && d >= 0;
                        u32 acc = 1;
                        if (usable) acc = uniformLen ? any_mismatch<NW>(Y[q], E, d, clU, tailU) : any_mismatch<NW>(Y[q], E, d, L2 >> 4, mask_top32(L2 & 15));
                        if (acc == 0) hit = true;                                                  // verified and consistent
                        else {
                            // rare: decide between "not an overlap" and "overlap, but inconsistent / contained"
                            const u32* X = (t == 0 || t == 1) ? X0 : X1;
                            int dd, lo, hi;
                            if (t == 0) { dd = j; lo = 0; hi = n; } else if (t == 3) { dd = off; lo = 0; hi = n; }
                            else if (t == 2) { dd = L1 - j - L2; lo = L2 - n; hi = L2; } else { dd = j + h - L2; lo = L2 - n; hi = L2; }
                            u32 diff = 0;
#pragma unroll
                            for (int c = 0; c < NW; c++) diff |= (Y[q][c] ^ get32(X, min(max(2 * (16 * c + dd), -32), 32 * D))) & range_mask32(lo - 16 * c, hi - 16 * c);
                            if (diff == 0) {
                                if (cont) atomicOr(&A.cflag[myEnt[q] >> 2], i > (u64)(myEnt[q] >> 2) ? 1u : 2u);   // economyGraph.cpp:735
                                else bad = true;                                                    // a true overlap that disagrees with the longest one
                            }
                        }
                    }
                    const u64 hb = __ballot(hit);
                    nhits += (u32)__popcll(hb);
                    if (hit) {
                        const int t = myEnt[q] & 3; const u32 ci = lane + 64 * q;
                        if (t == 0 || t == 2) selR = min(selR, ((u32)myJ[q] << 17) | ((u32)(0x3FF - myL2[q]) << 7) | ci);
                        else selL = min(selL, ((u32)(0x1FF - myJ[q]) << 17) | ((u32)(0x3FF - myL2[q]) << 7) | ci);
                    }
                }
                if (__any(bad)) tail = true;
            }
            if (tail && !TAILED) slowpath = true;
            if constexpr (TAILED) if (tail) {
                nhits = 0;
                // ------------------------------------------------------------ tail: some verified hit disagrees with the longest one (read errors,
                // repeats) or the speculation did not hold.  The reference's state machine (economyGraph.cpp:95-438) is then evaluated as it is
                // written, hit by hit in window / bucket order, but on data this wave already holds: every candidate is compared directly with
                // this read, the overhang of every hit goes to LDS, and one step of the machine is a dword-parallel prefix compare of two overhangs.
                bool thit[2] = {false, false}; bool tooLong = false;
                uint8_t* const tailSlot = L.tslotIdx;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if (q == 1 && !two) continue;
                    const u32 ci = lane + 64 * q;
                    u32* row = L.tslot[ci] + 1;
                    L.tslot[ci][0] = 0;
#pragma unroll
                    for (int c = 0; c < NW; c++) row[c] = Y[q][c];
                    row[NW] = 0; row[NW + 1] = 0;
                    u32 meta = 0; u32 ovd[TAIL_OVW];
#pragma unroll
                    for (int c = 0; c < TAIL_OVW; c++) ovd[c] = 0;
                    if (gate[q]) {
                        const int t = myEnt[q] & 3, j = myJ[q], L2 = myL2[q];
                        const bool rightSide = (t == 0 || t == 2);
                        const int span = rightSide ? (L1 - j) : (j + h);
                        const bool cont = L2 <= span; const int n = cont ? L2 : span;
                        const int off = L1 - j - h;
                        const u32* X = (t == 0 || t == 1) ? X0 : X1;
                        int dd, lo, hi;
                        if (t == 0) { dd = j; lo = 0; hi = n; } else if (t == 3) { dd = off; lo = 0; hi = n; }
                        else if (t == 2) { dd = L1 - j - L2; lo = L2 - n; hi = L2; } else { dd = j + h - L2; lo = L2 - n; hi = L2; }
                        u32 diff = 0;
#pragma unroll
                        for (int c = 0; c < NW; c++) diff |= (Y[q][c] ^ get32(X, min(max(2 * (16 * c + dd), -32), 32 * D))) & range_mask32(lo - 16 * c, hi - 16 * c);
                        if (diff == 0) {
                            if (cont) atomicOr(&A.cflag[myEnt[q] >> 2], i > (u64)(myEnt[q] >> 2) ? 1u : 2u);   // economyGraph.cpp:735
                            else {
                                const int ov = L2 - n;                                                   // bases of read 2 beyond this read
                                thit[q] = true; tooLong |= ov > 16 * TAIL_OVW;
                                meta = 0x80000000u | (rightSide ? 0u : 1u) | ((u32)((t == 2 || t == 3) ? 1 : 0) << 1) | ((u32)L2 << 2) | ((u32)j << 11) | ((u32)(ov & 0xFF) << 20);
#pragma unroll
                                for (int c = 0; c < TAIL_OVW; c++) {
                                    const int rm = ov - 16 * c;
                                    if (rm > 0) {
                                        if (t == 0 || t == 3) ovd[c] = get32(row, 2 * (n + 16 * c)) & mask_top32(rm);                 // tail of read 2
                                        else ovd[c] = rm >= 16 ? ~rev2_32(get32(row, 2 * (rm - 16))) : ((~rev2_32(row[0] >> (32 - 2 * rm))) & mask_top32(rm));   // revcomp of its head
                                    }
                                }
                            }
                        }
                    }
                    // the overhang replaces the candidate's dwords in its own slot; geometry, read id and slot of the hits are compacted in
                    // visiting order (candidate slots are in window / bucket order, so a ballot prefix is the position)
#pragma unroll
                    for (int c = 0; c < TAIL_OVW; c++) L.tslot[ci][c] = ovd[c];
                    const u64 hb = __ballot(thit[q]);
                    if (thit[q]) {
                        const u32 hx = nhits + (u32)__popcll(hb & ((1ull << lane) - 1ull));
                        L.candJ[hx] = meta; L.candSrc[hx] = myEnt[q] >> 2; tailSlot[hx] = (uint8_t)ci;
                    }
                    nhits += (u32)__popcll(hb);
                }
                if (__any(tooLong)) slowpath = true;
                wave_sync();
                if (!slowpath) {
                    u32 rightId = 0, rightO = 0, leftId = 0, leftO = 0, rightLen = 0, leftLen = 0;
                    u32 pRL = 0, pLL = 0, pRov = 0, pLov = 0;
                    bool ambR = false, ambL = false, mAR = false, mAL = false, mFR = false; int curJ = -1;
                    // One step = compare the hit's overhang with the overhang of the side's current `prev` over their common length.  The
                    // prev overhangs stay in registers (lane c < 8 holds dword c of both sides), the next hit's geometry and overhang are
                    // fetched from LDS one and two steps ahead: the serial chain per hit is a ballot and a few uniform updates.
                    const u32 lc = lane < (u32)TAIL_OVW ? lane : 0u;
                    const u32 nh = nhits;
                    u32 pRw = 0, pLw = 0;                                           // this lane's dword of the right / left prev overhang
                    u32 a0 = L.candJ[0], r20 = L.candSrc[0], a1 = L.candJ[nh > 1 ? 1 : 0], r21 = L.candSrc[nh > 1 ? 1 : 0];
                    u32 s1 = tailSlot[nh > 1 ? 1 : 0];
                    u32 cw0 = L.tslot[tailSlot[0]][lc];
                    for (u32 x = 0; x < nh; x++) {
                        const u32 xn = x + 2 < nh ? x + 2 : x;                              // two ahead: geometry, id, slot; one ahead: the overhang dword
                        const u32 a2 = L.candJ[xn], r22 = L.candSrc[xn], s2 = tailSlot[xn];
                        const u32 cw1 = L.tslot[s1][lc];
                        const u32 a = a0, r2 = r20, cw = cw0;
                        const bool isLeft = a & 1u; const u32 o = (a >> 1) & 1u, L2 = (a >> 2) & 0x1FFu, ov = (a >> 20) & 0xFFu; const int jw = (int)((a >> 11) & 0x1FFu);
                        if (jw != curJ) { curJ = jw; mAR = mAL = mFR = false; }
                        const u32 pov = isLeft ? pLov : pRov, pw = isLeft ? pLw : pRw;
                        const int m = (int)min(pov, ov);
                        const bool dif = lane < (u32)TAIL_OVW && ((pw ^ cw) & mask_top32(m - 16 * (int)lc)) != 0;
                        const bool cons = !__any(dif);
                        if (!isLeft) {
                            if (rightId == 0) { rightId = r2; rightO = o; rightLen = ov; pRw = cw; pRL = L2; pRov = ov; mAR = true; mFR = true; }
                            else if (cons) {
                                bool upd = false;
                                if (mAR) { if (L2 > pRL) { if (mFR) { rightId = r2; rightO = o; rightLen = ov; } upd = true; } }
                                else { upd = true; mAR = true; }
                                if (upd) { pRw = cw; pRL = L2; pRov = ov; }
                            } else ambR = true;
                        } else {
                            if (leftId == 0) { leftId = r2; leftO = o; leftLen = ov; pLw = cw; pLL = L2; pLov = ov; mAL = true; }
                            else if (cons) {
                                bool upd = false;
                                if (mAL) { if (L2 > pLL) upd = true; } else { upd = true; mAL = true; }
                                if (upd) { leftId = r2; leftO = o; leftLen = ov; pLw = cw; pLL = L2; pLov = ov; }
                            } else ambL = true;
                        }
                        a0 = a1; r20 = r21; cw0 = cw1; a1 = a2; r21 = r22; s1 = s2;
                    }
                    if (ambR || ambL) { rightLen = 0; leftLen = 0; }                                        // economyGraph.cpp:446-450
                    if (lane == 0) {
                        A.right[i] = (u64)rightId | ((u64)rightO << 40) | ((u64)(rightLen & 0x3FFFFFu) << 42);
                        A.left[i] = (u64)leftId | ((u64)leftO << 40) | ((u64)(leftLen & 0x3FFFFFu) << 42);
                        A.conn[i] = nhits;
                    }
                    continue;
                }
            }
        }
        STAMP(8);
        if (slowpath) {
            if (lane == 0) { u64 p = atomicAdd(&A.counters[6], 1ull); if (p < A.slow_cap) A.slow[p] = (u32)i; }
        } else if (HITS) {
            // (not reached: the hit-list variant leaves the loop body above)
        } else {
            // ---------------------------------------------------------------- 5. extension records
            selR = wave_min_dpp(selR); selL = wave_min_dpp(selL);
            u64 rv = 0, lv = 0;
            if (selR != ~0u) {
                const int ol = (int)(selR & 63u); const bool oq = ((selR >> 6) & 1u) != 0;
                const u32 en = (u32)__builtin_amdgcn_readlane((int)(oq ? myEnt[1] : myEnt[0]), ol);
                const int j = __builtin_amdgcn_readlane(oq ? myJ[1] : myJ[0], ol), L2 = __builtin_amdgcn_readlane(oq ? myL2[1] : myL2[0], ol);
                rv = (u64)(en >> 2) | ((u64)((en & 3) == 2 ? 1 : 0) << 40) | ((u64)((u32)(L2 - (L1 - j)) & 0x3FFFFFu) << 42);
            }
            if (selL != ~0u) {
                const int ol = (int)(selL & 63u); const bool oq = ((selL >> 6) & 1u) != 0;
                const u32 en = (u32)__builtin_amdgcn_readlane((int)(oq ? myEnt[1] : myEnt[0]), ol);
                const int j = __builtin_amdgcn_readlane(oq ? myJ[1] : myJ[0], ol), L2 = __builtin_amdgcn_readlane(oq ? myL2[1] : myL2[0], ol);
                lv = (u64)(en >> 2) | ((u64)((en & 3) == 3 ? 1 : 0) << 40) | ((u64)((u32)(L2 - j - h) & 0x3FFFFFu) << 42);
            }
            if (lane == 0) { A.right[i] = rv; A.left[i] = lv; A.conn[i] = nhits; }
        }
        STAMP(9);
    }
    if (HITS) {
        for (u32 x = lane; x < hLeft; x += 64) A.hits[hBase + x].from = 0;
        if (lane == 0 && hTotal) atomicAdd(&A.counters[5], hTotal);
    }
#ifdef SAGE2OV_STAMPS
    if (A.stamps && lane == 0) for (int x = 0; x < 10; x++) atomicAdd(&A.stamps[x], st_acc[x]);
#endif
}

// =============================================================================================
// reciprocal pass (economyGraph.cpp:455-480), order-independent restatement:
//   cond(i) does not depend on the serial order; the `exploredReads[x]!=4` test at the time read i
//   is visited is true iff NOT (x < i and cond(x)).
// =============================================================================================
constexpr u64 ID_MASK = (1ull << 40) - 1;
constexpr int COND_PER_THREAD = 8;                // reads per thread: the three log counters cost three atomics per 2048 reads
__global__ __launch_bounds__(256) void k_recip_cond(u64 N, const u64* __restrict__ right, const u64* __restrict__ left, const u32* __restrict__ conn,
                             const u32* __restrict__ cflag, uint8_t* status, u64* counters) {
    u64 ovs = 0; u32 ncond = 0, n6 = 0;
#pragma unroll
    for (int it = 0; it < COND_PER_THREAD; it++) {
        const u64 i = (u64)blockIdx.x * (256 * COND_PER_THREAD) + (u64)it * 256 + threadIdx.x + 1;
        if (i <= N) {
            const u32 c = conn[i], cf = cflag[i]; const bool over = c > CONN_LIMIT;
            const bool is6 = (cf & 1u) || ((cf & 2u) && !over);        // single-thread outcome of the :444 / :735 writes
            const u64 l = left[i], r = right[i]; const u64 lid = l & ID_MASK, rid = r & ID_MASK;
            bool cond = false;
            if (!is6 && (l >> 42) != 0 && (r >> 42) != 0) {
                const bool lrec = ((right[lid] & ID_MASK) == i) || ((left[lid] & ID_MASK) == i);
                const bool rrec = ((right[rid] & ID_MASK) == i) || ((left[rid] & ID_MASK) == i);
                cond = lrec && rrec;
            }
            status[i] = cond ? 4 : (is6 ? 6 : (over ? 5 : 0));
            ovs += c; ncond += cond; n6 += is6;
        }
    }
    // block reductions of the log counters: one atomic per counter per block
    __shared__ u64 red[3][4];
    u64 v0 = ovs, v1 = ncond, v2 = n6;
    for (int d = 32; d; d >>= 1) { v0 += __shfl_xor(v0, d); v1 += __shfl_xor(v1, d); v2 += __shfl_xor(v2, d); }
    const u32 w = threadIdx.x >> 6;
    if (lane_id() == 0) { red[0][w] = v0; red[1][w] = v1; red[2][w] = v2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        u64 t = 0; for (u32 x = 0; x < 4; x++) t += red[threadIdx.x][x];
        if (t) atomicAdd(&counters[1 + threadIdx.x], t);
    }
}
__device__ __forceinline__ u32 flip_type(u32 t) { return t == 0 ? 3u : (t == 3 ? 0u : t); }   // utils.cpp:212
// read length: a constant when the data set has a single length (the usual case), else the low 16 bits of the slot's last word
__device__ __forceinline__ int read_len(const u64* __restrict__ reads, int S, u64 id, int uniL) { return uniL ? uniL : (int)(reads[id * S + S - 1] & 0xFFFF); }
__device__ __forceinline__ EdgeCand make_edge(const u64* reads, int S, int uniL, u64 u, u64 v, u32 delta, u32 type) {
    EdgeCand e;
    if (u < v) { e.from = (u32)u; e.to = (u32)v; e.len = delta & 0xFFFFFu; e.type = type; }
    else {                                                                                  // the twin lives in the smaller id's list
        const int Lu = read_len(reads, S, u, uniL), Lv = read_len(reads, S, v, uniL);
        e.from = (u32)v; e.to = (u32)u; e.len = (u32)(Lu - (Lv - (int)delta)) & 0xFFFFFu; e.type = flip_type(type);   // economyGraph.cpp:821
    }
    return e;
}
constexpr int EMIT_PER_THREAD = 16;               // reads per thread: a block of 256 threads reserves space for 4096 reads with ONE atomic
__global__ __launch_bounds__(256) void k_recip_emit(u64 N, const u64* __restrict__ reads, int S, int uniL, const u64* __restrict__ right, const u64* __restrict__ left,
                             const uint8_t* __restrict__ status, EdgeCand* cand, u64 cap, u64* counters, u64 elo, u64 ehi) {
    __shared__ u32 sh[4]; __shared__ u64 shBase;
    const u64 tile0 = (u64)blockIdx.x * (256 * EMIT_PER_THREAD) + elo;
    u32 flags = 0, mine = 0;
#pragma unroll
    for (int it = 0; it < EMIT_PER_THREAD; it++) {
        const u64 i = tile0 + (u64)it * 256 + threadIdx.x;
        if (i < ehi && i <= N && status[i] == 4) {
            const u64 l = left[i], r = right[i], lid = l & ID_MASK, rid = r & ID_MASK;
            const bool eL = lid != i && !(lid < i && status[lid] == 4);                     // :462-467 (and u != v, :815)
            const bool eR = rid != i && !(rid < i && status[rid] == 4);                     // :468-473
            flags |= ((eL ? 1u : 0u) | (eR ? 2u : 0u)) << (2 * it); mine += (eL ? 1u : 0u) + (eR ? 1u : 0u);
        }
    }
    u32 total; const u32 excl = block_excl_scan(mine, sh, total);
    if (total == 0) return;
    if (threadIdx.x == 0) shBase = atomicAdd(&counters[0], (u64)total);
    __syncthreads();
    u64 p = shBase + excl;
#pragma unroll
    for (int it = 0; it < EMIT_PER_THREAD; it++) {
        const u32 f = (flags >> (2 * it)) & 3u; if (!f) continue;
        const u64 i = tile0 + (u64)it * 256 + threadIdx.x;
        const u64 l = left[i], r = right[i];
        if (f & 1u) { if (p < cap) cand[p] = make_edge(reads, S, uniL, i, l & ID_MASK, (u32)(l >> 42), ((l >> 40) & 3) == 0 ? 0u : 1u); p++; }
        if (f & 2u) { if (p < cap) cand[p] = make_edge(reads, S, uniL, i, r & ID_MASK, (u32)(r >> 42), ((r >> 40) & 3) == 0 ? 3u : 2u); p++; }
    }
}

// reduce-phase support: the host replay needs the lists of unresolved reads and of their neighbours
__global__ void k_red_mark(const EdgeCand* __restrict__ cand, u64 n, const uint8_t* __restrict__ status, uint8_t* need) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (x >= n) return;
    EdgeCand e = cand[x];
    if (status[e.from] == 0 || status[e.to] == 0) { need[e.from] = 1; need[e.to] = 1; }
}
__global__ void k_red_collect(EdgeCand* cand, u64 n, const uint8_t* __restrict__ status, const uint8_t* __restrict__ need, EdgeCand* out, u64 cap, u64* counter) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (x >= n) return;
    EdgeCand e = cand[x];
    if (need[e.from] || need[e.to]) { u64 p = atomicAdd(counter, 1ull); if (p < cap) out[p] = e; }
    if (status[e.from] == 0) cand[x].type = e.type | 0x80u;    // list of an unresolved read: rewritten by the replay
}
constexpr int UNRES_PER_THREAD = 16;
__global__ __launch_bounds__(256) void k_red_unresolved(u64 N, const uint8_t* __restrict__ status, u32* out, u64 cap, u64* counter) {
    __shared__ u32 sh[4]; __shared__ u64 shBase;
    const u64 tile0 = (u64)blockIdx.x * (256 * UNRES_PER_THREAD) + 1;
    u32 flags = 0, mine = 0;
#pragma unroll
    for (int it = 0; it < UNRES_PER_THREAD; it++) { const u64 i = tile0 + (u64)it * 256 + threadIdx.x; if (i <= N && status[i] == 0) { flags |= 1u << it; mine++; } }
    u32 total; const u32 excl = block_excl_scan(mine, sh, total);
    if (total == 0) return;
    if (threadIdx.x == 0) shBase = atomicAdd(counter, (u64)total);                 // one atomic per 4096 reads (most reads are unresolved on noisy data)
    __syncthreads();
    u64 p = shBase + excl;
#pragma unroll
    for (int it = 0; it < UNRES_PER_THREAD; it++) if (flags & (1u << it)) { if (p < cap) out[p] = (u32)(tile0 + (u64)it * 256 + threadIdx.x); p++; }
}

// =============================================================================================
// Reduce phase on the device (economyGraph.cpp:495-707) -- order-independent form, SURVEY A.6.
// Exact when discovery is symmetric, i.e. when no bucket is long (every S-S overlap is then seen from both
// sides, contained reads are never in S, and at the time markTransitiveEdge(r) runs in the serial BFS the lists
// of r and of all its neighbours are complete and unreduced).  The serial replay on the host remains the
// path for indexes with long buckets and for a handful of unresolved reads.
//   list[r] (all reads) = both directed entries of every reciprocal-pass candidate + for r in S its own hits;
//   for r in S: sort (len desc, id desc, type desc) (:853-871), mark (:643-679), drop marked (:681-707);
//   survivors with to > r replace the candidates owned by r.
// Entry = len:20 | to:32 | type:2 | position:9 (spare bits of the sort key carry the entry's place in the list).
// =============================================================================================
constexpr int RA_CAP = 512;                       // longest list handled on the device (connections <= 300 for S reads)
constexpr int RA_HT = 1024;
__device__ __forceinline__ u64 ra_key(u32 to, u32 type, u32 len) { return ((u64)(len & 0xFFFFFu) << 43) | ((u64)to << 11) | ((u64)(type & 3u) << 9); }
__device__ __forceinline__ u32 ra_to(u64 k) { return (u32)(k >> 11); }
__device__ __forceinline__ u32 ra_type(u64 k) { return (u32)(k >> 9) & 3u; }
__device__ __forceinline__ u32 ra_len(u64 k) { return (u32)(k >> 43) & 0xFFFFFu; }
__global__ void k_ra_degree_c(EdgeCand* cand, u64 n, const uint8_t* __restrict__ status, u32* deg) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (x >= n) return;
    EdgeCand e = cand[x]; e.type &= 0x7Fu;
    atomicAdd(&deg[e.from], 1u); atomicAdd(&deg[e.to], 1u);
    if (status[e.from] == 0) cand[x].type = e.type | 0x80u;            // list of an unresolved read: re-emitted after the reduction
}
__global__ void k_ra_degree_h(const u32* __restrict__ hitcount, u64 N, u32* deg) {       // deg[i] = candidate entries + own hits
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i > N) return;
    deg[i] += hitcount[i];
}
__global__ void k_ra_fill_c(const EdgeCand* __restrict__ cand, u64 n, const u64* __restrict__ reads, int S, int uniL, const u32* __restrict__ offs, u32* cursor, u64* ent) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (x >= n) return;
    EdgeCand e = cand[x]; e.type &= 0x7Fu;
    const int Lf = read_len(reads, S, e.from, uniL), Lt = read_len(reads, S, e.to, uniL);
    ent[offs[e.from] + atomicAdd(&cursor[e.from], 1u)] = ra_key(e.to, e.type, e.len);
    ent[offs[e.to] + atomicAdd(&cursor[e.to], 1u)] = ra_key(e.from, flip_type(e.type), (u32)(Lf - (Lt - (int)e.len)));   // the twin (economyGraph.cpp:821)
}
__global__ void k_ra_fill_h(const Hit* __restrict__ hits, u64 n, const u32* __restrict__ offs, const u32* __restrict__ deg, const u32* __restrict__ hitcount, u64* ent) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (x >= n) return;
    const Hit h = hits[x];                                               // the hits of a read are numbered 0.. by the probe kernel: no cursor
    if (h.from == 0) return;                                             // unused slot of a wave's chunk
    ent[offs[h.from] + (deg[h.from] - hitcount[h.from]) + h.seq] = ra_key(h.to, h.type, (u32)h.len);
}
struct RaLds { u64 key[RA_CAP]; u32 ht[RA_HT]; uint8_t mk[RA_HT]; unsigned short slot[RA_CAP]; };
__device__ __forceinline__ u32 ra_hash(u32 id) { return (id * 2654435761u) >> 22; }     // 10 bits
__device__ __forceinline__ void ra_mark_one(RaLds& L, u64 kb, u32 t1) {
    const u32 b = ra_to(kb), t2 = ra_type(kb);
    const bool compat = ((t1 == 0 || t1 == 2) && (t2 == 0 || t2 == 1)) || ((t1 == 1 || t1 == 3) && (t2 == 2 || t2 == 3));
    if (!compat) return;
    u32 sidx = ra_hash(b);
    for (;;) {
        const u32 v = L.ht[sidx];
        if (v == 0) break;
        if (v == b) { if (L.mk[sidx] == 1) L.mk[sidx] = 2; break; }
        sidx = (sidx + 1) & (RA_HT - 1);
    }
}
// counters: [0] lists longer than RA_CAP (-> host replay), [1] removed entries.  svn[w] = survivors with to > r of the w-th read.
__global__ __launch_bounds__(256) void k_ra_mark(const u32* __restrict__ ids, u64 nids, const u32* __restrict__ offs, const u32* __restrict__ deg,
                                                 const u64* __restrict__ ent, uint8_t* rm, u32* svn, u64* counters) {
    __shared__ RaLds lds[4];
    RaLds& L = lds[threadIdx.x >> 6];
    const u32 lane = lane_id();
    u64 removedTotal = 0;
    for (u64 w = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); w < nids; w += (u64)gridDim.x * 4) {
        const u32 r = ids[w]; const u32 n = deg[r]; const u32 o = offs[r];
        if (n == 0) { if (lane == 0) svn[w] = 0; continue; }
        if (n > (u32)RA_CAP) { if (lane == 0) { svn[w] = 0; atomicAdd(&counters[0], 1ull); } continue; }
        u32 P = 64; while (P < n) P <<= 1;
        wave_sync();
        for (u32 x = lane; x < P; x += 64) L.key[x] = x < n ? (ent[o + x] | (u64)x) : 0ull;
        for (u32 x = lane; x < (u32)RA_HT; x += 64) { L.ht[x] = 0; L.mk[x] = 0; }
        wave_sync();
        // bitonic sort, descending: (len, id, type) as in compareEdges (:853-871)
        for (u32 k = 2; k <= P; k <<= 1)
            for (u32 j = k >> 1; j > 0; j >>= 1) {
                for (u32 t = lane; t < P / 2; t += 64) {
                    const u32 i = ((t / j) * 2 * j) + (t % j), l = i + j;
                    const u64 a = L.key[i], b = L.key[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { L.key[i] = b; L.key[l] = a; }
                }
                wave_sync();
            }
        // node table: one mark per neighbour id (several entries may lead to the same read)
        for (u32 x = lane; x < n; x += 64) {
            const u32 id = ra_to(L.key[x]); u32 sidx = ra_hash(id);
            for (;;) {
                const u32 old = atomicCAS(&L.ht[sidx], 0u, id);
                if (old == 0 || old == id) break;
                sidx = (sidx + 1) & (RA_HT - 1);
            }
            L.slot[x] = (unsigned short)sidx; L.mk[sidx] = 1;
        }
        wave_sync();
        // markTransitiveEdge (:643-679): sequential over the sorted list; the neighbours' lists are fetched eight at a time
        // (first 64 entries of each, one per lane), so the loop pays one memory round trip per eight neighbours.
        // (Keeping the list locations in LDS and double-buffering the batches was measured: no gain, the loop is issue-bound.)
        for (u32 x0 = 0; x0 < n; x0 += 8) {
            u32 na[8], oa[8]; u64 kb[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const u32 a = x0 + u < n ? ra_to(L.key[x0 + u]) : 0u; na[u] = x0 + u < n ? deg[a] : 0u; oa[u] = offs[a]; }
#pragma unroll
            for (int u = 0; u < 8; u++) kb[u] = ent[oa[u] + (lane < na[u] ? lane : 0u)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const u32 x = x0 + u;
                if (x >= n) break;
                if (L.mk[L.slot[x]] != 1) continue;                         // (wave-uniform)
                const u32 t1 = ra_type(L.key[x]);
                if (lane < na[u]) ra_mark_one(L, kb[u], t1);
                for (u32 y = lane + 64; y < na[u]; y += 64) ra_mark_one(L, ent[oa[u] + y], t1);
                wave_sync();
            }
        }
        u32 nrm = 0, nsv = 0;
        for (u32 x = lane; x < n; x += 64) {
            const u64 kx = L.key[x]; const bool gone = L.mk[L.slot[x]] == 2;
            rm[o + (u32)(kx & 511u)] = gone ? 1 : 0;
            nrm += gone; nsv += (!gone && ra_to(kx) > r);
        }
        for (int dlt = 32; dlt; dlt >>= 1) { nrm += __shfl_xor(nrm, dlt); nsv += __shfl_xor(nsv, dlt); }
        if (lane == 0) svn[w] = nsv;
        removedTotal += nrm;
    }
    if (lane == 0 && removedTotal) atomicAdd(&counters[1], removedTotal);       // one atomic per wave, not per read
}
__global__ void k_ra_emit(const u32* __restrict__ ids, u64 nids, const u32* __restrict__ offs, const u32* __restrict__ deg, const u64* __restrict__ ent,
                          const uint8_t* __restrict__ rm, const u32* __restrict__ svoff, EdgeCand* cand, u64 base, u64 cap) {
    const u32 lane = lane_id();
    for (u64 w = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nids; w += ((u64)gridDim.x * blockDim.x) >> 6) {
        const u32 r = ids[w]; const u32 n = deg[r], o = offs[r];
        u64 pos0 = base + svoff[w];
        for (u32 x0 = 0; x0 < n; x0 += 64) {
            const u32 x = x0 + lane; bool keep = false; u64 k = 0;
            if (x < n) { k = ent[o + x]; keep = rm[o + x] == 0 && ra_to(k) > r; }
            const u64 bal = __ballot(keep);
            const u64 pos = pos0 + (u64)__popcll(bal & ((1ull << lane) - 1ull));
            if (keep && pos < cap) { EdgeCand e; e.from = r; e.to = ra_to(k); e.len = ra_len(k); e.type = ra_type(k); cand[pos] = e; }
            pos0 += (u64)__popcll(bal);
        }
    }
}

// =============================================================================================
// sortEconomyGraph + convertGraph (economyGraph.cpp:896, overlapGraph.cpp:84-111) on the candidate list
// =============================================================================================
__global__ void k_conv_degree(const EdgeCand* __restrict__ cand, u64 n, u32* deg) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n && !(cand[x].type & 0x80u)) atomicAdd(&deg[cand[x].from], 1u);
}
__global__ void k_conv_fill(const EdgeCand* __restrict__ cand, u64 n, const u32* __restrict__ offs, u32* cursor, u64* keys) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    EdgeCand e = cand[x]; if (e.type & 0x80u) return;
    u32 p = atomicAdd(&cursor[e.from], 1u);
    keys[offs[e.from] + p] = ((u64)e.to << 22) | ((u64)e.type << 20) | e.len;      // ascending = compareIdBased (economyGraph.cpp:875)
}
__global__ void k_conv_sort(u64 N, const u32* __restrict__ offs, const u32* __restrict__ deg, u64* keys, u32* keep) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    u32 d = deg[i]; if (!d) return;
    u64* a = keys + offs[i]; u32* kp = keep + offs[i];
    for (u32 x = 1; x < d; x++) { u64 v = a[x]; int j = (int)x - 1; while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; j--; } a[j + 1] = v; }
    for (u32 x = 0; x < d; x++) kp[x] = (x == 0 || (a[x] >> 20) != (a[x - 1] >> 20)) ? 1u : 0u;   // overlapGraph.cpp:101
}
__global__ void k_conv_owner(u64 N, const u32* __restrict__ offs, const u32* __restrict__ deg, u32* owner) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    u32 d = deg[i]; for (u32 x = 0; x < d; x++) owner[offs[i] + x] = (u32)i;
}
__global__ void k_conv_emit(u64 n, const u64* __restrict__ keys, const u32* __restrict__ keep, const u32* __restrict__ pos, const u32* __restrict__ owner,
                            const u64* __restrict__ reads, int S, int uniL, FinalEdge* out) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n || !keep[x]) return;
    u64 kx = keys[x]; FinalEdge f; f.from = owner[x]; f.to = (u32)(kx >> 22); f.type = (u32)(kx >> 20) & 3u; f.len = (u32)(kx & 0xFFFFFu);
    const u32 Lu = (u32)read_len(reads, S, f.from, uniL), Lv = (u32)read_len(reads, S, f.to, uniL);
    f.len_twin = Lu - (Lv - f.len);                                                         // overlapGraph.cpp:145-148 (u32 arithmetic)
    out[pos[x]] = f;
}

// =============================================================================================
// host-side launchers
// =============================================================================================
static inline unsigned grid_for(u64 n, unsigned block) { return (unsigned)std::max<u64>(1, (n + block - 1) / block); }

Device* dev_create(int ordinal, std::string& err) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) { err = std::string("no HIP device available: ") + hipGetErrorString(e); return nullptr; }
    int dev = ordinal;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= count) { err = "device ordinal out of range"; return nullptr; }
    if (hipSetDevice(dev) != hipSuccess) { err = "hipSetDevice failed"; return nullptr; }
    Device* d = new Device(); d->ordinal = dev;
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; delete d; return nullptr; }
    { const char* tb = getenv("SAGE2OV_TEST_TAG_BITS"); const int nb = tb ? atoi(tb) : 24;
      const u32 mask = (nb >= 1 && nb < 24) ? ((1u << nb) - 1u) : 0xFFFFFFu;
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_tag_mask), &mask, sizeof mask) != hipSuccess) { err = "tag mask upload failed"; delete d; return nullptr; }
      const char* mb = getenv("SAGE2OV_TEST_MTAG_BITS"); const int nm = mb ? atoi(mb) : 24;
      const u32 mmask = (nm >= 1 && nm < 24) ? ((1u << nm) - 1u) : 0xFFFFFFu;
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_mtag_mask), &mmask, sizeof mmask) != hipSuccess) { err = "tag mask upload failed"; delete d; return nullptr; } }
    for (auto& ev : d->ev) hipEventCreate(&ev);
    if (hipMalloc(&d->d_counters, 24 * sizeof(u64)) != hipSuccess) { err = "hipMalloc(counters) failed"; delete d; return nullptr; }
    hipMemset(d->d_counters, 0, 24 * sizeof(u64));
    return d;
}
static void free_reads(Device* d) {
    hipFree(d->reads); hipFree(d->right); hipFree(d->left); hipFree(d->conn); hipFree(d->cflag);
    hipFree(d->status); hipFree(d->cand);     // slots / csr / final_edges live in the workspace arena
    for (auto& b : d->ws) { if (b.p) hipFree(b.p); b.p = nullptr; b.cap = 0; }
    d->reads = d->slots = nullptr; d->csr = nullptr; d->right = d->left = nullptr; d->conn = d->cflag = nullptr; d->status = nullptr; d->cand = nullptr; d->final_edges = nullptr;
}
void dev_destroy(Device* d) {
    if (!d) return;
    hipSetDevice(d->ordinal);
    hipStreamSynchronize(d->stream);
    free_reads(d); hipFree(d->d_counters);
    for (auto& ev : d->ev) if (ev) hipEventDestroy(ev);
    hipStreamDestroy(d->stream);
    delete d;
}
void* dev_stream(Device* d) { return (void*)d->stream; }
int dev_sync(Device* d, std::string& err) { HIPCHK(hipStreamSynchronize(d->stream)); return 0; }
void dev_timings(Device* d, DevTimings* t) { *t = d->tm; }
void dev_reset_timings(Device* d) { d->tm = DevTimings(); }

int dev_upload_reads(Device* d, const uint64_t* words, uint64_t N, int S, int minL, int maxL, int k, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (S != 4 && S != 8 && S != 16) { err = "unsupported words-per-read (read length limit is 504 bases)"; return SAGE2OV_ERR_LIMIT; }
    if (N >= (1ull << 30)) { err = "more than 2^30-1 unique reads per context is not supported yet"; return SAGE2OV_ERR_LIMIT; }
    free_reads(d);
    d->N = N; d->S = S; d->maxL = maxL; d->k = k; d->h = k > 64 ? 64 : k; d->uniL = (N && minL == maxL) ? maxL : 0;
    HIPCHK(hipMalloc(&d->reads, (N + 1) * S * sizeof(u64)));
    HIPCHK(hipMemcpyAsync(d->reads, words, (N + 1) * S * sizeof(u64), hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipMalloc(&d->right, (N + 1) * sizeof(u64))); HIPCHK(hipMalloc(&d->left, (N + 1) * sizeof(u64)));
    HIPCHK(hipMalloc(&d->conn, (N + 1) * sizeof(u32))); HIPCHK(hipMalloc(&d->cflag, (N + 1) * sizeof(u32)));
    HIPCHK(hipMalloc(&d->status, (N + 1)));
    d->cand_cap = 2 * N + 1024;
    HIPCHK(hipMalloc(&d->cand, d->cand_cap * sizeof(EdgeCand)));
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}

static int scan_u32(Device* d, const u32* in, u64 n, u32* out, u64* total, std::string& err);
// Step 1 on the device: see k_org_canon.  On return the read store is resident exactly as after dev_upload_reads, and the
// host receives the image (for the .reads writer, lengths) and the frequencies.
int dev_organize_reads(Device* d, const uint64_t* pool, uint64_t pool_words, const uint64_t* off, const uint16_t* len, uint64_t n, int S, int minL, int maxL, int k,
                       uint64_t* N_out, std::vector<uint64_t>& words_out, std::vector<uint16_t>& freq_out, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (S != 4 && S != 8 && S != 16) { err = "unsupported words-per-read (read length limit is 504 bases)"; return SAGE2OV_ERR_LIMIT; }
    if (n >= (1ull << 32) - RS_TILE) { err = "too many reads for the device organiser"; return SAGE2OV_ERR_LIMIT; }
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, d->stream));
    u64 N = 0;
    u64* reads = nullptr; unsigned short* dfreq = nullptr;
    if (n) {
        WS(dpool, u64, WS_ORG_POOL, pool_words + 17); WS(doff, u64, WS_ORG_OFF, n); WS(dlen, unsigned short, WS_ORG_LEN, n);
        WS(img, u64, WS_ORG_IMG, n * S); WS(k0, u64, WS_ORG_K0, n); WS(k1, u64, WS_ORG_K1, n); WS(v0, u32, WS_ORG_V0, n); WS(v1, u32, WS_ORG_V1, n);
        HIPCHK(hipMemcpyAsync(dpool, pool, pool_words * sizeof(u64), hipMemcpyHostToDevice, d->stream));
        HIPCHK(hipMemsetAsync(dpool + pool_words, 0, 17 * sizeof(u64), d->stream));
        HIPCHK(hipMemcpyAsync(doff, off, n * sizeof(u64), hipMemcpyHostToDevice, d->stream));
        HIPCHK(hipMemcpyAsync(dlen, len, n * sizeof(uint16_t), hipMemcpyHostToDevice, d->stream));
        hipLaunchKernelGGL(k_org_canon, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, dpool, doff, dlen, (u64)n, S, img, k0, v0);
        const u32 nb = (u32)((n + RS_TILE - 1) / RS_TILE);
        WS(hist, u32, WS_ORG_HIST, (u64)256 * nb + 2); WS(hscan, u32, WS_ORG_HSCAN, (u64)256 * nb + 2);
        u64 *ka = k0, *kb = k1; u32 *va = v0, *vb = v1;
        for (int pass = 0; pass < 8; pass++) {
            hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(64), 0, d->stream, ka, (u64)n, 8 * pass, hist, nb);
            u64 tot = 0; int rc = scan_u32(d, hist, (u64)256 * nb, hscan, &tot, err); if (rc) return rc;
            hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(64), 0, d->stream, ka, va, (u64)n, 8 * pass, hscan, nb, kb, vb);
            std::swap(ka, kb); std::swap(va, vb);
        }
        hipLaunchKernelGGL(k_org_ties, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, ka, va, (u64)n, img, S);
        WS(flag, u32, WS_ORG_FLAG, n + 2); WS(uid, u32, WS_ORG_UID, n + 2);
        hipLaunchKernelGGL(k_org_heads, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, ka, va, (u64)n, img, S, flag);
        { int rc = scan_u32(d, flag, n, uid, &N, err); if (rc) return rc; }
        if (N >= (1ull << 30)) { err = "more than 2^30-1 unique reads per context is not supported yet"; return SAGE2OV_ERR_LIMIT; }
        WS(headPos, u32, WS_ORG_HEAD, N + 2);
        hipLaunchKernelGGL(k_org_headpos, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, flag, uid, (u64)n, headPos, (u64)N);
        HIPCHK(hipMalloc(&reads, (N + 1) * S * sizeof(u64))); HIPCHK(hipMalloc(&dfreq, (N + 1) * sizeof(unsigned short)));
        HIPCHK(hipMemsetAsync(reads, 0, S * sizeof(u64), d->stream)); HIPCHK(hipMemsetAsync(dfreq, 0, sizeof(unsigned short), d->stream));
        hipLaunchKernelGGL(k_org_gather, dim3(grid_for(N * S, 256)), dim3(256), 0, d->stream, va, headPos, (u64)N, img, S, reads, dfreq);
    } else {
        HIPCHK(hipMalloc(&reads, S * sizeof(u64))); HIPCHK(hipMalloc(&dfreq, sizeof(unsigned short)));
        HIPCHK(hipMemsetAsync(reads, 0, S * sizeof(u64), d->stream)); HIPCHK(hipMemsetAsync(dfreq, 0, sizeof(unsigned short), d->stream));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, d->stream));
    words_out.resize((N + 1) * S); freq_out.resize(N + 1);
    HIPCHK(hipMemcpyAsync(words_out.data(), reads, (N + 1) * S * sizeof(u64), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipMemcpyAsync(freq_out.data(), dfreq, (N + 1) * sizeof(unsigned short), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); d->tm.organize_ms += ms; hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(dfreq);
    // the organised store becomes the context's read store (same state as after dev_upload_reads)
    free_reads(d);
    d->N = N; d->S = S; d->maxL = maxL; d->k = k; d->h = k > 64 ? 64 : k; d->reads = reads; d->uniL = (N && minL == maxL) ? maxL : 0;
    HIPCHK(hipMalloc(&d->right, (N + 1) * sizeof(u64))); HIPCHK(hipMalloc(&d->left, (N + 1) * sizeof(u64)));
    HIPCHK(hipMalloc(&d->conn, (N + 1) * sizeof(u32))); HIPCHK(hipMalloc(&d->cflag, (N + 1) * sizeof(u32)));
    HIPCHK(hipMalloc(&d->status, (N + 1)));
    d->cand_cap = 2 * N + 1024;
    HIPCHK(hipMalloc(&d->cand, d->cand_cap * sizeof(EdgeCand)));
    *N_out = N;
    return 0;
}

int dev_build_index(Device* d, uint64_t* slots_out, uint64_t* keys_out, uint64_t* csr_out, uint64_t* nlong_out, uint32_t* rebuilds, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const u64 N = d->N; if (!d->reads) { err = "reads not resident"; return SAGE2OV_ERR_ARG; }
    d->T = std::max<u64>(1024, 8 * N);                                   // load <= 0.5, as hashTable.cpp:83 sizes it (even: pairs)
    if (d->T >= (1ull << 32)) { err = "table too large for 32-bit slot indices"; return SAGE2OV_ERR_LIMIT; }
    const u32 big_cap = 1u << 20;
    WS(slots_ws, u64, WS_SLOTS, d->T); d->slots = slots_ws;
    WS(where, u64, WS_WHERE, std::max<u64>(1, 4 * N));               // per entry: slot index | rank inside the bucket << 32
    WS(big, u64, WS_BIG, (u64)big_cap * 3);
    WS(csr_ws, u32, WS_CSR, std::max<u64>(1, 4 * N)); d->csr = csr_ws;
    const bool wantMI = !getenv("SAGE2OV_NO_MINIMIZER_INDEX") && (d->h - std::min(d->h, 16) + 1) >= 8;
    u64* slot_g = nullptr; u64* mi1 = nullptr; u64* krec = nullptr; u64 TL = 0;
    if (wantMI) {
        TL = 1024; while (TL < d->T / 4) TL <<= 1;                           // >= 2N group slots
        WS(smh, u64, WS_SLOTMH, std::max<u64>(1, 4 * N)); slot_g = smh;      // per entry (rank-0 entries only): group slot | rank inside the group << 32
        WS(m1, u64, WS_MI1, TL); mi1 = m1;
        WS(kr_, u64, WS_KREC, 4 * N + MI_SCAN_PAD); krec = kr_;              // one record per distinct key (<= 4N), written by the fill kernel
    }
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    *rebuilds = 0;
    for (int attempt = 0;; attempt++) {
        HIPCHK(hipMemsetAsync(d->slots, 0, d->T * sizeof(u64), d->stream));
        HIPCHK(hipMemsetAsync(d->d_counters + 8, 0, 9 * sizeof(u64), d->stream));
        if (wantMI) HIPCHK(hipMemsetAsync(mi1, 0, TL * sizeof(u64), d->stream));
        const unsigned gE = (unsigned)std::min<u64>(grid_for(4 * N, 256), 256 * 64);
        hipLaunchKernelGGL(k_index_count, dim3(gE), dim3(256), 0, d->stream, d->reads, N, d->S, d->h, d->seed, d->slots, d->T, where,
                           slot_g, mi1, TL, d->d_counters + 8);
        hipLaunchKernelGGL(k_index_alloc, dim3(grid_for(d->T, ALLOC_ITEMS)), dim3(256), 0, d->stream, d->slots, d->T, d->d_counters + 8, big, big_cap);
        if (wantMI) hipLaunchKernelGGL(k_mi_alloc, dim3(grid_for(TL, ALLOC_ITEMS)), dim3(256), 0, d->stream, mi1, TL, d->d_counters + 8);
        hipLaunchKernelGGL(k_index_fill, dim3(gE), dim3(256), 0, d->stream, N, d->slots, where, d->csr, slot_g, mi1, krec);
        hipLaunchKernelGGL(k_index_sort, dim3(grid_for(d->T, 256)), dim3(256), 0, d->stream, d->slots, d->T, d->csr);
        u64 c[9];
        HIPCHK(hipMemcpyAsync(c, d->d_counters + 8, sizeof c, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        if (c[8]) { err = "a key occurs in more than 16 M reads"; return SAGE2OV_ERR_LIMIT; }
        if (c[2] > big_cap) { err = "too many long buckets"; return SAGE2OV_ERR_LIMIT; }
        if (c[2]) {
            hipLaunchKernelGGL(k_index_purity, dim3(grid_for(c[2] * 64, 256)), dim3(256), 0, d->stream, d->reads, d->S, d->h, big, c[2], d->csr, d->d_counters + 8);
            HIPCHK(hipMemcpyAsync(c, d->d_counters + 8, 5 * sizeof(u64), hipMemcpyDeviceToHost, d->stream));
            HIPCHK(hipStreamSynchronize(d->stream));
        }
        if (c[3] == 0) { d->n_csr = c[0]; d->n_keys = c[1]; d->n_long = c[2]; break; }
        if (attempt >= 8) { err = "index build: tag collisions in long buckets persist after 8 reseeds"; return SAGE2OV_ERR_INTERNAL; }
        d->seed = d->seed * 0x9E3779B97F4A7C15ull + 12345; (*rebuilds)++;
    }
    // ---- stage B: minimiser groups over the distinct-key records
    d->mi1 = nullptr; d->krec = nullptr; d->TL = 0;
    if (wantMI && d->n_keys > 0) {
        u64 mc[3];
        HIPCHK(hipMemcpyAsync(mc, d->d_counters + 8 + 5, sizeof mc, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        // the probe scan may run past the last group: empty records behind ALL records (there are a few more records than slots
        // when different keys share a bucket: each key files its own copy)
        HIPCHK(hipMemsetAsync(krec + mc[0], 0, MI_SCAN_PAD * sizeof(u64), d->stream));
        d->n_groups = mc[1];
        if (getenv("SAGE2OV_VERIFY_MI")) {
            u64* vo = nullptr; HIPCHK(hipMalloc(&vo, 2 * sizeof(u64))); HIPCHK(hipMemsetAsync(vo, 0, 2 * sizeof(u64), d->stream));
            hipLaunchKernelGGL(k_mi_verify, dim3(grid_for(4 * N, 256)), dim3(256), 0, d->stream, d->reads, N, d->S, d->h, d->seed, d->slots, where, mi1, TL, krec, vo);
            u64 hv2[2]; HIPCHK(hipMemcpyAsync(hv2, vo, sizeof hv2, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream)); hipFree(vo);
            fprintf(stderr, "[verify-mi] %llu of %llu entries do not find their bucket in their key's group\n", (unsigned long long)hv2[0], (unsigned long long)hv2[1]);
        }
        if (mc[2] == 0 && mc[1] * 10 <= TL * 7) { d->mi1 = mi1; d->krec = krec; d->TL = TL; }   // else: too crowded, the fast kernel uses the uniform table
    }
    HIPCHK(hipEventRecord(d->ev[1], d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]); d->tm.index_ms += ms;
    *slots_out = d->T; *keys_out = d->n_keys; *csr_out = d->n_csr; *nlong_out = d->n_long;
    return 0;
}

int dev_lookup(Device* d, uint64_t hi, uint64_t lo, uint64_t* entries, uint32_t cap, uint32_t* count, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (!d->slots) { err = "index not built"; return SAGE2OV_ERR_ARG; }
    u64* out = nullptr; const u32 c2 = std::min<u32>(cap, 128);
    HIPCHK(hipMalloc(&out, (129) * sizeof(u64)));
    hipLaunchKernelGGL(k_lookup, dim3(1), dim3(1), 0, d->stream, d->slots, d->T, d->csr, d->seed, d->h, (u64)hi, (u64)lo, out, c2);
    u64 host[129];
    HIPCHK(hipMemcpyAsync(host, out, sizeof host, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    hipFree(out);
    *count = (u32)host[0];
    for (u32 x = 0; x < *count && x < c2; x++) entries[x] = host[1 + x];
    return 0;
}

template <int MODE>
static int launch_probe(Device* d, ProbeArgs& A, std::string& err) {
    const u64 nreads = A.ids ? A.n_ids : A.hi - A.lo; if (nreads == 0) return 0;
    // one wave per read, 4 waves per block (2 for the 16-word layout: LDS); enough blocks to fill
    // 256 CUs several times over, grid-stride beyond
    const unsigned wpb = d->S == 16 ? 2 : 4;
    const unsigned blocks = (unsigned)std::min<u64>((nreads + wpb - 1) / wpb, 256ull * 32);
    switch (d->S) {
        case 4: hipLaunchKernelGGL((k_probe<4, MODE, 4>), dim3(blocks), dim3(256), 0, d->stream, A); break;
        case 8: hipLaunchKernelGGL((k_probe<8, MODE, 4>), dim3(blocks), dim3(256), 0, d->stream, A); break;
        case 16: hipLaunchKernelGGL((k_probe<16, MODE, 2>), dim3(blocks), dim3(128), 0, d->stream, A); break;
        default: err = "bad S"; return SAGE2OV_ERR_INTERNAL;
    }
    HIPCHK(hipGetLastError());
    return 0;
}
static int scan_u32(Device* d, const u32* in, u64 n, u32* out, u64* total, std::string& err);
// processing order of ids [lo,hi): grouped by the reads' global minimiser (see k_minimizer)
static int build_locality_order(Device* d, u64 lo, u64 hi, u32** order_out, std::string& err) {
    const u64 n = hi - lo;
    u64 nbk = 1024; while (nbk < n / 16) nbk <<= 1;
    int lg = 0; while ((1ull << lg) < nbk) lg++;
    const int shift = 32 - lg;
    WS(minh, u32, WS_MINH, n); WS(cnt, u32, WS_OCNT, nbk); WS(offs, u32, WS_OOFF, nbk); WS(cur, u32, WS_OCUR, nbk); WS(order, u32, WS_ORDER, n);
    HIPCHK(hipMemsetAsync(cnt, 0, nbk * sizeof(u32), d->stream)); HIPCHK(hipMemsetAsync(cur, 0, nbk * sizeof(u32), d->stream));
    hipLaunchKernelGGL(k_minimizer, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, d->reads, (u64)lo, (u64)hi, d->S, minh);
    hipLaunchKernelGGL(k_order_count, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, minh, (u64)n, shift, cnt);
    u64 tot = 0; int rc = scan_u32(d, cnt, nbk, offs, &tot, err); if (rc) return rc;
    hipLaunchKernelGGL(k_order_fill, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, minh, (u64)n, (u64)lo, shift, offs, cur, order);
    *order_out = order;
    return 0;
}
static ProbeArgs base_args(Device* d) {
    ProbeArgs A; memset(&A, 0, sizeof A);
    A.reads = d->reads; A.N = d->N; A.S = d->S; A.k = d->k; A.h = d->h; A.slots = d->slots; A.T = d->T; A.csr = d->csr; A.seed = d->seed;
    A.right = d->right; A.left = d->left; A.conn = d->conn; A.cflag = d->cflag; A.status = d->status; A.counters = d->d_counters;
    A.mi1 = d->mi1; A.TL = d->TL; A.krec = d->krec;
    return A;
}

template <int S, int NW, int WPL, int WPB, int HITS = 0>
static void launch_fast(Device* d, ProbeArgs& A, unsigned blocks) { hipLaunchKernelGGL((k_probe_fast<S, NW, WPL, WPB, HITS>), dim3(blocks), dim3(64 * WPB), 0, d->stream, A); }
// picks the instantiation for the resident reads; false: the 16-word layout has no fast kernel
template <int HITS>
static bool launch_fast_any(Device* d, ProbeArgs& A, unsigned blocks) {
    constexpr int FW = SAGE2OV_FAST_WPB;      // waves per block: a whole CU's worth works on one locality chunk
    const int nwinMax = d->maxL - d->h + 1;                               // windows of the longest read
    if (d->S == 4 && nwinMax <= 128) launch_fast<4, 8, 2, FW, HITS>(d, A, blocks);
    else if (d->S == 8 && d->maxL <= 160 && nwinMax <= 128) launch_fast<8, 10, 2, FW, HITS>(d, A, blocks);
    else if (d->S == 8 && d->maxL <= 160) launch_fast<8, 10, 4, FW, HITS>(d, A, blocks);
    else if (d->S == 8 && nwinMax <= 128) launch_fast<8, 16, 2, FW, HITS>(d, A, blocks);
    else if (d->S == 8) launch_fast<8, 16, 4, FW, HITS>(d, A, blocks);
    else return false;
    return true;
}

int dev_probe(Device* d, uint64_t lo, uint64_t hi, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (!d->slots) { err = "index not built"; return SAGE2OV_ERR_ARG; }
    const u64 N = d->N;
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    HIPCHK(hipMemsetAsync(d->right, 0, (N + 1) * sizeof(u64), d->stream)); HIPCHK(hipMemsetAsync(d->left, 0, (N + 1) * sizeof(u64), d->stream));
    HIPCHK(hipMemsetAsync(d->conn, 0, (N + 1) * sizeof(u32), d->stream)); HIPCHK(hipMemsetAsync(d->cflag, 0, (N + 1) * sizeof(u32), d->stream));
    HIPCHK(hipMemsetAsync(d->d_counters + 6, 0, sizeof(u64), d->stream));
    ProbeArgs A = base_args(d); A.lo = lo; A.hi = hi;
    const u64 nreads = hi > lo ? hi - lo : 0;
    const bool seq_only = getenv("SAGE2OV_SEQUENTIAL_PROBE") != nullptr;
    if (nreads && !seq_only) {
        WS(slow, u32, WS_SLOW, nreads);
        A.slow = slow; A.slow_cap = nreads;
        const int nwinMax = d->maxL - d->h + 1;                               // windows of the longest read
        if (!getenv("SAGE2OV_NO_LOCALITY")) { u32* order = nullptr; int rc = build_locality_order(d, lo, hi, &order, err); if (rc) return rc; A.ids = order; A.n_ids = nreads; }
        const unsigned blocks = (unsigned)std::min<u64>((nreads + FAST_CHUNK - 1) / FAST_CHUNK, 256ull * 16);
#ifdef SAGE2OV_STAMPS
        static u64* d_stamps = nullptr;
        if (!d_stamps) HIPCHK(hipMalloc(&d_stamps, 10 * sizeof(u64)));
        HIPCHK(hipMemsetAsync(d_stamps, 0, 10 * sizeof(u64), d->stream)); A.stamps = d_stamps;
#endif
        HIPCHK(hipEventRecord(d->ev[2], d->stream));
        const bool launched = launch_fast_any<0>(d, A, blocks);               // false: 16-word layout, sequential kernel only (for now)
        if (!launched) { A.ids = nullptr; A.n_ids = 0; int rc = launch_probe<0>(d, A, err); if (rc) return rc; }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(d->ev[3], d->stream));
        u64 nslow = 0;
        if (launched) HIPCHK(hipMemcpyAsync(&nslow, d->d_counters + 6, sizeof nslow, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        float ms = 0; hipEventElapsedTime(&ms, d->ev[2], d->ev[3]); d->tm.probe_kernel_ms += ms; d->tm.probe_launches++;
        d->tm.slow_reads += nslow;
#ifdef SAGE2OV_STAMPS
        { u64 st[10]; HIPCHK(hipMemcpy(st, d_stamps, sizeof st, hipMemcpyDeviceToHost)); u64 tot = 0; for (u64 v : st) tot += v;
          fprintf(stderr, "[stamps] kernel %.2f ms; share of wave cycles:", ms); for (int x = 0; x < 10; x++) fprintf(stderr, " %d:%.1f%%", x, 100.0 * (double)st[x] / (double)tot); fprintf(stderr, "\n"); }
#endif
        if (nslow) {                                                          // ambiguous / overflowing reads: sequential state machine
            ProbeArgs B = base_args(d); B.ids = slow; B.n_ids = nslow;
            int rc = launch_probe<0>(d, B, err); if (rc) return rc;
        }
    } else if (nreads) {
        HIPCHK(hipEventRecord(d->ev[2], d->stream));
        int rc = launch_probe<0>(d, A, err); if (rc) return rc;
        HIPCHK(hipEventRecord(d->ev[3], d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        float ms = 0; hipEventElapsedTime(&ms, d->ev[2], d->ev[3]); d->tm.probe_kernel_ms += ms; d->tm.probe_launches++;
    }
    HIPCHK(hipEventRecord(d->ev[1], d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]); d->tm.probe_ms += ms;
    return 0;
}

struct Record { u64 right, left; u32 conn, cflag; };
__global__ void k_pack_records(u64 lo, u64 hi, const u64* right, const u64* left, const u32* conn, const u32* cflag, Record* out) {
    u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= hi) return;
    Record r; r.right = right[i]; r.left = left[i]; r.conn = conn[i]; r.cflag = cflag[i]; out[i - lo] = r;
}
__global__ void k_unpack_records(u64 first, u64 n, const Record* in, u64* right, u64* left, u32* conn, u32* cflag, bool own) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (x >= n) return;
    Record r = in[x]; u64 i = first + x; right[i] = r.right; left[i] = r.left; conn[i] = r.conn;
    (void)own; cflag[i] = r.cflag;
}
// containment flags travel as two byte planes (bit0 plane, bit1 plane) so that a MAX all-reduce is a bitwise OR
__global__ void k_flags_export(u64 n, const u32* __restrict__ cflag, uint8_t* out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    u32 f = cflag[i]; out[i] = f & 1u; out[n + i] = (f >> 1) & 1u;
}
__global__ void k_flags_import(u64 n, const uint8_t* __restrict__ in, u32* cflag) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    cflag[i] = (in[i] ? 1u : 0u) | (in[n + i] ? 2u : 0u);
}
int dev_export_flags(Device* d, void* dst, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    hipLaunchKernelGGL(k_flags_export, dim3(grid_for(d->N + 1, 256)), dim3(256), 0, d->stream, (u64)(d->N + 1), d->cflag, (uint8_t*)dst);
    HIPCHK(hipStreamSynchronize(d->stream)); return 0;
}
int dev_import_flags(Device* d, const void* src, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    hipLaunchKernelGGL(k_flags_import, dim3(grid_for(d->N + 1, 256)), dim3(256), 0, d->stream, (u64)(d->N + 1), (const uint8_t*)src, d->cflag);
    HIPCHK(hipStreamSynchronize(d->stream)); return 0;
}
uint64_t dev_cand_count(Device* d) { return d->n_cand; }
int dev_export_cands(Device* d, void* dst, uint64_t cap, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (d->n_cand > cap) { err = "candidate export buffer too small"; return SAGE2OV_ERR_ARG; }
    if (d->n_cand) HIPCHK(hipMemcpyAsync(dst, d->cand, d->n_cand * sizeof(EdgeCand), hipMemcpyDeviceToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream)); return 0;
}
int dev_set_cands(Device* d, const void* src, uint64_t n, std::string& err) {      // replace the candidate list (device source)
    HIPCHK(hipSetDevice(d->ordinal));
    if (n > d->cand_cap) { hipFree(d->cand); d->cand = nullptr; d->cand_cap = n + 1024; HIPCHK(hipMalloc(&d->cand, d->cand_cap * sizeof(EdgeCand))); }
    if (n) HIPCHK(hipMemcpyAsync(d->cand, src, n * sizeof(EdgeCand), hipMemcpyDeviceToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream)); d->n_cand = n; return 0;
}
int dev_export_records(Device* d, void* dst, uint64_t lo, uint64_t hi, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (hi > lo) hipLaunchKernelGGL(k_pack_records, dim3(grid_for(hi - lo, 256)), dim3(256), 0, d->stream, (u64)lo, (u64)hi, d->right, d->left, d->conn, d->cflag, (Record*)dst);
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}
int dev_import_records(Device* d, const void* src, uint64_t first, uint64_t n, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (n) hipLaunchKernelGGL(k_unpack_records, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, (u64)first, (u64)n, (const Record*)src, d->right, d->left, d->conn, d->cflag, false);
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}

int dev_reciprocal(Device* d, uint64_t emit_lo, uint64_t emit_hi, uint64_t* n_ov, uint64_t* contained, uint64_t* contained_size, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const u64 N = d->N;
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    HIPCHK(hipMemsetAsync(d->d_counters, 0, 8 * sizeof(u64), d->stream));
    HIPCHK(hipMemsetAsync(d->status, 0, N + 1, d->stream));
    hipLaunchKernelGGL(k_recip_cond, dim3(grid_for(N, 256 * COND_PER_THREAD)), dim3(256), 0, d->stream, N, d->right, d->left, d->conn, d->cflag, d->status, d->d_counters);
    if (emit_hi > emit_lo)
        hipLaunchKernelGGL(k_recip_emit, dim3(grid_for(emit_hi - emit_lo, 256 * EMIT_PER_THREAD)), dim3(256), 0, d->stream, N, d->reads, d->S, d->uniL, d->right, d->left, d->status, d->cand, d->cand_cap, d->d_counters, (u64)emit_lo, (u64)emit_hi);
    u64 c[8];
    HIPCHK(hipMemcpyAsync(c, d->d_counters, sizeof c, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipEventRecord(d->ev[1], d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]); d->tm.reciprocal_ms += ms;
    if (c[0] > d->cand_cap) { err = "edge candidate buffer overflow"; return SAGE2OV_ERR_INTERNAL; }
    d->n_cand = c[0]; *n_ov = c[1]; *contained = c[2]; *contained_size = c[3];
    return 0;
}

int dev_download_initial(Device* d, uint64_t* right, uint64_t* left, uint8_t* status, uint32_t* conn, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const u64 N = d->N;
    if (right) HIPCHK(hipMemcpy(right, d->right, (N + 1) * sizeof(u64), hipMemcpyDeviceToHost));
    if (left) HIPCHK(hipMemcpy(left, d->left, (N + 1) * sizeof(u64), hipMemcpyDeviceToHost));
    if (status) HIPCHK(hipMemcpy(status, d->status, N + 1, hipMemcpyDeviceToHost));
    if (conn) HIPCHK(hipMemcpy(conn, d->conn, (N + 1) * sizeof(u32), hipMemcpyDeviceToHost));
    return 0;
}
int dev_download_status(Device* d, std::vector<uint8_t>& status, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    status.resize(d->N + 1);
    HIPCHK(hipMemcpy(status.data(), d->status, d->N + 1, hipMemcpyDeviceToHost));
    return 0;
}

int dev_unresolved_ids(Device* d, std::vector<uint32_t>& ids, std::string& err);
int dev_unresolved_hits(Device* d, std::vector<Hit>& hits, uint64_t* n_unresolved, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const u64 N = d->N;
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    std::vector<uint32_t> ids; int rc = dev_unresolved_ids(d, ids, err); if (rc) return rc;
    u64 nun = ids.size();
    *n_unresolved = nun; hits.clear();
    if (nun == 0) return 0;
    u64 cap = std::max<u64>(1 << 16, nun * 80);
    for (int attempt = 0; attempt < 4; attempt++) {
        WS(dh, Hit, WS_HITS, cap);
        HIPCHK(hipMemsetAsync(d->d_counters + 4, 0, sizeof(u64), d->stream));
        ProbeArgs A = base_args(d); A.hits = dh; A.hits_cap = cap;
        { WS(idbuf, u32, WS_IDS, nun); A.ids = idbuf; A.n_ids = nun; }    // the list dev_unresolved_ids left on the device: only these reads are probed
        rc = launch_probe<1>(d, A, err); if (rc) return rc;
        u64 nh = 0;
        HIPCHK(hipMemcpyAsync(&nh, d->d_counters + 4, sizeof nh, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        if (nh <= cap) {
            hits.resize(nh);
            if (nh) HIPCHK(hipMemcpy(hits.data(), dh, nh * sizeof(Hit), hipMemcpyDeviceToHost));
            HIPCHK(hipEventRecord(d->ev[1], d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
            float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]); d->tm.hits_ms += ms;
            return 0;
        }
        cap = nh + 1024;
    }
    err = "hit buffer sizing failed"; return SAGE2OV_ERR_INTERNAL;
}

// Reduce phase on the device (see k_ra_mark).  *done = 0 when the preconditions do not hold (long buckets, too few
// unresolved reads to be worth it, a list longer than RA_CAP, 32-bit offsets exhausted): the caller then runs the
// serial replay on the host; nothing but the idempotent 0x80 flags has been changed in that case.
int dev_reduce_device(Device* d, uint64_t min_unresolved, uint64_t* n_unresolved, uint64_t* n_hits, uint64_t* inserted, uint64_t* removed, int* done, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    *done = 0; *inserted = 0; *removed = 0; *n_hits = 0;
    const u64 N = d->N;
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    // unresolved reads (status 0), any order
    u32* ids = nullptr; u64 nun = 0;
    {
        u64 cap = 1 << 20;
        for (int attempt = 0; attempt < 2; attempt++) {
            WS(buf, u32, WS_IDS, cap); ids = buf;
            HIPCHK(hipMemsetAsync(d->d_counters + 5, 0, sizeof(u64), d->stream));
            hipLaunchKernelGGL(k_red_unresolved, dim3(grid_for(N, 256 * UNRES_PER_THREAD)), dim3(256), 0, d->stream, (u64)N, d->status, buf, cap, d->d_counters + 5);
            HIPCHK(hipMemcpyAsync(&nun, d->d_counters + 5, sizeof nun, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
            if (nun <= cap) break;
            cap = nun;
        }
    }
    *n_unresolved = nun;
    if (nun == 0) { *done = 1; return 0; }
    if (d->n_long != 0 || nun < min_unresolved) return 0;
    // directional hits of the unresolved reads, device resident: the fast kernel in its hit-list form (locality order, minimiser
    // groups), the sequential kernel for the few reads it hands over (> 128 candidates, ambiguous tags) and for the 16-word layout
    Hit* dh = nullptr; u64 nh = 0, nslots = 0;
    WS(hitcount, u32, WS_RA_CUR, N + 2);
    {
        u32* order = nullptr; { int rc = build_locality_order(d, 1, N + 1, &order, err); if (rc) return rc; }
        WS(slow, u32, WS_SLOW, N + 1);
        const unsigned blocks = (unsigned)std::min<u64>((N + FAST_CHUNK - 1) / FAST_CHUNK, 256ull * 16);
        u64 cap = std::max<u64>(1 << 16, nun * 80) + (u64)blocks * SAGE2OV_FAST_WPB * HITS_CHUNK; bool ok = false;
        if (getenv("SAGE2OV_TEST_SMALL_BUFFERS")) cap = 8192;                      // tests: start far too small, the sizing loop must recover
        for (int attempt = 0; attempt < 4 && !ok; attempt++) {
            WS(hb, Hit, WS_HITS, cap); dh = hb;
            HIPCHK(hipMemsetAsync(d->d_counters + 4, 0, 3 * sizeof(u64), d->stream));
            HIPCHK(hipMemsetAsync(hitcount, 0, (N + 2) * sizeof(u32), d->stream));
            ProbeArgs A = base_args(d); A.lo = 1; A.hi = N + 1; A.hits = dh; A.hits_cap = cap; A.hitcount = hitcount;
            A.ids = order; A.n_ids = N; A.slow = slow; A.slow_cap = N + 1;
            u64 c3[3] = {0, 0, 0};
            if (launch_fast_any<1>(d, A, blocks)) {
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(c3, d->d_counters + 4, sizeof c3, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
                if (c3[0] > cap) { cap = c3[0] + c3[0] / 8 + 1024; continue; }            // some chunk did not fit: everything again
                if (c3[2]) {                                                                  // handed over: exact sequential kernel, appends behind
                    ProbeArgs B = base_args(d); B.hits = dh; B.hits_cap = cap; B.hitcount = hitcount; B.ids = slow; B.n_ids = c3[2];
                    int rc = launch_probe<1>(d, B, err); if (rc) return rc;
                }
            } else {
                A.ids = nullptr; A.n_ids = 0; A.slow = nullptr;
                int rc = launch_probe<1>(d, A, err); if (rc) return rc;
            }
            u64 used = 0;
            HIPCHK(hipMemcpyAsync(&used, d->d_counters + 4, sizeof used, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
            if (used > cap) { cap = used + used / 8 + 1024; continue; }
            nslots = used; nh = c3[1] + (used - c3[0]); ok = true;                           // real hits: fast kernel's count + what the sequential kernel appended
        }
        if (!ok) { err = "hit buffer sizing failed"; return SAGE2OV_ERR_INTERNAL; }
    }
    *n_hits = nh;
    const u64 nc = d->n_cand;
    WS(deg, u32, WS_RA_DEG, N + 2); WS(offs, u32, WS_RA_OFF, N + 2); WS(cur, u32, WS_CURSOR, N + 2);
    HIPCHK(hipMemsetAsync(deg, 0, (N + 2) * sizeof(u32), d->stream)); HIPCHK(hipMemsetAsync(cur, 0, (N + 2) * sizeof(u32), d->stream));
    if (nc) hipLaunchKernelGGL(k_ra_degree_c, dim3(grid_for(nc, 256)), dim3(256), 0, d->stream, d->cand, (u64)nc, d->status, deg);
    hipLaunchKernelGGL(k_ra_degree_h, dim3(grid_for(N + 1, 256)), dim3(256), 0, d->stream, hitcount, (u64)N, deg);
    u64 tot = 0; { int rc = scan_u32(d, deg, N + 2, offs, &tot, err); if (rc) return rc; }
    if (tot >= (1ull << 32) - 64) return 0;
    WS(ent, u64, WS_RA_ENT, tot + 64); WS(rm, uint8_t, WS_RA_RM, tot + 64);
    if (nc) hipLaunchKernelGGL(k_ra_fill_c, dim3(grid_for(nc, 256)), dim3(256), 0, d->stream, d->cand, (u64)nc, d->reads, d->S, d->uniL, offs, cur, ent);
    if (nslots) hipLaunchKernelGGL(k_ra_fill_h, dim3(grid_for(nslots, 256)), dim3(256), 0, d->stream, dh, (u64)nslots, offs, deg, hitcount, ent);
    HIPCHK(hipMemsetAsync(d->d_counters + 8, 0, 4 * sizeof(u64), d->stream));
    WS(svn, u32, WS_NEED, nun + 2); WS(svoff, u32, WS_OWNER, nun + 2);
    const unsigned gb = (unsigned)std::min<u64>((nun + 3) / 4, 256ull * 16);
    hipLaunchKernelGGL(k_ra_mark, dim3(gb), dim3(256), 0, d->stream, ids, (u64)nun, offs, deg, ent, rm, svn, d->d_counters + 8);
    u64 nsv = 0; { int rc = scan_u32(d, svn, nun, svoff, &nsv, err); if (rc) return rc; }
    u64 c[2];
    HIPCHK(hipMemcpyAsync(c, d->d_counters + 8, sizeof c, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
    HIPCHK(hipGetLastError());
    if (c[0] != 0) return 0;                                              // a list does not fit the device kernel: serial replay
    if (d->n_cand + nsv > d->cand_cap || getenv("SAGE2OV_TEST_SMALL_BUFFERS")) {
        EdgeCand* ncand = nullptr; const u64 ncap = d->n_cand + nsv + 1024;
        HIPCHK(hipMalloc(&ncand, ncap * sizeof(EdgeCand)));
        HIPCHK(hipMemcpyAsync(ncand, d->cand, d->n_cand * sizeof(EdgeCand), hipMemcpyDeviceToDevice, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
        hipFree(d->cand); d->cand = ncand; d->cand_cap = ncap;
    }
    hipLaunchKernelGGL(k_ra_emit, dim3(gb), dim3(256), 0, d->stream, ids, (u64)nun, offs, deg, ent, rm, svoff, d->cand, (u64)d->n_cand, (u64)d->cand_cap);
    HIPCHK(hipGetLastError());
    d->n_cand += nsv;
    *inserted = nh; *removed = c[1]; *done = 1;
    HIPCHK(hipEventRecord(d->ev[1], d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]); d->tm.hits_ms += ms;
    return 0;
}

// candidates that touch an unresolved read or one of its neighbours (their adjacency lists are what
// markTransitiveEdge reads, economyGraph.cpp:643-679); candidates owned by unresolved reads are flagged
// as dropped on the device: the replay re-emits the survivors.
int dev_collect_reduce_edges(Device* d, std::vector<EdgeCand>& out, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    out.clear(); const u64 n = d->n_cand; if (n == 0) return 0;
    EdgeCand* buf = nullptr; u64 cap = 1 << 16;
    WS(need, uint8_t, WS_NEED, d->N + 1); HIPCHK(hipMemsetAsync(need, 0, d->N + 1, d->stream));
    hipLaunchKernelGGL(k_red_mark, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, d->cand, (u64)n, d->status, need);
    // first a dry count (cap 0 keeps the flagging idempotent), then the real collection
    HIPCHK(hipMemsetAsync(d->d_counters + 5, 0, sizeof(u64), d->stream));
    hipLaunchKernelGGL(k_red_collect, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, d->cand, (u64)n, d->status, need, (EdgeCand*)nullptr, (u64)0, d->d_counters + 5);
    u64 cnt = 0; HIPCHK(hipMemcpyAsync(&cnt, d->d_counters + 5, sizeof cnt, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
    if (cnt) {
        cap = cnt; { WS(nb_, EdgeCand, WS_NEAR, cap); buf = nb_; }
        HIPCHK(hipMemsetAsync(d->d_counters + 5, 0, sizeof(u64), d->stream));
        hipLaunchKernelGGL(k_red_collect, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, d->cand, (u64)n, d->status, need, buf, cap, d->d_counters + 5);
        HIPCHK(hipStreamSynchronize(d->stream));
        out.resize(cnt); HIPCHK(hipMemcpy(out.data(), buf, cnt * sizeof(EdgeCand), hipMemcpyDeviceToHost));
        for (auto& e : out) e.type &= 0x7Fu;
    }
    return 0;
}
int dev_unresolved_ids(Device* d, std::vector<uint32_t>& ids, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    ids.clear(); u64 cap = 1 << 20; 
    for (int attempt = 0; attempt < 2; attempt++) {
        WS(buf, u32, WS_IDS, cap);
        HIPCHK(hipMemsetAsync(d->d_counters + 5, 0, sizeof(u64), d->stream));
        hipLaunchKernelGGL(k_red_unresolved, dim3(grid_for(d->N, 256 * UNRES_PER_THREAD)), dim3(256), 0, d->stream, (u64)d->N, d->status, buf, cap, d->d_counters + 5);
        u64 cnt = 0; HIPCHK(hipMemcpyAsync(&cnt, d->d_counters + 5, sizeof cnt, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
        if (cnt <= cap) { ids.resize(cnt); if (cnt) HIPCHK(hipMemcpy(ids.data(), buf, cnt * sizeof(u32), hipMemcpyDeviceToHost)); std::sort(ids.begin(), ids.end()); return 0; }
        cap = cnt;
    }
    err = "unresolved id collection failed"; return SAGE2OV_ERR_INTERNAL;
}

int dev_debug_table(Device* d, uint64_t* out5, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    u64* dk = nullptr; HIPCHK(hipMalloc(&dk, 8 * sizeof(u64))); HIPCHK(hipMemsetAsync(dk, 0, 8 * sizeof(u64), d->stream));
    hipLaunchKernelGGL(k_debug_table, dim3(grid_for(d->T, 256)), dim3(256), 0, d->stream, d->slots, (u64)d->T, dk);
    HIPCHK(hipStreamSynchronize(d->stream));
    HIPCHK(hipMemcpy(out5, dk, 5 * sizeof(u64), hipMemcpyDeviceToHost)); hipFree(dk); return 0;
}
int dev_debug_keys(Device* d, uint64_t* out, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    u64* dk = nullptr; HIPCHK(hipMalloc(&dk, 8 * d->N * sizeof(u64)));
    hipLaunchKernelGGL(k_debug_keys, dim3(grid_for(4 * d->N, 256)), dim3(256), 0, d->stream, d->reads, (u64)d->N, d->S, d->h, dk);
    HIPCHK(hipStreamSynchronize(d->stream));
    HIPCHK(hipMemcpy(out, dk, 8 * d->N * sizeof(u64), hipMemcpyDeviceToHost));
    hipFree(dk); return 0;
}
// diagnostic: hit lists (economyGraph.cpp:591-633 semantics) of EVERY read, as if all were unresolved
int dev_debug_all_hits(Device* d, std::vector<Hit>& hits, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const u64 N = d->N; uint8_t* saved = nullptr;
    HIPCHK(hipMalloc(&saved, N + 1)); HIPCHK(hipMemcpy(saved, d->status, N + 1, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemset(d->status, 0, N + 1));
    u64 cap = std::max<u64>(1 << 16, N * 128); Hit* dh = nullptr;
    HIPCHK(hipMalloc(&dh, cap * sizeof(Hit)));
    HIPCHK(hipMemsetAsync(d->d_counters + 4, 0, sizeof(u64), d->stream));
    ProbeArgs A = base_args(d); A.lo = 1; A.hi = N + 1; A.hits = dh; A.hits_cap = cap;
    int rc = launch_probe<1>(d, A, err);
    u64 nh = 0;
    if (!rc) { HIPCHK(hipMemcpyAsync(&nh, d->d_counters + 4, sizeof nh, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream)); }
    if (!rc && nh > cap) { err = "debug hit buffer too small"; rc = SAGE2OV_ERR_LIMIT; }
    if (!rc) { hits.resize(nh); if (nh) HIPCHK(hipMemcpy(hits.data(), dh, nh * sizeof(Hit), hipMemcpyDeviceToHost)); }
    HIPCHK(hipMemcpy(d->status, saved, N + 1, hipMemcpyDeviceToDevice));
    hipFree(saved); hipFree(dh);
    return rc;
}

int dev_append_edges(Device* d, const EdgeCand* e, uint64_t n, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (n == 0) return 0;
    if (d->n_cand + n > d->cand_cap) {
        EdgeCand* nc = nullptr; u64 ncap = d->n_cand + n + 1024;
        HIPCHK(hipMalloc(&nc, ncap * sizeof(EdgeCand)));
        HIPCHK(hipMemcpy(nc, d->cand, d->n_cand * sizeof(EdgeCand), hipMemcpyDeviceToDevice));
        hipFree(d->cand); d->cand = nc; d->cand_cap = ncap;
    }
    HIPCHK(hipMemcpy(d->cand + d->n_cand, e, n * sizeof(EdgeCand), hipMemcpyHostToDevice));
    d->n_cand += n;
    return 0;
}

static int scan_u32(Device* d, const u32* in, u64 n, u32* out, u64* total, std::string& err) {
    const u64 nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    WS(partial, u64, WS_PARTIAL, nb + 1);
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, d->stream, in, (u64)n, partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, d->stream, partial, (u64)nb, partial + nb);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, d->stream, in, (u64)n, partial, out);
    HIPCHK(hipMemcpyAsync(total, partial + nb, sizeof(u64), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}

int dev_download_edges(Device* d, std::vector<FinalEdge>& out, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    out.resize(d->n_final);
    if (d->n_final) HIPCHK(hipMemcpy(out.data(), d->final_edges, d->n_final * sizeof(FinalEdge), hipMemcpyDeviceToHost));
    return 0;
}
int dev_convert(Device* d, uint64_t* n_final, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const u64 N = d->N, n = d->n_cand;
    d->final_edges = nullptr; d->n_final = 0;
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    if (n) {
        if (n >= (1ull << 32)) { err = "too many edge candidates"; return SAGE2OV_ERR_LIMIT; }
        WS(deg, u32, WS_DEG, N + 2); WS(offs, u32, WS_OFFS, N + 2); WS(cursor, u32, WS_CURSOR, N + 2);
        WS(keys, u64, WS_KEYS, n); WS(keep, u32, WS_KEEP, n); WS(pos, u32, WS_POS, n); WS(owner, u32, WS_OWNER, n);
        HIPCHK(hipMemsetAsync(deg, 0, (N + 2) * sizeof(u32), d->stream)); HIPCHK(hipMemsetAsync(cursor, 0, (N + 2) * sizeof(u32), d->stream));
        hipLaunchKernelGGL(k_conv_degree, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, d->cand, (u64)n, deg);
        u64 tot = 0; int rc = scan_u32(d, deg, N + 1, offs, &tot, err); if (rc) return rc;
        hipLaunchKernelGGL(k_conv_fill, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, d->cand, (u64)n, offs, cursor, keys);
        hipLaunchKernelGGL(k_conv_sort, dim3(grid_for(N + 1, 256)), dim3(256), 0, d->stream, (u64)N, offs, deg, keys, keep);
        hipLaunchKernelGGL(k_conv_owner, dim3(grid_for(N + 1, 256)), dim3(256), 0, d->stream, (u64)N, offs, deg, owner);
        // `tot` = candidates still alive (dropped ones were never filled in): only keys[0..tot) are defined
        u64 nf = 0;
        if (tot) {
            rc = scan_u32(d, keep, tot, pos, &nf, err); if (rc) return rc;
            { WS(fe, FinalEdge, WS_FINAL, std::max<u64>(1, nf)); d->final_edges = fe; }
            hipLaunchKernelGGL(k_conv_emit, dim3(grid_for(tot, 256)), dim3(256), 0, d->stream, (u64)tot, keys, keep, pos, owner, d->reads, d->S, d->uniL, d->final_edges);
        }
        d->n_final = nf;
        HIPCHK(hipStreamSynchronize(d->stream));
    }
    HIPCHK(hipEventRecord(d->ev[1], d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]); d->tm.convert_ms += ms;
    *n_final = d->n_final;
    return 0;
}

}  // namespace s2
