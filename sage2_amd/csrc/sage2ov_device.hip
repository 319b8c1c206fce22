// sage2_amd/csrc/sage2ov_device.hip -- hand-written HIP kernels for gfx950 (MI355X) and their launchers.
//
// The kernels live in kernels_*.inc (one file per phase of the path, included below in dependency order: common helpers,
// locality order, step 1, index build, scan, sequential probe, fast probe, reciprocal pass, device reduce, convert); this file
// holds the device state, the workspace arena and the host-side launchers.
// Integer / bit / index work only (no MFMA): everything here is bound by HBM gathers.  Design notes
// (layouts, algorithmic bytes, rooflines) are in DESIGN.md; reference citations are relative to the
// SAGE2 tree.  Wavefront = 64 lanes everywhere; one wavefront owns one read in the probe kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <memory>
#include <atomic>
#include <thread>
#include <sched.h>
#include <pthread.h>
#include <sys/mman.h>
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

enum { WS_SLOTS, WS_CNT, WS_WHERE, WS_BIG, WS_CSR, WS_SLOW, WS_NEED, WS_DEG, WS_OFFS, WS_CURSOR, WS_KEYS, WS_KEEP, WS_POS, WS_OWNER, WS_FINAL,
       WS_PARTIAL, WS_IDS, WS_NEAR, WS_HITS, WS_MINH, WS_OCNT, WS_OOFF, WS_OCUR, WS_ORDER, WS_MI1, WS_MICNT, WS_MICUR, WS_KREC, WS_SLOTMH, WS_RA_DEG, WS_RA_OFF, WS_RA_CUR, WS_RA_ENT, WS_RA_RM, WS_ORG_POOL, WS_ORG_OFF, WS_ORG_LEN, WS_ORG_IMG, WS_ORG_K0, WS_ORG_K1, WS_ORG_V0, WS_ORG_V1, WS_ORG_HIST, WS_ORG_HSCAN, WS_ORG_FLAG, WS_ORG_UID, WS_ORG_HEAD, WS_RR_IN, WS_RR_DEGP, WS_RR_OFFP, WS_RR_ENTP, WS_RR_OUTP, WS_RR_WIDX, WS_RR_RANK, WS_RR_CUR, WS_RA_HEAVY, WS_RA_HSIZE, WS_RA_HSCR, WS_RR_LEN, WS_RR_COFF, WS_RR_OUTC,
       WS_PT_K0, WS_PT_K1, WS_PT_P0, WS_PT_P1, WS_PT_M0, WS_PT_M1, WS_PT_CNT, WS_PT_BASE, WS_PT_OFF, WS_PT_GOFF, WS_RR_LOC, WS_RR_RANKL, WS_LOC_READS, WS_LOC_IDOF, WS_LOC_POSOF, WS_LOC_STATUS, WS_ORG_GFLAG, WS_ORG_GPOS, WS_LOC_META, WS_SLOW2, WS_RA_ENT32, WS_RA_SPLIT, WS_PRE_BASE, WS_PRE_NONE, WS_COUNT };   // ids of the workspace arena (Device::ws)
struct Device {
    int ordinal = 0;
    Options opt;                 // the environment switches, as the context read them when it was created (sage2ov_internal.h)
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    // reads
    u64 N = 0; int S = 0, maxL = 0, k = 0, h = 0; double probeShare = 1.0;
    u64* d_runStarts = nullptr; u64* h_runStarts = nullptr; hipEvent_t evRunStarts = nullptr; bool runStartsValid = false; double runStartFrac = 0.0;      // share of the reads without a predecessor in the locality order (counted by k_loc_index into d_runStarts)
    u64* reads = nullptr;        // (N+1)*S words, slot i = read id i
    // the same reads in LOCALITY order (slot p = the read at position p of the order by global minimiser): what the index entries point at and
    // what the probe kernels gather from -- a read's overlap partners are neighbours in the genome, hence (mostly) neighbours here
    u64* readsLoc = nullptr; u32* idOf = nullptr; u32* posOf = nullptr; uint8_t* statusP = nullptr; unsigned short* metaP = nullptr;
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
    u64 memLow = ~0ull;          // lowest free device memory seen at the sampling points (mem_sample)
    // Memory-diet mode (round 3, BASELINE configs[4]: a billion reads per GPU): transient buffers are released at the end of their phase instead of kept for
    // the next step, the id-ordered read store is released once the locality-ordered one exists (uniform read length only; rebuilt for the next index
    // build), no minimiser groups, the per-read result arrays and the candidate list are allocated when they are first needed.  On when the whole
    // footprint of the default mode (~640 bytes per read) would pass 70 % of the device's memory, or SAGE2OV_MEMORY_DIET=1 (0 forbids it).
    bool diet = false;
    // hits written out by the initial pass (k_probe_fast<..., TAIL = 2> on noisy data, one context probing everything): the reduce phase filters them
    struct PreHits { bool valid = false; Hit* hits = nullptr; u64 cap = 0, used = 0; u64* base = nullptr; } pre;
    void* s4keep = nullptr;      // step 4: the simplified graph stays in HBM until the next call (S4Keep)
    // step 4: blocks a call released, kept for the next call (it asks for the same ~50 sizes; fresh HBM costs ~50 ms per GB to map, 0.7 s per call at BASELINE
    // configs[2]).  Emptied when the read set changes, when the workspace arena runs out of memory, and never filled in memory-diet mode.
    std::vector<std::pair<void*, size_t>> s4cache;
    void* rrStaging = nullptr;   // ranked reduce: the pinned 2 MB-page host buffers the potential lists are downloaded into, kept from step to step (RrStaging)
    // workspace arena: buffers of the timed path are allocated once and only ever grow (no hipMalloc/hipFree per step)
    struct Buf { void* p = nullptr; size_t cap = 0; u32 epoch = 0; };
    Buf ws[WS_COUNT];
    // MEMORY-DIET MODE (read sets that fill the HBM): the index build's sort buffers and the results of a step (records, candidates, final edges, convert's sort buffers)
    // take turns with the memory.  Until round 4 each side was hipFree'd when the other one came and hipMalloc'ed again a step later -- 3.4 of the 3.7 s of an index build
    // at 1.02 G reads, and erratic (tools/ubench/mempool.hip: the same 80 GB come back in 0.5 ms or in 5 s depending on their shape).  Now both sides are CARVED out of one
    // block that stays: phase A = the index build, phase B = everything from the probe pass to the next build; a phase starts with an empty block (ph_begin: the callers
    // have synchronised the stream, as they did in front of hipFree).  The first step sizes the block (what does not fit yet gets blocks of its own for that phase);
    // from the second step on nothing is allocated.
    struct Phase { char* blk = nullptr; size_t cap = 0, used = 0, virt = 0, need = 0; u32 epoch = 1; std::vector<void*> overflow; } ph;
    bool resCarved = false;      // right / left / conn / cflag / status / cand point into the phase block (diet mode): never hipFree'd
};
static void s4cache_release(Device* d) { for (auto& c : d->s4cache) hipFree(c.first); d->s4cache.clear(); }
static bool ws_phased(int id);
static void* ph_carve(Device* d, size_t bytes);
static void* ws_get(Device* d, int id, size_t bytes) {
    Device::Buf& b = d->ws[id];
    if (d->diet && ws_phased(id)) {
        if (b.p && b.epoch == d->ph.epoch && b.cap >= bytes) return b.p;
        if (b.p && !b.epoch) { hipFree(b.p); }                                 // (allocated before the context went on the diet)
        b.p = ph_carve(d, bytes + 256); b.cap = b.p ? bytes + 256 : 0; b.epoch = b.p ? d->ph.epoch : 0;
        return b.p;
    }
    if (b.epoch) { b.p = nullptr; b.cap = 0; b.epoch = 0; }                   // (a carved piece from an earlier diet: not ours to free)
    if (b.cap < bytes || !b.p) {
        if (b.p) hipFree(b.p);
        b.p = nullptr; b.cap = 0;
        size_t want = d->diet ? bytes + 256 : bytes + bytes / 16 + 256;        // (grow-only arena: 6 % of slack saves re-allocations; none when memory is what is short)
        if (hipMalloc(&b.p, want) != hipSuccess) {
            (void)hipGetLastError(); b.p = nullptr;
            if (d->s4cache.empty()) return nullptr;
            s4cache_release(d);                                                       // (step 4's spare blocks go first)
            if (hipMalloc(&b.p, want) != hipSuccess) { b.p = nullptr; return nullptr; }
        }
        b.cap = want;
    }
    return b.p;
}
// workspace ids that live in the phase block in diet mode (see Device::Phase)
static bool ws_phased(int id) {
    switch (id) { case WS_PT_K0: case WS_PT_K1: case WS_WHERE: case WS_MINH: case WS_OCUR: case WS_PT_CNT: case WS_PT_BASE: case WS_PARTIAL: case WS_PT_OFF:
                  case WS_KEYS: case WS_KEEP: case WS_POS: case WS_FINAL: case WS_SLOW: case WS_SLOW2: return true; default: return false; }
}
static void* ph_carve(Device* d, size_t bytes) {
    Device::Phase& P = d->ph;
    const size_t a = (P.virt + 255) & ~(size_t)255; P.virt = a + bytes; P.need = std::max(P.need, P.virt);
    const size_t off = (P.used + 255) & ~(size_t)255;
    if (P.blk && off + bytes <= P.cap) { P.used = off + bytes; return P.blk + off; }
    void* p = nullptr;                                                     // (first step, or a phase that outgrew the block: a block of its own until the phase ends)
    if (hipMalloc(&p, std::max<size_t>(bytes, 256)) != hipSuccess) {
        (void)hipGetLastError(); s4cache_release(d);
        if (P.blk && P.used == 0) { hipFree(P.blk); P.blk = nullptr; P.cap = 0; }      // (a block too small for this phase and not in use yet stands in the way)
        if (hipMalloc(&p, std::max<size_t>(bytes, 256)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    P.overflow.push_back(p);
    return p;
}
// a new phase: everything carved so far is dead (the caller has synchronised the stream); the block grows to what the largest phase so far needed
static void ph_begin(Device* d, bool mayGrow = true) {
    Device::Phase& P = d->ph;
    for (void* p : P.overflow) hipFree(p);
    P.overflow.clear();
    if (mayGrow && P.need > P.cap) { if (P.blk) hipFree(P.blk); P.blk = nullptr; P.cap = 0; const size_t want = P.need + P.need / 64 + 4096; void* p = nullptr;
                          if (hipMalloc(&p, want) == hipSuccess) { P.blk = (char*)p; P.cap = want; } else (void)hipGetLastError(); }
    P.used = 0; P.virt = 0; P.epoch++;
    for (int id = 0; id < WS_COUNT; id++) if (ws_phased(id)) { Device::Buf& b = d->ws[id]; if (b.epoch) { b.p = nullptr; b.cap = 0; b.epoch = 0; } }
    if (d->resCarved) { d->right = d->left = nullptr; d->conn = d->cflag = nullptr; d->status = nullptr; d->cand = nullptr; d->cand_cap = 0; d->n_cand = 0; d->resCarved = false; }
    d->final_edges = nullptr; d->n_final = 0;
}
static void ph_release(Device* d) {
    Device::Phase& P = d->ph;
    for (void* p : P.overflow) hipFree(p);
    P.overflow.clear(); if (P.blk) hipFree(P.blk); P.blk = nullptr; P.cap = P.used = P.virt = P.need = 0; P.epoch++;
    for (int id = 0; id < WS_COUNT; id++) { Device::Buf& b = d->ws[id]; if (b.epoch) { b.p = nullptr; b.cap = 0; b.epoch = 0; } }
    if (d->resCarved) { d->right = d->left = nullptr; d->conn = d->cflag = nullptr; d->status = nullptr; d->cand = nullptr; d->cand_cap = 0; d->resCarved = false; }
}
static void ws_free(Device* d, int id) { Device::Buf& b = d->ws[id]; if (b.epoch) return;      // (carved: it goes with its phase)
                                         if (b.p) hipFree(b.p); b.p = nullptr; b.cap = 0; }
// the edge candidates: a block of their own, or a piece of the phase block in diet mode (then the old piece is simply left behind)
static int cand_resize(Device* d, u64 ncap, u64 keep, std::string& err) {
    EdgeCand* nc = nullptr;
    if (d->diet) { nc = (EdgeCand*)ph_carve(d, ncap * sizeof(EdgeCand)); if (!nc) { err = "edge candidates: out of device memory"; return SAGE2OV_ERR_NOMEM; } }
    else HIPCHK(hipMalloc(&nc, ncap * sizeof(EdgeCand)));
    if (keep && d->cand) { HIPCHK(hipMemcpyAsync(nc, d->cand, keep * sizeof(EdgeCand), hipMemcpyDeviceToDevice, d->stream)); HIPCHK(hipStreamSynchronize(d->stream)); }
    if (d->cand && !d->resCarved) hipFree(d->cand);
    d->cand = nc; d->cand_cap = ncap; if (d->diet) d->resCarved = true;
    return 0;
}
static void decide_diet(Device* d) {
    size_t fr = 0, to = 0; d->diet = false;
    if (hipMemGetInfo(&fr, &to) == hipSuccess && to) d->diet = (double)(d->N + 1) * 640.0 > 0.7 * (double)to;
    if (const char* ev = d->opt.get("SAGE2OV_MEMORY_DIET")) d->diet = atoi(ev) != 0;
}
static void mem_sample(Device* d) { size_t fr = 0, to = 0; if (hipMemGetInfo(&fr, &to) == hipSuccess) d->memLow = std::min<u64>(d->memLow, (u64)fr); }
#define WS(var, type, id, count)                                                                     \
    type* var = (type*)ws_get(d, id, (size_t)(count) * sizeof(type));                                \
    if (!var) { err = std::string("workspace allocation failed: ") + #id; return SAGE2OV_ERR_NOMEM; }

#include "kernels_common.inc"
#include "kernels_order.inc"
#include "kernels_organize.inc"
#include "kernels_partition.inc"
#include "kernels_index.inc"
#include "kernels_scan.inc"
#include "kernels_probe_seq.inc"
#include "kernels_probe_fast.inc"
#include "kernels_reciprocal.inc"
#include "kernels_reduce.inc"
#include "kernels_convert.inc"
#include "kernels_simplify.inc"

// =============================================================================================
// host-side launchers
// =============================================================================================
// one thread per item.  A launch of 2^32 threads or more does NOT fail on this runtime: its size is taken modulo 2^32 and the tail of the work silently
// never runs (round 3, 1.02 G reads).  grid_for is therefore only for item counts that the documented limits keep below 2^32 (asserted in debug
// builds by the limit checks of the callers: reads < 2^30, tuples < 2^32); kernels whose item count can pass it are grid-stride and use grid_for_capped.
static inline unsigned grid_for(u64 n, unsigned block) { return (unsigned)std::max<u64>(1, (n + block - 1) / block); }
static inline unsigned grid_for_capped(u64 n, unsigned block) { return (unsigned)std::max<u64>(1, std::min<u64>((n + block - 1) / block, ((1ull << 32) / block) - 1)); }

void dev_set_options(Device* d, const Options& opt) { d->opt = opt; }
Device* dev_create(int ordinal, const Options& opt, std::string& err) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) { err = std::string("no HIP device available: ") + hipGetErrorString(e); return nullptr; }
    int dev = ordinal;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= count) { err = "device ordinal out of range"; return nullptr; }
    if (hipSetDevice(dev) != hipSuccess) { err = "hipSetDevice failed"; return nullptr; }
    Device* d = new Device(); d->ordinal = dev; d->opt = opt;
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; delete d; return nullptr; }
    { const char* tb = d->opt.get("SAGE2OV_TEST_TAG_BITS"); const int nb = tb ? atoi(tb) : 24;
      const u32 mask = (nb >= 1 && nb < 24) ? ((1u << nb) - 1u) : 0xFFFFFFu;
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_tag_mask), &mask, sizeof mask) != hipSuccess) { err = "tag mask upload failed"; delete d; return nullptr; }
      const char* mb = d->opt.get("SAGE2OV_TEST_MTAG_BITS"); const int nm = mb ? atoi(mb) : 24;
      const u32 mmask = (nm >= 1 && nm < 24) ? ((1u << nm) - 1u) : 0xFFFFFFu;
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_mtag_mask), &mmask, sizeof mmask) != hipSuccess) { err = "tag mask upload failed"; delete d; return nullptr; }
      const char* fb = d->opt.get("SAGE2OV_TEST_FP_BITS"); const int nf = fb ? atoi(fb) : 20;
      const u32 fmask = (nf >= 0 && nf < 20) ? ((1u << nf) - 1u) : 0xFFFFFu;
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_fp_mask), &fmask, sizeof fmask) != hipSuccess) { err = "fingerprint mask upload failed"; delete d; return nullptr; } }
    for (auto& ev : d->ev) hipEventCreate(&ev);
    if (hipMalloc(&d->d_counters, (24 + 64) * sizeof(u64)) != hipSuccess) { err = "hipMalloc(counters) failed"; delete d; return nullptr; }
    hipMemset(d->d_counters, 0, (24 + 64) * sizeof(u64));
    d->d_runStarts = d->d_counters + 24;
    if (hipHostMalloc((void**)&d->h_runStarts, 64 * sizeof(u64), hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&d->evRunStarts, hipEventDisableTiming) != hipSuccess) {
        err = "hipHostMalloc / hipEventCreate failed"; hipFree(d->d_counters); delete d; return nullptr; }
    return d;
}
static void rr_staging_release(Device* d);
static void free_reads(Device* d) {
    rr_staging_release(d);                     // (pinned host buffers sized by the previous read set)
    hipFree(d->reads);
    if (!d->resCarved) { hipFree(d->right); hipFree(d->left); hipFree(d->conn); hipFree(d->cflag); hipFree(d->status); hipFree(d->cand); }     // slots / csr / final_edges live in the workspace arena
    ph_release(d);                                                              // (diet mode: the phase block and what was carved out of it)
    for (auto& b : d->ws) { if (b.p && !b.epoch) hipFree(b.p); b.p = nullptr; b.cap = 0; b.epoch = 0; }
    s4cache_release(d);
    d->runStartFrac = 0.0; d->runStartsValid = false;          // (measured on the read set that just went: a new one decides for itself)
    d->readsLoc = nullptr; d->idOf = d->posOf = nullptr; d->statusP = nullptr; d->metaP = nullptr; d->mi1 = d->krec = nullptr; d->cand_cap = 0; d->n_cand = 0;
    d->reads = d->slots = nullptr; d->csr = nullptr; d->right = d->left = nullptr; d->conn = d->cflag = nullptr; d->status = nullptr; d->cand = nullptr; d->final_edges = nullptr;
}
void dev_destroy(Device* d) {
    if (!d) return;
    hipSetDevice(d->ordinal);
    hipStreamSynchronize(d->stream);
    dev_simplify_release(d); rr_staging_release(d);
    free_reads(d); hipFree(d->d_counters); if (d->h_runStarts) hipHostFree(d->h_runStarts); if (d->evRunStarts) hipEventDestroy(d->evRunStarts);
    for (auto& ev : d->ev) if (ev) hipEventDestroy(ev);
    hipStreamDestroy(d->stream);
    delete d;
}
void* dev_stream(Device* d) { return (void*)d->stream; }
int dev_sync(Device* d, std::string& err) { HIPCHK(hipStreamSynchronize(d->stream)); return 0; }
void dev_timings(Device* d, DevTimings* t) { *t = d->tm; }
void dev_reset_timings(Device* d) { d->tm = DevTimings(); }

// per-read results + candidate list: allocated when the probe pass first needs them (after the index build's transients are gone in diet mode)
static int ensure_results(Device* d, std::string& err) {
    const u64 N = d->N;
    if (!d->right && d->diet) {                                                // (pieces of the phase block: Device::Phase)
        d->right = (u64*)ph_carve(d, (N + 1) * sizeof(u64)); d->left = (u64*)ph_carve(d, (N + 1) * sizeof(u64));
        d->conn = (u32*)ph_carve(d, (N + 1) * sizeof(u32)); d->cflag = (u32*)ph_carve(d, (N + 1) * sizeof(u32)); d->status = (uint8_t*)ph_carve(d, N + 1);
        d->resCarved = true;
        if (!d->right || !d->left || !d->conn || !d->cflag || !d->status) { err = "result buffers: out of device memory"; return SAGE2OV_ERR_NOMEM; }
    }
    if (!d->right) { HIPCHK(hipMalloc(&d->right, (N + 1) * sizeof(u64))); HIPCHK(hipMalloc(&d->left, (N + 1) * sizeof(u64)));
                     HIPCHK(hipMalloc(&d->conn, (N + 1) * sizeof(u32))); HIPCHK(hipMalloc(&d->cflag, (N + 1) * sizeof(u32))); HIPCHK(hipMalloc(&d->status, (N + 1))); }
    if (!d->cand && !d->diet) { d->cand_cap = 2 * N + 1024; HIPCHK(hipMalloc(&d->cand, d->cand_cap * sizeof(EdgeCand))); }      // (diet: sized by a counting pass of the reciprocal kernel)
    return 0;
}
int dev_upload_reads(Device* d, const uint64_t* words, uint64_t N, int S, int minL, int maxL, int k, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (S != 4 && S != 8 && S != 16 && S != 32) { err = "unsupported words-per-read (read length limit is 1018 bases)"; return SAGE2OV_ERR_LIMIT; }
    if (N >= (1ull << 30)) { err = "more than 2^30-1 unique reads per context is not supported yet"; return SAGE2OV_ERR_LIMIT; }
    free_reads(d);
    d->N = N; d->S = S; d->maxL = maxL; d->k = k; d->h = k > 64 ? 64 : k; d->uniL = (N && minL == maxL) ? maxL : 0;
    HIPCHK(hipMalloc(&d->reads, (N + 1) * S * sizeof(u64)));
    HIPCHK(hipMemcpyAsync(d->reads, words, (N + 1) * S * sizeof(u64), hipMemcpyHostToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    decide_diet(d);
    if (!d->diet) { int rc = ensure_results(d, err); if (rc) return rc; }
    return 0;
}

static int scan_u32(Device* d, const u32* in, u64 n, u32* out, u64* total, std::string& err);
struct PtBufs { u32* E[2]; int W; const u32* src0 = nullptr; };      // two buffers of n tuples of W dwords each (kernels_partition.inc); src0: the first pass reads the tuples there (left untouched) and writes E[0]
static int partition_by_window(Device* d, PtBufs& B, int keyw, u32 n, int shiftW, u64 nWin, bool digit0Counted, u32* cnt, u32* base, u32* off, int* cur_out, std::string& err);
// Step 1 on the device: see k_org_canon.  On return the read store is resident exactly as after dev_upload_reads, and the
// host receives the image (for the .reads writer, lengths) and the frequencies.
int dev_organize_reads(Device* d, const uint64_t* pool, uint64_t pool_words, const uint64_t* off, const uint16_t* len, uint64_t n, int S, int minL, int maxL, int k,
                       uint64_t* N_out, RawU64& words_out, RawU16& freq_out, std::string& err, OrgAscii* ascii) {
    HIPCHK(hipSetDevice(d->ordinal));
    struct EvPair { hipEvent_t a = nullptr, b = nullptr; ~EvPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); } } evp;   // (destroyed on every return path)
    HIPCHK(hipEventCreate(&evp.a)); HIPCHK(hipEventCreate(&evp.b));
    const hipEvent_t e0 = evp.a, e1 = evp.b;
    const bool timing = d->opt.get("SAGE2OV_TIMING") != nullptr; auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; hipStreamSynchronize(d->stream); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[step 1/device] %-30s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tp).count()); tp = t; };
    HIPCHK(hipEventRecord(e0, d->stream));
    // ASCII input (sage2ov_reads_add_ascii): filter, 2-bit pack and canonical orientation on the device (utils.cpp:144-166, :96-119, readLoader.cpp:195)
    unsigned char* dbases = nullptr; u64* doffA = nullptr; u32* gflag = nullptr; u32* gpos = nullptr;
    if (ascii) {
        const u64 nin = ascii->n_in;
        if (nin >= (1ull << 32) - RS_TILE) { err = "too many reads for the device organiser"; return SAGE2OV_ERR_LIMIT; }
        WS(db, unsigned char, WS_ORG_POOL, ascii->nbytes + 64); WS(dof, u64, WS_ORG_OFF, nin + 2); WS(gf, u32, WS_ORG_GFLAG, nin + 2); WS(gp, u32, WS_ORG_GPOS, nin + 2);
        dbases = db; doffA = dof; gflag = gf; gpos = gp;
        HIPCHK(hipMemcpyAsync(dbases, ascii->bases, ascii->nbytes, hipMemcpyHostToDevice, d->stream));
        HIPCHK(hipMemcpyAsync(doffA, ascii->off, (nin + 1) * sizeof(u64), hipMemcpyHostToDevice, d->stream));
        u64 init[6] = {0, 0, 0, 0, ~0ull, 0}; HIPCHK(hipMemcpyAsync(d->d_counters + 16, init, sizeof init, hipMemcpyHostToDevice, d->stream));
        if (nin) hipLaunchKernelGGL(k_org_classify, dim3(grid_for(nin, 256)), dim3(256), 0, d->stream, dbases, doffA, (u64)nin, (u32)k, 1018u, gflag, d->d_counters + 16);
        u64 cc[6]; HIPCHK(hipMemcpyAsync(cc, d->d_counters + 16, sizeof cc, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
        if (cc[5]) { err = "reads longer than 1018 bases are not supported"; return SAGE2OV_ERR_LIMIT; }
        ascii->good = cc[1]; ascii->total_bp = cc[2]; ascii->small = cc[3]; ascii->maxL = (int)cc[0]; ascii->minL = cc[1] ? (int)cc[4] : 0;
        n = cc[1]; maxL = ascii->maxL; minL = ascii->minL;
        { int need = (2 * std::max(maxL, 1) + 9 + 63) / 64; S = 4; while (S < need) S *= 2; } ascii->S = S;
        if (nin) { u64 tot = 0; int rc = scan_u32(d, gflag, nin, gpos, &tot, err); if (rc) return rc; }
    }
    if (S != 4 && S != 8 && S != 16 && S != 32) { err = "unsupported words-per-read (read length limit is 1018 bases)"; return SAGE2OV_ERR_LIMIT; }
    if (n >= (1ull << 32) - RS_TILE) { err = "too many reads for the device organiser"; return SAGE2OV_ERR_LIMIT; }
    u64 N = 0;
    u64* reads = nullptr; unsigned short* dfreq = nullptr;
    if (n) {
        WS(img, u64, WS_ORG_IMG, n * S); WS(k0, u64, WS_ORG_K0, n); WS(k1, u64, WS_ORG_K1, n); WS(v0, u32, WS_ORG_V0, n); WS(v1, u32, WS_ORG_V1, n);
        if (ascii) hipLaunchKernelGGL(k_org_pack, dim3(grid_for(ascii->n_in, 256)), dim3(256), 0, d->stream, dbases, doffA, (u64)ascii->n_in, gflag, gpos, S, img, k0, v0);
        else {
            WS(dpool, u64, WS_ORG_POOL, pool_words + 17); WS(doff, u64, WS_ORG_OFF, n); WS(dlen, unsigned short, WS_ORG_LEN, n);
            HIPCHK(hipMemcpyAsync(dpool, pool, pool_words * sizeof(u64), hipMemcpyHostToDevice, d->stream));
            HIPCHK(hipMemsetAsync(dpool + pool_words, 0, 17 * sizeof(u64), d->stream));
            HIPCHK(hipMemcpyAsync(doff, off, n * sizeof(u64), hipMemcpyHostToDevice, d->stream));
            HIPCHK(hipMemcpyAsync(dlen, len, n * sizeof(uint16_t), hipMemcpyHostToDevice, d->stream));
            lap("work space + upload");
            hipLaunchKernelGGL(k_org_canon, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, dpool, doff, dlen, (u64)n, S, img, k0, v0);
        }
        const u32 nb = (u32)((n + RS_TILE - 1) / RS_TILE);
        WS(hist, u32, WS_ORG_HIST, (u64)256 * nb + 2); WS(hscan, u32, WS_ORG_HSCAN, (u64)256 * nb + 2);
        u64 *ka = k0, *kb = k1; u32 *va = v0, *vb = v1;
        for (int pass = 0; pass < 8; pass++) {
            hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(64), 0, d->stream, ka, (u64)n, 8 * pass, hist, nb);
            u64 tot = 0; int rc = scan_u32(d, hist, (u64)256 * nb, hscan, &tot, err); if (rc) return rc;
            hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(64), 0, d->stream, ka, va, (u64)n, 8 * pass, hscan, nb, kb, vb);
            std::swap(ka, kb); std::swap(va, vb);
        }
        WS(flag, u32, WS_ORG_FLAG, n + 2); WS(uid, u32, WS_ORG_UID, n + 2);
        u32 longrun = 0;
        HIPCHK(hipMemsetAsync(flag, 0, sizeof(u32), d->stream));
        hipLaunchKernelGGL(k_org_longrun, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, ka, (u64)n, flag);
        HIPCHK(hipMemcpyAsync(&longrun, flag, sizeof longrun, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
        if (longrun || d->opt.get("SAGE2OV_TEST_FULL_SORT")) {
            // many reads share their first 32 bases: order on every word instead (stable LSD radix, last word first; the pairs arrive
            // sorted by the first word, which the last eight passes simply reproduce)
            for (int c = S - 1; c >= 0; c--) {
                hipLaunchKernelGGL(k_org_wordkeys, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, img, S, c, va, (u64)n, ka);
                for (int pass = 0; pass < 8; pass++) {
                    hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(64), 0, d->stream, ka, (u64)n, 8 * pass, hist, nb);
                    u64 tot = 0; int rc = scan_u32(d, hist, (u64)256 * nb, hscan, &tot, err); if (rc) return rc;
                    hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(64), 0, d->stream, ka, va, (u64)n, 8 * pass, hscan, nb, kb, vb);
                    std::swap(ka, kb); std::swap(va, vb);
                }
            }
        } else hipLaunchKernelGGL(k_org_ties, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, ka, va, (u64)n, img, S);
        hipLaunchKernelGGL(k_org_heads, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, ka, va, (u64)n, img, S, flag);
        { int rc = scan_u32(d, flag, n, uid, &N, err); if (rc) return rc; }
        if (N >= (1ull << 30)) { err = "more than 2^30-1 unique reads per context is not supported yet"; return SAGE2OV_ERR_LIMIT; }
        WS(headPos, u32, WS_ORG_HEAD, N + 2);
        hipLaunchKernelGGL(k_org_headpos, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, flag, uid, (u64)n, headPos, (u64)N);
        lap("canonical, sort, heads");
        HIPCHK(hipMalloc(&reads, (N + 1) * S * sizeof(u64))); HIPCHK(hipMalloc(&dfreq, (N + 1) * sizeof(unsigned short)));
        HIPCHK(hipMemsetAsync(reads, 0, S * sizeof(u64), d->stream)); HIPCHK(hipMemsetAsync(dfreq, 0, sizeof(unsigned short), d->stream));
        hipLaunchKernelGGL(k_org_gather, dim3(grid_for_capped(N * S, 256)), dim3(256), 0, d->stream, va, headPos, (u64)N, img, S, reads, dfreq);
    } else {
        HIPCHK(hipMalloc(&reads, S * sizeof(u64))); HIPCHK(hipMalloc(&dfreq, sizeof(unsigned short)));
        HIPCHK(hipMemsetAsync(reads, 0, S * sizeof(u64), d->stream)); HIPCHK(hipMemsetAsync(dfreq, 0, sizeof(unsigned short), d->stream));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, d->stream));
    lap("read store allocation + gather");
    words_out.resize((N + 1) * S); freq_out.resize(N + 1);
    touch_pages(words_out.data(), words_out.size() * sizeof(u64), 8);
    lap("host vectors (resize, first touch)");
    HIPCHK(hipMemcpyAsync(words_out.data(), reads, (N + 1) * S * sizeof(u64), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipMemcpyAsync(freq_out.data(), dfreq, (N + 1) * sizeof(unsigned short), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    lap("download of the store");
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); d->tm.organize_ms += ms;
    hipFree(dfreq);
    // the organised store becomes the context's read store (same state as after dev_upload_reads)
    free_reads(d);
    d->N = N; d->S = S; d->maxL = maxL; d->k = k; d->h = k > 64 ? 64 : k; d->reads = reads; d->uniL = (N && minL == maxL) ? maxL : 0;
    mem_sample(d);
    // the organiser's work space (the staged reads, their slot image, the sort's key / value pairs: ~180 bytes per read) is not needed again
    for (int id : {WS_ORG_POOL, WS_ORG_OFF, WS_ORG_LEN, WS_ORG_IMG, WS_ORG_K0, WS_ORG_K1, WS_ORG_V0, WS_ORG_V1, WS_ORG_HIST, WS_ORG_HSCAN, WS_ORG_FLAG, WS_ORG_UID, WS_ORG_HEAD, WS_ORG_GFLAG, WS_ORG_GPOS}) ws_free(d, id);
    decide_diet(d);
    if (!d->diet) { int rc = ensure_results(d, err); if (rc) return rc; }
    lap("release + result buffers");
    *N_out = N;
    return 0;
}

static int build_locality_order(Device* d, u64 lo, u64 hi, const u32** order_out, std::string& err);
// The locality-ordered copy of the read store + the two translation tables (see Device::readsLoc).  Part of the index build (timed with it).
// Round 3: two kernels.  k_loc_index writes the translation tables from the order; k_loc_scatter then STREAMS the id-ordered store (coalesced reads) and
// writes every slot to its position (whole 32 / 64 / 128-byte slots: no read-modify-write at the memory side).  The gather it replaces pulled every
// 64-byte slot as a 128-byte line request: 5.4 GB of reads for 2.7 GB of reads at configs[2].
// runStarts (64 words, zeroed by the host; their sum, over every eighth block): positions whose read has another minimiser or another strand of it than the read before -- reads that cannot take their windows
// over from a predecessor (kernels_probe_fast.inc); their share decides whether the minimiser groups are built (dev_build_index)
__global__ void k_loc_index(const u32* __restrict__ order, u64 N, u32* idOf, u32* posOf, unsigned short* meta, unsigned long long* runStarts) {      // order: 3 dwords per position {hash, id, meta}
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) { idOf[0] = 0; posOf[0] = 0; idOf[N + 1] = 0; posOf[N + 1] = 0; meta[0] = 0xFFFF; meta[N + 1] = 0xFFFF; }
    if ((blockIdx.x & 7u) == 0) {                                         // (every eighth block of 256 positions: a sample is all the decision needs; one atomic per sampled block)
        const bool start = order && p < N && (p == 0 || order[3 * p] != order[3 * (p - 1)] || ((order[3 * p + 2] ^ order[3 * (p - 1) + 2]) & 0x100u) != 0);
        const int c = __syncthreads_count(start);
        if (threadIdx.x == 0 && c) atomicAdd(runStarts + ((blockIdx.x >> 3) & 63u), (unsigned long long)c);
    }
    if (p >= N) return;
    const u32 id = order ? order[3 * p + 1] : (u32)(p + 1); u32 mt = order ? order[3 * p + 2] : 0xFFFFu;
    // (bit 9, round 4: this read has ANOTHER minimiser hash than the read before it -- nothing of that read's run can be carried over; the fast probe kernel moves the
    //  boundaries of its waves' visits there)
    if (order && (p == 0 || order[3 * p] != order[3 * (p - 1)])) mt |= 0x200u;
    idOf[p + 1] = id; posOf[id] = (u32)(p + 1); meta[p + 1] = (unsigned short)mt;       // meta 0xFFFF: no minimiser information (no window reuse)
}
__global__ void k_loc_scatter(const u64* __restrict__ reads, const u32* __restrict__ posOf, u64 N, int S, u64* out) {
    // one 16-byte piece of a slot per thread and step; grid-stride: (N + 1) * S / 2 pieces reach 2^32 from 2^29 reads of the 16-word layout on, and a launch
    // of 2^32 threads or more runs modulo 2^32 (grid_for_capped)
    const int H = S / 2; const u64 total = (N + 1) * H, stride = (u64)gridDim.x * blockDim.x;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const u64 id = t / H; const int c = (int)(t % H);
        const ulonglong2 v = ((const ulonglong2*)reads)[t];        // (slot 0 is all zero and stays at position 0: posOf[0] = 0)
        ((ulonglong2*)out)[(u64)posOf[id] * H + c] = v;
    }
}
__global__ void k_status_by_pos(const u32* __restrict__ idOf, const uint8_t* __restrict__ status, u64 N, uint8_t* statusP) {
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (p > N) return; statusP[p] = p ? status[idOf[p]] : (uint8_t)0xFF;
}
__global__ void k_ids_to_pos(const u32* __restrict__ ids, u64 n, const u32* __restrict__ posOf, u32* out) { const u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (x < n) out[x] = posOf[ids[x]]; }
static int build_locality_store(Device* d, std::string& err) {
    const u64 N = d->N;
    WS(rl, u64, WS_LOC_READS, (N + 1) * d->S); WS(io, u32, WS_LOC_IDOF, N + 2); WS(po, u32, WS_LOC_POSOF, N + 2); WS(sp, uint8_t, WS_LOC_STATUS, N + 2);
    WS(me, unsigned short, WS_LOC_META, N + 2);
    d->readsLoc = rl; d->idOf = io; d->posOf = po; d->statusP = sp; d->metaP = me;
    const u32* order = nullptr;
    if (N && !d->opt.get("SAGE2OV_NO_LOCALITY")) { int rc = build_locality_order(d, 1, N + 1, &order, err); if (rc) return rc; }
    HIPCHK(hipMemsetAsync(d->d_runStarts, 0, 64 * sizeof(u64), d->stream));                   // run starts of the order (k_loc_index)
    hipLaunchKernelGGL(k_loc_index, dim3(grid_for(std::max<u64>(N, 1), 256)), dim3(256), 0, d->stream, order, (u64)N, io, po, me, (unsigned long long*)d->d_runStarts);
    d->runStartsValid = order != nullptr;
    if (!order) d->runStartFrac = 0.0;
    // (the count travels to a pinned word while k_loc_scatter below runs: dev_build_index waits for the copy's event, not for the stream)
    if (d->runStartsValid) { HIPCHK(hipMemcpyAsync(d->h_runStarts, d->d_runStarts, 64 * sizeof(u64), hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipEventRecord(d->evRunStarts, d->stream)); }
    if (d->diet) { HIPCHK(hipStreamSynchronize(d->stream)); ph_begin(d, false); }       // the order's sort buffers (24 bytes per read) go before the tuples come: a phase of their own
    hipLaunchKernelGGL(k_loc_scatter, dim3(grid_for_capped((N + 1) * (d->S / 2), 256)), dim3(256), 0, d->stream, d->reads, po, (u64)N, d->S, rl);
    HIPCHK(hipGetLastError());
    return 0;
}
static int refresh_status_by_pos(Device* d, std::string& err) {
    hipLaunchKernelGGL(k_status_by_pos, dim3(grid_for(d->N + 1, 256)), dim3(256), 0, d->stream, d->idOf, d->status, (u64)d->N, d->statusP);
    HIPCHK(hipGetLastError()); return 0;
}
// exclusive scan without the read-back of the total (no host synchronisation)
static int scan_u32_async(Device* d, const u32* in, u64 n, u32* out, std::string& err);

static int pt_digits(u64 nWin, int* bits, int* nd) {            // window id bits, number of <= 9-bit digits
    int wb = 0; while ((1ull << wb) < nWin) wb++;
    *bits = wb; *nd = wb == 0 ? 0 : (wb + 8) / 9; return 0;
}
// The tuples are sorted by the window id (dword keyw >> shiftW); `digit 0 counted` tells that cnt already holds the first digit's per-tile histogram (taken
// by the kernel that wrote the tuples).  On return B.E[*cur_out] is the sorted array and off[0..nWin] (if given) holds the window boundaries.
static int partition_by_window(Device* d, PtBufs& B, int keyw, u32 n, int shiftW, u64 nWin, bool digit0Counted, u32* cnt, u32* base, u32* off, int* cur_out, std::string& err) {
    int wb, nd; pt_digits(nWin, &wb, &nd);
    const u32 ntiles = (u32)((n + PT_TILE - 1) / PT_TILE);
    int cur = 0;
    if (n) {
        const int bper = nd ? (wb + nd - 1) / nd : 0;
        const u32* in = B.src0 ? B.src0 : B.E[0];
        if (B.src0 && nd == 0) { HIPCHK(hipMemcpyAsync(B.E[0], B.src0, (size_t)n * B.W * sizeof(u32), hipMemcpyDeviceToDevice, d->stream)); in = B.E[0]; }
        for (int j = 0; j < nd; j++) {
            const int shift = shiftW + j * bper; const int bj = std::min(bper, wb - j * bper); const u32 mask = (1u << bj) - 1u;
            if (!(j == 0 && digit0Counted)) hipLaunchKernelGGL(k_pt_hist, dim3(ntiles), dim3(PT_THREADS), 0, d->stream, in + keyw, n, shift, mask, cnt, ntiles, (u32)B.W);
            int rc = scan_u32_async(d, cnt, (u64)(mask + 1) * ntiles, base, err); if (rc) return rc;
            const int o = (j == 0 && B.src0) ? 0 : (cur ^ 1);
            if (B.W == 4 && keyw == 0) hipLaunchKernelGGL((k_pt_scatter<4, 0>), dim3(ntiles), dim3(PT_SC_THREADS), 0, d->stream, in, n, shift, mask, base, ntiles, B.E[o]);
            else if (B.W == 3 && keyw == 0) hipLaunchKernelGGL((k_pt_scatter<3, 0>), dim3(ntiles), dim3(PT_SC_THREADS), 0, d->stream, in, n, shift, mask, base, ntiles, B.E[o]);
            else if (B.W == 3 && keyw == 2) hipLaunchKernelGGL((k_pt_scatter<3, 2>), dim3(ntiles), dim3(PT_SC_THREADS), 0, d->stream, in, n, shift, mask, base, ntiles, B.E[o]);
            else { err = "partition: unsupported tuple format"; return SAGE2OV_ERR_INTERNAL; }
            cur = o; in = B.E[o];
        }
        if (off) hipLaunchKernelGGL(k_pt_bounds, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, B.E[cur] + keyw, n, shiftW, (u32)nWin, off, (u32)B.W);
    } else if (off) HIPCHK(hipMemsetAsync(off, 0, (nWin + 1) * sizeof(u32), d->stream));
    HIPCHK(hipGetLastError());
    *cur_out = cur;
    return 0;
}
// first-digit parameters of the tuple kernel (must agree with partition_by_window)
static void pt_first_digit(u64 nWin, int shiftW, int* shift0, u32* mask0, int* doHist) {
    int wb, nd; pt_digits(nWin, &wb, &nd);
    if (!nd) { *shift0 = 0; *mask0 = 0; *doHist = 0; return; }
    const int bper = (wb + nd - 1) / nd; *shift0 = shiftW; *mask0 = (1u << std::min(bper, wb)) - 1u; *doHist = 1;
}

// (probe time the minimiser groups save) / (what they cost to build) at full share, as a function of the share f of reads without a predecessor: see dev_build_index
static inline double RUN_START_RULE(double f) { return f > 0.1 ? 27.0 * (f - 0.1) : 0.0; }
int dev_build_index(Device* d, uint64_t* slots_out, uint64_t* keys_out, uint64_t* csr_out, uint64_t* nlong_out, uint32_t* rebuilds, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const u64 N = d->N; if (!d->reads && !d->readsLoc) { err = "reads not resident"; return SAGE2OV_ERR_ARG; }
    s4cache_release(d);                                                   // (a new step 2: step 4's spare blocks must not stand in the way of steps 2-3; empty unless step 4 ran since)
    HIPCHK(hipEventRecord(d->ev[0], d->stream));                          // (the locality store is part of the build and of index_ms)
    if (d->diet) {                                                            // phase A: the previous step's results go before the sort's buffers come (Device::Phase)
        HIPCHK(hipStreamSynchronize(d->stream));
        if (!d->resCarved) { hipFree(d->right); hipFree(d->left); hipFree(d->conn); hipFree(d->cflag); hipFree(d->status); hipFree(d->cand);     // (from before the context went on the diet)
                             d->right = d->left = nullptr; d->conn = d->cflag = nullptr; d->status = nullptr; d->cand = nullptr; d->cand_cap = 0; d->n_cand = 0; }
        ph_begin(d);
    }
    if (d->reads) { int rc = build_locality_store(d, err); if (rc) return rc; }
    // (diet mode, second and later builds: the id-ordered store was released after the first one; the locality-ordered store and its tables are what
    //  this build would produce again -- the order is a function of the reads -- so they are kept)
    if (d->diet && d->reads && d->uniL) { HIPCHK(hipStreamSynchronize(d->stream)); hipFree(d->reads); d->reads = nullptr; }      // every later reader takes lengths from uniL
    const bool byPos = d->reads == nullptr;
    // (diet mode: the previous step's results -- per-read records, candidates, final edges: ~60 bytes per read -- went when this build began, ph_begin above)
    d->T = std::max<u64>(IX_W, (8 * N + IX_W - 1) / IX_W * IX_W);            // load <= 0.5, as hashTable.cpp:83 sizes it; whole windows
    // (tests: SAGE2OV_TEST_TABLE_SLOTS forces a larger table, e.g. beyond 2^32 slots -- slot indices are 64-bit, pair indices and window ids 32-bit)
    if (const char* ev = d->opt.get("SAGE2OV_TEST_TABLE_SLOTS")) { const u64 want = strtoull(ev, nullptr, 10); if (want > d->T) d->T = (want + IX_W - 1) / IX_W * IX_W; }
    if ((d->T >> 1) >= (1ull << 32)) { err = "table too large: more than 2^32 slot pairs"; return SAGE2OV_ERR_LIMIT; }
    const u64 nW = d->T / IX_W; const u32 n = (u32)(4 * N);
    const u32 big_cap = 1u << 20;
    WS(slots_ws, u64, WS_SLOTS, d->T); d->slots = slots_ws;
    WS(big, u64, WS_BIG, (u64)big_cap * 3);
    WS(csr_ws, u32, WS_CSR, std::max<u64>(1, 4 * N)); d->csr = csr_ws;
    // The minimiser groups (second access path of the fast kernel) serve the reads that START a run of the locality order -- reads with another minimiser, or another
    // strand of it, than the read before: every one of their windows is a random line of the uniform table otherwise.  How many there are is a property of the data
    // (coverage: a minimiser's reads are a run per strand), counted by k_loc_index.  Measured at the end of round 3 with the order on all 32 hash bits
    // (tests/diag/groups_by_coverage.py, groups_by_size.py; share of such reads -> probe time saved at full share / build time, ms): 34-39 M reads at 10x / 20x / 30x / 50x
    // coverage 30.5 % -> 35.3 / 6.2, 18.8 % -> 13.0 / 5.8, 13.9 % -> 6.6 / 5.5, 9.4 % -> 2.0 / 5.2; 8.5-9.7 M reads 32.2 % -> 6.8 / 1.7, 19.4 % -> 2.2 / 1.5, 14.2 % -> 0.9 / 1.4,
    // 9.5 % -> -0.1 / 1.3; 50x (9.5 %) at 42.5 M, 68 M and 102 M reads 4.2 / 6.1, 8.5 / 10.0 and 12.1 / 16.2.  The ratio is close to 27 (f - 0.1) from 20 M reads on and 0.72
    // of that below (RUN_START_RULE); the saving scales with the share of the reads this context probes, the build does not: built iff share x ratio >= 1.
    if (d->runStartsValid) {
        HIPCHK(hipEventSynchronize(d->evRunStarts));
        u64 rs = 0; for (int x = 0; x < 64; x++) rs += d->h_runStarts[x];
        u64 sampled = 0; for (u64 b0 = 0; b0 * 256 < N; b0 += 8) sampled += std::min<u64>(256, N - b0 * 256);      // positions of the sampled blocks
        d->runStartFrac = sampled ? (double)rs / (double)sampled : 0.0; d->runStartsValid = false;      // (kept: a diet-mode rebuild reuses the order)
    }
    const double f_ = d->runStartFrac, sizeFactor = N >= 20000000ull ? 1.0 : 0.72;
    bool wantMI = !d->diet && N >= 2000000ull && d->probeShare * sizeFactor * RUN_START_RULE(f_) >= 1.0;
    if (d->opt.get("SAGE2OV_TIMING")) fprintf(stderr, "[index] reads without a predecessor in the locality order: %.1f %% -> minimiser groups %s\n", 100.0 * f_, wantMI ? "built" : "not built");
    if (const char* ev = d->opt.get("SAGE2OV_MINIMIZER_INDEX")) wantMI = atoi(ev) != 0;
    if (d->opt.get("SAGE2OV_NO_MINIMIZER_INDEX") || (d->h - std::min(d->h, 16) + 1) < 8) wantMI = false;
    u64 TL = 0; int tlBits = 0; u64 gW = 0;
    // >= 3N group words (round 3; 2N before): a group window takes 256 x MIW_R = 2048 group tuples in registers and ~3N distinct keys spread over TL / 2048 windows, so with
    // TL in [2N, 3N) most windows overflowed into the global scratch, the scratch ran out, and the groups were built and then given up ("crowded") for every read set
    // between 24 M and 33 M unique reads -- 4 ms of build for nothing, found by timing sizes between configs[1] and configs[2] (tests/diag/groups_by_size.py)
    if (wantMI) { TL = IX_GW; tlBits = IX_GWLOG; while (TL < d->T / 8 * 3) { TL <<= 1; tlBits++; } gW = TL / IX_GW; }
    const u64 nAlloc = std::max<u64>(4, (u64)n + 4);
    const u32 ntiles = (u32)((n + PT_TILE - 1) / PT_TILE);
    PtBufs B; B.W = wantMI ? 4 : 3;                                          // {K, M, entry, tag} with the groups, {K, entry, tag} without
    { WS(a, u32, WS_PT_K0, nAlloc * B.W); B.E[0] = a; } { WS(a, u32, WS_PT_K1, nAlloc * B.W); B.E[1] = a; }
    WS(cnt, u32, WS_PT_CNT, (u64)PT_NB_MAX * std::max<u32>(ntiles, 1) + 2); WS(base, u32, WS_PT_BASE, (u64)PT_NB_MAX * std::max<u32>(ntiles, 1) + 2);
    WS(winOff, u32, WS_PT_OFF, nW + 2);
    u64* mi1 = nullptr; u64* krec = nullptr; u32* gOff = nullptr;
    if (wantMI) {
        WS(m1, u64, WS_MI1, TL); mi1 = m1;
        WS(kr_, u64, WS_KREC, 4 * N + MI_SCAN_PAD); krec = kr_;              // one record per distinct key (<= 4N)
        WS(go, u32, WS_PT_GOFF, gW + 3); gOff = go;
    }
    *rebuilds = 0;
    const bool timing = d->opt.get("SAGE2OV_TIMING") != nullptr; auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; hipStreamSynchronize(d->stream); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[index] %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - tp).count()); tp = t; };
    u64 c[9]; u32 ixCntBits = (u32)IXW_CNT_BITS_DEFAULT;
    for (int attempt = 0;; attempt++) {
        HIPCHK(hipMemsetAsync(d->d_counters + 8, 0, 10 * sizeof(u64), d->stream));
        // ---- tuples of the 4N entries, sorted by the window of their home slot
        int shift0, doHist; u32 mask0; pt_first_digit(nW, IX_WPLOG, &shift0, &mask0, &doHist);
        if (n) hipLaunchKernelGGL(k_ix_tuples, dim3(ntiles), dim3(PT_THREADS), 0, d->stream, byPos ? d->readsLoc : d->reads, d->posOf, byPos ? 1 : 0, (u32)N, d->S, d->h, d->seed, (u32)(d->T >> 1), wantMI ? 1 : 0, shift0, mask0, doHist,
                                  B.E[0], cnt, ntiles);
        lap("tuples");
        int cur = 0;
        { int rc = partition_by_window(d, B, 0, n, IX_WPLOG, nW, doHist != 0, cnt, base, winOff, &cur, err); if (rc) return rc; }
        lap("partition by table window");
        if (d->opt.get("SAGE2OV_VERIFY_PARTITION") && n) {
            u64* vo = nullptr; HIPCHK(hipMalloc(&vo, 3 * sizeof(u64))); HIPCHK(hipMemsetAsync(vo, 0, 3 * sizeof(u64), d->stream));
            hipLaunchKernelGGL(k_pt_verify, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, B.E[cur], n, IX_WPLOG, (u32)B.W, winOff, (u32)nW, vo);
            u64 hv[3]; HIPCHK(hipMemcpyAsync(hv, vo, sizeof hv, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream)); hipFree(vo);
            fprintf(stderr, "[verify-partition] %llu tuples, %llu windows: %llu order violations, largest window %llu tuples, %llu negative windows\n", (unsigned long long)n, (unsigned long long)nW,
                    (unsigned long long)hv[0], (unsigned long long)hv[1], (unsigned long long)hv[2]);
        }
        // ---- the windows of the uniform table, built in LDS; group tuples into the free buffer set
        IxWinArgs A; A.T = B.E[cur]; A.W = B.W; A.winOff = winOff; A.nW = (u32)nW; A.slots = d->slots; A.csr = d->csr;
        A.counters = d->d_counters + 8; A.big = big; A.big_cap = big_cap; A.idOf = d->idOf; A.G = wantMI ? B.E[cur ^ 1] : nullptr; A.wh = nullptr; A.cntBits = ixCntBits;     // (group tuples, 12 bytes each, into the free 16-byte buffer)
        // (the scratch words of heavy windows need a buffer of their own)
        // one word per SURPLUS tuple of a heavy window (more than 3072 tuples where the mean is 2048: keys in thousands of reads); small inputs get the
        // worst case (every tuple in one window), big ones an eighth of it
        { const u64 whCap = n <= (64u << 20) ? nAlloc : nAlloc / 8; WS(whs, u64, WS_WHERE, whCap); A.wh = whs; A.wh_cap = whCap; }
        const u64 ixPerCu = d->opt.get("SAGE2OV_IXW_GRID_PER_CU") ? std::max(1, atoi(d->opt.get("SAGE2OV_IXW_GRID_PER_CU"))) : 6;
        hipLaunchKernelGGL(k_ix_window, dim3((unsigned)std::min<u64>(nW, 256ull * ixPerCu)), dim3(IXW_T), 0, d->stream, A);      // persistent: two rounds of 3 workgroups per CU, each with ONE pair of statistics atomics
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c, d->d_counters + 8, sizeof c, hipMemcpyDeviceToHost, d->stream));
        u64 c9 = 0; HIPCHK(hipMemcpyAsync(&c9, d->d_counters + 8 + 9, sizeof c9, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        lap("table windows"); mem_sample(d);
        if (c[8] && (c[8] & 0xFFFFFull) == 0 && (c[8] >> 40) == 0 && ixCntBits == (u32)IXW_CNT_BITS_DEFAULT) {      // only the per-key counter ran over: again, with 30 bits of count
            if (timing) fprintf(stderr, "[index] a key occurs in more than 2^20 reads: building again with the wide counter split\n");
            ixCntBits = (u32)IXW_CNT_BITS_WIDE; attempt--; continue; }
        if (c[8]) { char b_[256]; snprintf(b_, sizeof b_, "index build: %llu table windows overflowed, %llu keys occur in more than 2^30 reads, %llu heavy windows found no scratch (cursor %llu of %llu)",
                             (unsigned long long)(c[8] & 0xFFFFF), (unsigned long long)((c[8] >> 20) & 0xFFFFF), (unsigned long long)(c[8] >> 40), (unsigned long long)c9, (unsigned long long)A.wh_cap);
                    err = b_; return SAGE2OV_ERR_LIMIT; }
        if (c[2] > big_cap) { err = "too many long buckets"; return SAGE2OV_ERR_LIMIT; }
        if (c[2]) {
            hipLaunchKernelGGL(k_index_purity, dim3(grid_for(c[2] * 64, 256)), dim3(256), 0, d->stream, d->readsLoc, d->S, d->h, big, c[2], d->csr, d->d_counters + 8);
            HIPCHK(hipMemcpyAsync(c, d->d_counters + 8, 5 * sizeof(u64), hipMemcpyDeviceToHost, d->stream));
            HIPCHK(hipStreamSynchronize(d->stream));
        }
        if (c[3] == 0) { d->n_csr = c[0]; d->n_keys = c[1]; d->n_long = c[2];
            // ---- stage B: the group tuples sorted by the window of their group word, the windows of the group table built in LDS
            d->mi1 = nullptr; d->krec = nullptr; d->TL = 0; d->n_groups = 0;
            if (wantMI && d->n_keys > 0) {
                const u32 nG = n;                                        // one group tuple per entry tuple position; those without a record (GT_INVALID) sort behind window gW - 1
                PtBufs G; G.W = 3; G.E[0] = B.E[cur ^ 1]; G.E[1] = B.E[cur];                                // (the entry tuples are no longer needed)
                int gcur = 0; const int gshift = 31 - (tlBits - IX_GWLOG);                                  // keys are minimiser hashes >> 1
                { int rc = partition_by_window(d, G, 0, nG, gshift, gW + 1, false, cnt, base, gOff, &gcur, err); if (rc) return rc; }
                lap("partition by group window");
                MiWinArgs MA; MA.G = G.E[gcur]; MA.gOff = gOff; MA.gW = (u32)gW; MA.tlBits = tlBits; MA.mi1 = mi1; MA.krec = krec; MA.counters = d->d_counters + 8; MA.wh = A.wh; MA.wh_cap = A.wh_cap;
                HIPCHK(hipMemsetAsync(d->d_counters + 8 + 9, 0, sizeof(u64), d->stream));                  // (the scratch cursor starts over)
                const u64 miPerCu = d->opt.get("SAGE2OV_MIW_GRID_PER_CU") ? std::max(1, atoi(d->opt.get("SAGE2OV_MIW_GRID_PER_CU"))) : 8;
                hipLaunchKernelGGL(k_mi_window, dim3((unsigned)std::min<u64>(gW, 256ull * miPerCu)), dim3(256), 0, d->stream, MA);
                // the probe scan may run past the last group: empty records behind ALL records
                u64 mc[3];
                HIPCHK(hipMemcpyAsync(mc, d->d_counters + 8 + 5, sizeof mc, hipMemcpyDeviceToHost, d->stream));
                HIPCHK(hipStreamSynchronize(d->stream));
                HIPCHK(hipMemsetAsync(krec + mc[0], 0, MI_SCAN_PAD * sizeof(u64), d->stream));   // (mc[0] records, at krec[0 .. mc[0]))
                lap("group windows"); mem_sample(d);
                d->n_groups = mc[1];
                if (d->opt.get("SAGE2OV_VERIFY_MI")) {
                    u64* vo = nullptr; HIPCHK(hipMalloc(&vo, 2 * sizeof(u64))); HIPCHK(hipMemsetAsync(vo, 0, 2 * sizeof(u64), d->stream));
                    hipLaunchKernelGGL(k_mi_verify, dim3(grid_for(4 * N, 256)), dim3(256), 0, d->stream, d->readsLoc, N, d->S, d->h, d->seed, d->slots, d->T, mi1, TL, krec, vo);
                    u64 hv2[2]; HIPCHK(hipMemcpyAsync(hv2, vo, sizeof hv2, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream)); hipFree(vo);
                    fprintf(stderr, "[verify-mi] %llu of %llu entries do not find their bucket in their key's group\n", (unsigned long long)hv2[0], (unsigned long long)hv2[1]);
                }
                if (mc[2] == 0 && mc[1] * 10 <= TL * 7) { d->mi1 = mi1; d->krec = krec; d->TL = TL; }   // else: too crowded, the fast kernel uses the uniform table
                if (timing) fprintf(stderr, "[index] minimiser groups: %llu groups, %llu records, %llu group words (load %.2f), crowded windows %llu -> %s\n", (unsigned long long)mc[1], (unsigned long long)mc[0],
                                    (unsigned long long)TL, (double)mc[1] / (double)TL, (unsigned long long)mc[2], d->mi1 ? "used" : "NOT used");
            }
            break;
        }
        if (attempt >= 8) { err = "index build: tag collisions in long buckets persist after 8 reseeds"; return SAGE2OV_ERR_INTERNAL; }
        d->seed = d->seed * 0x9E3779B97F4A7C15ull + 12345; (*rebuilds)++;
    }
    HIPCHK(hipEventRecord(d->ev[1], d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]); d->tm.index_ms += ms;
    if (d->diet) ph_begin(d);                                                 // phase B: the sort's buffers (~150 bytes per read) go, the step's results come
    *slots_out = d->T; *keys_out = d->n_keys; *csr_out = d->n_csr; *nlong_out = d->n_long;
    return 0;
}

int dev_lookup(Device* d, uint64_t hi, uint64_t lo, uint64_t* entries, uint32_t cap, uint32_t* count, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (!d->slots) { err = "index not built"; return SAGE2OV_ERR_ARG; }
    u64* out = nullptr; const u32 c2 = std::min<u32>(cap, 128);
    HIPCHK(hipMalloc(&out, (129) * sizeof(u64)));
    hipLaunchKernelGGL(k_lookup, dim3(1), dim3(1), 0, d->stream, d->slots, d->T, d->csr, d->idOf, d->seed, d->h, (u64)hi, (u64)lo, out, c2);
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
    // one wave per read, 4 waves per block (2 for the 16-word layout, 1 for the 32-word layout: LDS); enough blocks to fill
    // 256 CUs several times over, grid-stride beyond
    const unsigned wpb = d->S == 32 ? 1 : (d->S == 16 ? 2 : 4);
    const unsigned blocks = (unsigned)std::min<u64>((nreads + wpb - 1) / wpb, 256ull * 32);
    switch (d->S) {
        case 4: hipLaunchKernelGGL((k_probe<4, MODE, 4>), dim3(blocks), dim3(256), 0, d->stream, A); break;
        case 8: hipLaunchKernelGGL((k_probe<8, MODE, 4>), dim3(blocks), dim3(256), 0, d->stream, A); break;
        case 16: hipLaunchKernelGGL((k_probe<16, MODE, 2>), dim3(blocks), dim3(128), 0, d->stream, A); break;
        case 32: hipLaunchKernelGGL((k_probe<32, MODE, 1>), dim3(blocks), dim3(64), 0, d->stream, A); break;     // 505 .. 1018 bases: 34 KB of LDS per wave
        default: err = "bad S"; return SAGE2OV_ERR_INTERNAL;
    }
    HIPCHK(hipGetLastError());
    return 0;
}
static int scan_u32(Device* d, const u32* in, u64 n, u32* out, u64* total, std::string& err);
// processing order of ids [lo,hi): grouped by the reads' global minimiser (see k_minimizer)
static int build_locality_order(Device* d, u64 lo, u64 hi, const u32** order_out, std::string& err) {
    const u64 n = hi - lo;
    // reads sorted by (their global minimiser's hash, strand of the minimiser, start of the read relative to it): LSD radix passes
    // (kernels_partition.inc), first the 9 bits of strand + offset, then the hash -- reads with one minimiser end up next to each
    // other, each starting a few bases after the one before.  Measured at BASELINE configs[2]: probe kernel 149 -> 142 ms from the finer processing
    // order alone, 120 ms with the read store in that order too.  *order_out: per position, meta << 32 | id.
    // ALL 32 bits of the hash since the end of round 3 (four passes of 8 bits; 27 bits = three passes of 9 before): two minimisers that share a bucket interleave
    // their reads and break each other's runs of shifted reads (window reuse, 5.2) -- hash bits -> probe pass at configs[2]: 18 -> 97.2 ms, 24 -> 78.8, 27 -> 62.5,
    // 30 -> 57.2, 32 -> 56.5, for 0.3 ms more of index build.
    int lg = 32;
    if (const char* ev = d->opt.get("SAGE2OV_ORDER_BITS")) lg = std::max(1, std::min(32, atoi(ev)));
    const u32 ntiles = (u32)((n + PT_TILE - 1) / PT_TILE);
    PtBufs B; B.W = 3;                                                        // {hash, id, meta}
    { WS(a, u32, WS_MINH, 3 * (n + 4)); B.E[0] = a; } { WS(a, u32, WS_OCUR, 3 * (n + 4)); B.E[1] = a; }
    WS(cnt, u32, WS_PT_CNT, (u64)PT_NB_MAX * std::max<u32>(ntiles, 1) + 2); WS(base, u32, WS_PT_BASE, (u64)PT_NB_MAX * std::max<u32>(ntiles, 1) + 2);
    // one element format for all four passes, written by the minimiser kernel itself (round 2 had a pack and a re-key kernel in between): pass 1 sorts by
    // the meta (dword 2), passes 2-4 by the hash (dword 0)
    if (d->S == 4) hipLaunchKernelGGL((k_minimizer_t<4>), dim3(grid_for(n, 256)), dim3(256), 0, d->stream, d->reads, (u64)lo, (u64)hi, B.E[0]);
    else if (d->S == 8) hipLaunchKernelGGL((k_minimizer_t<8>), dim3(grid_for(n, 256)), dim3(256), 0, d->stream, d->reads, (u64)lo, (u64)hi, B.E[0]);
    else hipLaunchKernelGGL(k_minimizer, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, d->reads, (u64)lo, (u64)hi, d->S, B.E[0]);
    int cur = 0; int rc = partition_by_window(d, B, 2, (u32)n, 0, 1ull << 9, false, cnt, base, nullptr, &cur, err); if (rc) return rc;
    PtBufs C; C.W = 3; C.E[0] = B.E[cur]; C.E[1] = B.E[cur ^ 1];
    int cur2 = 0; rc = partition_by_window(d, C, 0, (u32)n, 32 - lg, 1ull << lg, false, cnt, base, nullptr, &cur2, err); if (rc) return rc;
    *order_out = C.E[cur2];
    return 0;
}
static ProbeArgs base_args(Device* d) {
    ProbeArgs A; memset(&A, 0, sizeof A);
    A.reads = d->readsLoc; A.idOf = d->idOf; A.statusP = d->statusP; A.meta = d->opt.get("SAGE2OV_NO_WINDOW_REUSE") ? nullptr : d->metaP; A.N = d->N; A.S = d->S; A.k = d->k; A.h = d->h; A.slots = d->slots; A.T = d->T; A.csr = d->csr; A.seed = d->seed;
    A.right = d->right; A.left = d->left; A.conn = d->conn; A.cflag = d->cflag; A.status = d->status; A.counters = d->d_counters;
    A.mi1 = d->mi1; A.TL = d->TL; A.krec = d->krec; A.uniL = d->uniL;
    A.chunkShift = (u32)FAST_CHUNK_LOG;      // (plan_fast_grid may double the positions per block visit)
    A.noRun = d->opt.get("SAGE2OV_NO_RUN_MODE") ? 1u : 0u; A.noParTail = d->opt.get("SAGE2OV_NO_PAR_TAIL") ? 1u : 0u;
    return A;
}

template <int S, int NW, int WPL, int WPB, int HITS, int TAIL>
static void launch_fast(Device* d, ProbeArgs& A, unsigned blocks) {
    if constexpr (HITS == 0 && TAIL == 0 && NW <= 10) {               // the clean-data form has an instantiation for read sets of ONE length (no per-candidate length registers)
        if (A.uniL) { hipLaunchKernelGGL((k_probe_fast<S, NW, WPL, WPB, HITS, TAIL, true>), dim3(blocks), dim3(64 * WPB), 0, d->stream, A); return; }
    }
    hipLaunchKernelGGL((k_probe_fast<S, NW, WPL, WPB, HITS, TAIL, false>), dim3(blocks), dim3(64 * WPB), 0, d->stream, A);
}
// The sequential-groups form of the clean-data kernel (k_probe_fast<..., UNI, QN, SEQ>): read sets of one length in the 4- and 8-word layouts only; qn = 2 (128 candidates,
// as the standard form) or 4 (256: the reads of high-coverage data the standard form lists).  false: no such instantiation for these reads.
static bool launch_fast_seq_any(Device* d, ProbeArgs& A, unsigned blocks, int qn) {
    constexpr int FW = SAGE2OV_FAST_WPB;
    const int nwinMax = d->maxL - d->h + 1;
    if (!d->uniL || nwinMax > 128) return false;
    if (d->S == 4) { if (qn == 4) hipLaunchKernelGGL((k_probe_fast<4, 8, 2, FW, 0, 0, true, 4, true>), dim3(blocks), dim3(64 * FW), 0, d->stream, A);
                     else hipLaunchKernelGGL((k_probe_fast<4, 8, 2, FW, 0, 0, true, 2, true>), dim3(blocks), dim3(64 * FW), 0, d->stream, A); return true; }
    if (d->S == 8 && d->maxL <= 160) { if (qn == 4) hipLaunchKernelGGL((k_probe_fast<8, 10, 2, FW, 0, 0, true, 4, true>), dim3(blocks), dim3(64 * FW), 0, d->stream, A);
                                       else hipLaunchKernelGGL((k_probe_fast<8, 10, 2, FW, 0, 0, true, 2, true>), dim3(blocks), dim3(64 * FW), 0, d->stream, A); return true; }
    return false;
}
// picks the instantiation for the resident reads; false: the 32-word layout (505 .. 1018 bases) has no fast kernel
template <int HITS, int TAIL>
static bool launch_fast_any(Device* d, ProbeArgs& A, unsigned blocks) {
    constexpr int FW = SAGE2OV_FAST_WPB;      // waves per block: a whole CU's worth works on one locality chunk
    const int nwinMax = d->maxL - d->h + 1;                               // windows of the longest read
    if (d->S == 4 && nwinMax <= 128) launch_fast<4, 8, 2, FW, HITS, TAIL>(d, A, blocks);
    else if (d->S == 8 && d->maxL <= 160 && nwinMax <= 128) launch_fast<8, 10, 2, FW, HITS, TAIL>(d, A, blocks);
    else if (d->S == 8 && d->maxL <= 160) launch_fast<8, 10, 3, 4, HITS, TAIL>(d, A, blocks);       // 129..160 windows, e.g. 150-bp reads with k <= 22 (three waves per SIMD: blocks of four)
    // (the 16-dword layout's state-machine rows take 12 KB of LDS per wave: four waves per block keep three blocks on a CU)
    else if (d->S == 8 && nwinMax <= 128) launch_fast<8, 16, 2, ((HITS || !TAIL) ? FW : 4), HITS, TAIL>(d, A, blocks);
    else if (d->S == 8) launch_fast<8, 16, 4, ((HITS || !TAIL) ? FW : 4), HITS, TAIL>(d, A, blocks);
    // 16-word layout (252 .. 504 bases, round 3): 20-dword compares and five windows per lane up to 320 bases / 320 windows (2 x 300 MiSeq reads), 32-dword
    // compares and eight windows per lane beyond; blocks of four waves (the state machine's rows take 10 / 18 KB of LDS per wave)
    else if (d->S == 16 && d->maxL <= 320 && nwinMax <= 320) launch_fast<16, 20, 5, 4, HITS, TAIL>(d, A, blocks);
    else if (d->S == 16 && nwinMax <= 512) launch_fast<16, 32, 8, 4, HITS, TAIL>(d, A, blocks);
    else return false;
    return true;
}
// The grid of the fast kernel.  Every block walks its chunks in a fixed round-robin order and is resident until its slowest wave is done, so the grid's granularity
// decides how long CUs idle at the end of the launch.  Measured at BASELINE configs[2] (round 3, uniform grids, blocks per CU -> kernel ms): 4 -> 92.2, 8 -> 90.8, 16 -> 89.6
// (rounds 1-2), 32 -> 88.7, 64 -> 88.0, 256 -> 88.4, one chunk per block -> 93.2 (a block's start and drain cost ~5 us).  Hence a TAPERED grid: up to three phases of 4096
// blocks that take 3/4 of what is left each (60 / 15 / ... rounds at configs[2]), then the rest in blocks of a few chunks -- the blocks the hardware dispatches last are short.
// Launches that write hits out reserve a 2048-slot chunk of the hit buffer per wave at a time: they keep one phase of 16 blocks per CU, or the buffer's slack quadruples.
// SAGE2OV_FAST_BLOCKS_PER_CU=<n>: one uniform phase of n blocks per CU (diagnostic).
// SAGE2OV_FAST_BLOCKS_PER_CU=<n>: one uniform phase of n blocks per CU (diagnostic); SAGE2OV_TEST_PHASE_BLOCKS=<b>: phases of b blocks instead of 4096 (tests: several
// phases on a few thousand reads).
static unsigned plan_fast_grid(const Options& O, ProbeArgs& A, u64 n, bool writesHits = false) {
    const char* eu = O.get("SAGE2OV_FAST_BLOCKS_PER_CU"); const int uniform = eu ? std::max(1, atoi(eu)) : 0;
    const char* ep = O.get("SAGE2OV_TEST_PHASE_BLOCKS"); const u64 PB = ep ? (u64)std::max(1, atoi(ep)) : 4096;
    memset(A.phase, 0, sizeof A.phase);
    // positions per block visit: twice FAST_CHUNK for launches big enough that the coarser grid does not show (kernels_probe_fast.inc: 24 M positions and more;
    // SAGE2OV_FAST_CHUNK_SHIFT = 7 / 8 overrides); launches that write hits out keep the small size
    u32 cs = (u32)FAST_CHUNK_LOG + ((n >= 24000000ull && !writesHits) ? 1u : 0u);
    if (const char* ec = O.get("SAGE2OV_FAST_CHUNK_SHIFT")) cs = (u32)std::max(FAST_CHUNK_LOG, std::min(FAST_CHUNK_LOG + 3, atoi(ec)));
    A.chunkShift = cs;
    const u64 CH = 1ull << cs, C = (n + CH - 1) / CH;
    if (writesHits || uniform || C <= PB * 8) {
        const char* eh = O.get("SAGE2OV_FAST_HITS_BLOCKS_PER_CU"); const int hitsPerCu = eh ? std::max(1, atoi(eh)) : 16;
        const u64 nb = std::max<u64>(1, std::min<u64>(C, 256ull * (writesHits ? hitsPerCu : (uniform ? uniform : 16))));
        A.phase[0][0] = 0; A.phase[0][1] = (u32)nb; A.phase[0][2] = (u32)((C + nb - 1) / nb); A.phase[0][3] = 0;
        return (unsigned)nb;
    }
    u64 rem = C, chunk0 = 0, block0 = 0; int p = 0;
    for (; p < 3 && rem > PB * 8; p++) {
        const u64 r = rem * 3 / 4 / PB;
        A.phase[p][0] = (u32)block0; A.phase[p][1] = (u32)PB; A.phase[p][2] = (u32)r; A.phase[p][3] = (u32)chunk0;
        block0 += PB; chunk0 += PB * r; rem -= PB * r;
    }
    const u64 nb = std::min<u64>(rem, 2 * PB);
    A.phase[p][0] = (u32)block0; A.phase[p][1] = (u32)nb; A.phase[p][2] = (u32)((rem + nb - 1) / nb); A.phase[p][3] = (u32)chunk0;
    return (unsigned)(block0 + nb);
}

void dev_set_probe_share(Device* d, double share) { d->probeShare = share; }
bool dev_has_minimiser_groups(Device* d) { return d->mi1 != nullptr && d->TL != 0; }

int dev_probe(Device* d, uint64_t lo, uint64_t hi, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (!d->slots) { err = "index not built"; return SAGE2OV_ERR_ARG; }
    const u64 N = d->N;
    { int rc = ensure_results(d, err); if (rc) return rc; }
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    HIPCHK(hipMemsetAsync(d->right, 0, (N + 1) * sizeof(u64), d->stream)); HIPCHK(hipMemsetAsync(d->left, 0, (N + 1) * sizeof(u64), d->stream));
    HIPCHK(hipMemsetAsync(d->conn, 0, (N + 1) * sizeof(u32), d->stream)); HIPCHK(hipMemsetAsync(d->cflag, 0, (N + 1) * sizeof(u32), d->stream));
    HIPCHK(hipMemsetAsync(d->d_counters + 6, 0, 2 * sizeof(u64), d->stream));          // [6] reads handed on, [7] of them: more candidates than the form has slots
    ProbeArgs A = base_args(d); A.lo = lo; A.hi = hi;
    const u64 nreads = hi > lo ? hi - lo : 0;
    d->pre.valid = false;
    const bool seq_only = d->opt.get("SAGE2OV_SEQUENTIAL_PROBE") != nullptr;
    if (nreads && !seq_only) {
        WS(slow, u32, WS_SLOW, nreads);
        A.slow = slow; A.slow_cap = nreads;
        // (positions [lo, hi) of the locality order: consecutive items are neighbours in the genome AND in the read store)
#ifdef SAGE2OV_STAMPS
        static u64* d_stamps = nullptr;
        if (!d_stamps) HIPCHK(hipMalloc(&d_stamps, 32 * sizeof(u64)));
        HIPCHK(hipMemsetAsync(d_stamps, 0, 32 * sizeof(u64), d->stream)); A.stamps = d_stamps;
#endif
        // Which kernel?  The one that carries the state machine for inconsistent reads (TAIL = 1) is 10 % slower on every read; on error-free
        // data a read in a thousand needs it.  So the first 1/128 of the range runs without it (TAIL = 0: such reads are listed), the share of
        // listed reads decides for the rest, and the listed reads go through the TAIL = 1 kernel as an id list afterwards; what that one
        // cannot settle either (more than 128 candidates, overhangs beyond its rows) ends in the sequential kernel, as before.
        const char* evs = d->opt.get("SAGE2OV_PROBE_SAMPLE_MIN");                 // tests: sample on small inputs too
        const u64 sampleMin = evs ? strtoull(evs, nullptr, 10) : (128u << 10);
        const char* evt = d->opt.get("SAGE2OV_PROBE_TAIL");                       // "0" / "1" / "2": no sampling, that kernel for everything
        u64 nsample = (nreads >= 4 * sampleMin && !evt) ? std::max<u64>(nreads / 128, sampleMin) : 0;   // (on noisy data the sample is work done twice)
        nsample = (nsample + 2 * FAST_CHUNK - 1) / (2 * FAST_CHUNK) * (2 * FAST_CHUNK);
        int tailKernel = evt ? atoi(evt) : 1; bool anyListed = evt && tailKernel == 0;       // 0 / 1 / 2: see k_probe_fast
        bool launched = true, mainWide = false;
        u64 nslow = 0, ncap = 0; float kms = 0;
        // the clean-data launches of one-length read sets run the sequential-groups form (eight waves per SIMD: -2.4 % at configs[2]); SAGE2OV_PROBE_SEQ=0: the standard form
        const bool seqForm = !(d->opt.get("SAGE2OV_PROBE_SEQ") && atoi(d->opt.get("SAGE2OV_PROBE_SEQ")) == 0);
        auto timed = [&](auto&& launch) -> int {                              // one launch of the fast kernel between two events
            HIPCHK(hipEventRecord(d->ev[2], d->stream));
            launch();
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(d->ev[3], d->stream));
            u64 c2_[2] = {0, 0}; HIPCHK(hipMemcpyAsync(c2_, d->d_counters + 6, sizeof c2_, hipMemcpyDeviceToHost, d->stream));
            HIPCHK(hipStreamSynchronize(d->stream)); nslow = c2_[0]; ncap = c2_[1];
            float ms = 0; hipEventElapsedTime(&ms, d->ev[2], d->ev[3]); kms += ms; d->tm.probe_fast_launches++;
            if (d->opt.get("SAGE2OV_TIMING")) fprintf(stderr, "[probe] fast kernel launch %.3f ms, listed so far %llu\n", ms, (unsigned long long)nslow);
            return 0;
        };
        // noisy data, and this context probes every read: the hits of the unresolved reads ARE the verified hits of this pass -- written out
        // now (k_probe_fast<..., 2> with hitBase), filtered by the final statuses in the reduce phase (k_hits_filter)
        auto arm_prehits = [&](ProbeArgs& P, u64 nr, unsigned nb, bool& armed) -> int {
            armed = false;
            if (!(tailKernel == 2 && d->probeShare == 1.0 && lo == 1 && hi == N + 1 && !d->opt.get("SAGE2OV_NO_PREHITS"))) return 0;
            const u64 hcap = nr * 72 + (u64)nb * SAGE2OV_FAST_WPB * HITS_CHUNK;
            Hit* hb = (Hit*)ws_get(d, WS_HITS, hcap * sizeof(Hit));                  // (72 hits per read: 49 GB at 42 M reads -- no room: the reduce phase makes its own lists)
            if (!hb) { (void)hipGetLastError(); return 0; }
            WS(hbase, u64, WS_PRE_BASE, N + 2); WS(hcnt, u32, WS_RA_CUR, N + 2);
            HIPCHK(hipMemsetAsync(hbase, 0xFF, (N + 2) * sizeof(u64), d->stream)); HIPCHK(hipMemsetAsync(hcnt, 0, (N + 2) * sizeof(u32), d->stream));
            HIPCHK(hipMemsetAsync(d->d_counters + 4, 0, 2 * sizeof(u64), d->stream));
            P.hits = hb; P.hits_cap = hcap; P.hitBase = hbase; P.hitcount = hcnt; armed = true;
            d->pre.hits = hb; d->pre.cap = hcap; d->pre.base = hbase;
            return 0;
        };
        auto close_prehits = [&]() -> int {
            u64 used = 0; HIPCHK(hipMemcpyAsync(&used, d->d_counters + 4, sizeof used, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
            d->pre.used = used; d->pre.valid = used <= d->pre.cap;                   // (a chunk beyond the buffer: hits were dropped, the reduce phase makes its own lists)
            return 0;
        };
        if (nsample && nsample < nreads) {
            ProbeArgs As = A; As.hi = lo + nsample;
            int rc = timed([&] { const unsigned nbs = plan_fast_grid(d->opt, As, nsample); launched = (seqForm && launch_fast_seq_any(d, As, nbs, 2)) || launch_fast_any<0, 0>(d, As, nbs); }); if (rc) return rc;
            if (launched) {
                anyListed = true;
                // listed for being inconsistent: more than half -> state machine for every read; more than 3 % -> kernel that carries it.  Listed for having more candidates
                // than slots (high coverage), more than 3 %: the rest runs the WIDE form (256 slots) right away instead of listing a fifth of the reads again
                const u64 nInc = nslow - std::min(nslow, ncap);
                tailKernel = nInc * 2 > nsample ? 2 : (nInc * 32 > nsample ? 1 : 0);
                mainWide = tailKernel == 0 && ncap * 32 > nsample && d->uniL != 0 && !d->opt.get("SAGE2OV_NO_WIDE");
                ProbeArgs Ar = A; Ar.lo = lo + nsample; const unsigned nb = plan_fast_grid(d->opt, Ar, nreads - nsample, tailKernel == 2);
                bool armed = false; { int rca = arm_prehits(Ar, nreads - nsample, nb, armed); if (rca) return rca; }
                rc = timed([&] { if (tailKernel == 2) launch_fast_any<0, 2>(d, Ar, nb); else if (tailKernel == 1) launch_fast_any<0, 1>(d, Ar, nb); else if (!((mainWide && launch_fast_seq_any(d, Ar, nb, 4)) || (seqForm && launch_fast_seq_any(d, Ar, nb, 2)))) launch_fast_any<0, 0>(d, Ar, nb); }); if (rc) return rc;
                if (armed) { int rca = close_prehits(); if (rca) return rca; }
            }
        } else {
            ProbeArgs Aw = A; const unsigned nb = plan_fast_grid(d->opt, Aw, nreads, tailKernel == 2); bool armed = false; { int rca = arm_prehits(Aw, nreads, nb, armed); if (rca) return rca; }
            int rc = timed([&] { launched = tailKernel == 2 ? launch_fast_any<0, 2>(d, Aw, nb) : (tailKernel == 1 ? launch_fast_any<0, 1>(d, Aw, nb) : ((seqForm && launch_fast_seq_any(d, Aw, nb, 2)) || launch_fast_any<0, 0>(d, Aw, nb))); }); if (rc) return rc;
            if (armed && launched) { int rca = close_prehits(); if (rca) return rca; }
        }
        if (!launched) {                                                       // 32-word layout: sequential kernel only
            HIPCHK(hipEventRecord(d->ev[2], d->stream));
            int rc = launch_probe<0>(d, A, err); if (rc) return rc;
            HIPCHK(hipEventRecord(d->ev[3], d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
            float ms = 0; hipEventElapsedTime(&ms, d->ev[2], d->ev[3]); kms += ms; nslow = 0;
        }
        u32* list = slow;
        // Reads the 128-slot forms listed, many of them (high coverage: more than 128 candidates; tests/diag/coverage_sweep.py): first the WIDE sequential-groups form
        // (256 candidate slots, no state machine); what it lists in turn -- inconsistent reads, more than 256 candidates -- goes on as before.
        const u64 wideMin = d->opt.get("SAGE2OV_PROBE_WIDE_MIN") ? strtoull(d->opt.get("SAGE2OV_PROBE_WIDE_MIN"), nullptr, 10) : 4096;
        if (launched && anyListed && nslow >= wideMin && tailKernel != 2 && !mainWide && d->uniL && !d->opt.get("SAGE2OV_NO_WIDE")) {
            WS(slow2, u32, WS_SLOW2, nslow);
            HIPCHK(hipMemsetAsync(d->d_counters + 6, 0, 2 * sizeof(u64), d->stream));
            ProbeArgs B = base_args(d); B.ids = slow; B.n_ids = nslow; B.slow = slow2; B.slow_cap = nslow;
            const u64 nl = nslow; const unsigned nbl = plan_fast_grid(d->opt, B, nl); bool wide = false;
            int rc = timed([&] { wide = launch_fast_seq_any(d, B, nbl, 4); }); if (rc) return rc;
            if (wide) {                                                          // (the survivors' list becomes the list: copied back so that the steps below find it where they expect it)
                if (nslow) HIPCHK(hipMemcpyAsync(slow, slow2, nslow * sizeof(u32), hipMemcpyDeviceToDevice, d->stream));
            } else nslow = nl;
        }
        if (launched && anyListed && nslow) {                                  // listed by the TAIL = 0 kernel: the state machine, in the TAIL = 1 kernel
            WS(slow2, u32, WS_SLOW2, nslow);
            HIPCHK(hipMemsetAsync(d->d_counters + 6, 0, 2 * sizeof(u64), d->stream));
            ProbeArgs B = base_args(d); B.ids = slow; B.n_ids = nslow; B.slow = slow2; B.slow_cap = nslow;
#ifdef SAGE2OV_STAMPS
            B.stamps = A.stamps;
#endif
            const u64 nl = nslow;
            const unsigned nbl = plan_fast_grid(d->opt, B, nl); int rc = timed([&] { launch_fast_any<0, 1>(d, B, nbl); }); if (rc) return rc;
            list = slow2;
        }
        d->tm.probe_kernel_ms += kms; d->tm.probe_launches++;
        d->tm.slow_reads += nslow;
#ifdef SAGE2OV_STAMPS
        { u64 st[32]; HIPCHK(hipMemcpy(st, d_stamps, sizeof st, hipMemcpyDeviceToHost)); u64 tot = 0; for (int x = 0; x < 24; x++) if (x < 10 || x >= 14) tot += st[x];
          // stages 0-9 in order of the code; 16 = slot -> window -> entry (before the gathers), 17 = reach, 18 = broadcast of the speculated reads, 19 = compares (8 = what follows them)
          fprintf(stderr, "[stamps] kernel %.2f ms; share of wave cycles (cycles per read):", kms); for (int x = 0; x < 24; x++) if (x < 10 || x >= 14) fprintf(stderr, " %d:%.1f%% (%.0f)", x, 100.0 * (double)st[x] / (double)tot, st[10] ? (double)st[x] / (double)st[10] : 0.0);
          fprintf(stderr, "; window reuse: %llu of %llu reads (same minimiser strand as the previous read: %llu), mean shift %.1f\n", (unsigned long long)st[11], (unsigned long long)st[10], (unsigned long long)st[13], st[11] ? (double)st[12] / (double)st[11] : 0.0);
          fprintf(stderr, "[stamps] run mode: %llu reads off the frame, %llu runs ended on a read the general path took; strand changes carried over by mirroring the ring: %llu (refused: %llu -- too far %llu, own entry not in the ring %llu, a window with one read twice %llu), steps: %llu\n", (unsigned long long)st[24], (unsigned long long)st[25], (unsigned long long)st[26], (unsigned long long)st[27], (unsigned long long)st[29], (unsigned long long)st[30], (unsigned long long)st[31], (unsigned long long)st[28]); }
#endif
        if (launched && nslow) {                                              // ambiguous / overflowing reads: sequential state machine
            ProbeArgs B = base_args(d); B.ids = list; B.n_ids = nslow;
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
    mem_sample(d);
    return 0;
}

// The wire form of a read's record, 16 bytes (24 until the end of round 3: the record all-gather is the largest exchange of a multi-GPU step).  Exact because a
// position is < 2^30 (the context's limit), an extension length < 2^22, the flags are two bits, and the connection count is at most (windows of a read) x (bucket cap
// of 101 entries, hashTable.cpp:178) < 2^18:   w0 = right pos:30 | type:2 | len:22 | flags:2 | conn[7:0]:8      w1 = left pos:30 | type:2 | len:22 | conn[17:8]:10
struct Record { u64 w0, w1; };
__device__ __forceinline__ u64 rec_half(u64 e) { return (e & 0x3FFFFFFFull) | (((e >> 40) & 3ull) << 30) | (((e >> 42) & 0x3FFFFFull) << 32); }
__device__ __forceinline__ u64 rec_entry(u64 w) { return (w & 0x3FFFFFFFull) | (((w >> 30) & 3ull) << 40) | (((w >> 32) & 0x3FFFFFull) << 42); }
// (a rank probes a range of POSITIONS of the locality order; since round 3 the per-read arrays are indexed by position too and name neighbours by position --
//  every rank computes the same order -- so a rank's records are a contiguous slice)
__global__ void k_pack_records(u64 lo, u64 hi, const u64* right, const u64* left, const u32* conn, const u32* cflag, Record* out) {
    u64 p = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x; if (p >= hi) return;
    const u32 c = min(conn[p], (1u << 18) - 1u), f = cflag[p] & 3u;
    Record r; r.w0 = rec_half(right[p]) | ((u64)f << 54) | ((u64)(c & 0xFFu) << 56); r.w1 = rec_half(left[p]) | ((u64)(c >> 8) << 54); out[p - lo] = r;
}
__global__ void k_unpack_records(u64 first, u64 n, const Record* in, u64* right, u64* left, u32* conn, u32* cflag) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (x >= n) return;
    Record r = in[x]; const u64 p = first + x; right[p] = rec_entry(r.w0); left[p] = rec_entry(r.w1); conn[p] = (u32)(r.w0 >> 56) | ((u32)(r.w1 >> 54) << 8);
    cflag[p] |= (u32)(r.w0 >> 54) & 3u;   // containment marks (economyGraph.cpp:735) are OR-ed, never overwritten: the local probe may have marked this read too
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
    if (n > d->cand_cap) { int rc = cand_resize(d, n + 1024, 0, err); if (rc) return rc; }
    if (n) HIPCHK(hipMemcpyAsync(d->cand, src, n * sizeof(EdgeCand), hipMemcpyDeviceToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream)); d->n_cand = n; return 0;
}
// survivor buckets of a sharded reduce phase: candidates [first, first + n) of the list (what this rank's marks re-emitted)
int dev_export_cand_range(Device* d, void* dst, uint64_t first, uint64_t n, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (first + n > d->n_cand) { err = "candidate range out of bounds"; return SAGE2OV_ERR_ARG; }
    if (n) HIPCHK(hipMemcpyAsync(dst, d->cand + first, n * sizeof(EdgeCand), hipMemcpyDeviceToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream)); return 0;
}
// the list becomes its first `keep` candidates followed by the n candidates at src (device memory)
int dev_replace_cand_tail(Device* d, uint64_t keep, const void* src, uint64_t n, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (keep > d->n_cand) { err = "candidate range out of bounds"; return SAGE2OV_ERR_ARG; }
    if (keep + n > d->cand_cap) {
        HIPCHK(hipStreamSynchronize(d->stream));
        { int rc = cand_resize(d, keep + n + 1024, keep, err); if (rc) return rc; }
    }
    if (n) HIPCHK(hipMemcpyAsync(d->cand + keep, src, n * sizeof(EdgeCand), hipMemcpyDeviceToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream)); d->n_cand = keep + n; return 0;
}
int dev_export_records(Device* d, void* dst, uint64_t lo, uint64_t hi, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (hi > lo) hipLaunchKernelGGL(k_pack_records, dim3(grid_for(hi - lo, 256)), dim3(256), 0, d->stream, (u64)lo, (u64)hi, d->right, d->left, d->conn, d->cflag, (Record*)dst);
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}
int dev_import_records(Device* d, const void* src, uint64_t first, uint64_t n, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (n) hipLaunchKernelGGL(k_unpack_records, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, (u64)first, (u64)n, (const Record*)src, d->right, d->left, d->conn, d->cflag);
    HIPCHK(hipStreamSynchronize(d->stream));
    return 0;
}

int dev_reciprocal(Device* d, uint64_t emit_lo, uint64_t emit_hi, uint64_t* n_ov, uint64_t* contained, uint64_t* contained_size, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const u64 N = d->N;
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    HIPCHK(hipMemsetAsync(d->d_counters, 0, 8 * sizeof(u64), d->stream));
    // (records by position; the status goes out by position -- statusP, what the emit half and the hit-list kernels read -- and by id)
    hipLaunchKernelGGL(k_recip_cond, dim3(grid_for(N, 256 * COND_PER_THREAD)), dim3(256), 0, d->stream, N, d->right, d->left, d->conn, d->cflag, d->idOf, d->status, d->statusP, d->d_counters);
    HIPCHK(hipEventRecord(d->ev[4], d->stream));
    if (d->diet && emit_hi > emit_lo) {                                     // the list is sized by a counting pass (capacity 0: nothing is written, the cursor counts)
        hipLaunchKernelGGL(k_recip_emit, dim3(grid_for(emit_hi - emit_lo, 256 * EMIT_PER_THREAD)), dim3(256), 0, d->stream, N, d->readsLoc, d->S, d->uniL, d->right, d->left, d->statusP, d->idOf, (EdgeCand*)nullptr, (u64)0, d->d_counters, (u64)emit_lo, (u64)emit_hi);
        u64 want = 0; HIPCHK(hipMemcpyAsync(&want, d->d_counters, sizeof want, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
        if (want + 1024 > d->cand_cap || !d->cand) { int rc = cand_resize(d, want + 1024, 0, err); if (rc) return rc; }
        HIPCHK(hipMemsetAsync(d->d_counters, 0, sizeof(u64), d->stream));
    }
    if (emit_hi > emit_lo)
        hipLaunchKernelGGL(k_recip_emit, dim3(grid_for(emit_hi - emit_lo, 256 * EMIT_PER_THREAD)), dim3(256), 0, d->stream, N, d->readsLoc, d->S, d->uniL, d->right, d->left, d->statusP, d->idOf, d->cand, d->cand_cap, d->d_counters, (u64)emit_lo, (u64)emit_hi);
    u64 c[8];
    HIPCHK(hipMemcpyAsync(c, d->d_counters, sizeof c, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipEventRecord(d->ev[1], d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]); d->tm.reciprocal_ms += ms;
    hipEventElapsedTime(&ms, d->ev[0], d->ev[4]); d->tm.recip_cond_ms += ms;
    if (c[0] > d->cand_cap) { err = "edge candidate buffer overflow"; return SAGE2OV_ERR_INTERNAL; }
    d->n_cand = c[0]; *n_ov = c[1]; *contained = c[2]; *contained_size = c[3];
    return 0;
}

int dev_download_initial(Device* d, uint64_t* right, uint64_t* left, uint8_t* status, uint32_t* conn, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const u64 N = d->N;
    if (right || left || conn) {                                           // the C ABI's arrays are indexed by id and name neighbours by id (economyGraph.h:24-30)
        u64* ri = nullptr; u64* li = nullptr; u32* ci = nullptr;
        struct Free { u64*& a; u64*& b; u32*& c; ~Free() { hipFree(a); hipFree(b); hipFree(c); } } fr{ri, li, ci};
        HIPCHK(hipMalloc(&ri, (N + 1) * sizeof(u64))); HIPCHK(hipMalloc(&li, (N + 1) * sizeof(u64))); HIPCHK(hipMalloc(&ci, (N + 1) * sizeof(u32)));
        hipLaunchKernelGGL(k_records_to_ids, dim3(grid_for(N + 1, 256)), dim3(256), 0, d->stream, (u64)N, d->idOf, d->right, d->left, d->conn, ri, li, ci);
        HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(d->stream));
        if (right) HIPCHK(hipMemcpy(right, ri, (N + 1) * sizeof(u64), hipMemcpyDeviceToHost));
        if (left) HIPCHK(hipMemcpy(left, li, (N + 1) * sizeof(u64), hipMemcpyDeviceToHost));
        if (conn) HIPCHK(hipMemcpy(conn, ci, (N + 1) * sizeof(u32), hipMemcpyDeviceToHost));
    }
    if (status) HIPCHK(hipMemcpy(status, d->status, N + 1, hipMemcpyDeviceToHost));
    return 0;
}
int dev_download_status(Device* d, std::vector<uint8_t>& status, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    status.resize(d->N + 1);
    HIPCHK(hipMemcpy(status.data(), d->status, d->N + 1, hipMemcpyDeviceToHost));
    return 0;
}

int dev_unresolved_ids(Device* d, std::vector<uint32_t>& ids, std::string& err);
int dev_unresolved_hits(Device* d, std::vector<Hit>& hits, uint64_t* n_unresolved, std::string& err, std::vector<uint32_t>* ids_out) {
    HIPCHK(hipSetDevice(d->ordinal));
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    std::vector<uint32_t> ids; int rc = dev_unresolved_ids(d, ids, err); if (rc) return rc;
    u64 nun = ids.size();
    *n_unresolved = nun; hits.clear(); if (ids_out) *ids_out = ids;
    if (nun == 0) return 0;
    u64 cap = std::max<u64>(1 << 16, nun * 80);
    for (int attempt = 0; attempt < 4; attempt++) {
        WS(dh, Hit, WS_HITS, cap);
        HIPCHK(hipMemsetAsync(d->d_counters + 4, 0, sizeof(u64), d->stream));
        ProbeArgs A = base_args(d); A.hits = dh; A.hits_cap = cap;
        { WS(idbuf, u32, WS_IDS, nun); WS(pbuf, u32, WS_SLOW, nun + 1);      // the list dev_unresolved_ids left on the device, as positions: only these reads are probed
          hipLaunchKernelGGL(k_ids_to_pos, dim3(grid_for(nun, 256)), dim3(256), 0, d->stream, idbuf, (u64)nun, d->posOf, pbuf); A.ids = pbuf; A.n_ids = nun; }
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
// The serial part of the reduce phase when some bucket is long (economyGraph.cpp:513-564): the order in which the unresolved reads are
// explored.  plist[offp[w] .. offp[w+1]) = potential list of the w-th unresolved read, sorted like the reference sorts a list when the
// read is explored (:853-871); an entry is `to | twin << 31`.  An own hit is in the read's list iff the target was still unexplored
// when the read was explored, a twin iff its source had been explored before; candidates of the reciprocal pass (hasCand) are always
// there but their far ends are never explorable.  Returns rank[id] (1-based exploration order; 0: not an unresolved read).
// Both inner loops look for RARE entries (a still unexplored target; an explored but unmarked neighbour) among ~100 per list, so they are
// written as "find the next entry that satisfies the test": eight entries per step with AVX2 gathers of rank[] where the host has them
// (the walk is instruction-bound: 2 ns per entry visit with a scalar loop, 1.9 G visits per 10 M reads).  After every event the search
// restarts behind it with fresh values, so a batch never acts on state that an event of the same batch has changed.
// Tables of the walk live on 2 MB pages where the kernel hands them out (transparent huge pages, madvise mode): the walk's accesses are
// spread over ~2 GB (lists) + 170 MB (tables), far beyond what a TLB of 4 KB pages covers.
struct HugeBuf {
    void* p = nullptr; size_t bytes = 0; bool registered = false;
    void* get(size_t n) {
        bytes = (std::max<size_t>(n, 1) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) { p = nullptr; return nullptr; }
        madvise(p, bytes, MADV_HUGEPAGE);
        return p;
    }
    ~HugeBuf() { if (p) { if (registered) hipHostUnregister(p); munmap(p, bytes); } }
};
struct RrStaging {                                       // slice s of a call reuses buffer s of the previous call when it is large enough
    std::vector<std::unique_ptr<HugeBuf>> bufs; size_t next = 0;
    u32* get(size_t n) {
        const size_t want = (std::max<size_t>(n, 16) + 16) * sizeof(u32);
        if (next < bufs.size() && bufs[next]->p && bufs[next]->bytes >= want) return (u32*)bufs[next++]->p;
        std::unique_ptr<HugeBuf> b(new HugeBuf()); u32* p = (u32*)b->get(want + want / 8); if (!p) return nullptr;
        memset(p, 0, b->bytes);                                  // (touch: the pages exist before they are pinned)
        if (hipHostRegister(p, b->bytes, hipHostRegisterDefault) == hipSuccess) b->registered = true; else (void)hipGetLastError();
        if (next < bufs.size()) bufs[next] = std::move(b); else bufs.push_back(std::move(b));
        return (u32*)bufs[next++]->p;
    }
};
static void rr_staging_release(Device* d) { if (d->rrStaging) { delete (RrStaging*)d->rrStaging; d->rrStaging = nullptr; } }
// (the exploration walk itself -- explore_order, host code with AVX2 gathers where the host has them -- lives in sage2ov_walk.cpp since round 4: it never was device code)
// Multi-rank contexts (shareRank / shareWorld): hit lists, adjacency and -- with long buckets -- the exploration order are computed for ALL unresolved
// reads on every rank (they are read by everybody's marks), but the marks (:643-707), the removals and the re-emission of the surviving edges run for this
// rank's share of the unresolved reads only (a contiguous part of the list): d->n_cand grows by this rank's survivors, *removed counts this rank's
// removals; the caller all-gathers the survivor buckets (dev_export_survivors / dev_set_survivors) and sums the counters.
int dev_reduce_device(Device* d, uint64_t min_unresolved, uint64_t* n_unresolved, uint64_t* n_hits, uint64_t* inserted, uint64_t* removed, int* done, std::string& err,
                      uint32_t shareRank, uint32_t shareWorld) {
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
    if (nun < min_unresolved) return 0;
    const bool ranked = d->n_long != 0;                                          // one-sided discovery: the exploration order decides which edges exist
    const bool timing = d->opt.get("SAGE2OV_TIMING") != nullptr; auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; hipStreamSynchronize(d->stream); auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[reduce/device] %-30s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tp).count()); tp = t; };
    // directional hits of the unresolved reads, device resident: the fast kernel in its hit-list form (locality order, minimiser
    // groups), the sequential kernel for the few reads it hands over (> 128 candidates, ambiguous tags) and for the 16-word layout
    Hit* dh = nullptr; u64 nh = 0, nslots = 0; u64 dbgFastEnd = 0;
    WS(hitcount, u32, WS_RA_CUR, N + 2);
    u32* locDev = nullptr;                                                       // ranked form: read id -> 1-based position in the locality order
    {
        const u32* order = d->idOf + 1;                                            // ids in locality order (positions 1..N)
        if (ranked) locDev = d->posOf;
        if (ranked || shareWorld > 1) {                                            // (several ranks cut the SAME list into shares: it must not depend on the order of atomics)
            // the unresolved reads listed in LOCALITY order (a stable compaction of `order`): their potential lists are then laid out in that
            // order too, so the lists the host's walk visits one after the other sit next to each other in memory (cache lines, TLB)
            WS(flg, u32, WS_RR_IN, N + 2); WS(fpos, u32, WS_RR_WIDX, N + 2);
            hipLaunchKernelGGL(k_rr_unres_flag, dim3(grid_for(N, 256)), dim3(256), 0, d->stream, order, (u64)N, d->status, flg);
            u64 cnt2 = 0; { int rc = scan_u32(d, flg, N, fpos, &cnt2, err); if (rc) return rc; }
            if (cnt2 != nun) { err = "unresolved read count changed"; return SAGE2OV_ERR_INTERNAL; }
            hipLaunchKernelGGL(k_rr_unres_pick, dim3(grid_for(N, 256)), dim3(256), 0, d->stream, order, (u64)N, flg, fpos, ids);
        }
        WS(slow, u32, WS_SLOW, N + 1);
        const unsigned blocks = (unsigned)std::min<u64>((N + FAST_CHUNK - 1) / FAST_CHUNK, 256ull * 16);
        u64 cap = std::max<u64>(1 << 16, nun * 80) + (u64)blocks * SAGE2OV_FAST_WPB * HITS_CHUNK; bool ok = false;
        if (d->opt.get("SAGE2OV_TEST_SMALL_BUFFERS")) cap = 8192;                      // tests: start far too small, the sizing loop must recover
        // The initial pass may have written every read's hits out already (dev_probe, noisy data): drop the ones with a resolved end, in place, and
        // run the hit-list kernel only for the unresolved reads that pass did not cover (its sample, hand-overs).  Any shortage of room: the
        // ordinary way below, from scratch.
        if (d->pre.valid && !d->opt.get("SAGE2OV_TEST_SMALL_BUFFERS")) {
            d->pre.valid = false;                                                     // (consumed: the filter works in place)
            dh = d->pre.hits; const u64 pcap = d->pre.cap; u64 used = d->pre.used;
            WS(noneList, u32, WS_PRE_NONE, N + 2);
            HIPCHK(hipMemsetAsync(d->d_counters + 22, 0, 2 * sizeof(u64), d->stream));
            hipLaunchKernelGGL(k_hits_filter, dim3((unsigned)std::min<u64>((N + 3) / 4, 256ull * 64)), dim3(256), 0, d->stream, dh, d->pre.base, d->status, d->statusP, d->idOf, d->posOf, (u64)N, hitcount, noneList, d->d_counters + 22);
            u64 fc[2] = {0, 0}; HIPCHK(hipMemcpyAsync(fc, d->d_counters + 22, sizeof fc, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
            HIPCHK(hipGetLastError());
            nh = fc[0]; ok = true;
            if (fc[1]) {                                                              // unresolved reads without written hits: the hit-list kernel over their positions, appending
                u64 c3[3] = {used, 0, 0}; HIPCHK(hipMemcpyAsync(d->d_counters + 4, c3, sizeof c3, hipMemcpyHostToDevice, d->stream));
                ProbeArgs A = base_args(d); A.ids = noneList; A.n_ids = fc[1]; A.hits = dh; A.hits_cap = pcap; A.hitcount = hitcount; A.slow = slow; A.slow_cap = N + 1;
                if (launch_fast_any<1, 0>(d, A, plan_fast_grid(d->opt, A, fc[1], true))) {
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipMemcpyAsync(c3, d->d_counters + 4, sizeof c3, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
                    if (c3[0] > pcap) ok = false;
                    else if (c3[2]) { ProbeArgs B = base_args(d); B.hits = dh; B.hits_cap = pcap; B.hitcount = hitcount; B.ids = slow; B.n_ids = c3[2]; int rc = launch_probe<1>(d, B, err); if (rc) return rc; }
                } else { A.slow = nullptr; int rc = launch_probe<1>(d, A, err); if (rc) return rc; }
                if (ok) {
                    u64 used2 = 0; HIPCHK(hipMemcpyAsync(&used2, d->d_counters + 4, sizeof used2, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
                    if (used2 > pcap) ok = false; else { nh += c3[1] + (used2 - std::max(c3[0], used)); used = used2; }
                }
            }
            if (ok) nslots = used;
        }
        for (int attempt = 0; attempt < 4 && !ok; attempt++) {
            WS(hb, Hit, WS_HITS, cap); dh = hb;
            HIPCHK(hipMemsetAsync(d->d_counters + 4, 0, 3 * sizeof(u64), d->stream));
            HIPCHK(hipMemsetAsync(hitcount, 0, (N + 2) * sizeof(u32), d->stream));
            ProbeArgs A = base_args(d); A.lo = 1; A.hi = N + 1; A.hits = dh; A.hits_cap = cap; A.hitcount = hitcount;
            A.slow = slow; A.slow_cap = N + 1;                                       // (all positions; the kernel skips what is not status 0)
            u64 c3[3] = {0, 0, 0};
            if (launch_fast_any<1, 0>(d, A, blocks)) {
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(c3, d->d_counters + 4, sizeof c3, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
                if (c3[0] > cap) { cap = c3[0] + c3[0] / 8 + 1024; continue; }            // some chunk did not fit: everything again
                dbgFastEnd = c3[0];
                if (c3[2]) {                                                                  // handed over: exact sequential kernel, appends behind
                    ProbeArgs B = base_args(d); B.hits = dh; B.hits_cap = cap; B.hitcount = hitcount; B.ids = slow; B.n_ids = c3[2];
                    int rc = launch_probe<1>(d, B, err); if (rc) return rc;
                }
            } else {
                A.slow = nullptr;
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
    lap("hit lists");
    if (d->opt.get("SAGE2OV_DEBUG_HITS") && nslots) {                                    // diagnostic: an order-independent checksum of the hit lists
        std::vector<Hit> hh(nslots); HIPCHK(hipMemcpy(hh.data(), dh, nslots * sizeof(Hit), hipMemcpyDeviceToHost));
        u64 sum = 0, cnt = 0, sumseq = 0, sumF = 0, cntF = 0; u64 idx = 0;
        for (const Hit& h : hh) { const bool fastPart = idx++ < dbgFastEnd; if (h.from) { u64 x = ((u64)h.from << 32) ^ ((u64)h.to * 0x9E3779B97F4A7C15ull) ^ ((u64)(u32)h.len << 8) ^ h.type; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; sum += x; sumseq += h.seq; cnt++; if (fastPart) { sumF += x; cntF++; } } }
        if (const char* path = d->opt.get("SAGE2OV_DEBUG_HITS_FILE")) { FILE* f = fopen(path, "wb"); if (f) { fwrite(hh.data(), sizeof(Hit), (size_t)std::min<u64>(dbgFastEnd, nslots), f); fclose(f); } }
        fprintf(stderr, "[reduce/device] hits %llu in %llu slots, checksum %016llx, seq sum %llu, nh %llu; fast kernel's part: %llu hits, checksum %016llx\n", (unsigned long long)cnt, (unsigned long long)nslots, (unsigned long long)sum, (unsigned long long)sumseq, (unsigned long long)nh, (unsigned long long)cntF, (unsigned long long)sumF);
    }
    const u64 nc = d->n_cand;
    WS(deg, u32, WS_RA_DEG, N + 2); WS(offs, u32, WS_RA_OFF, N + 2); WS(cur, u32, WS_CURSOR, N + 2);
    HIPCHK(hipMemsetAsync(deg, 0, (N + 2) * sizeof(u32), d->stream)); HIPCHK(hipMemsetAsync(cur, 0, (N + 2) * sizeof(u32), d->stream));
    if (nc) hipLaunchKernelGGL(k_ra_degree_c, dim3(grid_for(nc, 256)), dim3(256), 0, d->stream, d->cand, (u64)nc, d->status, deg);
    u32* rankDev = nullptr; u64 present = 0;
    HIPCHK(hipMemsetAsync(d->d_counters + 8, 0, 4 * sizeof(u64), d->stream));
    if (ranked) {
        // potential lists (own hits + twins of incoming hits), sorted and merged on the device, slice by slice (a slice stays below 2^30
        // entries); exploration order on the host; ranks back
        WS(incount, u32, WS_RR_IN, N + 2); WS(widx, u32, WS_RR_WIDX, N + 2); WS(degp, u32, WS_RR_DEGP, nun + 2); WS(pcur, u32, WS_RR_CUR, N + 2);
        HIPCHK(hipMemsetAsync(incount, 0, (N + 2) * sizeof(u32), d->stream)); HIPCHK(hipMemsetAsync(pcur, 0, (N + 2) * sizeof(u32), d->stream)); HIPCHK(hipMemsetAsync(widx, 0, (N + 2) * sizeof(u32), d->stream));
        hipLaunchKernelGGL(k_rr_widx, dim3(grid_for(nun, 256)), dim3(256), 0, d->stream, ids, (u64)nun, widx);
        if (nslots) hipLaunchKernelGGL(k_rr_incount, dim3(grid_for(nslots, 256)), dim3(256), 0, d->stream, dh, (u64)nslots, incount);
        hipLaunchKernelGGL(k_rr_degp, dim3(grid_for(nun, 256)), dim3(256), 0, d->stream, ids, (u64)nun, hitcount, incount, degp);
        std::vector<u32> hIds(nun), hLen(nun, 0), hDeg(N + 2), hDegp(nun), hLoc(N + 2); std::vector<uint8_t> hasCand(nun, 0); std::vector<const u32*> listPtr(nun, nullptr);
        HIPCHK(hipMemcpyAsync(hLoc.data(), locDev, (N + 2) * sizeof(u32), hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipMemcpyAsync(hDegp.data(), degp, nun * sizeof(u32), hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipMemcpyAsync(hIds.data(), ids, nun * sizeof(u32), hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipMemcpyAsync(hDeg.data(), deg, (N + 2) * sizeof(u32), hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));   // (deg so far: candidate entries only)
        for (u64 w = 0; w < nun; w++) hasCand[w] = hDeg[hIds[w]] != 0;
        if (timing) { u64 c512 = 0, c1k = 0, c4k = 0; u32 mx = 0; for (u32 v : hDegp) { c512 += v > 512; c1k += v > 1024; c4k += v > 4096; mx = std::max(mx, v); }
            fprintf(stderr, "[reduce/device] potential lists: %llu reads, > 512: %llu, > 1024: %llu, > 4096: %llu, longest %u\n", (unsigned long long)nun, (unsigned long long)c512, (unsigned long long)c1k, (unsigned long long)c4k, mx); }
        // gigabytes at 10 M reads: 2 MB pages for the host's walk, registered so that the download is a DMA; kept in the context from step to step (allocating, touching
        // and registering 1.7 GB took 100 ms of every call)
        if (!d->rrStaging) d->rrStaging = new RrStaging();
        RrStaging& staging = *(RrStaging*)d->rrStaging; staging.next = 0;
        std::vector<std::vector<u32>> heavyLists;                                 // lists beyond the device sort (reads that thousands of others see): sorted and merged on the host
        u64 sliceEntries = 1ull << 30; if (const char* ev = d->opt.get("SAGE2OV_TEST_RANK_SLICE")) sliceEntries = std::max<u64>(1024, strtoull(ev, nullptr, 10));
        WS(lenp, u32, WS_RR_LEN, nun + 2);
        for (u64 w0 = 0; w0 < nun;) {
            u64 w1 = w0, tot0 = 0; while (w1 < nun && (w1 == w0 || tot0 + hDegp[w1] <= sliceEntries)) tot0 += hDegp[w1++];
            if (tot0 >= (1ull << 32) - 64) return 0;                              // one read with 2^32 potential neighbours: not this path
            const u64 nw = w1 - w0;
            WS(offp, u32, WS_RR_OFFP, nw + 2); WS(coff, u32, WS_RR_COFF, nw + 2);
            u64 totp = 0; { int rc = scan_u32(d, degp + w0, nw, offp, &totp, err); if (rc) return rc; }
            WS(entp, u64, WS_RR_ENTP, totp + 64); WS(outp, u32, WS_RR_OUTP, totp + 64);
            HIPCHK(hipMemsetAsync(entp, 0, (totp + 64) * sizeof(u64), d->stream));
            if (nslots) hipLaunchKernelGGL(k_rr_fillp, dim3(grid_for(nslots, 256)), dim3(256), 0, d->stream, dh, (u64)nslots, d->reads, d->S, d->uniL, widx, (u32)w0, (u32)w1, offp, hitcount, pcur, entp);
            hipLaunchKernelGGL(k_rr_sortp, dim3((unsigned)std::min<u64>((nw + 3) / 4, 256ull * 16)), dim3(256), 0, d->stream, (u64)nw, offp, degp + w0, entp, outp, lenp + w0, d->d_counters + 8 + 3);
            u64 totc = 0; { int rc = scan_u32(d, lenp + w0, nw, coff, &totc, err); if (rc) return rc; }
            WS(outc, u32, WS_RR_OUTC, totc + 64);
            hipLaunchKernelGGL(k_rr_compact, dim3((unsigned)std::min<u64>((nw + 3) / 4, 256ull * 16)), dim3(256), 0, d->stream, (u64)nw, offp, lenp + w0, coff, outp, locDev, outc);
            lap("  potential lists: kernels");
            u32* hbuf = staging.get(totc); if (!hbuf) { err = "host staging buffer allocation failed"; return SAGE2OV_ERR_NOMEM; }
            lap("  potential lists: staging buffer (2 MB pages, touched, registered)");
            HIPCHK(hipMemcpyAsync(hLen.data() + w0, lenp + w0, nw * sizeof(u32), hipMemcpyDeviceToHost, d->stream));
            if (totc) HIPCHK(hipMemcpyAsync(hbuf, outc, totc * sizeof(u32), hipMemcpyDeviceToHost, d->stream));
            HIPCHK(hipStreamSynchronize(d->stream));
            u64 at = 0, rel = 0; std::vector<u64> seg;
            for (u64 w = w0; w < w1; w++) {
                const u32 n = hDegp[w];
                if (n <= (u32)RR_CAP) { listPtr[w] = hbuf + at; at += hLen[w]; }
                else {                                                            // (its device length is 0)
                    seg.resize(n); HIPCHK(hipMemcpy(seg.data(), entp + rel, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost));
                    std::sort(seg.begin(), seg.end(), std::greater<u64>());
                    heavyLists.emplace_back(); std::vector<u32>& hl = heavyLists.back(); hl.reserve(n);
                    for (u32 x = 0; x < n; x++) {
                        const u64 kx = seg[x]; if (kx == 0) break;
                        const bool twin = kx & 1ull;
                        if (!twin && x > 0 && (seg[x - 1] & 1ull) && (seg[x - 1] >> 1) == (kx >> 1)) continue;      // the own hit behind its twin: merged
                        const bool sym = twin && x + 1 < n && seg[x + 1] != 0 && !(seg[x + 1] & 1ull) && (seg[x + 1] >> 1) == (kx >> 1);
                        hl.push_back(hLoc[(u32)(kx >> 11) & 0x3FFFFFFFu] | ((sym ? 0u : (twin ? 2u : 1u)) << 30));
                    }
                    hLen[w] = (u32)hl.size();
                }
                rel += n;
            }
            w0 = w1;
        }
        { size_t hx = 0; for (u64 w = 0; w < nun; w++) if (hDegp[w] > (u32)RR_CAP) listPtr[w] = heavyLists[hx++].data(); }      // (after the last push_back: the vectors no longer move)
        lap("potential lists (build + sort + download)");
        std::vector<u32> rankByPos, posOf(nun), startOrder(nun);
        {   // positions of the unresolved reads in list order and in ascending id order (the walk's starts): two gathers and a compaction, on the device
            WS(sflg, u32, WS_RR_IN, N + 2); WS(sfpos, u32, WS_RR_WIDX, N + 2); WS(sout, u32, WS_RR_CUR, N + 2); WS(pout, u32, WS_RR_DEGP, nun + 2);
            hipLaunchKernelGGL(k_ids_to_pos, dim3(grid_for(nun, 256)), dim3(256), 0, d->stream, ids, (u64)nun, d->posOf, pout);
            hipLaunchKernelGGL(k_rr_start_flag, dim3(grid_for(N, 256)), dim3(256), 0, d->stream, (u64)N, d->status, sflg);
            u64 cnt3 = 0; { int rc = scan_u32(d, sflg, N, sfpos, &cnt3, err); if (rc) return rc; }
            if (cnt3 != nun) { err = "unresolved read count changed"; return SAGE2OV_ERR_INTERNAL; }
            hipLaunchKernelGGL(k_rr_start_pick, dim3(grid_for(N, 256)), dim3(256), 0, d->stream, (u64)N, sflg, sfpos, d->posOf, sout);
            HIPCHK(hipMemcpyAsync(posOf.data(), pout, nun * sizeof(u32), hipMemcpyDeviceToHost, d->stream));
            HIPCHK(hipMemcpyAsync(startOrder.data(), sout, nun * sizeof(u32), hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
        }
        lap("walk set-up (device + download)");
        explore_order(d->opt, posOf, listPtr, hLen, hasCand, N, startOrder, rankByPos);
        if (rankByPos.size() != N + 2) { err = "exploration walk: table allocation failed"; return SAGE2OV_ERR_NOMEM; }
        lap("exploration order (host)");
        { WS(rk, u32, WS_RR_RANK, N + 2); rankDev = rk; }
        { WS(rl, u32, WS_RR_RANKL, N + 2);
          HIPCHK(hipMemcpyAsync(rl, rankByPos.data(), (N + 2) * sizeof(u32), hipMemcpyHostToDevice, d->stream));
          hipLaunchKernelGGL(k_rr_rank_by_id, dim3(grid_for(N + 2, 256)), dim3(256), 0, d->stream, rl, locDev, (u64)(N + 2), rankDev);
          HIPCHK(hipStreamSynchronize(d->stream)); }
        if (nslots) hipLaunchKernelGGL(k_rr_degree_h, dim3(grid_for(nslots, 256)), dim3(256), 0, d->stream, dh, (u64)nslots, rankDev, deg, d->d_counters + 8);
    } else hipLaunchKernelGGL(k_ra_degree_h, dim3(grid_for(N + 1, 256)), dim3(256), 0, d->stream, hitcount, (u64)N, deg);
    lap("  ranks to the device, degrees of the final lists");
    u64 tot = 0; { int rc = scan_u32(d, deg, N + 2, offs, &tot, err); if (rc) return rc; }
    if (tot >= (1ull << 32) - 64) return 0;
    WS(ent, u64, WS_RA_ENT, tot + 64); WS(rm, uint8_t, WS_RA_RM, tot + 64); WS(ent32, u32, WS_RA_ENT32, tot + 64);
    if (nc) hipLaunchKernelGGL(k_ra_fill_c, dim3(grid_for(nc, 256)), dim3(256), 0, d->stream, d->cand, (u64)nc, d->reads, d->S, d->uniL, offs, cur, ent, ent32);
    if (ranked) { if (nslots) hipLaunchKernelGGL(k_rr_fill_h, dim3(grid_for(nslots, 256)), dim3(256), 0, d->stream, dh, (u64)nslots, rankDev, d->reads, d->S, d->uniL, offs, cur, ent, ent32); }
    else if (nslots) hipLaunchKernelGGL(k_ra_fill_h, dim3(grid_for(nslots, 256)), dim3(256), 0, d->stream, dh, (u64)nslots, offs, deg, hitcount, ent, ent32);
    if (ranked) { u64 c3[3]; HIPCHK(hipMemcpyAsync(c3, d->d_counters + 8, sizeof c3, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream)); present = c3[2]; }
    HIPCHK(hipMemsetAsync(d->d_counters + 8, 0, 5 * sizeof(u64), d->stream));
    // this rank's share of the unresolved reads: entries [wlo, whi) of the list (any contiguous cut is exact: a read's marks depend on the lists only)
    const u64 wlo = shareWorld > 1 ? nun * shareRank / shareWorld : 0, whi = shareWorld > 1 ? nun * (shareRank + 1) / shareWorld : nun;
    const u32* const idsAll = ids; ids = ids + wlo; const u64 nunAll = nun; nun = whi - wlo; (void)idsAll; (void)nunAll;
    WS(svn, u32, WS_NEED, nun + 2); WS(svoff, u32, WS_OWNER, nun + 2);
    // (the marks' grid: a wave per read, blocks walk the list round-robin; blocks per CU -> reduce phase at configs[1] + 0.1 % errors: 16 -> 69.5 ms, 64 -> 66.0, 256 -> 65.2, 1024 -> 65.2: the tail again)
    const u64 raPerCu = d->opt.get("SAGE2OV_RA_GRID_PER_CU") ? std::max(1, atoi(d->opt.get("SAGE2OV_RA_GRID_PER_CU"))) : 256;
    const unsigned gb = (unsigned)std::max<u64>(1, std::min<u64>((nun + 3) / 4, 256ull * raPerCu));
    const u64 heavyCap = 1 << 16; WS(heavy, u32, WS_RA_HEAVY, heavyCap);
    lap("  final lists filled");
    HIPCHK(hipEventRecord(d->ev[5], d->stream));                          // (marks_ms: the sharded part of the phase -- marks, removals, re-emission)
    const u32 noShortcut = d->opt.get("SAGE2OV_RA_NO_SHORTCUT") ? 1u : 0u;      // (tests: the walk of every list, as until round 4)
    WS(mid, u32, WS_RA_SPLIT, nun + 2);                                          // the reads with more than 128 entries, listed by the first launch for the second
    if (nun) {
    hipLaunchKernelGGL((k_ra_mark<128, 8, 0, false>), dim3(gb), dim3(256), 0, d->stream, ids, (u64)nun, offs, deg, ent, ent32, rm, svn, d->d_counters + 8, heavy, heavyCap, noShortcut, mid);     // lists of <= 128 entries
    hipLaunchKernelGGL((k_ra_mark<RA_CAP, 10, 128, true>), dim3(gb), dim3(256), 0, d->stream, ids, (u64)nun, offs, deg, ent, ent32, rm, svn, d->d_counters + 8, heavy, heavyCap, noShortcut, mid);   // 129 .. RA_CAP (the reads the first launch listed); longer: k_ra_mark_big
    }
    u64 c[4];
    HIPCHK(hipMemcpyAsync(c, d->d_counters + 8, sizeof c, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
    HIPCHK(hipGetLastError());
    if (timing) fprintf(stderr, "[reduce/device] marks of %llu of %llu reads by the short cut (k_ra_mark: ra_shortcut), %.2f lists read per read\n", (unsigned long long)c[3], (unsigned long long)nun, nun ? (double)c[2] / (double)nun : 0.0);
    if (c[0] > heavyCap) { if (shareWorld > 1) { err = "reduce: too many oversized lists in this rank's share"; return SAGE2OV_ERR_LIMIT; } return 0; }   // (never seen) that many oversized lists: serial replay (a rank of many cannot decide that alone)
    if (c[0]) {                                                           // lists beyond the LDS kernel: same marking out of global scratch, one wavefront each
        const u32 nhv = (u32)c[0];
        WS(hsize, u32, WS_RA_HSIZE, nhv); WS(hscr, u64, WS_RA_HSCR, nhv);
        hipLaunchKernelGGL(k_ra_heavy_sizes, dim3(grid_for(nhv, 256)), dim3(256), 0, d->stream, heavy, nhv, ids, deg, hsize);
        std::vector<u32> hs(nhv); HIPCHK(hipMemcpyAsync(hs.data(), hsize, nhv * sizeof(u32), hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));   // (the stream is non-blocking: the null stream does not wait for it)
        std::vector<u64> so(nhv); u64 words = 0;
        for (u32 b = 0; b < nhv; b++) { u64 P = 64; while (P < hs[b]) P <<= 1; so[b] = words; words += P + 2 * P + 2 * P; }      // key[P] u64 + ht[4P] u32 + mk[4P] u32
        HIPCHK(hipMemcpyAsync(hscr, so.data(), nhv * sizeof(u64), hipMemcpyHostToDevice, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
        u64* scratch = nullptr; HIPCHK(hipMalloc(&scratch, words * sizeof(u64)));
        hipLaunchKernelGGL(k_ra_mark_big, dim3(nhv), dim3(64), 0, d->stream, ids, heavy, hscr, scratch, offs, deg, ent, rm, svn, d->d_counters + 8);
        hipError_t e2 = hipStreamSynchronize(d->stream); hipFree(scratch);
        if (e2 != hipSuccess) { err = std::string("k_ra_mark_big: ") + hipGetErrorString(e2); return SAGE2OV_ERR_DEVICE; }
        HIPCHK(hipMemcpy(c, d->d_counters + 8, sizeof c, hipMemcpyDeviceToHost));
    }
    u64 nsv = 0; if (nun) { int rc = scan_u32(d, svn, nun, svoff, &nsv, err); if (rc) return rc; }
    if (d->n_cand + nsv > d->cand_cap || d->opt.get("SAGE2OV_TEST_SMALL_BUFFERS")) {
        { int rc = cand_resize(d, d->n_cand + nsv + 1024, d->n_cand, err); if (rc) return rc; }
    }
    lap("final lists + marks");
    if (nun) hipLaunchKernelGGL(k_ra_emit, dim3(gb), dim3(256), 0, d->stream, ids, (u64)nun, offs, deg, ent, rm, svoff, d->cand, (u64)d->n_cand, (u64)d->cand_cap);
    HIPCHK(hipGetLastError());
    d->n_cand += nsv;
    *inserted = ranked ? 2 * present : nh; *removed = c[1]; *done = 1;
    HIPCHK(hipEventRecord(d->ev[1], d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]); d->tm.hits_ms += ms;
    hipEventElapsedTime(&ms, d->ev[5], d->ev[1]); d->tm.marks_ms += ms;
    mem_sample(d);
    return 0;
}

// candidates that touch an unresolved read or one of its neighbours (their adjacency lists are what
// markTransitiveEdge reads, economyGraph.cpp:643-679); candidates owned by unresolved reads are flagged
// as dropped on the device: the replay re-emits the survivors.
int dev_collect_reduce_edges(Device* d, const std::vector<uint32_t>& unresolved, std::vector<EdgeCand>& out, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    out.clear(); const u64 n = d->n_cand; if (n == 0) return 0;
    EdgeCand* buf = nullptr; u64 cap = 1 << 16;
    const u32 nUn = (u32)std::min<size_t>(unresolved.size(), 1u << 30);
    if (nUn && nUn <= FEW_MAX / 3 && !d->opt.get("SAGE2OV_TEST_GENERAL_COLLECT")) {
        // a handful of unresolved reads (see k_red_collect_few): the short list = these reads + their two extension partners, found from their records
        WS(dIds, u32, WS_IDS, 4 * (u64)FEW_MAX); WS(dRec, u64, WS_NEED, 2 * (u64)nUn + 2);
        u32* dNeed = dIds + FEW_MAX;
        HIPCHK(hipMemcpyAsync(dIds, unresolved.data(), nUn * sizeof(u32), hipMemcpyHostToDevice, d->stream));
        hipLaunchKernelGGL(k_gather_records, dim3(grid_for(nUn, 256)), dim3(256), 0, d->stream, dIds, nUn, d->posOf, d->idOf, d->right, d->left, dRec);
        std::vector<u64> rec(2 * (size_t)nUn); HIPCHK(hipMemcpyAsync(rec.data(), dRec, rec.size() * sizeof(u64), hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
        std::vector<u32> need(unresolved.begin(), unresolved.begin() + nUn);
        for (u64 r : rec) { const u32 id = (u32)(r & ((1ull << 40) - 1)); if (id) need.push_back(id); }
        std::sort(need.begin(), need.end()); need.erase(std::unique(need.begin(), need.end()), need.end());
        HIPCHK(hipMemcpyAsync(dNeed, need.data(), need.size() * sizeof(u32), hipMemcpyHostToDevice, d->stream));
        const unsigned blocks = (unsigned)std::min<u64>(grid_for(n, 256), 256ull * 8);
        // (every candidate that touches a read of the short list was emitted by a read of the list or by a partner of one: at most 2 per read and partner)
        cap = 2 * (u64)need.size() * 3 + 64; { WS(nb_, EdgeCand, WS_NEAR, cap); buf = nb_; }
        HIPCHK(hipMemsetAsync(d->d_counters + 5, 0, sizeof(u64), d->stream));
        hipLaunchKernelGGL(k_red_collect_few, dim3(blocks), dim3(256), 0, d->stream, d->cand, (u64)n, dNeed, (u32)need.size(), dIds, nUn, buf, cap, d->d_counters + 5);
        HIPCHK(hipGetLastError());
        u64 cnt = 0; HIPCHK(hipMemcpyAsync(&cnt, d->d_counters + 5, sizeof cnt, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
        if (cnt <= cap) {                                                  // (else: the general form below -- the 0x80 flags are idempotent)
            out.resize(cnt); if (cnt) HIPCHK(hipMemcpy(out.data(), buf, cnt * sizeof(EdgeCand), hipMemcpyDeviceToHost));
            return 0;
        }
    }
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

int dev_meminfo(Device* d, uint64_t* out4, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    size_t fr = 0, to = 0; HIPCHK(hipMemGetInfo(&fr, &to)); mem_sample(d);
    u64 arena = 0; for (auto& b : d->ws) if (!b.epoch) arena += b.cap;
    arena += d->ph.cap;                                                            // (diet mode: the phase block; what was carved out of it is not counted twice)
    if (d->opt.get("SAGE2OV_MEMINFO")) {                                         // diagnostic: the big buffers of the arena, by workspace id (enum order)
        fprintf(stderr, "[meminfo] free %.2f GB of %.2f, lowest %.2f, arena %.2f GB, N %llu, diet %d:", fr / 1e9, to / 1e9, d->memLow / 1e9, arena / 1e9, (unsigned long long)d->N, (int)d->diet);
        for (int x = 0; x < WS_COUNT; x++) if (d->ws[x].cap >= (64u << 20)) fprintf(stderr, " ws%d=%.2f", x, d->ws[x].cap / 1e9);
        fprintf(stderr, "\n");
    }
    out4[0] = fr; out4[1] = to; out4[2] = d->memLow; out4[3] = arena; return 0;
}
int dev_debug_table(Device* d, uint64_t* out5, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    u64* dk = nullptr; HIPCHK(hipMalloc(&dk, 8 * sizeof(u64))); HIPCHK(hipMemsetAsync(dk, 0, 8 * sizeof(u64), d->stream));
    hipLaunchKernelGGL(k_debug_table, dim3(grid_for_capped(d->T, 256)), dim3(256), 0, d->stream, d->slots, (u64)d->T, dk);
    HIPCHK(hipStreamSynchronize(d->stream));
    HIPCHK(hipMemcpy(out5, dk, 5 * sizeof(u64), hipMemcpyDeviceToHost)); hipFree(dk); return 0;
}
int dev_debug_keys(Device* d, uint64_t* out, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (!d->reads) { err = "the id-ordered read store has been released (memory-diet mode)"; return SAGE2OV_ERR_ARG; }
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
    HIPCHK(hipMalloc(&saved, N + 1)); HIPCHK(hipStreamSynchronize(d->stream));
    HIPCHK(hipMemcpyAsync(saved, d->status, N + 1, hipMemcpyDeviceToDevice, d->stream));
    HIPCHK(hipMemsetAsync(d->status, 0, N + 1, d->stream));
    u64 cap = std::max<u64>(1 << 16, N * 128); Hit* dh = nullptr;
    HIPCHK(hipMalloc(&dh, cap * sizeof(Hit)));
    HIPCHK(hipMemsetAsync(d->d_counters + 4, 0, sizeof(u64), d->stream));
    ProbeArgs A = base_args(d); A.lo = 1; A.hi = N + 1; A.hits = dh; A.hits_cap = cap;
    int rc = refresh_status_by_pos(d, err); if (!rc) rc = launch_probe<1>(d, A, err);
    u64 nh = 0;
    if (!rc) { HIPCHK(hipMemcpyAsync(&nh, d->d_counters + 4, sizeof nh, hipMemcpyDeviceToHost, d->stream)); HIPCHK(hipStreamSynchronize(d->stream)); }
    if (!rc && nh > cap) { err = "debug hit buffer too small"; rc = SAGE2OV_ERR_LIMIT; }
    if (!rc) { hits.resize(nh); if (nh) HIPCHK(hipMemcpy(hits.data(), dh, nh * sizeof(Hit), hipMemcpyDeviceToHost)); }
    HIPCHK(hipMemcpyAsync(d->status, saved, N + 1, hipMemcpyDeviceToDevice, d->stream));
    { int rc2 = refresh_status_by_pos(d, err); if (!rc) rc = rc2; }      // (statusP is what the reciprocal pass left again)
    HIPCHK(hipStreamSynchronize(d->stream));
    hipFree(saved); hipFree(dh);
    return rc;
}

int dev_append_edges(Device* d, const EdgeCand* e, uint64_t n, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (n == 0) return 0;
    if (d->n_cand + n > d->cand_cap) {
        { int rc = cand_resize(d, d->n_cand + n + 1024, d->n_cand, err); if (rc) return rc; }
    }
    HIPCHK(hipMemcpyAsync(d->cand + d->n_cand, e, n * sizeof(EdgeCand), hipMemcpyHostToDevice, d->stream)); HIPCHK(hipStreamSynchronize(d->stream));
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

static int scan_u32_async(Device* d, const u32* in, u64 n, u32* out, std::string& err) {
    const u64 nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    WS(partial, u64, WS_PARTIAL, nb + 1);
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, d->stream, in, (u64)n, partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, d->stream, partial, (u64)nb, partial + nb);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, d->stream, in, (u64)n, partial, out);
    HIPCHK(hipGetLastError());
    return 0;
}

int dev_download_edges(Device* d, std::vector<FinalEdge>& out, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    out.resize(d->n_final);
    if (d->n_final) HIPCHK(hipMemcpy(out.data(), d->final_edges, d->n_final * sizeof(FinalEdge), hipMemcpyDeviceToHost));
    return 0;
}
int dev_upload_edges(Device* d, const std::vector<FinalEdge>& in, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    WS(fe, FinalEdge, WS_FINAL, std::max<size_t>(1, in.size()));
    if (!in.empty()) HIPCHK(hipMemcpy(fe, in.data(), in.size() * sizeof(FinalEdge), hipMemcpyHostToDevice));
    d->final_edges = fe; d->n_final = in.size();
    return 0;
}
int dev_convert(Device* d, uint64_t* n_final, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const u64 N = d->N, n = d->n_cand;
    d->final_edges = nullptr; d->n_final = 0;
    HIPCHK(hipEventRecord(d->ev[0], d->stream));
    if (n) {
        if (n >= (1ull << 32) - PT_TILE) { err = "too many edge candidates"; return SAGE2OV_ERR_LIMIT; }
        // Round 3: the candidates arrive in POSITION order of their emitters (their `from` ids are random): a stable LSD radix sort by `from`
        // (kernels_partition.inc, <= 9 bits per pass; the candidate list itself is only read) makes every read's list contiguous without an atomic --
        // the counting form this replaces (two atomics per candidate on per-read counters) took 5.4 ms at BASELINE configs[2], 3.4 when the
        // candidates still came in id order.
        PtBufs B; B.W = 4; B.src0 = (const u32*)d->cand;
        { WS(a, u32, WS_PT_K0, 4 * (n + 4)); B.E[0] = a; } { WS(a, u32, WS_PT_K1, 4 * (n + 4)); B.E[1] = a; }
        const u32 ntiles = (u32)((n + PT_TILE - 1) / PT_TILE);
        WS(cnt, u32, WS_PT_CNT, (u64)PT_NB_MAX * std::max<u32>(ntiles, 1) + 2); WS(base, u32, WS_PT_BASE, (u64)PT_NB_MAX * std::max<u32>(ntiles, 1) + 2);
        WS(keys, u64, WS_KEYS, n); WS(keep, u32, WS_KEEP, n); WS(pos, u32, WS_POS, n);
        int cur = 0; int rc = partition_by_window(d, B, 0, (u32)n, 0, N + 2, false, cnt, base, nullptr, &cur, err); if (rc) return rc;
        const EdgeCand* sorted = (const EdgeCand*)B.E[cur];
        hipLaunchKernelGGL(k_conv_group, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, sorted, (u64)n, keys, keep);
        u64 nf = 0; rc = scan_u32(d, keep, n, pos, &nf, err); if (rc) return rc;
        { WS(fe, FinalEdge, WS_FINAL, std::max<u64>(1, nf)); d->final_edges = fe; }
        hipLaunchKernelGGL(k_conv_emit, dim3(grid_for(n, 256)), dim3(256), 0, d->stream, (u64)n, keys, keep, pos, sorted, d->reads, d->S, d->uniL, d->final_edges);
        HIPCHK(hipGetLastError());
        d->n_final = nf;
        HIPCHK(hipStreamSynchronize(d->stream));
    }
    HIPCHK(hipEventRecord(d->ev[1], d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]); d->tm.convert_ms += ms;
    mem_sample(d);
    if (d->diet) for (int id : {WS_PT_K0, WS_PT_K1, WS_PT_CNT, WS_PT_BASE, WS_KEYS, WS_KEEP, WS_POS, WS_PARTIAL}) ws_free(d, id);
    *n_final = d->n_final;
    return 0;
}

// =============================================================================================
// step 4: simplification of the overlap graph in HBM (main.cpp:139-172).  Input: the final edges of step 3 (device resident, .graph3 order).
// =============================================================================================
namespace {
struct S4Mem {                                   // released on every exit path -- into the device's cache of step-4 blocks (Device::s4cache), or freed in diet mode
    Device* d = nullptr;
    std::vector<std::pair<void*, size_t>> ptrs;
    void* take(size_t bytes) {
        for (auto& c : d->s4cache) if (c.second == bytes) { void* p = c.first; c = d->s4cache.back(); d->s4cache.pop_back(); return p; }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) == hipSuccess) return p;
        (void)hipGetLastError();
        s4cache_release(d);                                                           // (blocks of other sizes are in the way)
        return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr;
    }
    template <class T> T* get(size_t n, std::string& err) { const size_t bytes = std::max<size_t>(n, 1) * sizeof(T); void* p = take(bytes); if (!p) { (void)hipGetLastError(); err = "step 4: device allocation failed"; return nullptr; } ptrs.push_back({p, bytes}); return (T*)p; }
    void give_back(void* p, size_t bytes) { if (d->diet) hipFree(p); else d->s4cache.push_back({p, bytes}); }
    void drop(void* p) { for (auto& q : ptrs) if (q.first == p) { give_back(q.first, q.second); q.first = nullptr; } }
    ~S4Mem() { for (auto& q : ptrs) if (q.first) give_back(q.first, q.second); }
};
}
#define S4GET(var, type, n) type* var = mem.get<type>((n), err); if (!var) return SAGE2OV_ERR_NOMEM;
struct S4Keep { S4Mem mem; S4Graph g; u32 nh = 0; u64 listUsed = 0; };
void dev_simplify_release(Device* d) { if (d->s4keep) { hipSetDevice(d->ordinal); delete (S4Keep*)d->s4keep; d->s4keep = nullptr; } }

int dev_simplify(Device* d, SimplifiedGraph& out, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    const bool timing = d->opt.get("SAGE2OV_TIMING") != nullptr; auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!timing) return; auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[step 4] %-40s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - tp).count()); tp = t; };
    dev_simplify_release(d);
    lap("release of the previous result");
    std::unique_ptr<S4Keep> keep(new S4Keep());
    S4Mem& mem = keep->mem; S4Graph& g = keep->g; mem.d = d;
    const u64 N64 = d->N, np = d->n_final;
    if (2 * np + 2 * (N64 + 1) >= (1ull << 32)) { err = "step 4: too many edges"; return SAGE2OV_ERR_LIMIT; }
    const u32 N = (u32)N64, capH = (u32)(2 * np + 2 * (N64 + 1));
    { S4GET(a, u32, capH) g.from = a; } { S4GET(a, u32, capH) g.to = a; } { S4GET(a, u32, capH) g.len = a; } { S4GET(a, u32, capH) g.cnt = a; } { S4GET(a, u32, capH) g.off = a; }
    { S4GET(a, uint8_t, capH) g.type = a; } { S4GET(a, uint8_t, capH) g.alive = a; }
    { S4GET(a, u32, N + 2) g.deg = a; } { S4GET(a, u32, N + 2) g.adjOff = a; } { S4GET(a, u32, capH) g.adj = a; }
    u64 listCap = std::max<u64>(4 * (N64 + 1), 1 << 16), listUsed = 0;
    { S4GET(a, u64, listCap) g.lists = a; }
    S4GET(cursor, u32, N + 2) S4GET(cont, uint8_t, N + 2) S4GET(forced, uint8_t, N + 2) S4GET(role, uint8_t, N + 2) S4GET(h0, u32, N + 2) S4GET(h1, u32, N + 2)
    S4GET(stA, S4State, 2 * (size_t)(N + 1)) S4GET(stB, S4State, 2 * (size_t)(N + 1)) S4GET(stC, S4State, 2 * (size_t)(N + 1))
    S4GET(spFlag, u32, 2 * (size_t)(N + 1) + 2) S4GET(spPos, u32, 2 * (size_t)(N + 1) + 2) S4GET(spList, u32, 2 * (size_t)(N + 1) + 2)
    S4GET(isNew, u32, N + 2) S4GET(newCnt, u32, N + 2) S4GET(rank, u32, N + 2) S4GET(cntScan, u32, N + 2)
    S4GET(chain, S4Chain, capH) S4GET(jobs, S4Job, 4 * (size_t)(N + 1)) S4GET(jobLen, u32, 4 * (size_t)(N + 1)) S4GET(jobStart, u32, 4 * (size_t)(N + 1))
    S4GET(decA, uint8_t, N + 2) S4GET(decB, uint8_t, N + 2) S4GET(othA, u32, N + 2) S4GET(othB, u32, N + 2) S4GET(remA, u32, capH) S4GET(remB, u32, capH)
    S4GET(dctr, u32, 8)
    u32 tabBits = 10; while ((1ull << tabBits) < 2ull * (N64 + 2)) tabBits++;
    const u32 tabMask = (u32)((1ull << tabBits) - 1);
    S4GET(tab, S4Pair, (size_t)1 << tabBits)
    u32 nh = (u32)(2 * np);
    hipStream_t st = d->stream;
    lap("allocations");
    HIPCHK(hipEventRecord(d->ev[0], st));
    if (np) hipLaunchKernelGGL(k_s4_init, dim3(grid_for(np, 256)), dim3(256), 0, st, d->final_edges, (u64)np, g);
    auto rd = [&](u32* v, int n) -> int { return hipMemcpyAsync(v, dctr, n * sizeof(u32), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess ? 0 : -1; };
    bool adjValid = false;                                                         // the lists only change when a sweep changes the graph
    auto adjacency = [&]() -> int {
        if (adjValid) return 0;
        adjValid = true;
        HIPCHK(hipMemsetAsync(g.deg, 0, (N + 2) * sizeof(u32), st)); HIPCHK(hipMemsetAsync(cursor, 0, (N + 2) * sizeof(u32), st));
        if (nh) hipLaunchKernelGGL(k_s4_degree, dim3(grid_for(nh, 256)), dim3(256), 0, st, g, nh);
        u64 tot = 0; int rc = scan_u32(d, g.deg, (u64)N + 1, g.adjOff, &tot, err); if (rc) return rc;
        if (nh) hipLaunchKernelGGL(k_s4_fill, dim3(grid_for(nh, 256)), dim3(256), 0, st, g, nh, cursor);
        hipLaunchKernelGGL(k_s4_sortadj, dim3(grid_for((u64)N + 1, 256)), dim3(256), 0, st, g, N);
        return 0;
    };
    const dim3 gN(grid_for((u64)N + 1, 256)), gS(grid_for(2 * ((u64)N + 1), 256)), b256(256);
    int jumpRounds = 1; while ((1ull << jumpRounds) < 2ull * (N64 + 2)) jumpRounds++;
    // ---- contractCompositePaths (simplification.cpp:14-53)
    auto contract = [&](u64* merged) -> int {
        int rc = adjacency(); if (rc) return rc;
        HIPCHK(hipMemsetAsync(forced, 0, N + 2, st));
        const S4State* fin = nullptr; bool plainJumping = d->opt.get("SAGE2OV_S4_PLAIN_JUMPING") != nullptr;
        for (int round = 0;; round++) {
            if (round > 64) { err = "step 4: chain promotion does not settle"; return SAGE2OV_ERR_INTERNAL; }
            hipLaunchKernelGGL(k_s4_contractible, gN, b256, 0, st, g, N, forced, cont, h0, h1);
            hipLaunchKernelGGL(k_s4_state_init, gS, b256, 0, st, g, N, cont, h0, h1, stA);
            bool ranked = false;
            if (!plainJumping) {                                                     // O(n): splitters, jumping over the splitter states only, hand-out
                hipLaunchKernelGGL(k_s4_split_flags, gS, b256, 0, st, N, cont, spFlag);
                u64 nS = 0; rc = scan_u32(d, spFlag, 2 * ((u64)N + 1), spPos, &nS, err); if (rc) return rc;
                HIPCHK(hipMemsetAsync(dctr, 0, 8 * sizeof(u32), st));
                if (nS) {
                    hipLaunchKernelGGL(k_s4_split_list, gS, b256, 0, st, N, spFlag, spPos, spList);
                    hipLaunchKernelGGL(k_s4_walk_a, dim3(grid_for(nS, 64)), dim3(64), 0, st, spList, (u32)nS, stA, stB, dctr + 2);
                }
                S4State* a = stB; S4State* b = stC;
                for (int r = 0; nS && r < jumpRounds; r++) {
                    HIPCHK(hipMemsetAsync(dctr, 0, sizeof(u32), st));
                    hipLaunchKernelGGL(k_s4_jump_list, dim3(grid_for(nS, 256)), b256, 0, st, spList, (u32)nS, a, b, dctr); std::swap(a, b);
                    u32 open = 0; if (rd(&open, 1)) { err = "step 4: counter read failed"; return SAGE2OV_ERR_DEVICE; }
                    if (!open) break;
                }
                HIPCHK(hipMemsetAsync(b, 0xFF, 2 * (size_t)(N + 1) * sizeof(S4State), st));
                hipLaunchKernelGGL(k_s4_walk_b, dim3(grid_for(2 * ((u64)N + 1), 64)), dim3(64), 0, st, N, cont, stA, a, b, dctr + 2);
                u32 cz[3] = {0, 0, 0}; if (rd(cz, 3)) { err = "step 4: counter read failed"; return SAGE2OV_ERR_DEVICE; }
                if (cz[2]) plainJumping = true; else { fin = b; ranked = true; }
            }
            if (!ranked) {
                S4State* a = stA; S4State* b = stB;
                for (int r = 0; r < jumpRounds; r++) {                               // stops as soon as every state has reached its chain end (cycles never do)
                    HIPCHK(hipMemsetAsync(dctr, 0, sizeof(u32), st));
                    hipLaunchKernelGGL(k_s4_jump, gS, b256, 0, st, N, cont, a, b, dctr); std::swap(a, b);
                    u32 open = 0; if (rd(&open, 1)) { err = "step 4: counter read failed"; return SAGE2OV_ERR_DEVICE; }
                    if (!open) break;
                }
                fin = a;
            }
            HIPCHK(hipMemsetAsync(tab, 0xFF, sizeof(S4Pair) << tabBits, st)); HIPCHK(hipMemsetAsync(dctr, 0, 8 * sizeof(u32), st));
            hipLaunchKernelGGL(k_s4_chain, gN, b256, 0, st, g, N, cont, fin, stA, role, tab, tabMask, dctr);
            hipLaunchKernelGGL(k_s4_parallel, gN, b256, 0, st, g, N, fin, role, tab, tabMask, dctr);
            u32 pz[2] = {0, 0}; if (rd(pz, 2)) { err = "step 4: counter read failed"; return SAGE2OV_ERR_DEVICE; }
            if (pz[1]) { if (plainJumping) { err = "step 4: chain states left unwritten"; return SAGE2OV_ERR_INTERNAL; } plainJumping = true; round--; continue; }   // a cycle without a splitter
            const u32 promoted = pz[0];
            if (!promoted) break;
            hipLaunchKernelGGL(k_s4_promote, gN, b256, 0, st, N, role, forced);
        }
        hipLaunchKernelGGL(k_s4_creator_counts, gN, b256, 0, st, N, role, fin, isNew, newCnt);
        u64 nNew = 0, nList = 0;
        rc = scan_u32(d, isNew, (u64)N + 1, rank, &nNew, err); if (rc) return rc;
        rc = scan_u32(d, newCnt, (u64)N + 1, cntScan, &nList, err); if (rc) return rc;
        if (!nNew) { *merged = 0; return 0; }
        if ((u64)nh + 2 * nNew > capH) { err = "step 4: half-edge pool exhausted"; return SAGE2OV_ERR_INTERNAL; }
        if (listUsed + 2 * nList >= (1ull << 32)) { err = "step 4: read lists exceed 2^32 entries"; return SAGE2OV_ERR_LIMIT; }
        if (listUsed + 2 * nList > listCap) {                                       // grow the list pool (old segments stay where they are)
            const u64 want = std::max<u64>(2 * listCap, listUsed + 2 * nList);
            u64* nl = mem.get<u64>(want, err); if (!nl) return SAGE2OV_ERR_NOMEM;
            if (listUsed) HIPCHK(hipMemcpyAsync(nl, g.lists, listUsed * sizeof(u64), hipMemcpyDeviceToDevice, st));
            HIPCHK(hipStreamSynchronize(st)); mem.drop(g.lists); g.lists = nl; listCap = want;
        }
        HIPCHK(hipMemsetAsync(dctr, 0, 8 * sizeof(u32), st));
        hipLaunchKernelGGL(k_s4_emit, gN, b256, 0, st, g, N, role, fin, h0, h1, rank, cntScan, nh, (u32)listUsed, chain);
        hipLaunchKernelGGL(k_s4_entries, gN, b256, 0, st, g, N, role, fin, h0, h1, chain, jobs, dctr);
        u32 nj = 0; if (rd(&nj, 1)) { err = "step 4: counter read failed"; return SAGE2OV_ERR_DEVICE; }
        if (nj) {
            hipLaunchKernelGGL(k_s4_job_len, dim3(grid_for(nj, 256)), b256, 0, st, jobs, nj, jobLen);
            u64 total = 0; rc = scan_u32(d, jobLen, nj, jobStart, &total, err); if (rc) return rc;
            if (total) hipLaunchKernelGGL(k_s4_copy, dim3(grid_for(total, 256)), b256, 0, st, jobs, nj, jobStart, (u64)total, g.lists);
        }
        nh += (u32)(2 * nNew); listUsed += 2 * nList; adjValid = false;
        *merged = nNew;                                                            // (non-zero: the caller counts the merged nodes)
        return 0;
    };
    // merged nodes of a sweep = nodes with role 1 or 2 (every one of them is one successful mergeEdges)
    auto count_roles = [&](u64* n) -> int {
        HIPCHK(hipMemsetAsync(dctr, 0, 8 * sizeof(u32), st));
        hipLaunchKernelGGL(k_s4_count, gN, b256, 0, st, N, role, dctr);
        u32 v = 0; if (rd(&v, 1)) { err = "step 4: counter read failed"; return SAGE2OV_ERR_DEVICE; }
        *n = v; return 0;
    };
    auto contract_counted = [&](u64* merged) -> int { u64 m = 0; int rc = contract(&m); if (rc) return rc; if (!m) { *merged = 0; return 0; } return count_roles(merged); };
    // ---- removeDeadEnds (simplification.cpp:58-113)
    auto dead_ends = [&](int threshold, u64* removed) -> int {
        int rc = adjacency(); if (rc) return rc;
        HIPCHK(hipMemsetAsync(decA, 0, N + 2, st)); HIPCHK(hipMemsetAsync(decB, 0, N + 2, st));
        uint8_t* in = decA; uint8_t* o = decB;
        for (int it = 0;; it++) {
            if (it > 100000) { err = "step 4: dead-end sweep does not settle"; return SAGE2OV_ERR_INTERNAL; }
            HIPCHK(hipMemsetAsync(dctr, 0, 8 * sizeof(u32), st));
            hipLaunchKernelGGL(k_s4_dead, gN, b256, 0, st, g, N, threshold, in, o, dctr);
            u32 ch = 0; if (rd(&ch, 1)) { err = "step 4: counter read failed"; return SAGE2OV_ERR_DEVICE; }
            std::swap(in, o);
            if (!ch) break;
        }
        HIPCHK(hipMemsetAsync(dctr, 0, 8 * sizeof(u32), st));
        hipLaunchKernelGGL(k_s4_count, gN, b256, 0, st, N, in, dctr);
        if (nh) hipLaunchKernelGGL(k_s4_dead_apply, dim3(grid_for(nh, 256)), b256, 0, st, g, nh, in);
        u32 v = 0; if (rd(&v, 1)) { err = "step 4: counter read failed"; return SAGE2OV_ERR_DEVICE; }
        *removed = v; if (v) hipLaunchKernelGGL(k_s4_filteradj, gN, b256, 0, st, g, N);     // deletions only: the lists are compacted, not rebuilt
        return 0;
    };
    // ---- removeBubbles (simplification.cpp:118-194)
    auto bubbles = [&](long long closeLength, u64* removed) -> int {
        int rc = adjacency(); if (rc) return rc;
        HIPCHK(hipMemsetAsync(decA, 0, N + 2, st)); HIPCHK(hipMemsetAsync(othA, 0, (N + 2) * sizeof(u32), st)); HIPCHK(hipMemsetAsync(remA, 0xFF, (size_t)capH * sizeof(u32), st));
        uint8_t* din = decA; uint8_t* dout = decB; u32* oin = othA; u32* oout = othB; u32* rin = remA; u32* rout = remB;
        for (int it = 0;; it++) {
            if (it > 100000) { err = "step 4: bubble sweep does not settle"; return SAGE2OV_ERR_INTERNAL; }
            HIPCHK(hipMemsetAsync(dctr, 0, 8 * sizeof(u32), st));
            hipLaunchKernelGGL(k_s4_bubble, gN, b256, 0, st, g, N, closeLength, rin, din, oin, dout, oout, dctr);
            u32 ch = 0; if (rd(&ch, 1)) { err = "step 4: counter read failed"; return SAGE2OV_ERR_DEVICE; }
            if (!ch) break;                                                        // rin is the removal map of exactly these decisions
            HIPCHK(hipMemsetAsync(rout, 0xFF, (size_t)capH * sizeof(u32), st));
            hipLaunchKernelGGL(k_s4_bubble_rem, gN, b256, 0, st, g, N, rin, dout, oout, rout);
            std::swap(din, dout); std::swap(oin, oout); std::swap(rin, rout);
        }
        HIPCHK(hipMemsetAsync(dctr, 0, 8 * sizeof(u32), st));
        hipLaunchKernelGGL(k_s4_count, gN, b256, 0, st, N, din, dctr);
        if (nh) hipLaunchKernelGGL(k_s4_bubble_apply, dim3(grid_for(nh, 256)), b256, 0, st, g, nh, rin);
        u32 v = 0; if (rd(&v, 1)) { err = "step 4: counter read failed"; return SAGE2OV_ERR_DEVICE; }
        *removed = v; if (v) hipLaunchKernelGGL(k_s4_filteradj, gN, b256, 0, st, g, N);
        return 0;
    };
    // ---- main.cpp:150-172
    int threshold = 0; long long closeValue = 10; u64 contracted = 0, removed = 0, iters = 0, x = 0;
    int rc;
    if ((rc = contract_counted(&x))) return rc; contracted += x;
    if ((rc = dead_ends(threshold, &x))) return rc; removed += x;
    if ((rc = bubbles(closeValue, &x))) return rc; removed += x;
    if ((rc = contract_counted(&x))) return rc; contracted += x;
    for (;;) {
        u64 a = 0, b = 0, c = 0;
        if ((rc = dead_ends(threshold, &a))) return rc;
        if ((rc = bubbles(closeValue, &b))) return rc;
        if ((rc = contract_counted(&c))) return rc;
        removed += a + b; contracted += c; iters++;
        if (a + b + c == 0) break;
        if (closeValue < 50) closeValue += 10;
        if (threshold < 3) threshold++;
    }
    HIPCHK(hipEventRecord(d->ev[1], st)); HIPCHK(hipStreamSynchronize(st));
    float ms = 0; hipEventElapsedTime(&ms, d->ev[0], d->ev[1]);
    lap("sweeps (host view)");
    S4GET(dstat, u64, 2)
    HIPCHK(hipMemsetAsync(dstat, 0, 2 * sizeof(u64), st));
    if (nh) hipLaunchKernelGGL(k_s4_stats, dim3(grid_for(nh / 2 + 1, 256)), b256, 0, st, g, nh, dstat);
    u64 hs[2] = {0, 0}; HIPCHK(hipMemcpyAsync(hs, dstat, sizeof hs, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
    out = SimplifiedGraph();
    out.n_half_edges = nh; out.contracted = contracted; out.removed = removed; out.iterations = iters; out.device_ms = ms; out.N = N64; out.pairs_alive = hs[0]; out.reads_on_edges = hs[1];
    // the graph stays in HBM (downloaded only when it is written); the work buffers go
    for (void* p : {(void*)stA, (void*)stB, (void*)stC, (void*)spFlag, (void*)spPos, (void*)spList, (void*)tab, (void*)jobs, (void*)jobLen, (void*)jobStart, (void*)remA, (void*)remB, (void*)chain, (void*)g.adj, (void*)cursor, (void*)h0, (void*)h1,
                    (void*)isNew, (void*)newCnt, (void*)rank, (void*)cntScan, (void*)othA, (void*)othB}) mem.drop(p);
    g.adj = nullptr;
    lap("statistics + release of the work buffers");
    keep->nh = nh; keep->listUsed = listUsed;
    d->s4keep = keep.release();
    return 0;
}
int dev_simplify_download(Device* d, SimplifiedGraph& out, std::string& err) {
    HIPCHK(hipSetDevice(d->ordinal));
    if (!d->s4keep) { err = "step 4: no simplified graph on the device"; return SAGE2OV_ERR_ARG; }
    if (out.downloaded) return 0;
    const S4Keep& k = *(S4Keep*)d->s4keep; const S4Graph& g = k.g; const u32 nh = k.nh;
    out.from.resize(nh); out.to.resize(nh); out.len.resize(nh); out.cnt.resize(nh); out.off.resize(nh); out.type.resize(nh); out.alive.resize(nh); out.lists.resize(k.listUsed);
    if (nh) {
        HIPCHK(hipMemcpy(out.from.data(), g.from, nh * sizeof(u32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out.to.data(), g.to, nh * sizeof(u32), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(out.len.data(), g.len, nh * sizeof(u32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out.cnt.data(), g.cnt, nh * sizeof(u32), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(out.off.data(), g.off, nh * sizeof(u32), hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out.type.data(), g.type, nh, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(out.alive.data(), g.alive, nh, hipMemcpyDeviceToHost));
    }
    if (k.listUsed) HIPCHK(hipMemcpy(out.lists.data(), g.lists, k.listUsed * sizeof(u64), hipMemcpyDeviceToHost));
    out.downloaded = true;
    return 0;
}

}  // namespace s2
