// sage2_amd/csrc/sage2ov_multi.cpp -- steps 2-3 on G GPUs of one node, driven from C++ (the `--gpus G` mode of the sage2ov CLI).
//
// One host thread per GPU, each with its own context (rank r of G, include/sage2ov.h: sage2ov_config.rank / world); the read set and the index are
// replicated, the probe pass is range-partitioned over positions of the locality order, and four one-shot exchanges carry the results:
//   1. all-gather of the 16-byte per-read records (the reciprocal test reads the NEIGHBOUR's record, economyGraph.cpp:460),
//   2. MAX all-reduce (= OR) of the two containment byte planes (economyGraph.cpp:735: a mark lands on a read of any rank),
//   3. all-gather of the per-rank edge buckets (counts first, buckets padded to the largest),
//   4. after the reduce phase, whose marks (economyGraph.cpp:643-707) are sharded over the ranks: all-gather of the per-rank survivor buckets and
//      the sum of the removal counters.
// The collectives are RCCL (ncclAllGather / ncclAllReduce over xGMI; one communicator per GPU from ncclCommInitAll, every thread issuing its own
// rank's call on its context's stream).  The same sequence with torch.distributed is sage2_amd/dist.py::run_steps23_sharded.
// A second transport exists for boxes with fewer GPUs than ranks (tests, rehearsals): all ranks share ONE device and the "collectives" are device
// copies between the ranks' buffers behind a thread barrier -- it exercises everything but RCCL itself, which in turn runs with one rank there.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "sage2ov.h"
#include "sage2ov_multi.h"

namespace sage2ov_multi {
namespace {

struct Barrier {                                               // (C++17: no std::barrier); a rank that failed releases everybody: wait() then returns false
    std::mutex m; std::condition_variable cv; int n, waiting = 0; unsigned gen = 0; bool broken = false;
    explicit Barrier(int n_) : n(n_) {}
    bool wait() { std::unique_lock<std::mutex> l(m); if (broken) return false; const unsigned g = gen; if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); } else cv.wait(l, [&] { return gen != g || broken; }); return !broken; }
    void release_all() { { std::lock_guard<std::mutex> l(m); broken = true; cv.notify_all(); } failed.store(true, std::memory_order_release); }
    std::atomic<bool> failed{false};                           // (read without the lock by ranks that poll a collective: see Rank::finish)
};
struct DevBuf {                                                // device memory of one rank, freed on scope exit
    void* p = nullptr; size_t cap = 0;
    int need(size_t bytes) { if (bytes <= cap) return 0; if (p) hipFree(p); p = nullptr; cap = 0; if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return -1; cap = bytes; return 0; }
    ~DevBuf() { if (p) hipFree(p); }
};

struct Shared {                                                // what the rank threads see of each other
    int G; bool shareGpu; Barrier bar;
    std::vector<ncclComm_t> comms;
    std::vector<const void*> sendPtr; std::vector<uint64_t> u64Slot;          // shared-GPU transport: published send buffers / scalars
    std::vector<int> rc; std::vector<std::string> err;
    Shared(int g, bool sg) : G(g), shareGpu(sg), bar(g), comms(g, nullptr), sendPtr(g, nullptr), u64Slot(g, 0), rc(g, 0), err(g) {}
};

struct Rank {
    Shared& S; int r; sage2ov_ctx* ctx; hipStream_t st;
    Rank(Shared& s, int r_, sage2ov_ctx* c) : S(s), r(r_), ctx(c), st((hipStream_t)sage2ov_stream(c)) {}
    int fail(const char* what) { S.err[r] = what; S.bar.release_all(); return S.rc[r] = SAGE2OV_ERR_DEVICE; }
    int other_failed() { if (!S.rc[r]) { S.rc[r] = SAGE2OV_ERR_INTERNAL; S.err[r] = "another rank failed"; } return S.rc[r]; }
    // The ranks are threads of one process, so a rank that fails anywhere (a library error on its GPU, a failed allocation) can tell the others: they meet at the
    // thread barrier BEFORE every RCCL collective -- nobody enters one that a failed rank will never join -- and a rank that is already inside one polls its stream
    // instead of blocking in hipStreamSynchronize, aborts its communicator when the flag goes up and returns (`sage2ov --gpus G` used to hang).
    int finish() {
        for (;;) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return 0;
            if (q != hipErrorNotReady) return fail("stream query");
            if (S.bar.failed.load(std::memory_order_acquire)) { if (S.comms[r]) { ncclCommAbort(S.comms[r]); S.comms[r] = nullptr; } return other_failed(); }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
#define BAR() do { if (!S.bar.wait()) return other_failed(); } while (0)
    // every rank contributes `bytes` at send; recv gets G x bytes in rank order
    int allgather(const void* send, void* recv, size_t bytes) {
        if (!S.shareGpu) { BAR(); if (ncclAllGather(send, recv, bytes, ncclUint8, S.comms[r], st) != ncclSuccess) return fail("ncclAllGather"); return finish(); }
        S.sendPtr[r] = send; BAR();
        for (int q = 0; q < S.G; q++) if (bytes && hipMemcpyAsync((char*)recv + (size_t)q * bytes, S.sendPtr[q], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("device copy");
        if (hipStreamSynchronize(st) != hipSuccess) return fail("stream sync");       // (a device-to-device hipMemcpy may return before it has run, and nothing orders the null stream with the library's stream)
        BAR(); return 0;
    }
    int allreduce_max_bytes(void* buf, size_t bytes) {          // in place
        if (!S.shareGpu) { BAR(); if (ncclAllReduce(buf, buf, bytes, ncclUint8, ncclMax, S.comms[r], st) != ncclSuccess) return fail("ncclAllReduce"); return finish(); }
        std::vector<unsigned char> mine(bytes), other(bytes);     // (rehearsal transport: through the host)
        S.sendPtr[r] = buf; BAR();
        if (hipMemcpy(mine.data(), buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("copy");
        for (int q = 0; q < S.G; q++) if (q != r) { if (hipMemcpy(other.data(), S.sendPtr[q], bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("copy"); for (size_t x = 0; x < bytes; x++) if (other[x] > mine[x]) mine[x] = other[x]; }
        BAR();                                                    // (everybody has read everybody's original planes)
        if (hipMemcpy(buf, mine.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return fail("copy");
        BAR(); return 0;
    }
    // counts / counters: G x u64 on the host, through a tiny device buffer with RCCL (keeps every exchange on the same transport)
    int allgather_u64(uint64_t v, std::vector<uint64_t>& all, DevBuf& scratch) {
        all.assign(S.G, 0);
        if (S.shareGpu) { S.u64Slot[r] = v; BAR(); all = S.u64Slot; BAR(); return 0; }
        if (scratch.need((size_t)(S.G + 1) * 8)) return fail("hipMalloc");
        if (hipMemcpyAsync(scratch.p, &v, 8, hipMemcpyHostToDevice, st) != hipSuccess) return fail("copy");
        BAR();
        if (ncclAllGather(scratch.p, (char*)scratch.p + 8, 8, ncclUint8, S.comms[r], st) != ncclSuccess) return fail("ncclAllGather");
        if (hipMemcpyAsync(all.data(), (char*)scratch.p + 8, (size_t)S.G * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return fail("copy");
        return finish();
    }
    // ragged buckets of 16-byte records: counts, then buckets padded to the largest; `out` gets the concatenation in rank order
    int allgather_buckets(const void* bucket, uint64_t n, DevBuf& padded, DevBuf& gathered, DevBuf& out, uint64_t* total, DevBuf& scratch) {
        std::vector<uint64_t> cnt; if (int rc = allgather_u64(n, cnt, scratch)) return rc;
        uint64_t mx = 1, tot = 0; for (uint64_t c : cnt) { mx = c > mx ? c : mx; tot += c; }
        if (padded.need(mx * 16) || gathered.need((size_t)S.G * mx * 16) || out.need((tot ? tot : 1) * 16)) return fail("hipMalloc");
        if (n && hipMemcpyAsync(padded.p, bucket, n * 16, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("copy");
        if (hipStreamSynchronize(st) != hipSuccess) return fail("stream sync");
        if (int rc = allgather(padded.p, gathered.p, mx * 16)) return rc;
        uint64_t at = 0;
        for (int q = 0; q < S.G; q++) { if (cnt[q] && hipMemcpyAsync((char*)out.p + at * 16, (char*)gathered.p + (size_t)q * mx * 16, cnt[q] * 16, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("copy"); at += cnt[q]; }
        if (hipStreamSynchronize(st) != hipSuccess) return fail("stream sync");
        *total = tot; return 0;
    }
#define S2(call) do { int rc_ = (call); if (rc_) { S.rc[r] = rc_; S.err[r] = sage2ov_last_error(ctx); S.bar.release_all(); return rc_; } } while (0)
    // the timed region of SURVEY 8(d) on this rank; every rank ends with the complete canonical edge list
    int steps23() {
        S2(sage2ov_index_build(ctx));
        S2(sage2ov_overlap_probe_shard(ctx));
        sage2ov_read_stats rs; S2(sage2ov_reads_stats(ctx, &rs));
        const uint64_t N = rs.unique_reads;
        uint64_t maxShard = 0; for (int q = 0; q < S.G; q++) { const uint64_t lo = 1 + N * q / S.G, hi = 1 + N * (q + 1) / S.G; if (hi - lo > maxShard) maxShard = hi - lo; }   // = sage2ov_shard_range of rank q
        DevBuf send, recv, planes, bucket, padded, gathered, all, scratch;
        uint64_t RB = 0; S2(sage2ov_shard_record_bytes(ctx, &RB));                           // bytes of a read's record on the wire
        // 1. records
        if (send.need((maxShard ? maxShard : 1) * RB) || recv.need((size_t)S.G * (maxShard ? maxShard : 1) * RB)) return fail("hipMalloc");
        // (on the context's own stream: a hipMemset on the null stream is not ordered with the library's non-blocking stream -- it could land on top of the exported records)
        if (hipMemsetAsync(send.p, 0, (maxShard ? maxShard : 1) * RB, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("memset");
        S2(sage2ov_shard_export_records(ctx, send.p, maxShard));
        // 2. containment planes (exported before other ranks' records arrive; the import ORs the flags a record carries)
        uint64_t fb = 0; S2(sage2ov_shard_flags_bytes(ctx, &fb));
        if (planes.need(fb)) return fail("hipMalloc");
        S2(sage2ov_shard_export_flags(ctx, planes.p));
        if (int rc = allgather(send.p, recv.p, (maxShard ? maxShard : 1) * RB)) return rc;
        for (int q = 0; q < S.G; q++) { const uint64_t lo = 1 + N * q / S.G, hi = 1 + N * (q + 1) / S.G; if (hi > lo) S2(sage2ov_shard_import_records(ctx, (char*)recv.p + (size_t)q * (maxShard ? maxShard : 1) * RB, lo, hi - lo)); }
        if (int rc = allreduce_max_bytes(planes.p, fb)) return rc;
        S2(sage2ov_shard_import_flags(ctx, planes.p));
        S2(sage2ov_overlap_reciprocal(ctx));
        // 3. edge buckets
        uint64_t ne = 0, total = 0; S2(sage2ov_shard_edges_count(ctx, &ne));
        if (bucket.need((ne ? ne : 1) * 16)) return fail("hipMalloc");
        S2(sage2ov_shard_edges_export(ctx, bucket.p, ne ? ne : 1));
        if (int rc = allgather_buckets(bucket.p, ne, padded, gathered, all, &total, scratch)) return rc;
        S2(sage2ov_shard_edges_set(ctx, total ? all.p : nullptr, total));
        // 4. reduce phase with sharded marks, survivor buckets, removal counters
        S2(sage2ov_overlap_reduce(ctx));
        uint64_t ns = 0, rem = 0, stotal = 0; S2(sage2ov_shard_survivors_count(ctx, &ns, &rem));
        if (bucket.need((ns ? ns : 1) * 16)) return fail("hipMalloc");
        S2(sage2ov_shard_survivors_export(ctx, bucket.p, ns ? ns : 1));
        if (int rc = allgather_buckets(bucket.p, ns, padded, gathered, all, &stotal, scratch)) return rc;
        std::vector<uint64_t> rems; if (int rc = allgather_u64(rem, rems, scratch)) return rc;
        uint64_t remTotal = 0; for (uint64_t x : rems) remTotal += x;
        S2(sage2ov_shard_survivors_set(ctx, stotal ? all.p : nullptr, stotal, remTotal));
        S2(sage2ov_overlap_convert(ctx));
        return 0;
    }
#undef S2
#undef BAR
};

}  // namespace

int run_steps23(const std::vector<sage2ov_ctx*>& ctx, const std::vector<int>& devices, bool share_gpu, std::string& err, int fail_rank) {
    const int G = (int)ctx.size();
    if (G < 1 || (int)devices.size() != G) { err = "run_steps23: one device per context"; return SAGE2OV_ERR_ARG; }
    Shared S(G, share_gpu);
    if (!share_gpu) {
        if (ncclCommInitAll(S.comms.data(), G, devices.data()) != ncclSuccess) { err = "ncclCommInitAll failed (RCCL needs one distinct GPU per rank)"; return SAGE2OV_ERR_DEVICE; }
    }
    std::vector<std::thread> th;
    for (int r = 0; r < G; r++) th.emplace_back([&, r] {
        if (hipSetDevice(devices[r]) != hipSuccess) { S.rc[r] = SAGE2OV_ERR_DEVICE; S.err[r] = "hipSetDevice"; S.bar.release_all(); return; }      // (the others must not wait for this rank)
        Rank me(S, r, ctx[r]);
        if (fail_rank == r) { me.fail("failure injected for this rank (test hook)"); return; }
        me.steps23(); });
    for (auto& t : th) t.join();
    if (!share_gpu) for (auto& c : S.comms) if (c) ncclCommDestroy(c);
    // (the rank that failed first is the one to report, not a rank that merely saw the flag)
    for (int r = 0; r < G; r++) if (S.rc[r] && S.err[r] != "another rank failed") { err = "rank " + std::to_string(r) + ": " + S.err[r]; return S.rc[r]; }
    for (int r = 0; r < G; r++) if (S.rc[r]) { err = "rank " + std::to_string(r) + ": " + S.err[r]; return S.rc[r]; }
    return SAGE2OV_OK;
}

}  // namespace sage2ov_multi
