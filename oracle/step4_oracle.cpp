// ============================================================================
// oracle/step4_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of SAGE2's step 4 (overlap-graph simplification, main.cpp:139-172 over
// overlapGraph/simplification.cpp and the list primitives of overlapGraph/overlapGraph.cpp).
// It exists to CHECK a device implementation of that step; nothing under sage2_amd/ links,
// imports or executes it.
//
// Parity status: PINNED.  tests/test_step4_oracle.py compares the file this restatement
// writes byte-for-byte with tests/golden/*.graph4.gz, which were dumped by the reference's
// own classes (oracle/ref_driver.cpp::sage2ref_run_step4, fixtures by oracle/make_golden_step4.py).
//
// Serial by nature: every sweep visits the nodes in ascending id and sees the graph as the
// earlier nodes of the same sweep left it.  The per-node edge lists are "newest first"
// (overlapGraph.cpp:191 inserts at the head) and that order is observable -- in the file
// order (overlapGraph.cpp:356-358 walks each list from its tail) and in the bubble test's
// "first edge a->b" (simplification.cpp:153).  Edges live in one pool with index links.
// ============================================================================
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct OnEdge { uint64_t id; uint32_t orientation, flag, distPrev, distNext; };     // overlapGraph.h:15-23 (40/1/1/11/11 bits)
struct Edge {
    uint64_t from, to; uint32_t type, reducible; uint32_t len;                       // overlapGraph.h:25-40 (lengthOfEdge: 32 bits)
    int next, prev, twin; bool alive;
    std::vector<OnEdge> reads;                                                       // listOfReads[1..count]
};

struct Graph {
    uint64_t N = 0; std::string header[3];
    std::vector<Edge> pool; std::vector<int> head;                                   // head[i]: newest edge of node i, -1 if none

    int new_edge() { pool.emplace_back(); Edge& e = pool.back(); e.next = e.prev = e.twin = -1; e.alive = true; return (int)pool.size() - 1; }
    // overlapGraph.cpp:191-204 insertIntoList: at the head
    void push(int e) {
        const uint64_t u = pool[e].from;
        if (head[u] >= 0) { pool[e].next = head[u]; pool[head[u]].prev = e; }
        head[u] = e;
    }
    // overlapGraph.cpp:271-295 deleteEdge
    void unlink(int e) {
        Edge& x = pool[e];
        if (x.prev < 0) head[x.from] = x.next; else pool[x.prev].next = x.next;
        if (x.next >= 0) pool[x.next].prev = x.prev;
        x.alive = false; x.next = x.prev = -1; x.reads.clear(); x.reads.shrink_to_fit();
    }
    // utils.cpp:212 reverseEdgeType: 0 <-> 3, 1 and 2 stay
    static uint32_t rev_type(uint32_t t) { return t == 0 ? 3 : (t == 3 ? 0 : t); }
    // overlapGraph.cpp:162-185 insertEdge (both directions, forward first)
    int insert_pair(uint64_t u, uint64_t v, uint32_t type, std::vector<OnEdge>&& lf, std::vector<OnEdge>&& lr, uint32_t d1, uint32_t d2) {
        const int f = new_edge(), r = new_edge();
        Edge& F = pool[f]; Edge& R = pool[r];
        F.from = u; F.to = v; F.type = type; F.reducible = 1; F.len = d1; F.reads = std::move(lf); F.twin = r;
        R.from = v; R.to = u; R.type = rev_type(type); R.reducible = 1; R.len = d2; R.reads = std::move(lr); R.twin = f;
        push(f); push(r);
        return f;
    }
    // overlapGraph.cpp:260-267 combinedEdgeType
    static int combined(uint32_t a, uint32_t b) {
        if ((a == 0 && b == 0) || (a == 1 && b == 2)) return 0;
        if ((a == 0 && b == 1) || (a == 1 && b == 3)) return 1;
        if ((a == 2 && b == 0) || (a == 3 && b == 2)) return 2;
        if ((a == 2 && b == 1) || (a == 3 && b == 3)) return 3;
        return -1;
    }
    // overlapGraph.cpp:300-336 getListOfReads: list(e1) + the node between + list(e2)
    std::vector<OnEdge> joined(int e1, int e2) const {
        const Edge& A = pool[e1]; const Edge& B = pool[e2];
        std::vector<OnEdge> out; out.reserve(A.reads.size() + B.reads.size() + 1);
        const uint32_t dPrev = A.reads.empty() ? A.len : A.reads.back().distNext;
        const uint32_t dNext = B.reads.empty() ? B.len : B.reads.front().distPrev;
        out.insert(out.end(), A.reads.begin(), A.reads.end());
        out.push_back(OnEdge{A.to, (A.type == 1 || A.type == 3) ? 1u : 0u, 0u, dPrev & 0x7FFu, dNext & 0x7FFu});
        out.insert(out.end(), B.reads.begin(), B.reads.end());
        return out;
    }
    // overlapGraph.cpp:208-255 mergeEdges with all flows 0 (step 4 runs before the flow): both inputs are consumed
    bool merge(int e1, int e2) {
        const int type = combined(pool[e1].type, pool[e2].type);
        if (e1 == e2 || e1 == pool[e2].twin || pool[e1].reducible == 0 || pool[e2].reducible == 0 || type == -1) return false;
        const int t1 = pool[e1].twin, t2 = pool[e2].twin;
        std::vector<OnEdge> lf = joined(e1, e2), lr = joined(t2, t1);
        const uint64_t u = pool[e1].from, v = pool[e2].to;
        const uint32_t d1 = pool[e1].len + pool[e2].len, d2 = pool[t1].len + pool[t2].len;
        insert_pair(u, v, (uint32_t)type, std::move(lf), std::move(lr), d1, d2);
        unlink(t1); unlink(e1); unlink(t2); unlink(e2);
        return true;
    }

    // simplification.cpp:14-53 contractCompositePaths
    uint64_t contract() {
        uint64_t removed = 0;
        for (uint64_t i = 1; i <= N; i++) {
            const int a = head[i]; if (a < 0) continue;
            const int b = pool[a].next; if (b < 0 || pool[b].next >= 0) continue;            // exactly two edges
            bool adjacent = false;                                                           // :27-34 the two neighbours already share an edge
            for (int u = head[pool[a].to]; u >= 0; u = pool[u].next) if (pool[u].to == pool[b].to) { adjacent = true; break; }
            if (adjacent) continue;
            if (merge(pool[a].twin, b)) removed++;
        }
        return removed;
    }
    // simplification.cpp:58-113 removeDeadEnds
    uint64_t dead_ends(int threshold) {
        uint64_t deleted = 0;
        for (uint64_t i = 1; i <= N; i++) {
            if (head[i] < 0) continue;
            int in = 0, out = 0; bool keep = false;
            for (int v = head[i]; v >= 0; v = pool[v].next) {
                if ((int64_t)pool[v].reads.size() > threshold) { keep = true; break; }       // :77-85 a composite edge with more than `threshold` reads
                if (pool[v].from == pool[v].to) { keep = true; break; }                      // :86-91 a loop
                if (pool[v].type == 0 || pool[v].type == 1) in++; else out++;
            }
            if (!keep && ((in == 0 && out > 0) || (in > 0 && out == 0))) {
                for (int v = head[i], nx; v >= 0; v = nx) { nx = pool[v].next; unlink(pool[v].twin); unlink(v); }
                deleted++;
            }
        }
        return deleted;
    }
    // simplification.cpp:118-194 removeBubbles
    uint64_t bubbles(int64_t closeLength) {
        uint64_t deleted = 0;
        for (uint64_t i = 1; i <= N; i++) {
            if (head[i] < 0) continue;
            int in = 0, out = 0, inE = -1, outE = -1;
            for (int v = head[i]; v >= 0; v = pool[v].next) {
                if (pool[v].type == 0 || pool[v].type == 1) { in++; inE = pool[v].twin; } else { out++; outE = v; }
            }
            if (in != 1 || out != 1) continue;
            const int64_t d1 = (int64_t)pool[inE].len + (int64_t)pool[outE].len;
            const uint64_t a = pool[inE].from, b = pool[outE].to;
            for (int v = head[a]; v >= 0; v = pool[v].next) {
                if (pool[v].to != b) continue;
                const int64_t d2 = pool[v].len;
                const int64_t n1 = 1 + (int64_t)pool[inE].reads.size() + (int64_t)pool[outE].reads.size(), n2 = (int64_t)pool[v].reads.size();
                if (llabs(d1 - d2) < closeLength) {
                    if (n1 < n2 / 2) { unlink(pool[inE].twin); unlink(inE); unlink(pool[outE].twin); unlink(outE); deleted++; }
                    if (n2 < n1 / 2) { unlink(pool[v].twin); unlink(v); deleted++; }
                }
                break;
            }
        }
        return deleted;
    }

    // overlapGraph.cpp:371-442 loadOverlapGraphFromFile (text format of overlapGraph.cpp:12-20)
    bool load(const char* path) {
        FILE* f = fopen(path, "r"); if (!f) return false;
        char line[256];
        for (int x = 0; x < 3; x++) { if (!fgets(line, sizeof line, f)) { fclose(f); return false; } header[x] = line; }
        struct Rec { uint64_t from, to; unsigned type, red; unsigned long len; double flow; unsigned long long cnt; std::vector<OnEdge> l; };
        std::vector<Rec> recs; uint64_t maxid = 0;
        for (;;) {
            Rec r; if (fscanf(f, "%lu %lu %u %u %lu %lf %llu", &r.from, &r.to, &r.type, &r.red, &r.len, &r.flow, &r.cnt) != 7) break;
            for (unsigned long long j = 0; j < r.cnt; j++) { OnEdge o; unsigned long id; if (fscanf(f, "%lu %u %u %u %u", &id, &o.orientation, &o.flag, &o.distPrev, &o.distNext) != 5) { fclose(f); return false; } o.id = id; r.l.push_back(o); }
            if (r.from > maxid) maxid = r.from; if (r.to > maxid) maxid = r.to;
            recs.push_back(std::move(r));
        }
        fclose(f);
        if (recs.size() & 1) return false;
        if (N < maxid) N = maxid;
        head.assign(N + 1, -1); pool.clear(); pool.reserve(recs.size() * 2);
        for (size_t x = 0; x < recs.size(); x += 2) {                                        // :427-438 edge, then its twin
            const int v = new_edge(), u = new_edge();
            Edge& V = pool[v]; Edge& U = pool[u];
            V.from = recs[x].from; V.to = recs[x].to; V.type = recs[x].type; V.reducible = recs[x].red; V.len = (uint32_t)recs[x].len; V.reads = std::move(recs[x].l); V.twin = u;
            U.from = recs[x + 1].from; U.to = recs[x + 1].to; U.type = recs[x + 1].type; U.reducible = recs[x + 1].red; U.len = (uint32_t)recs[x + 1].len; U.reads = std::move(recs[x + 1].l); U.twin = v;
            push(v); push(u);
        }
        return true;
    }
    // overlapGraph.cpp:338-369 saveOverlapGraphInFile + operator<< :12-20
    void put(FILE* f, const Edge& e) const {
        fprintf(f, "%lu\t%lu\t%u\t%u\t%u\t0\t%zu\n", e.from, e.to, e.type, e.reducible, e.len, e.reads.size());
        for (const OnEdge& o : e.reads) fprintf(f, "%lu\t%u\t%u\t%u\t%u\n", o.id, o.orientation, o.flag, o.distPrev, o.distNext);
        fputc('\n', f);
    }
    bool save(const char* path) const {
        FILE* f = fopen(path, "w"); if (!f) return false;
        for (int x = 0; x < 3; x++) fputs(header[x].c_str(), f);
        for (uint64_t i = 1; i <= N; i++) {
            int u = head[i]; if (u < 0) continue;
            while (pool[u].next >= 0) u = pool[u].next;                                       // oldest first
            for (; u >= 0; u = pool[u].prev) if (i <= pool[u].to) { put(f, pool[u]); put(f, pool[pool[u].twin]); }
        }
        fclose(f); return true;
    }
};

}  // namespace

extern "C" {
// main.cpp:150-172: contract, dead ends (0), bubbles (10), contract; then dead ends / bubbles / contract with growing thresholds until nothing changes.
// counters: [0] N, [1] loop iterations, [2] nodes contracted, [3] dead ends + bubbles removed, [4] surviving directed edges
int orc4_run_files(const char* graph3_path, unsigned long long n_unique, const char* graph4_path, unsigned long long* counters) {
    Graph g; g.N = n_unique;
    if (!g.load(graph3_path)) return -1;
    int threshold = 0, closeValue = 10; unsigned long long contracted = 0, removed = 0, iters = 0;
    contracted += g.contract(); removed += g.dead_ends(threshold); removed += g.bubbles(closeValue); contracted += g.contract();
    for (;;) {
        const uint64_t a = g.dead_ends(threshold), b = g.bubbles(closeValue), c = g.contract();
        removed += a + b; contracted += c; iters++;
        if (a + b + c == 0) break;
        if (closeValue < 50) closeValue += 10;
        if (threshold < 3) threshold++;
    }
    if (graph4_path && !g.save(graph4_path)) return -2;
    if (counters) { counters[0] = g.N; counters[1] = iters; counters[2] = contracted; counters[3] = removed; unsigned long long alive = 0; for (const Edge& e : g.pool) alive += e.alive; counters[4] = alive; }
    return 0;
}
}
