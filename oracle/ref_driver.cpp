// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (our code, not the reference's).
// Calls the REFERENCE's own classes in process, exactly in the order main.cpp:37-131 does
// for steps 1-3, with steady_clock timers around each call (the reference logs whole
// seconds only).  Linked against objects compiled from /root/reference by oracle/Makefile
// into oracle/_ref/.  Used to (a) make golden fixtures and (b) time the CPU baseline
// ("kind": "reference") in bench.py.
#include <chrono>
#include "main.h"   // reference header: defines logStream, genomeSize, averageReadLength (main.h:28-36)

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" int sage2ref_run_steps123(const char* fasta, int k, int threads, const char* out_prefix /* may be NULL */,
                                     double* t /* [6]: load, organize, index, initial, reduce, sort+convert */,
                                     unsigned long long* c /* [3]: unique reads, good reads, avg len */) {
    if (threads > 0) omp_set_num_threads(threads);
    std::string logp = out_prefix ? std::string(out_prefix) + ".log" : std::string("/dev/null");
    if (logStream.is_open()) logStream.close();
    logStream.open(logp.c_str());
    genomeSize = 0; averageReadLength = 0;
    double t0 = now_s();
    ReadLoader* loader = new ReadLoader((uint16_t)k);          // main.cpp:44
    loader->readDatasetInBytes(fasta);                          // main.cpp:48
    double t1 = now_s();
    loader->organizeReads();                                    // main.cpp:49
    double t2 = now_s();
    if (out_prefix) loader->saveReadsInFile(std::string(out_prefix) + ".reads");
    double t2b = now_s();
    HashTable* hash = new HashTable((uint16_t)k, loader);       // main.cpp:76
    hash->hashPrefixesAndSuffix();                              // main.cpp:77
    double t3 = now_s();
    EconomyGraph* eco = new EconomyGraph((uint16_t)k, hash);    // main.cpp:108
    eco->buildInitialOverlapGraph();                            // main.cpp:109
    double t4 = now_s();
    eco->buildOverlapGraphEconomy();                            // main.cpp:111
    delete hash;                                                // main.cpp:112
    double t5 = now_s();
    eco->sortEconomyGraph();                                    // main.cpp:114
    OverlapGraph* graph = new OverlapGraph(eco, loader);        // main.cpp:116
    graph->convertGraph();                                      // main.cpp:117
    delete eco;
    double t6 = now_s();
    if (out_prefix) graph->saveOverlapGraphInFile(std::string(out_prefix) + ".graph3");
    t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2b; t[3] = t4 - t3; t[4] = t5 - t4; t[5] = t6 - t5;
    c[0] = loader->numberOfUniqueReads; c[1] = loader->numberOfReads; c[2] = averageReadLength;
    delete graph; delete loader;
    logStream.close();
    return 0;
}

// Step 4 of the reference (main.cpp:139-172) on files: load <in_prefix>.reads + <in_prefix>.graph3 with the reference's own loaders,
// run its simplification loop exactly as main.cpp does, and dump the in-memory graph with the reference's own writer.  The reference
// itself never writes the post-step-4 graph (its step 5 would read "<prefix>.graph4", main.cpp:196); this is how the fixtures
// under tests/golden/*.graph4.gz were made.  c[0..3]: unique reads, loop iterations, nodes contracted, edges/reads removed.
extern "C" int sage2ref_run_step4(const char* in_prefix, int k, int threads, const char* out_graph4, double* t /* [2]: load, simplify */,
                                  unsigned long long* c /* [4] */) {
    if (threads > 0) omp_set_num_threads(threads);
    if (logStream.is_open()) logStream.close();
    logStream.open((std::string(out_graph4) + ".log").c_str());
    genomeSize = 0; averageReadLength = 0;
    double t0 = now_s();
    ReadLoader* loader = new ReadLoader((uint16_t)k);                              // main.cpp:143
    loader->loadReadsFromFile(std::string(in_prefix) + ".reads");                  // main.cpp:144
    OverlapGraph* graph = new OverlapGraph(loader);                                // main.cpp:146
    graph->loadOverlapGraphFromFile(std::string(in_prefix) + ".graph3");           // main.cpp:147
    double t1 = now_s();
    int threshold = 0, closeValue = 10;                                            // main.cpp:150
    unsigned long long contracted = 0, removed = 0, iters = 0;
    contracted += contractCompositePaths(graph, loader);                           // main.cpp:151-154
    removed += removeDeadEnds(graph, loader, threshold);
    removed += removeBubbles(graph, loader, closeValue);
    contracted += contractCompositePaths(graph, loader);
    for (;;) {                                                                     // main.cpp:158-172
        uint64_t a = removeDeadEnds(graph, loader, threshold), b = removeBubbles(graph, loader, closeValue), cc = contractCompositePaths(graph, loader);
        removed += a + b; contracted += cc; iters++;
        if (a + b + cc == 0) break;
        if (closeValue < 50) closeValue += 10;
        if (threshold < 3) threshold++;
    }
    double t2 = now_s();
    graph->saveOverlapGraphInFile(out_graph4);
    t[0] = t1 - t0; t[1] = t2 - t1;
    c[0] = loader->numberOfUniqueReads; c[1] = iters; c[2] = contracted; c[3] = removed;
    delete graph; delete loader;
    logStream.close();
    return 0;
}

// The in-memory hand-over of INTEGRATION.md section 2: the reference's OWN graph object is filled from a canonical edge list (what
// sage2ov_edges_export returns) through its own insertEdgeInGraph (overlapGraph.cpp:120), in list order -- the order convertGraph would have
// produced (overlapGraph.cpp:93-111) -- and then runs ITS step 4 (main.cpp:150-172) and ITS writer.  If the hand-over is right, the dump is the
// graph4 fixture (made from the reference's own files).  edges: n x {from, to, type, length} u64.
extern "C" int sage2ref_step4_from_edges(const char* reads_path, int k, int threads, const unsigned long long* edges, unsigned long long n,
                                         unsigned long long good_reads, unsigned long long avg_len, const char* out_graph4, unsigned long long* c /* [4] */) {
    if (threads > 0) omp_set_num_threads(threads);
    if (logStream.is_open()) logStream.close();
    logStream.open((std::string(out_graph4) + ".log").c_str());
    genomeSize = 0; averageReadLength = avg_len;                                   // (what loadOverlapGraphFromFile takes from the graph3 header, overlapGraph.cpp:382-385)
    ReadLoader* loader = new ReadLoader((uint16_t)k);
    loader->loadReadsFromFile(reads_path);                                         // main.cpp:144
    loader->numberOfReads = good_reads;
    OverlapGraph* graph = new OverlapGraph(loader);                                // main.cpp:146
    for (unsigned long long x = 0; x < n; x++)
        graph->insertEdgeInGraph(edges[4 * x], edges[4 * x + 1], (uint32_t)edges[4 * x + 3], (uint8_t)edges[4 * x + 2]);      // overlapGraph.cpp:101-108
    int threshold = 0, closeValue = 10;
    unsigned long long contracted = 0, removed = 0, iters = 0;
    contracted += contractCompositePaths(graph, loader);
    removed += removeDeadEnds(graph, loader, threshold);
    removed += removeBubbles(graph, loader, closeValue);
    contracted += contractCompositePaths(graph, loader);
    for (;;) {
        uint64_t a = removeDeadEnds(graph, loader, threshold), b = removeBubbles(graph, loader, closeValue), cc = contractCompositePaths(graph, loader);
        removed += a + b; contracted += cc; iters++;
        if (a + b + cc == 0) break;
        if (closeValue < 50) closeValue += 10;
        if (threshold < 3) threshold++;
    }
    graph->saveOverlapGraphInFile(out_graph4);
    c[0] = loader->numberOfUniqueReads; c[1] = iters; c[2] = contracted; c[3] = removed;
    delete graph; delete loader;
    logStream.close();
    return 0;
}
