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
