// sage2_amd/csrc/sage2ov.hpp -- host-side C++ mirror of the four reference classes main.cpp:44-131 drives for
// steps 1-3, as thin wrappers over the C ABI (include/sage2ov.h).  Same names, argument meaning and call order as
// the reference (inputReader/readLoader.h:44-55, economyGraph/hashTable.h:34-43, economyGraph/economyGraph.h:43-53,
// overlapGraph/overlapGraph.h:54-67); errors become exceptions instead of exit() (utils.cpp:36).
#pragma once
#include <stdexcept>
#include <string>
#include "sage2ov.h"

namespace sage2ov {

struct Error : std::runtime_error { int code; Error(int c, const std::string& m) : std::runtime_error(m), code(c) {} };

class Context {
public:
    explicit Context(uint16_t minOvlp, int device = SAGE2OV_DEVICE_CURRENT, unsigned hostThreads = 0, unsigned rank = 0, unsigned world = 1, unsigned flags = 0) {
        sage2ov_config cfg{}; cfg.min_overlap = minOvlp; cfg.device = device; cfg.rank = rank; cfg.world = world; cfg.host_threads = hostThreads; cfg.flags = flags;
        int rc = sage2ov_ctx_create(&cfg, &c_); if (rc) throw Error(rc, sage2ov_last_error(nullptr));
    }
    ~Context() { sage2ov_ctx_destroy(c_); }
    Context(const Context&) = delete; Context& operator=(const Context&) = delete;
    sage2ov_ctx* get() const { return c_; }
    void check(int rc) const { if (rc) throw Error(rc, sage2ov_last_error(c_)); }
private:
    sage2ov_ctx* c_ = nullptr;
};

// ReadLoader (readLoader.h:44-55)
class ReadLoader {
public:
    uint64_t numberOfUniqueReads = 0, numberOfReads = 0, totalBP = 0, averageReadLength = 0;
    explicit ReadLoader(Context& ctx) : ctx_(ctx) {}
    void loadFromList(const std::string& listPath) { ctx_.check(sage2ov_reads_add_list(ctx_.get(), listPath.c_str())); }
    void readDatasetInBytes(const std::string& mateFile1, const std::string& mateFile2 = "") {
        ctx_.check(sage2ov_reads_add_file(ctx_.get(), mateFile1.c_str(), mateFile2.empty() ? nullptr : mateFile2.c_str()));
    }
    void organizeReads() { ctx_.check(sage2ov_reads_organize(ctx_.get())); refresh(); }
    void saveReadsInFile(const std::string& path) { ctx_.check(sage2ov_reads_save(ctx_.get(), path.c_str())); }
    void loadReadsFromFile(const std::string& path) { ctx_.check(sage2ov_reads_load(ctx_.get(), path.c_str())); refresh(); }
    sage2ov_read_stats stats() const { sage2ov_read_stats s{}; ctx_.check(sage2ov_reads_stats(ctx_.get(), &s)); return s; }
    Context& context() { return ctx_; }
private:
    void refresh() { auto s = stats(); numberOfUniqueReads = s.unique_reads; numberOfReads = s.good_reads; totalBP = s.total_bp; averageReadLength = s.average_read_length; }
    Context& ctx_;
};

// HashTable (hashTable.h:20-43)
class HashTable {
public:
    explicit HashTable(ReadLoader* loader1) : loaderObj(loader1) {}
    void hashPrefixesAndSuffix() { loaderObj->context().check(sage2ov_index_build(loaderObj->context().get())); }
    void saveHashTableInFile(const std::string& path) { loaderObj->context().check(sage2ov_hashtable_save(loaderObj->context().get(), path.c_str())); }   // hashTable.cpp:256
    sage2ov_index_stats stats() const { sage2ov_index_stats s{}; loaderObj->context().check(sage2ov_index_stats_get(loaderObj->context().get(), &s)); return s; }
    ReadLoader* loaderObj;
};

// EconomyGraph (economyGraph.h:32-54)
class EconomyGraph {
public:
    explicit EconomyGraph(HashTable* hash1) : hashObj(hash1) {}
    void buildInitialOverlapGraph() { ctx().check(sage2ov_overlap_initial(ctx().get())); }
    void buildOverlapGraphEconomy() { ctx().check(sage2ov_overlap_reduce(ctx().get())); }
    void sortEconomyGraph() {}   // folded into OverlapGraph::convertGraph (one device pass does both, economyGraph.cpp:896 + overlapGraph.cpp:84)
    sage2ov_overlap_stats stats() const { sage2ov_overlap_stats s{}; ctx().check(sage2ov_overlap_stats_get(ctx().get(), &s)); return s; }
    HashTable* hashObj;
    Context& ctx() const { return hashObj->loaderObj->context(); }
};

// OverlapGraph, steps-1-3 part (overlapGraph.h:54-67: convertGraph, saveOverlapGraphInFile)
class OverlapGraph {
public:
    OverlapGraph(EconomyGraph* economy1, ReadLoader* loader1) : economyObj(economy1), ctx_(&loader1->context()) {}
    explicit OverlapGraph(ReadLoader* loader1) : economyObj(nullptr), ctx_(&loader1->context()) {}          // overlapGraph.cpp:47 (graph from a file)
    void convertGraph() { ctx_->check(sage2ov_overlap_convert(ctx_->get())); }
    void saveOverlapGraphInFile(const std::string& path) { ctx_->check(sage2ov_graph_save(ctx_->get(), path.c_str())); }
    void loadOverlapGraphFromFile(const std::string& path) { ctx_->check(sage2ov_graph_load(ctx_->get(), path.c_str())); }
    // step 4: the loop of main.cpp:150-172 over contractCompositePaths / removeDeadEnds / removeBubbles (simplification.cpp), on the device
    sage2ov_simplify_stats simplify() { ctx_->check(sage2ov_graph_simplify(ctx_->get())); sage2ov_simplify_stats s{}; ctx_->check(sage2ov_simplify_stats_get(ctx_->get(), &s)); return s; }
    void saveSimplifiedGraphInFile(const std::string& path) { ctx_->check(sage2ov_graph4_save(ctx_->get(), path.c_str())); }   // what step 5 loads (main.cpp:196)
    EconomyGraph* economyObj;
private:
    Context* ctx_;
};

}  // namespace sage2ov
