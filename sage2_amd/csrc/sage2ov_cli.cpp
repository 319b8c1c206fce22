// sage2_amd/csrc/sage2ov_cli.cpp -- `sage2ov`: drop-in producer of <outdir>/<prefix>.reads and <prefix>.graph3.
// Mimics the SAGE2 command line for steps 1-3 (main.cpp:384-521): -f | -l, -k, -o, -p, -i, -m, -M, -s, -d, -h.
// Continue with the unchanged reference:  SAGE2 -f reads.fa -k K -o out -p P -i P -m 4      (main.cpp:141-148)
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <memory>
#include <vector>
#include "sage2ov.hpp"
#include "sage2ov_multi.h"

using namespace std;
using namespace sage2ov;

static void printUsage() {
    cout << "USAGE:\n\tsage2ov [options] -f <inputFile> -k <minOverlap>   (steps 1-3 of SAGE2 on an MI355X)\n"
            "\tsage2ov [options] -l <listInput> -k <minOverlap>\n\n";
}
static void printListOfArgs() {
    cout << "\t-f|--fileInput <string>\tinterleaved FASTA/FASTQ(.gz)\n\t-l|--listInput <string>\tlist file (f1=/f2=/f=)\n"
            "\t-k|--minOverlap <int>\tminimum overlap (required)\n\t-o|--outputDir <string>\n\t-p|--prefix <string>\t[untitled]\n"
            "\t-i|--inputPrefix <string>\t[prefix]\n\t-m|--minStep <int>\t[1]\n\t-M|--maxStep <int>\t[3] (4 = simplified graph P.graph4; steps 5-7: run SAGE2 -m 5 on the files written here)\n"
            "\t-s|--saveAll\n\t-d|--debug\n\t-g|--gpu <int>\tHIP device ordinal [0]\n"
            "\t-G|--gpus <int>\tsteps 2-3 on this many GPUs of the node (devices gpu .. gpu+G-1): reads and index replicated, probe pass\n"
            "\t\t\trange-partitioned, records / flags / edge and survivor buckets exchanged with RCCL over xGMI [1]\n\t-h|--help\n\n";
}
static string trimBack(string s, const string& pat) { size_t e = s.find_last_not_of(pat); return e == string::npos ? "" : s.substr(0, e + 1); }
static double now() { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char* argv[]) {
    const double tMain = now(); const bool timing = getenv("SAGE2OV_TIMING") != nullptr;
    int minStep = 1, maxStep = 3, gpu = 0, gpus = 1, failRank = -1; bool shareGpu = false, forceMulti = false; unsigned minOverlap = 0; bool saveAll = false, debugging = false, fFlag = false, lFlag = false;
    string fileInput, listInput, outputDir, prefixName = "untitled", inputPrefix;
    static struct option opts[] = {{"help", no_argument, 0, 'h'}, {"fileInput", required_argument, 0, 'f'}, {"minOverlap", required_argument, 0, 'k'},
        {"listInput", required_argument, 0, 'l'}, {"outputDir", required_argument, 0, 'o'}, {"prefix", required_argument, 0, 'p'},
        {"inputPrefix", required_argument, 0, 'i'}, {"minStep", required_argument, 0, 'm'}, {"maxStep", required_argument, 0, 'M'},
        {"saveAll", no_argument, 0, 's'}, {"debug", no_argument, 0, 'd'}, {"gpu", required_argument, 0, 'g'}, {"gpus", required_argument, 0, 'G'},
        {"share-gpu", no_argument, 0, 1001},      // rehearsal: all ranks on device `gpu`, exchanges by device copies instead of RCCL (a box with fewer GPUs than ranks)
        {"fail-rank", required_argument, 0, 1003},  // tests: this rank of a multi-GPU run fails before its first step (the run must end non-zero, not hang)
        {"force-multi", no_argument, 0, 1002},    // the multi-GPU code path (RCCL communicator, the four exchanges) even with one GPU
        {0, 0, 0, 0}};
    int c, oi = 0;
    while ((c = getopt_long(argc, argv, "hf:l:k:o:p:i:m:M:sdg:G:", opts, &oi)) != -1) {
        switch (c) {
            case 'h': cout << "\n"; printUsage(); printListOfArgs(); exit(0);
            case 'f': fileInput = optarg; fFlag = true; break;
            case 'l': listInput = optarg; lFlag = true; break;
            case 'k': minOverlap = atoi(optarg); break;
            case 'o': outputDir = optarg; if (outputDir != "") outputDir = trimBack(outputDir, "/") + "/"; break;   // main.cpp:438-442
            case 'p': prefixName = optarg; break;
            case 'i': inputPrefix = optarg; break;
            case 'm': minStep = atoi(optarg); if (minStep < 1) minStep = 1; break;
            case 'M': maxStep = atoi(optarg); if (maxStep > 7) maxStep = 7; break;
            case 's': saveAll = true; break;
            case 'd': debugging = true; break;
            case 'g': gpu = atoi(optarg); break;
            case 'G': gpus = atoi(optarg); if (gpus < 1) gpus = 1; break;
            case 1001: shareGpu = true; break;
            case 1002: forceMulti = true; break;
            case 1003: failRank = atoi(optarg); break;
            case '?': cout << "\n"; exit(0);
            default: cout << "[ERROR] Wrong command line arguments!\n\n"; exit(0);
        }
    }
    if (fFlag && lFlag) { cout << "[ERROR] Options -f|--fileInput and -l|--listInput are mutually exclusive!\n\n"; exit(0); }
    bool allSet = true;                                                                     // main.cpp:498-521
    if (fileInput == "" && listInput == "" && minStep <= 1) { cout << "[ERROR] One of the options -f|--fileInput or -l|--listInput is required.\n"; allSet = false; }
    if (minOverlap == 0) { cout << "[ERROR] Option -k|--minOverlap is required.\n"; allSet = false; }
    if (maxStep < minStep) { cout << "[ERROR] maxStep should not be smaller than minStep!\n\n"; allSet = false; }
    if (inputPrefix == "") inputPrefix = prefixName;
    if (!allSet) { cout << "\n"; printUsage(); cout << "(For more information run sage2ov -h)\n\n"; exit(0); }
    if (minStep > 4) { cout << "[ERROR] sage2ov implements steps 1-4; continue with: SAGE2 -m " << minStep << " -i " << inputPrefix << " ...\n"; exit(0); }
    (void)debugging;
    if (outputDir != "") { string cmd = "mkdir -p " + outputDir; if (system(cmd.c_str())) {} }     // utils.cpp:45
    ofstream logStream((outputDir + prefixName + ".log").c_str());
    logStream << "***********************************************************************************************************\n"
              << "\tEXECUTING PROGRAM: sage2ov (MI355X-native SAGE2 steps 1-4), " << sage2ov_version() << "\n"
              << (listInput != "" ? "\t  INPUT LIST PATH: " + listInput : "\t  INPUT FILE PATH: " + fileInput) << "\n"
              << "\t OUTPUT DIRECTORY: " << outputDir << "\n\t    OUTPUT PREFIX: " << prefixName << "\n\t  MINIMUM OVERLAP: " << minOverlap << "\n"
              << "\t       START STEP: " << minStep << "\n\t         END STEP: " << maxStep << "\n\t   SAVE ALL FILES: " << (saveAll ? "TRUE" : "FALSE") << "\n"
              << "***********************************************************************************************************\n\n";
    const int lastStep = maxStep > 4 ? 4 : maxStep;
    try {
        const bool multi = (gpus > 1 || forceMulti) && lastStep >= 3 && minStep <= 3;
        // (the device opens in the background while step 1 reads its files)
        Context ctx((uint16_t)minOverlap, lastStep == 1 ? SAGE2OV_DEVICE_NONE : gpu, 0, 0, multi ? (unsigned)gpus : 1u, minStep <= 1 ? SAGE2OV_FLAG_ASYNC_DEVICE : 0u);
        ReadLoader loaderObj(ctx);
        double t0 = now();
        if (timing) fprintf(stderr, "[cli] start-up (context, device) %8.1f ms\n", 1e3 * (t0 - tMain));
        if (minStep <= 1) {                                                                  // main.cpp:37-61
            logStream << "STEP 1: organizing reads\n";
            if (listInput != "") loaderObj.loadFromList(listInput); else loaderObj.readDatasetInBytes(fileInput);
            loaderObj.organizeReads();
            auto s = loaderObj.stats();
            logStream << "\tTotal reads: " << s.total_reads << "\n\tGood reads: " << s.good_reads << "\n\tNumber of unique reads: " << s.unique_reads
                      << "\n\tAverage read length: " << s.average_read_length << "\n\tStep 1 in " << now() - t0 << " sec.\n";
            if (lastStep == 1 || saveAll) { double tw = now(); loaderObj.saveReadsInFile(outputDir + prefixName + ".reads"); logStream << "\t" << prefixName << ".reads written in " << now() - tw << " sec.\n"; }
        } else {
            loaderObj.loadReadsFromFile(outputDir + inputPrefix + ".reads");                 // main.cpp:70-74 / :101-104
            logStream << "\tNumber of unique reads: " << loaderObj.numberOfUniqueReads << " (loaded)\n";
        }
        auto step4 = [&](OverlapGraph& graphObj) {                                            // main.cpp:134-181
            double t4 = now();
            auto ss = graphObj.simplify();
            logStream << "STEP 4: simplify overlap graph\n\t   Nodes removed: " << ss.nodes_contracted << "\n\tDead ends and bubbles removed: " << ss.removed
                      << "\n\tLoop iterations: " << ss.loop_iterations << "\n\tEdges left: " << ss.edges << " carrying " << ss.reads_on_edges << " reads\n\tStep 4 in " << now() - t4
                      << " sec (device " << ss.device_ms / 1000.0 << ").\n";
            { double tw = now(); graphObj.saveSimplifiedGraphInFile(outputDir + prefixName + ".graph4"); logStream << "\t" << prefixName << ".graph4 written in " << now() - tw << " sec.\n"; }   // the file step 5 loads (main.cpp:196)
        };
        if (minStep == 4) {                                                                   // main.cpp:141-148
            OverlapGraph graphObj(&loaderObj);
            graphObj.loadOverlapGraphFromFile(outputDir + inputPrefix + ".graph3");
            step4(graphObj);
        } else if (multi) {
            // steps 2-3 on `gpus` GPUs (sage2ov_multi.cpp): rank 0 = this context; the others import the organised read store
            t0 = now();
            vector<unique_ptr<Context>> others; vector<sage2ov_ctx*> all{ctx.get()}; vector<int> devs{gpu};
            {
                auto s = loaderObj.stats();
                vector<uint64_t> words((s.unique_reads + 1) * (uint64_t)s.words_per_read); vector<uint16_t> freq(s.unique_reads + 1);
                ctx.check(sage2ov_reads_export_words(ctx.get(), words.data(), words.size(), freq.data()));
                for (int r = 1; r < gpus; r++) {
                    const int dev = shareGpu ? gpu : gpu + r;
                    others.emplace_back(new Context((uint16_t)minOverlap, dev, 0, (unsigned)r, (unsigned)gpus));
                    others.back()->check(sage2ov_reads_import_words(others.back()->get(), words.data(), s.unique_reads, s.words_per_read, s.max_read_length, freq.data(), s.good_reads, s.total_bp));
                    all.push_back(others.back()->get()); devs.push_back(dev);
                }
            }
            string merr; int mrc = sage2ov_multi::run_steps23(all, devs, shareGpu, merr, failRank);
            if (mrc) throw Error(mrc, merr);
            sage2ov_index_stats is{}; ctx.check(sage2ov_index_stats_get(ctx.get(), &is)); sage2ov_overlap_stats os{}; ctx.check(sage2ov_overlap_stats_get(ctx.get(), &os));
            logStream << "STEPS 2-3 on " << gpus << (shareGpu ? " ranks sharing GPU " : " GPUs starting at ") << gpu << (shareGpu ? " (rehearsal transport)" : " (RCCL)") << "\n\t         Hash string length: " << is.hash_string_length
                      << "\n\t            Hash table size: " << is.slots << "\n\t Number of hash elements over threshold: " << is.long_buckets
                      << "\n     Total contained by extension: " << os.contained_extension << "\n          Total contained by size: " << os.contained_size
                      << "\n            Total left to explore: " << os.left_to_explore << "\n              Verified overlaps: " << os.verified_overlaps
                      << "\n     Total edges inserted: " << os.edges_inserted << "\n  Transitive edge removed: " << os.transitive_removed << "\n     Edges in the graph: " << os.edges
                      << "\n\tSteps 2-3 in " << now() - t0 << " sec.\n";
            OverlapGraph graphObj(&loaderObj);
            if (minStep == 1 && !saveAll) { double tw = now(); loaderObj.saveReadsInFile(outputDir + prefixName + ".reads"); logStream << "\t" << prefixName << ".reads written in " << now() - tw << " sec.\n"; }
            if (saveAll) { HashTable hashObj(&loaderObj); hashObj.saveHashTableInFile(outputDir + prefixName + ".hashTable"); }
            if (lastStep == 3 || saveAll) { double tw = now(); graphObj.saveOverlapGraphInFile(outputDir + prefixName + ".graph3"); logStream << "\t" << prefixName << ".graph3 written in " << now() - tw << " sec.\n"; }
            if (lastStep >= 4) step4(graphObj);
        } else if (lastStep >= 2) {
            HashTable hashObj(&loaderObj);
            t0 = now(); hashObj.hashPrefixesAndSuffix();                                      // main.cpp:76-77 (always rebuilt: P.hashTable is not read)
            auto is = hashObj.stats();
            logStream << "STEP 2: building hash table\n\t         Hash string length: " << is.hash_string_length << "\n\t            Hash table size: " << is.slots
                      << "\n\t Number of hash elements over threshold: " << is.long_buckets << "\n\tStep 2 in " << now() - t0 << " sec.\n";
            if (lastStep == 2) {
                if (minStep == 1 && !saveAll) { double tw = now(); loaderObj.saveReadsInFile(outputDir + prefixName + ".reads"); logStream << "\t" << prefixName << ".reads written in " << now() - tw << " sec.\n"; }
            }
            if (lastStep == 2 || saveAll) {                                                   // main.cpp:80-81: P.hashTable with -s or when the run ends here
                double tw = now(); hashObj.saveHashTableInFile(outputDir + prefixName + ".hashTable"); logStream << "\t" << prefixName << ".hashTable written in " << now() - tw << " sec.\n";
            }
            if (lastStep >= 3) {                                                              // main.cpp:92-132
                EconomyGraph economyObj(&hashObj);
                t0 = now(); economyObj.buildInitialOverlapGraph();
                auto os = economyObj.stats();
                logStream << "STEP 3: building overlap graph\n     Total contained by extension: " << os.contained_extension << "\n          Total contained by size: " << os.contained_size
                          << "\n            Total left to explore: " << os.left_to_explore << "\n              Verified overlaps: " << os.verified_overlaps << "\n";
                economyObj.buildOverlapGraphEconomy();
                os = economyObj.stats();
                logStream << "     Total edges inserted: " << os.edges_inserted << "\n  Transitive edge removed: " << os.transitive_removed << "\n";
                economyObj.sortEconomyGraph();
                OverlapGraph graphObj(&economyObj, &loaderObj);
                graphObj.convertGraph();
                logStream << "     Edges in the graph: " << economyObj.stats().edges << "\n\tStep 3 in " << now() - t0 << " sec.\n";
                if (minStep == 1 && !saveAll) { double tw = now(); loaderObj.saveReadsInFile(outputDir + prefixName + ".reads"); logStream << "\t" << prefixName << ".reads written in " << now() - tw << " sec.\n"; }
                if (lastStep == 3 || saveAll) { double tw = now(); graphObj.saveOverlapGraphInFile(outputDir + prefixName + ".graph3"); logStream << "\t" << prefixName << ".graph3 written in " << now() - tw << " sec.\n"; }      // main.cpp:120-131
                if (lastStep >= 4) step4(graphObj);
            }
        }
        if (timing) fprintf(stderr, "[cli] until the last file is written %8.1f ms\n", 1e3 * (now() - tMain));
        if (maxStep > 4) {
            cout << "sage2ov: steps 1-4 done; continue with the reference: SAGE2 " << (listInput != "" ? "-l " + listInput : "-f " + fileInput) << " -k " << minOverlap
                 << " -o " << (outputDir == "" ? "." : outputDir) << " -p " << prefixName << " -i " << prefixName << " -m 5 -M " << maxStep << "\n";
        }
        // every file is written and closed: leave without tearing down gigabytes of host vectors and the HIP runtime one allocation at a time
        // (0.25 s on a 10 M-read run; the kernel reclaims host and device memory of an exiting process at once)
        logStream.flush(); logStream.close(); cout.flush(); fflush(nullptr);
        if (!getenv("SAGE2OV_FULL_TEARDOWN")) _exit(0);
    } catch (const Error& e) {
        logStream << "sage2ov error " << e.code << " : " << e.what() << "!\n";               // utils.cpp:36-40 printError
        cerr << "sage2ov error " << e.code << ": " << e.what() << "\n";
        return EXIT_FAILURE;
    }
    return 0;
}
