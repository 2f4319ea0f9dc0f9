// d2_options.h -- option structs and parsing of the drop-in `dashing2 sketch|cmp` CLI.
// Flag names, defaults and validation mirror the reference: src/options.h:63-171 (SHARED_OPTS),
// :175-304 (VALID_LONG_OPTION_STRINGS / validate_options), :308-449 (SHARED_FIELDS),
// src/sketch_main.cpp:23-152, src/cmp_main.cpp:200-366.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace d2h {

enum OutputKind { SYMMETRIC_ALL_PAIRS, PHYLIP, ASYMMETRIC_ALL_PAIRS, KNN_GRAPH, NN_GRAPH_THRESHOLD, PANEL, DEDUP };  // enums.h:86-94
enum OutputFormat { MACHINE_READABLE, HUMAN_READABLE };                                                             // enums.h:96-100
enum KmerResult { ONE_PERM = 0, FULL_SETSKETCH = 1 };                                                                // enums.h:70-84 (in scope)
enum SketchSpace { SPACE_SET = 0, SPACE_MULTISET = 1, SPACE_PSET = 2 };                                              // enums.h:35-42

struct Options {
    bool is_cmp = false;
    int k = -1, w = -1, nt = -1;
    bool canon = true, cache = false, presketched = false;
    bool parse_by_seq = false;            // --parse-by-seq (options.h:378): one sketch per FASTX record of ONE input file
    size_t sketchsize = 1024;
    uint64_t seedseed = 0;
    size_t batch_size = 0;
    unsigned count_threshold = 0;         // -m/--threshold/--count-threshold (d2.h:103); --multiset only in this build
    std::string ffile, qfile, outfile, cmpout, outprefix;
    OutputKind ok = SYMMETRIC_ALL_PAIRS;
    OutputFormat of = HUMAN_READABLE;
    int measure = 0;                      // d2g_measure / cmp_main.h:8-17
    KmerResult kmer_result = ONE_PERM;
    SketchSpace sspace = SPACE_SET;
    int verbosity = 0;
    std::vector<std::string> paths;       // references then queries
    size_t nq = 0;                        // number of query paths (-Q)
    int device = 0;                       // D2G_DEVICE env (not a reference flag)
    std::string gpu_stats;                // --gpu-stats FILE (not a reference flag): machine-readable record of the run (SURVEY 5 "Metrics")
    int fmt_compat = 0;                   // --fmt-compat {10,11} (not a reference flag): float text layout of fmt < 11 / >= 11; 0 = not given (10)

    unsigned nthreads() const { return nt < 1 ? 1u : unsigned(nt); }      // as requested (-p / OMP_NUM_THREADS): what is printed
    // threads actually started: the request, cut to the CPUs this process may use (affinity mask, cgroup CPU quota -- a
    // container that shows 256 CPUs and grants 16 makes 112 requested threads throttle each other: 0.70 s instead of 0.36 s
    // for 5 GB of FASTA).  D2G_NO_CPU_CAP=1 keeps the request.
    unsigned workers() const;
    std::string to_string() const;        // Dashing2Options::to_string, src/d2.cpp:10-43
};

// returns 0 to continue, or the process exit code (+1) when parsing decided to stop (usage/errors)
int parse_options(int argc, char **argv, Options &o);
void sketch_usage();
void cmp_usage();

}  // namespace d2h
