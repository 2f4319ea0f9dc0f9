#include "d2_options.h"
#include <sched.h>
#include <cstdio>
#include <fstream>
#include <thread>
#include "../../include/d2g.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <getopt.h>
#include <set>

namespace d2h {

// every --long flag the reference accepts (src/options.h:175-286); anything else is rejected with
// the reference's message (src/options.h:290-304).
static const char *const VALID_LONG[] = {
    "128bit", "BMH", "PMH", "asymmetric", "asymmetric-all-pairs", "bagminhash", "batch-size", "bbit-sigs", "bed",
    "bigwig", "binary", "binary-output", "bmh", "by-chrom", "cache", "cache-sketches", "cmp-outfile", "cmpout",
    "compute-edit-distance", "containment", "count-threshold", "countdict", "countmin-size", "countsketch-size",
    "distance", "distout", "doph", "downsample", "edit-distance", "emit-binary", "enable-protein", "entmin",
    "exact-kmer-dist", "fastcmp", "fastcmp-bytes", "fastcmp-nibbles", "fastcmp-shorts", "fastcmp-words", "ffile",
    "filterset", "full", "full-setsketch", "greedy", "help", "hp-compress", "intersection", "intersection-size",
    "kmer-length", "leafcutter", "long-kmers", "mash-distance", "maxcand", "multiset", "nLSH", "nlsh", "no-canon",
    "normalize-intervals", "one-perm", "oneperm", "oneperm-setsketch", "oph", "outfile", "outprefix", "pairlist",
    "parse-by-seq", "phylip", "pmh", "pminhash", "poisson-distance", "prefix", "prob", "probminhash", "probs",
    "protein", "protein14", "protein20", "protein6", "protein8", "qfile", "refine-exact", "regbytes", "regsize",
    "save-kmercounts", "save-kmers", "seed", "seq", "set", "setsketch-ab", "sig-ram-limit", "similarity-threshold",
    "sketch-size-l2", "sketchsize", "spacing", "square", "symmetric-containment", "threads", "threshold", "top-k",
    "topk", "union-size", "verbose", "window-size", "seqs-in-ram"};

enum {
    OPT_CMPOUT = 1000, OPT_OUTPREF, OPT_BINARY, OPT_PHYLIP, OPT_ASYM, OPT_ISZ, OPT_USZ, OPT_MASH, OPT_SYMCONTAIN,
    OPT_CONTAIN, OPT_SEED, OPT_HELP, OPT_BATCH, OPT_PRESKETCHED, OPT_MULTISET, OPT_PARSEBYSEQ, OPT_FMTCOMPAT, OPT_GPUSTATS, OPT_UNSUPPORTED
};

void sketch_usage() {
    std::fprintf(stderr, "dashing2 sketch <opts> [fastas... (optional)]\n"
                         "MI355X build: One-Permutation SetSketch (default) or --multiset BagMinHash of k-mers (k <= 32),\n"
                         "optional all-pairs comparison.\n"
                         "  -k/--kmer-length k   -S/--sketchsize S   -L/--sketch-size-l2 l   -p/--threads n\n"
                         "  -F/--ffile paths.txt -Q/--qfile queries.txt  -o/--outfile stacked.bin\n"
                         "  --cmpout/--distout/--cmp-outfile out   --phylip   --binary-output   --asymmetric-all-pairs\n"
                         "  --no-canon/-C  --seed s  --cache/-W  --outprefix dir  --oph/-Z\n"
                         "  --multiset/--bagminhash/-B [-m/--count-threshold c]   --parse-by-seq (one sketch per record of ONE file)\n"
                         "  --distance/--mash-distance --containment --symmetric-containment --intersection --union-size\n"
                         "  --batch-size n  -v\n"
                         "  --fmt-compat {10,11}   float text of PHYLIP/TSV output as fmt < 11 (default: fixed notation below 1e16) or\n"
                         "                         fmt >= 11 (exponent form from 1e7) prints it -- the reference's fmt is an unpinned submodule\n"
                         "  --gpu-stats file.json  one JSON object per run: device(s), kernel milliseconds, bit-plane counts, algorithmic bytes, wall phases\n"
                         "  D2G_DEVICES=all|0,1,..  spread `sketch` (inputs dealt to the GPUs) and `cmp` (rows of the matrix) over several GPUs\n");
}
void cmp_usage() {
    std::fprintf(stderr, "dashing2 cmp <opts> [fastas... (optional)]\n"
                         "--presketched\t To compute distances using a pre-sketched method (e.g., dashing2 sketch -o path), "
                         "use this flag and pass in a single positional argument.\n");
    sketch_usage();
}

static bool validate_long_flags(char **argv, bool is_cmp) {
    for (char **p = argv; *p; ++p) {
        const size_t len = std::strlen(*p);
        if (len > 2 && std::memcmp(*p, "--", 2) == 0) {
            const std::string flag(*p + 2);
            if (is_cmp && flag == "presketched") continue;
            if (flag == "fmt-compat") continue;              // this build's own flag (float text generation of the fmt library)
            if (flag == "gpu-stats") continue;               // this build's own flag (machine-readable per-run record)
            bool ok = false;
            for (const char *v : VALID_LONG) if (flag == v) { ok = true; break; }
            if (!ok) {   // src/options.h:298-301
                std::fprintf(stderr, "flag %s not found in expected set. See usage.\n", flag.c_str());
                std::fprintf(stderr, "Exception Flag %s not found\n", flag.c_str());
                return false;
            }
        }
    }
    return true;
}

std::string Options::to_string() const {    // src/d2.cpp:10-43 (fields that exist in this scope)
    char buf[4096];
    int pos = std::snprintf(buf, sizeof buf, "Dashing2Options;k:%d", k);
    if (w > 0) pos += std::snprintf(buf + pos, sizeof buf - pos, ";w:%d", w);
    pos += std::snprintf(buf + pos, sizeof buf - pos, ";%s", parse_by_seq ? "parsebyseq" : "parsebyfile");
    pos += std::snprintf(buf + pos, sizeof buf - pos, ";trimchr");
    pos += std::snprintf(buf + pos, sizeof buf - pos, ";sketchsize:%zu", sketchsize);
    if (count_threshold > 0) pos += std::snprintf(buf + pos, sizeof buf - pos, ";%u", count_threshold);
    pos += std::snprintf(buf + pos, sizeof buf - pos, ";sketchtype:%s",
                         kmer_result == ONE_PERM ? "onepermsetsketch"
                         : (sspace == SPACE_SET ? "fullsetsketch" : sspace == SPACE_MULTISET ? "bagminhash" : "probminhash"));
    pos += std::snprintf(buf + pos, sizeof buf - pos, ";%s", "Fastx");
    if (!outprefix.empty()) pos += std::snprintf(buf + pos, sizeof buf - pos, ";outprefix:%s", outprefix.c_str());
    if (canon) pos += std::snprintf(buf + pos, sizeof buf - pos, ";canon");
    return std::string(buf, pos);
}

int parse_options(int argc, char **argv, Options &o) {
    if (!validate_long_flags(argv, o.is_cmp)) return 1 + 1;
    if (o.is_cmp) o.w = 0;                                    // src/cmp_main.cpp:202
    static const struct option longopts[] = {
        {"ffile", required_argument, 0, 'F'}, {"qfile", required_argument, 0, 'Q'}, {"threads", required_argument, 0, 'p'},
        {"sketchsize", required_argument, 0, 'S'}, {"cmpout", required_argument, 0, OPT_CMPOUT},
        {"distout", required_argument, 0, OPT_CMPOUT}, {"cmp-outfile", required_argument, 0, OPT_CMPOUT},
        {"outprefix", required_argument, 0, OPT_OUTPREF}, {"prefix", required_argument, 0, OPT_OUTPREF},
        {"kmer-length", required_argument, 0, 'k'}, {"outfile", required_argument, 0, 'o'},
        {"window-size", required_argument, 0, 'w'},
        {"binary-output", no_argument, 0, OPT_BINARY}, {"emit-binary", no_argument, 0, OPT_BINARY}, {"binary", no_argument, 0, OPT_BINARY},
        {"intersection", no_argument, 0, OPT_ISZ}, {"intersection-size", no_argument, 0, OPT_ISZ}, {"union-size", no_argument, 0, OPT_USZ},
        {"mash-distance", no_argument, 0, OPT_MASH}, {"distance", no_argument, 0, OPT_MASH}, {"poisson-distance", no_argument, 0, OPT_MASH},
        {"symmetric-containment", no_argument, 0, OPT_SYMCONTAIN}, {"containment", no_argument, 0, OPT_CONTAIN},
        {"phylip", no_argument, 0, OPT_PHYLIP},
        {"asymmetric-all-pairs", no_argument, 0, OPT_ASYM}, {"asymmetric", no_argument, 0, OPT_ASYM}, {"square", no_argument, 0, OPT_ASYM},
        {"oneperm-setsketch", no_argument, 0, 'Z'}, {"oneperm", no_argument, 0, 'Z'}, {"one-perm", no_argument, 0, 'Z'},
        {"oph", no_argument, 0, 'Z'}, {"doph", no_argument, 0, 'Z'},
        {"cache", no_argument, 0, 'W'}, {"cache-sketches", no_argument, 0, 'W'}, {"no-canon", no_argument, 0, 'C'},
        {"seed", required_argument, 0, OPT_SEED}, {"help", no_argument, 0, OPT_HELP},
        {"batch-size", required_argument, 0, OPT_BATCH}, {"sketch-size-l2", required_argument, 0, 'L'},
        {"verbose", no_argument, 0, 'v'}, {"presketched", no_argument, 0, OPT_PRESKETCHED},
        {"multiset", no_argument, 0, OPT_MULTISET}, {"bagminhash", no_argument, 0, OPT_MULTISET}, {"bmh", no_argument, 0, OPT_MULTISET},
        {"BMH", no_argument, 0, OPT_MULTISET},
        {"count-threshold", required_argument, 0, 'm'}, {"threshold", required_argument, 0, 'm'},
        {"parse-by-seq", no_argument, 0, OPT_PARSEBYSEQ},
        {"fmt-compat", required_argument, 0, OPT_FMTCOMPAT},
        {"gpu-stats", required_argument, 0, OPT_GPUSTATS},
        {0, 0, 0, 0}};
    // every other valid reference flag is recognised but outside the hot-path scope
    std::vector<struct option> all(longopts, longopts + sizeof(longopts) / sizeof(longopts[0]) - 1);
    std::set<std::string> have;
    for (auto &x : all) have.insert(x.name);
    static const std::set<std::string> with_arg = {"topk", "top-k", "similarity-threshold", "fastcmp", "regsize", "regbytes",
        "countsketch-size", "countmin-size", "count-threshold", "threshold", "downsample", "spacing", "filterset", "greedy",
        "nlsh", "nLSH", "sig-ram-limit", "maxcand", "setsketch-ab", "pairlist"};
    for (const char *v : VALID_LONG)
        if (!have.count(v)) all.push_back({v, with_arg.count(v) ? required_argument : no_argument, 0, OPT_UNSUPPORTED});
    all.push_back({0, 0, 0, 0});
    int c, idx = 0;
    optind = 1;
    while ((c = getopt_long(argc, argv, "m:p:k:w:c:f:S:F:Q:o:L:CNs2BPWh?ZJGHv", all.data(), &idx)) >= 0) {
        switch (c) {
            case 'F': o.ffile = optarg; break;
            case 'Q': o.qfile = optarg; o.ok = PANEL; break;
            case 'p': o.nt = std::atoi(optarg); break;
            case 'S': o.sketchsize = size_t(std::atoi(optarg)); break;
            case 'k': o.k = std::atoi(optarg); break;
            case 'w': o.w = std::atoi(optarg); break;
            case 'o': o.outfile = optarg; break;
            case 'C': o.canon = false; break;
            case 'W': o.cache = true; break;
            case 'Z': o.kmer_result = ONE_PERM; break;
            case 'v': ++o.verbosity; break;
            case 'L': {
                const int l = std::atoi(optarg);
                if (l <= 0 || l >= 64) { std::fprintf(stderr, "Error: ssl2 is out of bounds. ssl2: %d.\n", l); return 1 + 1; }
                o.sketchsize = size_t(1) << l;
                std::fprintf(stderr, "Using log_2 sketchsize = %d, yielding sketchsize = %zu\n", l, o.sketchsize);
            } break;
            case OPT_CMPOUT: o.cmpout = optarg; break;
            case OPT_OUTPREF: o.outprefix = optarg; break;
            case OPT_BINARY: o.of = MACHINE_READABLE; break;
            case OPT_PHYLIP: o.ok = PHYLIP; break;
            case OPT_ASYM: o.ok = ASYMMETRIC_ALL_PAIRS; break;
            case OPT_ISZ: o.measure = D2G_INTERSECTION; break;
            case OPT_USZ: o.measure = D2G_UNION_SIZE; break;
            case OPT_MASH: o.measure = D2G_POISSON_LLR; break;
            case OPT_SYMCONTAIN: o.measure = D2G_SYMMETRIC_CONTAINMENT; break;
            case OPT_CONTAIN: o.measure = D2G_CONTAINMENT; break;
            case OPT_SEED: o.seedseed = std::strtoull(optarg, 0, 10); break;
            case OPT_BATCH: o.batch_size = std::strtoull(optarg, 0, 10); break;
            case OPT_PRESKETCHED: o.presketched = true; break;
            case OPT_MULTISET: case 'B': o.sspace = SPACE_MULTISET; break;         // options.h:111
            case 'm': o.count_threshold = unsigned(std::atoi(optarg)); break;      // options.h:352
            case OPT_PARSEBYSEQ: o.parse_by_seq = true; break;                     // options.h:378
            case OPT_FMTCOMPAT:
                o.fmt_compat = std::atoi(optarg);
                if (o.fmt_compat != 10 && o.fmt_compat != 11) {
                    std::fprintf(stderr, "dashing2 (MI355X): --fmt-compat takes 10 (float text of fmt < 11, default) or 11 (fmt >= 11)\n");
                    return 1 + 1;
                }
                break;
            case OPT_GPUSTATS: o.gpu_stats = optarg; break;
            case OPT_HELP: case 'h': case '?': o.is_cmp ? cmp_usage() : sketch_usage(); return 1 + 1;
            case OPT_UNSUPPORTED:
                std::fprintf(stderr, "dashing2 (MI355X): option --%s is outside the hot-path scope of this build "
                                     "(OPH sketching + dense all-pairs comparison).\n", all[idx].name);
                return 1 + 1;
            default:
                std::fprintf(stderr, "dashing2 (MI355X): option -%c is outside the hot-path scope of this build.\n", c);
                return 1 + 1;
        }
    }
    if (o.k < 0) o.k = 32;                                     // nregperitem(DNA): sketch_main.cpp:70, options.h:474
    if (o.nt < 0) {                                            // sketch_main.cpp:71-74
        if (const char *s = std::getenv("OMP_NUM_THREADS")) o.nt = std::max(std::atoi(s), 1);
    }
    if (o.nt < 1) o.nt = 1;
    if (const char *d = std::getenv("D2G_DEVICE")) o.device = std::atoi(d);
    for (int i = optind; i < argc; ++i) o.paths.push_back(argv[i]);
    if (!o.ffile.empty()) {
        std::ifstream ifs(o.ffile);
        if (!ifs) { std::fprintf(stderr, "Exception No path found at %s\n", o.ffile.c_str()); return 1 + 1; }
        for (std::string l; std::getline(ifs, l);) o.paths.push_back(l);
    }
    const size_t nref = o.paths.size();
    if (!o.qfile.empty()) {
        std::ifstream ifs(o.qfile);
        for (std::string l; std::getline(ifs, l);) o.paths.push_back(l);
    }
    o.nq = o.paths.size() - nref;
    if (o.sspace != SPACE_SET && o.kmer_result == ONE_PERM) o.kmer_result = FULL_SETSKETCH;   // sketch_main.cpp:124-127
    if (o.sspace == SPACE_SET && o.count_threshold > 0) {
        std::fprintf(stderr, "dashing2 (MI355X): -m/--count-threshold with set sketches (OPH min-count filtering, oph.h:186-205) "
                             "is outside this build's hot-path scope; it is supported with --multiset.\n");
        return 1 + 1;
    }
    if (o.k > 32) {
        std::fprintf(stderr, "dashing2 (MI355X): k = %d > 32 uses the reference's rolling-hash encoder "
                             "(fastxsketch.cpp:420), which is outside this build's hot-path scope.\n", o.k);
        return 1 + 1;
    }
    if (o.w > o.k) {
        std::fprintf(stderr, "dashing2 (MI355X): windowed minimizers (-w > k) are outside this build's hot-path scope.\n");
        return 1 + 1;
    }
    return 0;
}

}  // namespace d2h


namespace d2h {

static unsigned usable_cpus() {
    static const unsigned cached = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n, unsigned(c)); }
        {   // cgroup v2: "<quota> <period>" or "max <period>"
            std::ifstream f("/sys/fs/cgroup/cpu.max");
            std::string q; long long per = 0;
            if (f >> q >> per && q != "max" && per > 0) {
                const long long quota = std::atoll(q.c_str());
                if (quota > 0) n = std::min(n, unsigned(std::max<long long>(1, (quota + per / 2) / per)));
            }
        }
        {   // cgroup v1
            std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
            long long quota = -1, per = 0;
            if (fq >> quota && fp >> per && quota > 0 && per > 0) n = std::min(n, unsigned(std::max<long long>(1, (quota + per / 2) / per)));
        }
        return std::max(1u, n);
    }();
    return cached;
}

unsigned Options::workers() const {
    const unsigned req = nthreads();
    if (const char *e = std::getenv("D2G_NO_CPU_CAP")) if (e[0] == '1') return req;
    return std::min(req, usable_cpus());
}

}  // namespace d2h
