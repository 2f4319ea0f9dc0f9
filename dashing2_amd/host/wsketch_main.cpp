// wsketch_main.cpp -- `dashing2 wsketch`: weighted-set sketching of binary id / weight / indptr files
// (reference src/wsketch.cpp:264-377 wsketch_main, :193-228 wmh_from_file, :131-191 wmh_from_file_csr).
//
// The device work is d2g_bmh_from_weighted_ids (BagMinHash of explicit weighted sets).  Everything here is the
// reference's file shell: flag letters, input element types, which sketch type a flag combination selects,
// output file names and layouts.  Only the BagMinHash selections are in this build's scope (SURVEY 2.2:
// ProbMinHash and FullSetSketch are out); the others are refused with a message that says so.
//
// Which sketch a flag selects is NOT what the usage text says -- the code decides (and that is what a drop-in
// must follow): `sketchtype` is 1 by default, 0 with -B, -1 with -q (wsketch.cpp:267,276-277), and
//   one or two inputs  : minhash() picks ProbMinHash for 0, BagMinHash for 1   (wsketch.cpp:80-84)
//   three inputs (CSR) : ProbMinHash for 1, BagMinHash for 0                    (wsketch.cpp:148-150)
// so BagMinHash is the DEFAULT of the 1-D form and needs -B in the CSR form.
//
// BagMinHash register VALUES follow this repository's BMH-D2G spec (the reference's sketch/bmh.h is absent):
// files written here carry valid weighted-minhash sketches that are comparable among themselves, not with
// files written by a stock dashing2.  The reference's `to_sigs<uint64_t>()` ("hashes") is also absent source:
// the bit patterns of the register doubles are written in its place (injective, equality-preserving).
#include "../../include/d2g.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>

namespace d2h {

namespace {

int wsketch_usage() {                                             // wsketch.cpp:230-262 (abridged to what this build does)
    std::fprintf(stderr, "Sketch raw IDs, with optional weights added\n"
                         "Usage: dashing2 wsketch [input.bin] <Optional: input.weights.bin> <Optional: indptr.bin for CSR data>\n"
                         "-S: set sketch size\n"
                         "-u: Read 32-bit identifiers from input.bin rather than 64-bit\n"
                         "-f: Read 32-bit floating point weights from [input.weights.bin]\n"
                         "-H: Read 16-bit data weights from [input.weight.bin] (Default: float64)\n"
                         "-U: Read 32-bit data weights from [input.weight.bin] (Default: float64)\n"
                         "-P: Read 32-bit indptr integers [indptr.bin] (Default: uint64_t)\n"
                         "-B / -q: sketch type selection (see the reference; this MI355X build implements the BagMinHash selections:\n"
                         "         no flag with one or two inputs, -B with three inputs)\n"
                         "-o: outprefix. If unset, uses [input.bin]\n"
                         "-p: Set number of threads (ignored: the device does the work)\n");
    return 1;
}

[[noreturn]] void wdie(const std::string &msg) {
    std::fprintf(stderr, "Exception %s\n", msg.c_str());
    std::exit(1);
}

// whole file as T (reference: FReader::getvec / fromfile, wsketch.cpp:113-129; .gz/.xz/.bz2 go through the
// same external decompressors, wsketch.cpp:92-109)
template <class T>
std::vector<T> read_vec(const std::string &path) {
    auto ends = [&](const char *suf) { const size_t n = std::strlen(suf); return path.size() >= n && path.compare(path.size() - n, n, suf) == 0; };
    std::string cmd;
    if (ends(".gz")) cmd = "gzip -dc ";
    else if (ends(".xz")) cmd = "xz -dc ";
    else if (ends(".bz2")) cmd = "bzip2 -dc ";
    std::FILE *fp = cmd.empty() ? std::fopen(path.c_str(), "rb") : ::popen((cmd + path).c_str(), "r");
    if (!fp) wdie("Failed to open path '" + path + "' for reading");
    std::vector<T> ret;
    T buf[4096];
    for (size_t n; (n = std::fread(buf, sizeof(T), 4096, fp)) > 0;) ret.insert(ret.end(), buf, buf + n);
    if (cmd.empty()) std::fclose(fp); else ::pclose(fp);
    return ret;
}

template <class T>
std::vector<double> weights_as_double(const std::string &path) {
    const std::vector<T> v = read_vec<T>(path);
    return std::vector<double>(v.begin(), v.end());               // h.update(i, weights[i]) converts to the sketch's double
}

std::vector<double> read_weights(const std::string &path, int f32) {
    // usef32: 1 float, -1 uint16_t, -2 uint32_t (CSR form only; the 1-D form reads double for -2: wsketch.cpp:217-223), else double
    if (f32 == 1) return weights_as_double<float>(path);
    if (f32 == -1) return weights_as_double<uint16_t>(path);
    if (f32 == -2) return weights_as_double<uint32_t>(path);
    return read_vec<double>(path);
}

std::vector<uint64_t> read_ids(const std::string &path, bool u32) {
    if (!u32) return read_vec<uint64_t>(path);
    const std::vector<uint32_t> v = read_vec<uint32_t>(path);
    return std::vector<uint64_t>(v.begin(), v.end());
}

void write_file(const std::string &path, const void *data, size_t nbytes) {
    std::FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp) wdie("Failed to open " + path);
    if (nbytes && std::fwrite(data, 1, nbytes, fp) != nbytes) wdie("Failed to write " + path);
    std::fclose(fp);
}

const char *kOutOfScope = "dashing2 (MI355X): this flag combination selects %s in the reference (src/wsketch.cpp:%s), which is outside this "
                          "build's hot-path scope; BagMinHash is the default with one or two inputs and needs -B with three inputs.\n";

}  // namespace

int wsketch_main(int argc, char **argv) {
    uint64_t sketchsize = 1024;
    int sketchtype = 1;
    bool u32 = false, ip32 = false;
    int f32 = 0;
    std::string outpref;
    optind = 1;
    for (int c; (c = getopt(argc, argv, "p:o:S:UPqBHPufh?")) >= 0;) {
        switch (c) {
            case 'p': break;                                       // host threads: nothing to parallelise on the host
            case 'S': sketchsize = std::strtoull(optarg, nullptr, 10); break;
            case 'B': sketchtype = 0; break;
            case 'q': sketchtype = -1; break;
            case 'u': u32 = true; break;
            case 'f': f32 = 1; break;
            case 'H': f32 = -1; break;
            case 'U': f32 = -2; break;
            case 'o': outpref = optarg; break;
            case 'P': ip32 = true; break;
            case '?': case 'h': return wsketch_usage();
        }
    }
    const int diff = argc - optind;
    if (diff < 1 || diff > 3) {
        std::fprintf(stderr, "Required: between one and three positional arguments. All flags must come before positional arguments. Diff: %d\n", diff);
        return wsketch_usage();
    }
    if (outpref.empty()) outpref = argv[optind];
    if (sketchsize < 1) wdie("sketch size must be positive");

    auto open_ctx = []() {
        d2g_ctx *ctx = nullptr;
        const char *dv = std::getenv("D2G_DEVICE");
        const int rc = d2g_ctx_create(dv ? std::atoi(dv) : 0, &ctx);
        if (rc != D2G_OK) wdie(std::string("dashing2 (MI355X) needs a gfx950 GPU; d2g_ctx_create: ") + d2g_strerror(rc) + " (there is no CPU fallback)");
        return ctx;
    };
    auto run = [&](d2g_ctx *ctx, const std::vector<uint64_t> &set_off, const std::vector<double> *w, std::vector<double> &sigs,
                   std::vector<double> &tw, std::vector<uint64_t> &owner) {
        // the reference feeds update() the element's POSITION within its set and maps the sampled positions back through the
        // id array afterwards (wsketch.cpp:31-37,57-67): positions are the element ids the sketch sees
        const size_t nsets = set_off.size() - 1;
        std::vector<uint64_t> pos(set_off.back());
        for (size_t i = 0; i < nsets; ++i)
            for (uint64_t e = set_off[i]; e < set_off[i + 1]; ++e) pos[e] = e - set_off[i];
        sigs.assign(nsets * sketchsize, 0.); tw.assign(nsets, 0.); owner.assign(nsets * sketchsize, ~0ull);
        const int rc = d2g_bmh_from_weighted_ids(ctx, pos.data(), w ? w->data() : nullptr, set_off.data(), nsets, sketchsize, sigs.data(),
                                                 tw.data(), owner.data());
        if (rc != D2G_OK) wdie(std::string("d2g_bmh_from_weighted_ids: ") + d2g_strerror(rc) + " (" + d2g_last_error(ctx) + ")");
    };

    if (diff == 3) {                                               // CSR: wsketch.cpp:297-349
        if (sketchtype != 0) {
            std::fprintf(stderr, kOutOfScope, sketchtype == 1 ? "ProbMinHash" : "FullSetSketch", "148-150");
            return 1;
        }
        const std::string idpath = argv[optind], cpath = argv[optind + 1], ippath = argv[optind + 2];
        const std::vector<uint64_t> ids = read_ids(idpath, u32);
        std::vector<uint64_t> indptr = ip32 ? [&] { const auto v = read_vec<uint32_t>(ippath); return std::vector<uint64_t>(v.begin(), v.end()); }()
                                            : read_vec<uint64_t>(ippath);
        if (indptr.size() < 2) wdie("No sketches found in file; this suggests there was an error.");
        std::vector<double> weights;
        const bool have_w = !cpath.empty() && cpath != "-";
        if (have_w) weights = read_weights(cpath, f32);
        if (indptr.back() > ids.size() || (have_w && weights.size() < indptr.back())) wdie("indptr runs past the id / weight arrays");
        for (size_t i = 0; i + 1 < indptr.size(); ++i) if (indptr[i] > indptr[i + 1]) wdie("indptr is not monotone");
        const uint64_t nsketches = indptr.size() - 1;
        // sets are addressed relative to indptr[0] (the reference indexes weights[j], indices[j] for j in [indptr[i], indptr[i+1]))
        const uint64_t base = indptr.front();
        std::vector<uint64_t> set_off(indptr.size());
        for (size_t i = 0; i < indptr.size(); ++i) set_off[i] = indptr[i] - base;
        std::vector<double> wsub;
        if (have_w) wsub.assign(weights.begin() + base, weights.begin() + indptr.back());
        d2g_ctx *ctx = open_ctx();
        std::vector<double> sigs, tw;
        std::vector<uint64_t> owner;
        run(ctx, set_off, have_w ? &wsub : nullptr, sigs, tw, owner);
        d2g_ctx_destroy(ctx);
        const std::string tail = "." + std::to_string(nsketches) + "." + std::to_string(sketchsize);
        {   // sampled ids: ind[x] with ind = indices + b (wsketch.cpp:36-37); a register no element reached keeps id ~0
            std::vector<uint64_t> sampled(nsketches * sketchsize);
            for (uint64_t i = 0; i < nsketches; ++i)
                for (uint64_t r = 0; r < sketchsize; ++r) {
                    const uint64_t o = owner[i * sketchsize + r];
                    sampled[i * sketchsize + r] = o == ~0ull ? ~0ull : ids[indptr[i] + o];
                }
            write_file(outpref + ".sampled.indices.stacked" + tail + ".i64", sampled.data(), sampled.size() * 8);
        }
        {   // stacked registers: [u64 n][u64 S][f64 total weight x n][f64 x n*S]  -- the `cmp --presketched` layout
            std::FILE *fp = std::fopen((outpref + ".sampled.regs.stacked" + tail + ".f64").c_str(), "wb");
            if (!fp) wdie("Failed to open " + outpref + ".sampled.regs.stacked" + tail + ".f64");
            const uint64_t hdr[2] = {nsketches, sketchsize};
            if (std::fwrite(hdr, 8, 2, fp) != 2 || std::fwrite(tw.data(), 8, tw.size(), fp) != tw.size() ||
                std::fwrite(sigs.data(), 8, sigs.size(), fp) != sigs.size())
                wdie("Failed to write MH registers to disk.");
            std::fclose(fp);
        }
        write_file(outpref + ".sampled.hashes.stacked" + tail + ".i64", sigs.data(), sigs.size() * 8);   // to_sigs<uint64_t>(): see header
        {
            std::FILE *fp = std::fopen((outpref + ".sampled.info.txt").c_str(), "wb");
            if (!fp) wdie("Failed to open " + outpref + ".sampled.info.txt");
            for (double t : tw) std::fprintf(fp, "%0.30Lg\n", (long double)t);                           // nlfmt<long double>, enums.h:164
            std::fclose(fp);
        }
        return 0;
    }

    // one or two inputs: wsketch.cpp:350-377
    if (sketchtype != 1) {
        std::fprintf(stderr, kOutOfScope, sketchtype == 0 ? "ProbMinHash" : "FullSetSketch", "80-84");
        return 1;
    }
    const std::string idpath = argv[optind], cpath = diff == 2 ? argv[optind + 1] : "";
    const std::vector<uint64_t> ids = read_ids(idpath, u32);
    std::vector<double> weights;
    if (!cpath.empty()) {
        weights = read_weights(cpath, f32 == -2 ? 0 : f32);        // the 1-D reader has no uint32 branch (wsketch.cpp:217-223)
        if (weights.size() != ids.size()) wdie("weight and id files hold different numbers of elements");
    }
    d2g_ctx *ctx = open_ctx();
    std::vector<uint64_t> set_off = {0, ids.size()};
    std::vector<double> sigs, tw;
    std::vector<uint64_t> owner;
    run(ctx, set_off, cpath.empty() ? nullptr : &weights, sigs, tw, owner);
    d2g_ctx_destroy(ctx);
    write_file(outpref + ".sampled.indices.u64", sigs.data(), sigs.size() * 8);      // tuple element 1 = to_sigs<uint64_t>(): see header
    {
        std::FILE *fp = std::fopen((outpref + ".sampled.hashes.f64").c_str(), "wb");
        if (!fp) wdie("Failed to open sigpath " + outpref + ".sampled.hashes.f64");
        const double t = tw[0];
        std::fwrite(&t, 8, 1, fp);
        std::fwrite(sigs.data(), 8, sigs.size(), fp);
        std::fclose(fp);
    }
    {
        std::vector<uint64_t> sampled(sketchsize);
        for (uint64_t r = 0; r < sketchsize; ++r) sampled[r] = owner[r] == ~0ull ? ~0ull : ids[owner[r]];
        write_file(outpref + ".sampled.ids.u64", sampled.data(), sampled.size() * 8);
    }
    {
        // wsketch.cpp:368-374.  The last `+=` adds four chars as INTEGERS and appends the sum as one char: reproduced
        std::string msg = std::string("Total weight: ") + std::to_string((long double)tw[0]) + ";" + argv[optind];
        if (optind + 1 < argc) msg += std::string(";") + argv[optind + 1];
        msg += char(';' + (f32 == 1 ? 'f' : f32 == 0 ? 'd' : 'H') + ';' + (u32 ? 'W' : 'L'));
        msg += '\n';
        write_file(outpref + ".sampled.tw.txt", msg.data(), msg.size());
        std::fputs(msg.c_str(), stderr);
    }
    return 0;
}

}  // namespace d2h
