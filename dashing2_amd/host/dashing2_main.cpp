// dashing2_main.cpp -- drop-in `dashing2 sketch` / `dashing2 cmp` for the MI355X hot paths.
// Host C++ over the C ABI of libd2g.so (include/d2g.h); mirrors, for the in-scope options,
//   main / dispatch          src/d2.cpp:133-151
//   sketch_main              src/sketch_main.cpp:23-152
//   sketch_core + formats    src/sketch_core.cpp:14-31,108-161 ; src/fastxsketch.cpp:302-424,554-610
//   makedest (cache names)   src/fastxmerge.cpp:70-120
//   cmp_main / load_results  src/cmp_main.cpp:24-198,200-366
//   cmp_core (densify)       src/cmp_core.cpp:686-718,746-751
//   emit_rectangular         src/emitrect.cpp:108-403
#include "../../include/d2g.h"
#include "d2_options.h"
#include "fmtfloat.h"
#include "slot_queue.h"
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <stdexcept>
#include <memory>
#include <string>
#include <functional>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef DASHING2_VERSION
#define DASHING2_VERSION "v2.1.20-mi355x"
#endif

using namespace d2h;

namespace {

struct Result {                              // SketchingResult, src/fastxsketch.h:23-58 (in-scope fields)
    std::vector<std::string> names, destination_files;
    std::vector<double> cardinalities;
    std::vector<double> signatures;          // [N][S] row-major
    double *sigs() { return signatures.data(); }
    const double *sigs() const { return signatures.data(); }
    size_t nsigs() const { return signatures.size(); }
    size_t nq = 0;
};

// The process leaves through _exit once its outputs are closed (main): releasing device memory and unpinning hundreds of MB
// one buffer at a time just before that costs tens of milliseconds and buys nothing.  D2G_FULL_TEARDOWN=1 releases everything.
static const bool g_release_at_exit = std::getenv("D2G_FULL_TEARDOWN") != nullptr;
[[noreturn]] void die(const std::string &msg) {          // THROW_EXCEPTION: src/enums.h:59-63
    std::fprintf(stderr, "Exception %s\n", msg.c_str());
    std::exit(1);
}
void check(d2g_ctx *ctx, int rc, const char *what) {
    if (rc == D2G_OK) return;
    die(std::string(what) + ": " + d2g_strerror(rc) + (ctx ? std::string(" (") + d2g_last_error(ctx) + ")" : std::string()));
}
bool isfile(const std::string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
size_t filesize(const std::string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0 ? size_t(st.st_size) : 0; }
std::string trim_folder(const std::string &s) {          // src/enums.cpp:22-26
    const auto pos = s.find_last_of('/');
    return pos == std::string::npos ? s : s.substr(pos + 1);
}
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// --gpu-stats FILE (SURVEY 5 "Metrics / logging"; the reference only has its verbosity levels, src/enums.h:106-111, and the
// banner of src/d2.cpp:136): what -v prints, machine-readable -- ONE JSON object per run with the device(s), the HIP-event
// milliseconds of every timed kernel family (d2g_set_timing / d2g_kernel_ms), bit-plane counts, algorithmic bytes, wall phases.
struct Stats {
    bool on = false;
    std::string path;
    std::mutex mu;
    std::vector<std::pair<std::string, std::string>> kv;      // key -> value already rendered as JSON
    static std::string esc(const std::string &x) {
        std::string r = "\"";
        for (unsigned char c : x) {
            if (c == '"' || c == '\\') { r += '\\'; r += char(c); }
            else if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); r += b; }
            else r += char(c);
        }
        return r + "\"";
    }
    static std::string numstr(double v) { char b[40]; if (!std::isfinite(v)) return "null"; std::snprintf(b, sizeof b, "%.9g", v); return b; }
    void raw(const std::string &k, const std::string &json) { if (!on) return; std::lock_guard<std::mutex> lk(mu); kv.emplace_back(k, json); }
    void num(const std::string &k, double v) { raw(k, numstr(v)); }
    void str(const std::string &k, const std::string &v) { raw(k, esc(v)); }
    // {"launches": n, "avg_ms": a, "total_ms": n a} of one timed kernel family on one context (synchronises on its events)
    static std::string kernel_json(d2g_ctx *ctx, const char *which) {
        int n = 0; float avg = 0, last = 0;
        if (d2g_kernel_ms(ctx, which, 1, &n, &avg, &last) != D2G_OK) return "null";
        return "{\"launches\": " + std::to_string(n) + ", \"avg_ms\": " + numstr(avg) + ", \"total_ms\": " + numstr(double(avg) * n) + "}";
    }
    void write() {
        if (!on) return;
        std::FILE *fp = std::fopen(path.c_str(), "wb");
        if (!fp) { std::fprintf(stderr, "dashing2 (MI355X): cannot write --gpu-stats file %s\n", path.c_str()); return; }
        std::fputs("{", fp);
        for (size_t i = 0; i < kv.size(); ++i) std::fprintf(fp, "%s%s: %s", i ? ", " : "", esc(kv[i].first).c_str(), kv[i].second.c_str());
        std::fputs("}\n", fp);
        std::fclose(fp);
    }
};
Stats g_stats;
constexpr int TIME_ALL = D2G_TIME_K0 | D2G_TIME_K1 | D2G_TIME_K2 | D2G_TIME_K2PREP | D2G_TIME_K3;

// D2G_DEVICES = "all" | "0,1,2": the GPUs a job may spread over -- `sketch` deals its input groups to them (no collectives),
// `cmp` shards the rows of the matrix (one exchange).  Default: the one device D2G_DEVICE names.  A list that repeats
// a device is allowed (cmp: loopback transport; used by the tests on one GPU).
std::vector<int> job_devices(const Options &o) {
    std::vector<int> d;
    const char *e = std::getenv("D2G_DEVICES");
    if (!e || !*e) return {o.device};
    if (std::strcmp(e, "all") == 0) { for (int i = 0; i < d2g_device_count(); ++i) d.push_back(i); }
    else for (const char *p = e; *p;) { char *q; const long v = std::strtol(p, &q, 10); if (q == p) break; d.push_back(int(v)); p = *q == ',' ? q + 1 : q; }
    if (d.empty()) d.push_back(o.device);
    return d;
}
std::string device_label(int dev) {
    char b[256];
    return d2g_device_name(dev, b, sizeof b) == D2G_OK ? std::string(b) : std::string("?");
}

// src/fastxmerge.cpp:70-120 for DNA, unspaced, w <= k: OPH set sketches and exact-counting multiset sketches
std::string makedest(const Options &o, const std::string &path) {
    std::string ret = path.substr(0, path.find_first_of(' '));
    if (!o.outprefix.empty()) ret = o.outprefix + '/' + trim_folder(path);
    if (o.seedseed != 0) ret += ".seed" + std::to_string(o.seedseed);
    if (o.canon) ret += ".rc_canon";
    ret += ".sketchsize" + std::to_string(o.sketchsize);
    ret += ".k" + std::to_string(o.k);
    if (o.count_threshold > 0) {
        // fastxmerge.cpp:91-95 prints std::to_string(double) when fmod(threshold, 1) != 0 -- but Dashing2Options::count_threshold_ is a
        // uint32_t (d2.h:103) filled by std::atoi (options.h:352), so that branch cannot be reached from the reference's CLI
        // either: `-m 2.7` is 2 there and here.  The integer branch: std::to_string(int(count_threshold_)).
        ret += ".ct_threshold" + std::to_string(int(o.count_threshold));
    }
    if (o.sspace != SPACE_SET) ret += ".ExactCounting";   // to_string(ct()), src/enums.cpp:47; fastxmerge.cpp:96-100
    ret += o.sspace == SPACE_SET ? ".SetSpace" : ".MultisetSpace";   // to_string(sspace), src/enums.cpp:40-46
    ret += ".DNA";                                        // bns::to_string(rht_) (absent bonsai; expected "DNA")
    // to_suffix, src/enums.cpp:28-38, gives ".bmh" for BagMinHash.  Multiset sketches of this build follow the repository's
    // BMH-D2G spec (the reference's sketch/bmh.h is absent): same layout, incomparable register values.  They get their own
    // suffix so that a --cache directory shared with a stock dashing2 can never mix the two silently.
    ret += o.sspace == SPACE_SET ? ".opss" : ".d2gbmh";
    return ret;
}

// one cached sketch: [f64 card][f64 x S]   (src/fastxsketch.cpp:60-112,556-607)
bool load_cached(const std::string &path, double *sig, double *card, size_t S) {
    if (filesize(path) != 8 + 8 * S) {
        if (isfile(path)) std::fprintf(stderr, "Expected %zu bytes of sketch, found %zu\n", S * 8, filesize(path) - 8);
        return false;
    }
    std::FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) return false;
    const bool ok = std::fread(card, 8, 1, fp) == 1 && std::fread(sig, 8, S, fp) == S;
    std::fclose(fp);
    return ok;
}
void write_cached(const std::string &path, const double *sig, double card, size_t S) {
    std::FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp) die("Failed to open file " + path + " for writing sketch.");
    if (std::fwrite(&card, 8, 1, fp) != 1 || std::fwrite(sig, 8, S, fp) != S) die("Failed to write sketch " + path);
    std::fclose(fp);
}

// ------------------------------------------------------------------------------------ sketch
// stacked output: [u64 N][u64 S][f64 card x N][f64 x N*S]   (sketch_core.cpp:130-140, fastxsketch.cpp:236-240)
d2g_ctx *make_ctx(const Options &o);
// The GPU context (HIP runtime start-up, 0.06-0.2 s) is created on a helper thread as soon as the options are parsed, while
// this thread stats / reads / parses the inputs; get() joins.  There is still no CPU fallback: a failure ends the process.
struct LazyCtx {
    const Options &o; std::thread th; d2g_ctx *ctx = nullptr; double t_create = 0, t_warm = 0;
    // `warm`: one-time costs the helper pays right after the context exists (D2G_WARM_*): the first host<->device copy of a process
    // costs ~30 ms whatever its size (tools/cmp_setup_time2.py), code objects ~1 ms per kernel family -- under the input reading
    LazyCtx(const Options &oo, int warm) : o(oo) {
        th = std::thread([this, warm] {
            double t = now();
            ctx = make_ctx(o);
            t_create = now() - t;
            t = now();
            if (warm) (void)d2g_warmup(ctx, warm);
            if (g_stats.on) (void)d2g_set_timing(ctx, TIME_ALL);
            t_warm = now() - t;
        });
    }
    d2g_ctx *get() { if (th.joinable()) th.join(); return ctx; }
    // the context is only torn down on request: main() leaves through _exit once every output is flushed and closed (the HIP
    // runtime's orderly shutdown costs tens of milliseconds that buy a CLI process nothing); D2G_FULL_TEARDOWN=1 keeps it
    ~LazyCtx() { get(); if (ctx && g_release_at_exit) d2g_ctx_destroy(ctx); }
};

void write_stacked(const Result &res, const Options &o) {
    const size_t N = res.names.size(), S = o.sketchsize;
    if (!o.outfile.empty()) {
        if (o.outfile == "-" || o.outfile == "/dev/stdout")
            die("Not yet supported: writing stacked sketches to file streams. This may change.");     // sketch_core.cpp:141-144
        std::FILE *fp = std::fopen(o.outfile.c_str(), "wb");
        if (!fp) die("Failed to open file " + o.outfile + " for in-place modification");
        const uint64_t hdr[2] = {uint64_t(N), uint64_t(S)};
        if (std::fwrite(hdr, 8, 2, fp) != 2 || std::fwrite(res.cardinalities.data(), 8, N, fp) != N ||
            std::fwrite(res.signatures.data(), 8, N * S, fp) != N * S) die("Failed to write " + o.outfile);
        std::fclose(fp);
        // <out>.names.txt (sketch_core.cpp:146-161, enums.h:160 "%0.24g")
        const std::string nf = o.outfile + ".names.txt";
        if (!(fp = std::fopen(nf.c_str(), "wb"))) die("Failed to open outfile at " + nf);
        std::fputs("#Name\tCardinality\n", fp);
        for (size_t i = 0; i < N; ++i) {
            std::fwrite(res.names[i].data(), 1, res.names[i].size(), fp);
            std::fprintf(fp, "\t%0.24g", res.cardinalities[i]);
            std::fputc('\n', fp);
        }
        std::fclose(fp);
    }
}

void sketch_core(Result &res, const Options &o, LazyCtx &lctx) {
    d2g_ctx *ctx = nullptr;                                             // joined once the parser threads are running
    const double t_enter = now();
    const size_t N = o.paths.size(), S = o.sketchsize, m = d2g_oph_m(S);
    if (!N) die("Can't sketch empty path set");
    res.names = o.paths;                                                // fastxsketch.cpp:625
    res.destination_files.resize(N);
    res.cardinalities.assign(N, -1.);
    res.signatures.assign(N * S, 0.);
    const uint64_t xormask = d2g_seed_mask(o.seedseed);                 // d2.h:224 -> enums.cpp:131-140
    std::vector<size_t> todo;
    for (size_t i = 0; i < N; ++i) {
        res.destination_files[i] = makedest(o, o.paths[i]);
        if (o.cache && isfile(res.destination_files[i]) &&
            load_cached(res.destination_files[i], &res.signatures[i * S], &res.cardinalities[i], S))
            continue;                                                   // fastxsketch.cpp:327-373
        todo.push_back(i);
    }
    // groups of inputs bounded by input bytes: read in parallel on the host, sketched one group per launch
    struct FileRef { std::string path; size_t size; };
    std::vector<std::vector<FileRef>> files_of(todo.size());            // the space-separated paths of every input line
    std::vector<std::pair<size_t, size_t>> groups;                      // [begin,end) into todo
    std::vector<size_t> group_bytes;
    size_t limit = size_t(48) << 20;
    if (const char *e = std::getenv("D2G_GROUP_BYTES")) { const long long v = std::atoll(e); if (v >= 1) limit = size_t(v); }   // tests: many small groups
    {
        size_t b = 0, acc = 0;
        for (size_t t = 0; t < todo.size(); ++t) {
            size_t fs = 0;
            const std::string &line = o.paths[todo[t]];
            for (size_t s = 0; s <= line.size();) {
                size_t e = line.find(' ', s);
                if (e == std::string::npos) e = line.size();
                if (e > s) { const std::string p = line.substr(s, e - s); const size_t z = filesize(p); files_of[t].push_back({p, z}); fs += (z + 15) / 16 * 16; }
                s = e + 1;
            }
            if (acc && acc + fs > limit) { groups.emplace_back(b, t); group_bytes.push_back(acc); b = t; acc = 0; }
            acc += fs;
        }
        if (b < todo.size()) { groups.emplace_back(b, todo.size()); group_bytes.push_back(acc); }
    }
    const double t_setup = now();
    // Host ingest pipeline.  Reader threads read() whole groups of FASTA files into page-locked staging
    // buffers; ONE device thread (this one) owns the GPU context: it uploads a group's raw bytes, K0 parses and 2-bit-packs them
    // on the device (d2g_sketcher_ingest_fasta), K1 / K3 sketch the stream, the registers are finalised (x87) and cached.
    // Inputs the device parser refuses (gz members, FASTQ, leading junk: first byte is not '>') are parsed by the host parser
    // (d2g_seqpack) on the reader thread instead, group by group -- the round-2 path.
    // WHICH PARSER IS THE DEFAULT -- measured, 1000 x 5 Mbp FASTA in the page cache, 16 usable cores (profiles/r03_e2e_cli.txt):
    // read()ing a group into staging costs a core as much as read()ing + packing it (22 vs 19 ms per 43 MB: the copy out of the
    // page cache into memory that is not cache-resident is the expensive half, and the packer works on a 5 MB buffer that
    // stays in L2), the device threads are not the bottleneck either way, and 0.8 GB of page-locked staging adds ~0.07 s of
    // teardown when the process exits.  So the device parser buys nothing end to end on this host and the HOST parser stays
    // the default; D2G_DEVICE_PARSE=1 selects the hybrid (device parser whenever a staging buffer is free).
    // The staging buffers are plain memory the readers fill at once; they are page-locked (d2g_host_register) as soon as the
    // GPU context exists, so neither the context creation nor the pinning delays the reading.
    double t_parse = 0, t_gpu = 0, t_fin = 0, t_read_raw = 0, t_host_pack = 0;
    uint64_t total_bases = 0;
    size_t n_dev_groups = 0, n_host_groups = 0;
    d2g_sketcher *sk = nullptr;
    struct Ready { size_t g; d2g_seqpack *sp; int buf; std::vector<uint64_t> foff, flen, gfo; size_t raw_bytes; double tparse; };
    std::deque<Ready> ready;
    std::vector<d2g_seqpack *> pool;                                    // recycled packers (allocations kept)
    std::mutex mu;
    std::condition_variable cv_ready, cv_space, cv_buf;
    std::atomic<size_t> next_group{0};
    std::string parse_error;
    const bool force_host = std::getenv("D2G_DEVICE_PARSE") == nullptr || std::getenv("D2G_HOST_PARSE") != nullptr || job_devices(o).size() > 1;
    const size_t nparsers = std::max<size_t>(1, std::min<size_t>({size_t(o.workers()), groups.size(), size_t(192)}));
    // (A byte-bounded queue deep enough to parse all of 1000 x 5 Mbp before the first launch, with four device threads to drain it,
    // was measured: the 112 freshly allocated packers fault in 1.3 GB and the pipeline went 0.21 -> 0.34 s.  The recycled pool stays.)
    const size_t max_ready = 2 * nparsers + 2;
    size_t max_group = 16;
    for (size_t gb : group_bytes) max_group = std::max(max_group, gb);
    const size_t buf_bytes = (max_group + 4095) / 4096 * 4096 + 4096;
    const size_t nbufs = force_host ? 0 : std::min<size_t>(groups.size(), std::max<size_t>(3, std::min<size_t>(nparsers + 2, (size_t(1) << 30) / buf_bytes)));
    std::vector<uint8_t *> bufs(nbufs, nullptr);
    std::deque<int> free_bufs;
    for (size_t i = 0; i < nbufs; ++i) {
        void *p = nullptr;
        if (posix_memalign(&p, 4096, buf_bytes) != 0) die("out of memory (ingest staging)");
        bufs[i] = static_cast<uint8_t *>(p);
        free_bufs.push_back(int(i));
    }
    std::vector<std::thread> parsers;
    for (size_t t = 0; t < nparsers; ++t) parsers.emplace_back([&]() {
        for (;;) {
            const size_t g = next_group.fetch_add(1);
            if (g >= groups.size()) break;
            const double t0 = now();
            Ready r{g, nullptr, -1, {}, {}, {}, 0, 0.0};
            // eligible for the device parser: every file of the group is a plain file whose first byte is '>'
            bool dev = !force_host && nbufs > 0;
            for (size_t x = groups[g].first; dev && x < groups[g].second; ++x)
                for (const FileRef &fr : files_of[x]) {
                    if (!isfile(fr.path)) { dev = false; break; }         // missing / not a regular file (FIFO, ...): the host path reports or reads it
                    if (fr.size == 0) continue;
                    char c0 = 0;
                    std::FILE *fp = std::fopen(fr.path.c_str(), "rb");
                    if (!fp || std::fread(&c0, 1, 1, fp) != 1 || c0 != '>') dev = false;
                    if (fp) std::fclose(fp);
                }
            std::string bad;
            int rc = D2G_OK;
            if (dev) {
                // a staging buffer that is free RIGHT NOW, else this thread packs the group itself: while the GPU context is still
                // being created (or the device threads are behind) the host cores keep producing sketchable groups instead of waiting
                std::lock_guard<std::mutex> lk(mu);
                if (!free_bufs.empty()) { r.buf = free_bufs.front(); free_bufs.pop_front(); } else dev = false;
            }
            if (dev) {
                uint8_t *dst = bufs[r.buf];
                size_t pos = 0;
                r.gfo.push_back(0);
                for (size_t x = groups[g].first; rc == D2G_OK && x < groups[g].second; ++x) {
                    for (const FileRef &fr : files_of[x]) {
                        r.foff.push_back(pos); r.flen.push_back(fr.size);
                        if (fr.size) {
                            std::FILE *fp = std::fopen(fr.path.c_str(), "rb");
                            // a file that changed size since the stat goes to the host parser's error handling
                            if (!fp || pos + fr.size > buf_bytes || std::fread(dst + pos, 1, fr.size, fp) != fr.size) { rc = D2G_ERR_IO; bad = fr.path; }
                            if (fp) std::fclose(fp);
                        }
                        pos += (fr.size + 15) / 16 * 16;
                    }
                    r.gfo.push_back(r.foff.size());
                }
                r.raw_bytes = pos;
            } else {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!pool.empty()) { r.sp = pool.back(); pool.pop_back(); }
                }
                rc = r.sp ? int(D2G_OK) : d2g_seqpack_create(o.k, &r.sp);
                for (size_t x = groups[g].first; rc == D2G_OK && x < groups[g].second; ++x) {
                    rc = d2g_seqpack_add_path(r.sp, o.paths[todo[x]].c_str());
                    if (rc) bad = o.paths[todo[x]];
                }
                if (rc == D2G_OK) (void)d2g_seqpack_packed_bytes(r.sp);          // pad now, off the device thread
            }
            const double t_work = now() - t0;
            std::unique_lock<std::mutex> lk(mu);
            (r.buf >= 0 ? t_read_raw : t_host_pack) += t_work;
            if (rc) {
                if (parse_error.empty()) parse_error = "Failed to open " + bad;
                if (r.sp) d2g_seqpack_destroy(r.sp);
                r.sp = nullptr;
                if (r.buf >= 0) { free_bufs.push_back(r.buf); r.buf = -1; cv_buf.notify_one(); }
            }
            cv_space.wait(lk, [&] { return ready.size() < max_ready; });
            r.tparse = now() - t0;
            ready.push_back(std::move(r));
            cv_ready.notify_one();
        }
    });
    ctx = lctx.get();                                                    // the readers are busy: now wait for the GPU context
    double t_pin = now();
    for (size_t i = 0; i < nbufs; ++i) check(ctx, d2g_host_register(ctx, bufs[i], buf_bytes), "d2g_host_register");
    t_pin = now() - t_pin;
    // x87 finalisation (getcard / data, src/oph.h:240-263) and cache files leave the device threads through a small queue
    struct Fin { size_t g; std::vector<uint64_t> regs; std::vector<double> sigs, cards; };
    std::deque<Fin> finq;
    std::mutex fmu;
    std::condition_variable fcv;
    bool fin_closing = false;
    auto store_group = [&](size_t g, const double *sg, const double *cd) {
        const size_t b = groups[g].first, e = groups[g].second;
        for (size_t t = b; t < e; ++t) {
            const size_t i = todo[t];
            std::memcpy(&res.signatures[i * S], &sg[(t - b) * S], S * sizeof(double));   // fastxsketch.cpp:610
            res.cardinalities[i] = cd[t - b];
            if (o.cache) write_cached(res.destination_files[i], &sg[(t - b) * S], cd[t - b], S);
        }
    };
    std::thread finisher([&] {
        for (;;) {
            Fin f;
            {
                std::unique_lock<std::mutex> lk(fmu);
                fcv.wait(lk, [&] { return fin_closing || !finq.empty(); });
                if (finq.empty()) return;
                f = std::move(finq.front()); finq.pop_front();
            }
            const double t0 = now();
            const size_t n = groups[f.g].second - groups[f.g].first;
            if (!f.regs.empty()) {
                f.sigs.resize(n * S); f.cards.resize(n);
                check(nullptr, d2g_oph_finalize(f.regs.data(), n, m, S, f.sigs.data(), f.cards.data(), 2), "d2g_oph_finalize");
            }
            store_group(f.g, f.sigs.data(), f.cards.data());
            t_fin += now() - t0;                                            // (only this thread writes it)
        }
    });
    // Device threads: two per GPU, each with its own context + sketcher (a d2g_ctx is used by one thread at a time) -- the upload
    // of one group overlaps the kernels and the synchronisations of the other.  D2G_DEVICES names several GPUs: every GPU gets its
    // pair of threads and all of them take groups from the one queue (inputs dealt to the GPUs as they come free: file-sharded, no
    // collectives -- SURVEY 8e; the loop being sharded is the reference's `for` over files, src/fastxsketch.cpp:302); results land by
    // input index, so the stacked output is in input order whatever GPU sketched a group.
    const std::vector<int> devs = job_devices(o);
    int ndev = groups.size() > 1 ? 2 : 1;
    if (const char *e = std::getenv("D2G_DEVICE_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) ndev = v; }
    const size_t nthreads_dev = std::max<size_t>(1, std::min<size_t>(size_t(ndev) * devs.size(), std::max<size_t>(groups.size(), 1)));
    std::atomic<size_t> taken{0};
    std::mutex smu;
    struct KAcc { int launches = 0; double total_ms = 0; };
    std::vector<std::array<KAcc, 3>> kacc(devs.size());                    // per device: k0, k1, k3
    std::vector<size_t> groups_of(devs.size(), 0);
    auto device_loop = [&](d2g_ctx *dctx, size_t di) {
        if (!dctx) {                                                        // every thread but the first makes its own context, in parallel
            const int rc2 = d2g_ctx_create(devs[di], &dctx);
            if (rc2 != D2G_OK) die(std::string("d2g_ctx_create (device thread, GPU ") + std::to_string(devs[di]) + "): " + d2g_strerror(rc2));
            if (g_stats.on) (void)d2g_set_timing(dctx, TIME_ALL);
        }
        d2g_sketcher *dsk = nullptr;
        check(dctx, d2g_sketcher_create(dctx, &dsk), "d2g_sketcher_create");
        double gpu = 0; uint64_t bases = 0; size_t ndevg = 0, nhostg = 0; double tp = 0;
        for (;;) {
            if (taken.fetch_add(1) >= groups.size()) break;
            Ready r;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_ready.wait(lk, [&] { return !ready.empty(); });
                r = std::move(ready.front());
                ready.pop_front();
                cv_space.notify_one();
            }
            if (!r.sp && r.buf < 0) continue;                               // error recorded; drain the queue
            const size_t b = groups[r.g].first, e = groups[r.g].second, n = e - b;
            const double t1 = now();
            Fin f{r.g, {}, {}, {}};
            // the packed run stream of the group: parsed on the device (packed == NULL below), or by the host parser
            const uint8_t *packed = nullptr; size_t packed_bytes = 0, nrun = 0;
            const uint64_t *run_start = nullptr, *goff = nullptr; const uint32_t *run_len = nullptr;
            uint64_t nb = 0;
            if (r.buf >= 0) {
                const int rc = d2g_sketcher_ingest_fasta(dsk, bufs[r.buf], r.raw_bytes, r.foff.data(), r.flen.data(), r.foff.size(), r.gfo.data(), n, o.k);
                if (rc == D2G_OK) check(dctx, d2g_sketcher_ingested_runs(dsk, &run_start, &run_len, &nrun, &goff, nullptr, &nb), "d2g_sketcher_ingested_runs");
                { std::lock_guard<std::mutex> lk(mu); free_bufs.push_back(r.buf); }
                if (rc == D2G_ERR_UNSUPPORTED) {                            // e.g. a '+' line further down: the host parser takes the group
                    check(dctx, d2g_seqpack_create(o.k, &r.sp), "d2g_seqpack_create");
                    for (size_t x = b; x < e; ++x)
                        if (d2g_seqpack_add_path(r.sp, o.paths[todo[x]].c_str()) != D2G_OK) die("Failed to open " + o.paths[todo[x]]);
                } else check(dctx, rc, "d2g_sketcher_ingest_fasta");
                if (rc == D2G_OK) ++ndevg;
            }
            if (r.sp) {
                packed = d2g_seqpack_packed(r.sp); packed_bytes = d2g_seqpack_packed_bytes(r.sp);
                run_start = d2g_seqpack_run_start(r.sp); run_len = d2g_seqpack_run_len(r.sp); nrun = d2g_seqpack_nruns(r.sp);
                goff = d2g_seqpack_genome_run_off(r.sp); nb = d2g_seqpack_nbases(r.sp);
                ++nhostg;
            }
            if (o.sspace == SPACE_MULTISET) {
                // fastxsketch.cpp:425-445: Counter -> BagMinHash; cardinality = total weight, signature = data()[0..S)
                f.sigs.resize(n * S); f.cards.resize(n);
                check(dctx, d2g_sketcher_run_bmh(dsk, packed, packed_bytes, run_start, run_len, nrun, goff, n, o.k, o.canon, xormask, S,
                                                 double(o.count_threshold), f.sigs.data(), f.cards.data()), "d2g_sketcher_run_bmh");
            } else {
                f.regs.resize(n * m);
                check(dctx, d2g_sketcher_run(dsk, packed, packed_bytes, run_start, run_len, nrun, goff, n, o.k, o.canon, xormask, S, f.regs.data()),
                      "d2g_sketcher_run");
            }
            gpu += now() - t1; bases += nb; tp += r.tparse;
            { std::lock_guard<std::mutex> lk(fmu); finq.push_back(std::move(f)); }
            fcv.notify_one();
            if (r.sp) {
                d2g_seqpack_clear(r.sp);
                std::lock_guard<std::mutex> lk(mu);
                pool.push_back(r.sp);
            }
        }
        std::array<KAcc, 3> mine;
        if (g_stats.on) {
            const char *names[3] = {"k0", "k1", "k3"};
            for (int x = 0; x < 3; ++x) { int n = 0; float avg = 0, last = 0; if (d2g_kernel_ms(dctx, names[x], 1, &n, &avg, &last) == D2G_OK) { mine[x].launches = n; mine[x].total_ms = double(avg) * n; } }
        }
        if (g_release_at_exit) { d2g_sketcher_destroy(dsk); if (dctx != ctx) d2g_ctx_destroy(dctx); }
        std::lock_guard<std::mutex> lk(smu);
        t_gpu += gpu; total_bases += bases; n_dev_groups += ndevg; n_host_groups += nhostg; t_parse += tp;
        groups_of[di] += ndevg + nhostg;
        for (int x = 0; x < 3; ++x) { kacc[di][x].launches += mine[x].launches; kacc[di][x].total_ms += mine[x].total_ms; }
    };
    std::vector<std::thread> more;
    const double t_dev0 = now();
    for (size_t t = 1; t < nthreads_dev; ++t) more.emplace_back(device_loop, (d2g_ctx *)nullptr, t / size_t(ndev));
    const double t_dev1 = now();
    device_loop(ctx, 0);
    for (auto &th : more) th.join();
    const double t_dev2 = now();
    { std::lock_guard<std::mutex> lk(fmu); fin_closing = true; }
    fcv.notify_all();
    finisher.join();
    (void)sk;
    if (o.verbosity) std::fprintf(stderr, "[d2g] device side: %zu device threads over %zu GPU(s) (started in %.3fs), device loops %.3fs wall, drain of the finisher %.3fs\n", nthreads_dev,
                                  devs.size(), t_dev1 - t_dev0, t_dev2 - t_dev1, now() - t_dev2);
    for (auto &th : parsers) th.join();
    for (d2g_seqpack *p : pool) d2g_seqpack_destroy(p);
    const double t_pipe = now();
    // the staging buffers stay page-locked until the process ends (it leaves through _exit): unpinning 0.8 GB costs more than
    // the whole device work of a small job; D2G_FULL_TEARDOWN=1 releases them
    if (g_release_at_exit) for (size_t i = 0; i < nbufs; ++i) { (void)d2g_host_unregister(ctx, bufs[i]); std::free(bufs[i]); }
    if (!parse_error.empty()) die(parse_error);
    if (o.verbosity) std::fprintf(stderr, "[d2g] sketched %zu inputs (%" PRIu64 " bases in the packed streams) in %zu groups (%zu parsed on the device, %zu by the host "
                                          "parser): reader threads %.3fs in all (%.3fs reading raw groups, %.3fs reading + packing, the rest waiting for queue space) over %zu threads, device threads: H2D+K0+K1+D2H %.3fs busy in all, finisher thread: x87 finalise+cache %.3fs; "
                                          "%zu staging buffers of %zu MiB page-locked in %.3fs\n",
                                  todo.size(), total_bases, groups.size(), n_dev_groups, n_host_groups, t_parse, t_read_raw, t_host_pack, nparsers, t_gpu, t_fin,
                                  nbufs, buf_bytes >> 20, t_pin);
    if (o.verbosity) std::fprintf(stderr, "[d2g] sketch wall: setup (stat, cache probe) %.3fs, ingest pipeline %.3fs\n", t_setup - t_enter, t_pipe - t_setup);
    if (g_stats.on) {
        std::string dj = "[";
        const char *names[3] = {"k0", "k1", "k3"};
        for (size_t d = 0; d < devs.size(); ++d) {
            dj += std::string(d ? ", " : "") + "{\"index\": " + std::to_string(devs[d]) + ", \"name\": " + Stats::esc(device_label(devs[d])) + ", \"groups\": " + std::to_string(groups_of[d]);
            for (int x = 0; x < 3; ++x)
                dj += std::string(", \"") + names[x] + "\": {\"launches\": " + std::to_string(kacc[d][x].launches) + ", \"total_ms\": " + Stats::numstr(kacc[d][x].total_ms) + "}";
            dj += "}";
        }
        dj += "]";
        const size_t mm = d2g_oph_m(S);
        // ceil(L/4) + 8 m per input (set sketches), + 8 for the total weight of a multiset sketch
        const double alg = double((total_bases + 3) / 4) + double(todo.size()) * (8.0 * double(o.sspace == SPACE_MULTISET ? S : mm) + (o.sspace == SPACE_MULTISET ? 8.0 : 0.0));
        g_stats.raw("sketch", std::string("{\"inputs\": ") + std::to_string(N) + ", \"sketched\": " + std::to_string(todo.size()) + ", \"from_cache\": " + std::to_string(N - todo.size()) +
                    ", \"k\": " + std::to_string(o.k) + ", \"sketchsize\": " + std::to_string(S) + ", \"space\": " + (o.sspace == SPACE_MULTISET ? "\"multiset (K3: counts + BagMinHash)\"" : "\"set (K1: OPH)\"") +
                    ", \"groups\": " + std::to_string(groups.size()) + ", \"groups_parsed_on_device\": " + std::to_string(n_dev_groups) + ", \"bases\": " + std::to_string(total_bases) +
                    ", \"algorithmic_bytes\": " + Stats::numstr(alg) + ", \"device_threads\": " + std::to_string(nthreads_dev) + ", \"parser_threads\": " + std::to_string(nparsers) +
                    ", \"devices\": " + dj +
                    ", \"wall_s\": {\"setup\": " + Stats::numstr(t_setup - t_enter) + ", \"ingest_pipeline\": " + Stats::numstr(t_pipe - t_setup) + ", \"device_loops\": " + Stats::numstr(t_dev2 - t_dev1) +
                    ", \"device_threads_busy_sum\": " + Stats::numstr(t_gpu) + ", \"parser_threads_sum\": " + Stats::numstr(t_parse) + ", \"finisher_x87_and_cache\": " + Stats::numstr(t_fin) + "}}");
    }
    write_stacked(res, o);
}

// --parse-by-seq (sketch_core.cpp:23-29 -> fastxsketchbyseq.cpp:102-268,270-531): one sketch per record of
// ONE input file; OPH set sketches (cardinality = exact distinct k-mer count when the estimate is below
// 10 S, lines 415-430) or multiset sketches; names are the record names.
void sketch_core_byseq(Result &res, const Options &o, LazyCtx &lctx) {
    if (o.paths.size() != 1)
        die("parse-by-seq currently only handles one file at a time. To process multiple files, simply concatenate them into one file, and run dashing2 on that.");
    const size_t S = o.sketchsize, m = d2g_oph_m(S);
    const uint64_t xormask = d2g_seed_mask(o.seedseed);
    d2g_seqpack *sp = nullptr;
    check(nullptr, d2g_seqpack_create(o.k, &sp), "d2g_seqpack_create");
    if (d2g_seqpack_add_path_by_record(sp, o.paths[0].c_str()) != D2G_OK) die("Failed to read from " + o.paths[0]);
    d2g_ctx *ctx = lctx.get();
    const size_t N = d2g_seqpack_ngenomes(sp);
    res.names.resize(N);
    for (size_t i = 0; i < N; ++i) res.names[i] = d2g_seqpack_name(sp, i);
    res.destination_files.assign(N, std::string());
    res.cardinalities.assign(N, 0.);
    res.signatures.assign(N * S, 0.);
    const size_t total_bytes = d2g_seqpack_packed_bytes(sp);
    const uint8_t *packed = d2g_seqpack_packed(sp);
    const uint64_t *run_start = d2g_seqpack_run_start(sp), *goff = d2g_seqpack_genome_run_off(sp);
    const uint32_t *run_len = d2g_seqpack_run_len(sp);
    d2g_sketcher *sk = nullptr;
    check(ctx, d2g_sketcher_create(ctx, &sk), "d2g_sketcher_create");
    std::vector<uint64_t> regs, rs_rel, goff_rel, ndist;
    std::vector<double> sigs, cards;
    const size_t max_rec = std::max<size_t>(1, (size_t(64) << 20) / m);            // <= 512 MiB of registers per launch
    for (size_t g0 = 0; g0 < N;) {
        // batch [g0, g1): bounded by records and by packed bytes (the slice is re-based so that only it is uploaded)
        size_t g1 = g0;
        const uint64_t r0 = goff[g0];
        const uint64_t base0 = r0 < goff[N] ? (run_start[r0] & ~uint64_t(15)) : 0;
        while (g1 < N && g1 - g0 < max_rec) {
            const uint64_t r1 = goff[g1 + 1];
            const uint64_t endb = r1 > r0 ? run_start[r1 - 1] + run_len[r1 - 1] : base0;
            if (g1 > g0 && endb - base0 > (uint64_t(192) << 20)) break;            // ~48 MB of packed bases
            ++g1;
        }
        const size_t n = g1 - g0, r1 = goff[g1], nrun = r1 - r0;
        rs_rel.resize(nrun); goff_rel.resize(n + 1);
        for (size_t r = 0; r < nrun; ++r) rs_rel[r] = run_start[r0 + r] - base0;
        for (size_t g = 0; g <= n; ++g) goff_rel[g] = goff[g0 + g] - r0;
        const uint8_t *pk = packed + base0 / 4;
        const uint64_t endb = nrun ? run_start[r1 - 1] + run_len[r1 - 1] : base0;
        const size_t pk_bytes = std::min<size_t>(total_bytes - base0 / 4, (endb - base0 + 3) / 4 + 64);   // slice + its 64 readable pad bytes
        sigs.resize(n * S); cards.resize(n);
        if (o.sspace == SPACE_MULTISET) {
            check(ctx, d2g_sketcher_run_bmh(sk, pk, pk_bytes, rs_rel.data(), run_len + r0, nrun, goff_rel.data(), n, o.k, o.canon,
                                            xormask, S, double(o.count_threshold), sigs.data(), cards.data()), "d2g_sketcher_run_bmh");
        } else {
            regs.resize(n * m);
            check(ctx, d2g_sketcher_run(sk, pk, pk_bytes, rs_rel.data(), run_len + r0, nrun, goff_rel.data(), n, o.k, o.canon,
                                        xormask, S, regs.data()), "d2g_sketcher_run");
            check(ctx, d2g_oph_finalize(regs.data(), n, m, S, sigs.data(), cards.data(), int(o.workers())), "d2g_oph_finalize");
            bool need = false;
            for (size_t i = 0; i < n; ++i) {
                if (std::isnan(cards[i])) cards[i] = 0.;                              // fastxsketchbyseq.cpp:410-414
                need |= cards[i] < 10. * double(S);
            }
            if (need) {                                                               // lines 415-430: exact distinct count
                ndist.resize(n);
                check(ctx, d2g_sketcher_run_distinct(sk, pk, pk_bytes, rs_rel.data(), run_len + r0, nrun, goff_rel.data(), n, o.k,
                                                     o.canon, xormask, ndist.data()), "d2g_sketcher_run_distinct");
                for (size_t i = 0; i < n; ++i) if (cards[i] < 10. * double(S)) cards[i] = double(ndist[i]);
            }
        }
        std::memcpy(&res.signatures[g0 * S], sigs.data(), n * S * sizeof(double));
        std::memcpy(&res.cardinalities[g0], cards.data(), n * sizeof(double));
        g0 = g1;
    }
    d2g_sketcher_destroy(sk);
    d2g_seqpack_destroy(sp);
    write_stacked(res, o);
}

// ------------------------------------------------------------------------------------ cmp: load
void load_results(Options &o, Result &res) {               // src/cmp_main.cpp:24-198
    const auto &paths = o.paths;
    if (paths.empty()) die("No paths provided to --presketched");
    const std::string &pf = paths.front();
    if (paths.size() == 1) {
        const std::string namesf = pf + ".names.txt";
        if (isfile(namesf)) {
            std::ifstream ifs(namesf);
            for (std::string l; std::getline(ifs, l);) {
                if (l.empty() || l.front() == '#') continue;
                res.names.emplace_back(l.substr(0, l.find_first_of('\t')));
            }
        }
        std::FILE *fp = std::fopen(pf.c_str(), "rb");
        if (!fp) die("Failed to open " + pf);
        uint64_t hdr[2];
        if (filesize(pf) < 16 || std::fread(hdr, 8, 2, fp) != 2)
            die("Failed to read num_entities from file " + pf + " of size " + std::to_string(filesize(pf)));
        const size_t N = hdr[0], S = hdr[1];
        o.sketchsize = S;
        if (res.names.empty()) for (size_t i = 0; i < N; ++i) res.names.push_back(std::to_string(i));
        res.cardinalities.resize(N);
        if (std::fread(res.cardinalities.data(), 8, N, fp) != N) die("Failed to read cardinalities from disk");
        const size_t nreg = (filesize(pf) - (N + 2) * 8) / 8;
        // (a private mapping of the file instead of this copy was measured: the page faults it moves into densify and the
        // upload cost more than the read -- config 4: 1.26 -> 1.43 s)
        res.signatures.resize(nreg);
        if (std::fread(res.signatures.data(), 8, nreg, fp) != nreg) die("Failed to read signatures from disk");
        std::fclose(fp);
        if (nreg != N * S) die("stacked sketch file " + pf + " has " + std::to_string(nreg) + " registers, expected " + std::to_string(N * S));
    } else {
        const size_t N = paths.size();
        std::vector<size_t> fs(N);
        for (size_t i = 0; i < N; ++i) {
            if (!isfile(paths[i])) { std::fprintf(stderr, "File does not exist at %s/%zu\n", paths[i].c_str(), i); std::exit(EXIT_FAILURE); }
            fs[i] = (filesize(paths[i]) - 8) / 8;
        }
        if (!std::all_of(fs.begin(), fs.end(), [&](size_t x) { return x == fs[0]; }))
            die("presketched files have uneven sizes; only sketches (not k-mer sets) are in this build's scope");
        o.sketchsize = fs[0];
        std::fprintf(stderr, "Sketchsize is now %zd\n", o.sketchsize);
        res.signatures.resize(N * fs[0]);
        res.cardinalities.resize(N);
        for (size_t i = 0; i < N; ++i) {
            std::FILE *fp = std::fopen(paths[i].c_str(), "rb");
            if (!fp || std::fread(&res.cardinalities[i], 8, 1, fp) != 1 ||
                std::fread(&res.signatures[i * fs[0]], 8, fs[0], fp) != fs[0]) {
                std::fprintf(stderr, "Failed to read at path %s\n", paths[i].c_str());
                std::exit(1);
            }
            std::fclose(fp);
        }
        // the reference leaves names_ empty here, so rows/sources are printed as E<i> (emitrect.cpp:145,176)
        for (size_t i = 0; i < N; ++i) res.names.push_back("E" + std::to_string(i));
    }
}

// ------------------------------------------------------------------------------------ cmp: emit
struct Emitter {
    const Options &o;
    const Result &res;
    std::FILE *fp = nullptr;
    bool own = false;
    // Binary matrices leave through ONE buffered stream.  Measured and dropped: (round 3) a shared mapping of the output filled by
    // all threads -- copy 0.74 -> 0.57 s for config 4's 5 GB, but unmapping the dirty pages cost another 0.51 s on the box's overlay
    // file system; (round 4) the batch cut into 4 MiB pieces that 8 threads pwrite() at their offsets -- no gain at all (config 3:
    // 22 ms either way, config 4: 0.50 s either way, ~10 GB/s): buffered writes to one file serialise on its inode lock.
    Emitter(const Options &oo, const Result &r) : o(oo), res(r) {
        const std::string outp = (o.cmpout.empty() || o.cmpout.front() == '-') ? "/dev/stdout" : o.cmpout;   // emitrect.cpp:114-115
        if (outp == "/dev/stdout") fp = stdout;
        else { fp = std::fopen(outp.c_str(), "wb"); own = true; }
        if (!fp) die("Failed to open path " + outp + " for writing");
        static std::vector<char> buf(1 << 22);
        std::setvbuf(fp, buf.data(), _IOFBF, buf.size());
    }
    ~Emitter() { if (fp) { std::fflush(fp); if (own) std::fclose(fp); } }
    void header() {                                                     // emitrect.cpp:136-151
        if (o.of != HUMAN_READABLE) return;
        const size_t ns = res.names.size();
        if (o.ok == PHYLIP) { std::fprintf(fp, "%zu\n", ns); return; }
        const char *label = o.ok == ASYMMETRIC_ALL_PAIRS ? "Asymmetric pairwise" : o.ok == PANEL ? "Panel (Query/Refernce)" : "Symmetric pairwise";
        std::fprintf(fp, "#Dashing2 %s Output\n", label);
        std::fprintf(fp, "#Dashing2Options: %s\n", o.to_string().c_str());
        // not a reference line: written only when --fmt-compat was given, so that the default output stays byte-identical to
        // emitrect.cpp:138-147 while a deliberate choice of float layout is on record in the file it shaped
        if (o.fmt_compat) std::fprintf(fp, "#Dashing2FloatText: fmt-compat=%d\n", o.fmt_compat);
        std::fputs("#Sources", fp);
        for (size_t i = 0; i < ns; ++i) { std::fputc('\t', fp); std::fwrite(res.names[i].data(), 1, res.names[i].size(), fp); }
        std::fputc('\n', fp);
    }
    // rows [r0, r1); row i has nvals(i) values starting at data + off(i)
    template <class NV> void rows(size_t r0, size_t r1, const float *data, NV nvals) {
        if (o.of == MACHINE_READABLE) {                                  // emitrect.cpp:189-192
            size_t tot = 0;
            for (size_t i = r0; i < r1; ++i) tot += nvals(i);
            if (std::fwrite(data, sizeof(float), tot, fp) != tot) die("Failed to write rows " + std::to_string(r0) + "-" + std::to_string(r1) + " to disk");
            return;
        }
        const size_t n = r1 - r0;
        std::vector<size_t> off(n + 1, 0);
        for (size_t i = 0; i < n; ++i) off[i + 1] = off[i] + nvals(r0 + i);
        std::vector<std::string> text(n);
#ifdef _OPENMP
        #pragma omp parallel for schedule(dynamic, 8) num_threads(o.workers())
#endif
        for (size_t r = 0; r < n; ++r) {                                 // emitrect.cpp:172-187
            const size_t i = r0 + r;
            std::string &s = text[r];
            std::string fn = (res.names.size() > i && !res.names[i].empty()) ? res.names[i] : std::string("E") + std::to_string(i);
            if (fn.size() < 9) fn.append(9 - fn.size(), ' ');
            const size_t nv = off[r + 1] - off[r];
            s.reserve(fn.size() + 2 * (i + 1) + nv * 12 + 2);
            s = fn;
            if (o.ok == SYMMETRIC_ALL_PAIRS) for (size_t t = 0; t < i + 1; ++t) s += "\t-";     // print_tabs, emitrect.cpp:40-66
            char buf[FMT_MAX_FLOAT_CHARS + 1];
            const float *p = data + off[r];
            for (size_t j = 0; j < nv; ++j) {
                buf[0] = '\t';
                const size_t l = format_float(p[j], buf + 1);
                s.append(buf, l + 1);
            }
            s += '\n';
        }
        for (const auto &s : text)
            if (std::fwrite(s.data(), 1, s.size(), fp) != s.size()) die("Failed to write text rows");
    }
};

struct DevBuf {
    d2g_ctx *ctx; void *p = nullptr;
    DevBuf(d2g_ctx *c, size_t n) : ctx(c) { check(c, d2g_malloc(c, n ? n : 4, &p), "d2g_malloc"); }
    ~DevBuf() { if (g_release_at_exit) d2g_free(ctx, p); }
};
// Host staging of a row batch: PLAIN page-aligned memory.  Rounds 2-3 page-locked these slots (hipHostMalloc); measured on MI355X /
// ROCm 7 (tools/cmp_setup_time2.py): a pageable D2H of 16 MiB takes 0.32 ms (1.1 ms the first time a buffer is touched) -- the
// same 50 GB/s as from page-locked memory -- while page-locking costs 0.2-0.28 ms per MiB (3 x 64 MiB = 56 ms) AND serialises
// with the operand upload inside the runtime (the upload of config 3 took 86 ms next to it instead of 30).
struct HostBuf {
    void *p = nullptr;
    explicit HostBuf(size_t n) { if (posix_memalign(&p, 4096, std::max<size_t>(n, 4096)) != 0) die("out of memory (row-batch staging)"); }
    ~HostBuf() { if (g_release_at_exit) std::free(p); }
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    template <class T> T *as() { return static_cast<T *>(p); }
};

// Row batches flow device -> pinned slot -> emitter thread: the kernel + D2H (+ host x87 epilogue) of batch i+1 run while
// batch i is formatted / written (the CLI ran kernel -> D2H -> emit strictly in series per 512 MiB batch).
// The queue itself is slot_queue.h (exercised under ThreadSanitizer by `make tsan`).
struct EmitJob { size_t r0, r1; const float *data; std::function<size_t(size_t)> nvals; };
struct EmitQueue : SlotQueue<EmitJob> {
    EmitQueue(Emitter &em, int nslots) : SlotQueue<EmitJob>(nslots, [&em](const EmitJob &j) { em.rows(j.r0, j.r1, j.data, j.nvals); }) {}
    void submit_rows(int slot, size_t r0, size_t r1, const float *data, std::function<size_t(size_t)> nvals) {
        submit(slot, EmitJob{r0, r1, data, std::move(nvals)});
    }
};

// values per row batch: a slot is one batch's staging; three slots cycle device -> host epilogue -> emitter.  Small jobs take small
// slots (more batches cost little: the pair kernel is launched per row range), big jobs 64 MiB ones.
size_t cmp_slot_values(size_t total_vals) {
    size_t v = std::min<size_t>(size_t(1) << 24, std::max<size_t>(size_t(1) << 22, total_vals / 12));
    if (const char *e = std::getenv("D2G_CMP_SLOT_VALUES")) { const long long x = std::atoll(e); if (x >= 1) v = size_t(x); }   // tests: tiny slots
    return v;
}

// the shape of a dense comparison job (src/emitrect.cpp:211-323): which rows are emitted, how many values each has
struct CmpShape {
    bool symmetric; size_t ns, nrows, c0, c1, ncol, total_vals, widest;
    CmpShape(const Options &o, const Result &res) {
        ns = res.names.size();
        symmetric = o.ok == SYMMETRIC_ALL_PAIRS || o.ok == PHYLIP;
        const size_t nq = res.nq, nf = o.ok == PANEL ? ns - nq : ns;
        c0 = o.ok == PANEL ? nf : 0; c1 = ns; ncol = symmetric ? 0 : c1 - c0;
        nrows = symmetric ? ns : nf;
        total_vals = symmetric ? ns * (ns - 1) / 2 : nf * ncol;
        widest = symmetric ? (ns ? ns - 1 : 0) : ncol;
    }
    // the batch of rows starting at r0 that fits `cap` values -> (r1, values)
    std::pair<size_t, size_t> batch(size_t r0, size_t cap) const {
        size_t r1 = r0, cnt = 0;
        if (symmetric) while (r1 < ns && (cnt == 0 || cnt + (ns - 1 - r1) <= cap)) { cnt += ns - 1 - r1; ++r1; }
        else { r1 = std::min(nrows, r0 + std::max<size_t>(1, cap / std::max<size_t>(ncol, 1))); cnt = (r1 - r0) * ncol; }
        return {r1, cnt};
    }
};

// counts of a rectangular batch -> floats (emitrect.cpp:211-268 call compare(i, j) per cell: cmp_core.cpp:458-517)
void rect_epilogue(const Options &o, const CmpShape &sh, size_t r0, size_t r1, const uint32_t *ca, const uint32_t *cb, const double *cards, size_t S,
                   bool have_lut, const std::vector<float> &lut, bool multiset, bool need_gtlt, float *out) {
#ifdef _OPENMP
    #pragma omp parallel for schedule(dynamic, 4) num_threads(o.workers())
#endif
    for (size_t i = r0; i < r1; ++i)
        for (size_t j = sh.c0; j < sh.c1; ++j) {
            const size_t p = (i - r0) * sh.ncol + (j - sh.c0);
            out[p] = have_lut ? lut[ca[p]]
                   : multiset ? d2g_epilogue_neq(ca[p], S, cards[i], cards[j], o.measure, o.k)
                   : need_gtlt ? d2g_epilogue_gtlt(ca[p], cb[p], S, cards[i], cards[j], o.measure, o.k)
                               : d2g_epilogue_gtlt(S - ca[p], 0, S, cards[i], cards[j], o.measure, o.k);
        }
}

// the D2G_* switches the context resolved when it was created (include/d2g.h: d2g_ctx_tuning)
std::string tuning_json(d2g_ctx *ctx) {
    const int n = d2g_ctx_tuning(ctx, nullptr, 0);
    if (n < 0) return "null";
    std::string buf((size_t)n + 1, '\0');
    d2g_ctx_tuning(ctx, &buf[0], buf.size());
    buf.resize((size_t)n);
    return buf;
}

// what the sparse-tile path did on the last upper-triangle launch of the set (include/d2g.h: d2g_cmp_set_sparse_info)
std::string sparse_json(d2g_ctx *ctx, const d2g_cmp_set *set) {
    uint32_t i4[4] = {0, 0, 0, 0};
    if (d2g_cmp_set_sparse_info(ctx, set, nullptr, i4) != D2G_OK) return "null";
    return std::string("{\"sorted_operand\": ") + (i4[0] ? "true" : "false") + ", \"tiles_listed_last_launch\": " + std::to_string(i4[1]) + ", \"dense_decided_by_prepare\": " +
           ((i4[2] & 1) ? "true" : "false") + ", \"dense_kernel_ran\": " +  ((i4[2] & 2) ? "true" : "false") + ", \"tiles_and_pair_list\": " + ((i4[2] & 4) ? "true" : "false") + ", \"callers_order_kept\": " + ((i4[2] & 8) ? "true" : "false") +
           ", \"pairs_listed\": " + std::to_string(i4[3]) + "}";
}

std::string planes_json(d2g_ctx *ctx, const d2g_cmp_set *set) {
    unsigned md = 0; int nb = 0; float mean = 0;
    if (d2g_cmp_set_planes(ctx, set, nullptr, &md, &nb, &mean) != D2G_OK) return "null";
    return "{\"max\": " + std::to_string(nb) + ", \"mean\": " + Stats::numstr(mean) + ", \"max_shared_values_per_column_plus1\": " + std::to_string(md) + "}";
}

// A dense comparison job over several GPUs from ONE process (the reference's seam is the single call
// emit_rectangular(opts, result), src/cmp_core.cpp:746-751).  Every GPU gets a contiguous block of rows of the signature matrix; one
// exchange (d2g_allpairs_prepare_all: all-to-all of column slices, sharded prepare, all-gather of the bit planes over RCCL/xGMI)
// leaves the whole bit-sliced operand on every GPU; row batches -- of the condensed triangle (emitrect.cpp:290-323), of the square
// matrix (--square, :249-268) or of the reference x query panel (-Q, :211-247) -- are then dealt to the GPUs round-robin, computed
// concurrently, and emitted in row order through the slot queue: byte-identical to the single-GPU output.
// Returns false (nothing emitted yet) when the sharded bit-sliced prepare overflowed its rank table on some rank -- an adversarial /
// extremely skewed register column at N > 21 845: the caller then takes the single-GPU path, whose AUTO algorithm falls back to the
// direct kernel.
bool cmp_core_multi(const Options &o, Result &res, const std::vector<int> &devs, bool have_lut, const std::vector<float> &lut,
                    bool multiset) {
    const CmpShape sh(o, res);
    const size_t ns = sh.ns, S = o.sketchsize;
    const int W = int(devs.size());
    const uint64_t *bits = reinterpret_cast<const uint64_t *>(res.sigs());
    const double *cards = res.cardinalities.data();
    const double t0 = now();
    std::vector<d2g_ctx *> ctxs(W, nullptr);
    for (int r = 0; r < W; ++r) {
        const int rc = d2g_ctx_create(devs[r], &ctxs[r]);
        if (rc != D2G_OK) die(std::string("D2G_DEVICES: d2g_ctx_create(") + std::to_string(devs[r]) + "): " + d2g_strerror(rc));
        if (g_stats.on) (void)d2g_set_timing(ctxs[r], TIME_ALL);
    }
    std::vector<d2g_comm *> comms(W, nullptr);
    check(ctxs[0], d2g_comm_create_all(ctxs.data(), W, comms.data()), "d2g_comm_create_all");
    std::vector<d2g_allpairs *> engs(W, nullptr);
    std::vector<const uint64_t *> rows(W, nullptr);
    std::vector<void *> rowbuf(W, nullptr);
    for (int r = 0; r < W; ++r) {
        check(ctxs[r], d2g_allpairs_create(ctxs[r], comms[r], ns, S, &engs[r]), "d2g_allpairs_create");
        size_t lo = 0, hi = 0;
        d2g_allpairs_rows_held(engs[r], &lo, &hi);
        check(ctxs[r], d2g_malloc(ctxs[r], std::max<size_t>((hi - lo) * S, 1) * 8, &rowbuf[r]), "d2g_malloc");
        if (hi > lo) check(ctxs[r], d2g_memcpy_h2d(ctxs[r], rowbuf[r], bits + lo * S, (hi - lo) * S * 8, nullptr), "h2d rows");
        rows[r] = static_cast<const uint64_t *>(rowbuf[r]);
    }
    check(ctxs[0], d2g_allpairs_prepare_all(engs.data(), W, rows.data(), nullptr), "d2g_allpairs_prepare_all");
    bool overflow = false;
    for (int r = 0; r < W; ++r) {
        const int st = d2g_allpairs_status(engs[r], nullptr);      // every rank sees every rank's status word
        if (st == D2G_ERR_INTERNAL) overflow = true;
        else check(ctxs[r], st, "d2g_allpairs_status");
        check(ctxs[r], d2g_free(ctxs[r], rowbuf[r]), "d2g_free");
    }
    if (overflow) {
        std::fprintf(stderr, "[d2g] multi-GPU bit-sliced prepare overflowed on a skewed register column: falling back to one GPU\n");
        for (int r = 0; r < W; ++r) { d2g_allpairs_destroy(engs[r]); d2g_comm_destroy(comms[r]); d2g_ctx_destroy(ctxs[r]); }
        return false;
    }
    const double t_prep = now() - t0;
    Emitter em(o, res);
    em.header();
    const size_t cap = std::max<size_t>(1, std::min(std::max(cmp_slot_values(sh.total_vals), sh.widest), std::max<size_t>(sh.total_vals, 1)));
    const bool fused = have_lut && sh.symmetric;                    // device floats (table epilogue inside the pair kernel)
    std::vector<std::unique_ptr<DevBuf>> da(W), dlut(W);
    for (int r = 0; r < W; ++r) {
        da[r].reset(new DevBuf(ctxs[r], cap * 4));
        dlut[r].reset(new DevBuf(ctxs[r], (S + 1) * sizeof(float)));
        if (have_lut) check(ctxs[r], d2g_memcpy_h2d(ctxs[r], dlut[r]->p, lut.data(), (S + 1) * sizeof(float), nullptr), "h2d lut");
    }
    const int NSLOT = 2 * W + 1;
    std::vector<std::unique_ptr<HostBuf>> hout(NSLOT), hca(NSLOT);
    for (int i = 0; i < NSLOT; ++i) { hout[i].reset(new HostBuf(cap * 4)); hca[i].reset(new HostBuf(fused ? 4 : cap * 4)); }
    double t_dev = 0;
    const double t_loop = now();
    struct Batch { size_t r0, r1, cnt; };
    size_t nbatches = 0;
    double emit_busy = 0;
    {
        EmitQueue eq(em, NSLOT);
        for (size_t next = 0; next < sh.nrows;) {
            // one round: up to W consecutive row batches, one per GPU, launched back to back (asynchronous) ...
            std::vector<Batch> round;
            const double ta = now();
            for (int r = 0; r < W && next < sh.nrows; ++r) {
                const auto b = sh.batch(next, cap);
                round.push_back({next, b.first, b.second});
                if (b.second) {
                    const d2g_cmp_set *set = d2g_allpairs_operand(engs[r]);
                    if (!sh.symmetric) check(ctxs[r], d2g_cmp_eqcount_rect_dev(ctxs[r], set, next, b.first, sh.c0, sh.c1, (uint32_t *)da[r]->p, nullptr), "d2g_cmp_eqcount_rect_dev");
                    else if (have_lut) check(ctxs[r], d2g_cmp_lut_ut_dev(ctxs[r], set, next, b.first, (const float *)dlut[r]->p, (float *)da[r]->p, nullptr), "d2g_cmp_lut_ut_dev");
                    else check(ctxs[r], d2g_cmp_eqcount_ut_dev(ctxs[r], set, next, b.first, (uint32_t *)da[r]->p, nullptr), "d2g_cmp_eqcount_ut_dev");
                }
                next = b.first;
            }
            // ... drained in row order into free slots; the emitter thread writes slot i while the next ones are copied / finished
            for (size_t b = 0; b < round.size(); ++b) {
                const Batch &bt = round[b];
                const int si = eq.acquire();
                float *out = hout[si]->as<float>();
                if (bt.cnt) {
                    if (fused) check(ctxs[b], d2g_memcpy_d2h(ctxs[b], out, da[b]->p, bt.cnt * 4, nullptr), "d2h");
                    else {
                        uint32_t *ca = hca[si]->as<uint32_t>();
                        check(ctxs[b], d2g_memcpy_d2h(ctxs[b], ca, da[b]->p, bt.cnt * 4, nullptr), "d2h");
                        if (sh.symmetric) check(ctxs[b], d2g_epilogue_ut(ca, nullptr, cards, ns, S, bt.r0, bt.r1, o.measure, o.k, multiset, int(o.workers()), out), "d2g_epilogue_ut");
                        else rect_epilogue(o, sh, bt.r0, bt.r1, ca, nullptr, cards, S, have_lut, lut, multiset, false, out);
                    }
                }
                if (sh.symmetric) eq.submit_rows(si, bt.r0, bt.r1, out, [ns](size_t i) { return ns - 1 - i; });
                else { const size_t ncol = sh.ncol; eq.submit_rows(si, bt.r0, bt.r1, out, [ncol](size_t) { return ncol; }); }
                ++nbatches;
            }
            t_dev += now() - ta;
        }
        eq.finish();
        emit_busy = eq.t_busy;
    }
    if (o.verbosity) std::fprintf(stderr, "[d2g] cmp on %d GPUs (%s): %zu sketches x S=%zu: upload+exchange+prepare %.3fs, %zu batches %.3fs wall (device+D2H+epilogue %.3fs busy, emit %.3fs busy, overlapped)\n",
                                  W, d2g_comm_is_rccl(comms[0]) ? "RCCL" : "loopback", ns, S, t_prep, nbatches, now() - t_loop, t_dev, emit_busy);
    if (g_stats.on) {
        std::string dj = "[";
        for (int r = 0; r < W; ++r)
            dj += std::string(r ? ", " : "") + "{\"index\": " + std::to_string(devs[r]) + ", \"name\": " + Stats::esc(device_label(devs[r])) +
                  ", \"k2\": " + Stats::kernel_json(ctxs[r], "k2") + ", \"k2prep\": " + Stats::kernel_json(ctxs[r], "k2prep") + "}";
        dj += "]";
        g_stats.raw("cmp", std::string("{\"sketches\": ") + std::to_string(ns) + ", \"sketchsize\": " + std::to_string(S) + ", \"values\": " + std::to_string(sh.total_vals) +
                    ", \"shape\": " + (sh.symmetric ? "\"upper triangle\"" : o.ok == PANEL ? "\"panel\"" : "\"square\"") + ", \"algo\": \"bitslice\", \"transport\": " +
                    (d2g_comm_is_rccl(comms[0]) ? "\"RCCL\"" : "\"loopback\"") + ", \"exchange_chunks\": " + std::to_string(d2g_allpairs_chunks(engs[0])) +
                    ", \"bit_planes\": " + planes_json(ctxs[0], d2g_allpairs_operand(engs[0])) +
                    ", \"algorithmic_bytes\": " + Stats::numstr(8.0 * double(S) * double(ns) + 4.0 * double(sh.total_vals)) + ", \"batches\": " + std::to_string(nbatches) +
                    ", \"slot_values\": " + std::to_string(cap) + ", \"devices\": " + dj +
                    ", \"wall_s\": {\"upload_exchange_prepare\": " + Stats::numstr(t_prep) + ", \"batches\": " + Stats::numstr(now() - t_loop) + ", \"device_d2h_epilogue_busy\": " +
                    Stats::numstr(t_dev) + ", \"emit_busy\": " + Stats::numstr(emit_busy) + "}}");
    }
    da.clear(); dlut.clear();
    for (int r = 0; r < W; ++r) { d2g_allpairs_destroy(engs[r]); d2g_comm_destroy(comms[r]); d2g_ctx_destroy(ctxs[r]); }
    return true;
}

void cmp_core(const Options &o, Result &res, d2g_ctx *ctx) {      // src/cmp_core.cpp:615-751 (dense outputs)
    const size_t ns = res.names.size(), S = o.sketchsize;
    if (res.nsigs() != ns * S) die("Empty signatures; trying to compare but don't have any");
    const bool multiset = o.sspace != SPACE_SET;
    double t_densify = 0;
    if (o.kmer_result == ONE_PERM) {                                // cmp_core.cpp:686-718
        size_t nfilled = 0;
        const double td = now();
        check(ctx, d2g_densify(res.sigs(), ns, S, &nfilled, int(o.workers())), "d2g_densify");
        t_densify = now() - td;
        if (o.verbosity) std::fprintf(stderr, "[d2g] densify scan %.3fs\n", t_densify);
        if (o.verbosity && nfilled) std::fprintf(stderr, "Densified a total of %zu/%zu entries\n", nfilled, S * ns);
    }
    const uint64_t *bits = reinterpret_cast<const uint64_t *>(res.sigs());
    const double *cards = res.cardinalities.data();
    std::vector<float> lut(S + 1);
    const bool have_lut = d2g_epilogue_lut(S, o.measure, o.k, multiset, lut.data()) == D2G_OK;
    const bool need_gtlt = !multiset && (S & (S - 1)) != 0;
    {
        const std::vector<int> devs = job_devices(o);
        if (devs.size() > 1 && !need_gtlt && ns >= 2) {
            if (cmp_core_multi(o, res, devs, have_lut, lut, multiset)) return;
        } else if (devs.size() > 1) {
            // said without -v: the user asked for several GPUs and gets one
            std::fprintf(stderr, "[d2g] D2G_DEVICES ignored for this job (%s): it runs on GPU %d alone\n",
                         need_gtlt ? "a sketch size that is not a power of two needs (gt, lt) counts from the raw registers, which the gathered bit-plane operand does not hold"
                                   : "fewer than two sketches", o.device);
        }
    }
    d2g_cmp_set *set = nullptr;
    const double t0 = now();
    const CmpShape sh(o, res);
    const size_t cap = std::max<size_t>(1, std::min(std::max(cmp_slot_values(sh.total_vals), sh.widest), std::max<size_t>(sh.total_vals, 1)));
    constexpr int NSLOT = 3;
    const bool fused = have_lut && sh.symmetric;
    struct Slot { std::unique_ptr<HostBuf> out, ca, cb; };
    Slot slots[NSLOT];
    for (auto &sl : slots) {                                        // plain memory: nothing to page-lock, nothing for a helper thread to do
        sl.out.reset(new HostBuf(cap * 4));
        sl.ca.reset(new HostBuf(fused ? 4 : cap * 4));
        sl.cb.reset(new HostBuf(need_gtlt ? cap * 4 : 4));
    }
    Emitter em(o, res);
    em.header();
    check(ctx, d2g_cmp_set_create(ctx, bits, ns, S, need_gtlt ? int(D2G_CMP_DIRECT) : int(D2G_CMP_AUTO), &set), "d2g_cmp_set_create");
    const double t_set = now();
    DevBuf dlut(ctx, (S + 1) * sizeof(float));
    if (have_lut) check(ctx, d2g_memcpy_h2d(ctx, dlut.p, lut.data(), (S + 1) * sizeof(float), nullptr), "h2d lut");
    DevBuf da(ctx, cap * 4), db(ctx, need_gtlt ? cap * 4 : 4);
    const double t_bufs = now();
    if (o.verbosity) std::fprintf(stderr, "[d2g] cmp set-up: host slots + header + operand upload + prepare %.3fs, device buffers %.3fs (slots of %zu values)\n",
                                  t_set - t0, t_bufs - t_set, cap);
    double t_dev = 0, emit_busy = 0;
    size_t nbatches = 0;
    const double t_loop = now();
    {
        EmitQueue eq(em, NSLOT);
        for (size_t r0 = 0; r0 < sh.nrows;) {
            const auto bt = sh.batch(r0, cap);
            const size_t r1 = bt.first, cnt = bt.second;
            const int si = eq.acquire();
            Slot &sl = slots[si];
            float *out = sl.out->as<float>();
            uint32_t *ca = sl.ca->as<uint32_t>(), *cb = sl.cb->as<uint32_t>();
            const double ta = now();
            if (cnt && sh.symmetric) {                                  // emitrect.cpp:290-323
                if (have_lut) {
                    check(ctx, d2g_cmp_lut_ut_dev(ctx, set, r0, r1, (const float *)dlut.p, (float *)da.p, nullptr), "d2g_cmp_lut_ut_dev");
                    check(ctx, d2g_memcpy_d2h(ctx, out, da.p, cnt * 4, nullptr), "d2h");
                } else {
                    if (need_gtlt) {
                        check(ctx, d2g_cmp_gtlt_ut_dev(ctx, set, r0, r1, (uint32_t *)da.p, (uint32_t *)db.p, nullptr), "d2g_cmp_gtlt_ut_dev");
                        check(ctx, d2g_memcpy_d2h(ctx, cb, db.p, cnt * 4, nullptr), "d2h");
                    } else {
                        check(ctx, d2g_cmp_eqcount_ut_dev(ctx, set, r0, r1, (uint32_t *)da.p, nullptr), "d2g_cmp_eqcount_ut_dev");
                    }
                    check(ctx, d2g_memcpy_d2h(ctx, ca, da.p, cnt * 4, nullptr), "d2h");
                    // x87 epilogue on the host (cmp_core.cpp:458-517)
                    check(ctx, d2g_epilogue_ut(ca, need_gtlt ? cb : nullptr, cards, ns, S, r0, r1, o.measure, o.k, multiset,
                                               int(o.workers()), out), "d2g_epilogue_ut");
                }
            } else if (cnt) {                                           // asymmetric / panel: emitrect.cpp:211-268
                if (need_gtlt) {
                    check(ctx, d2g_cmp_gtlt_rect_dev(ctx, set, r0, r1, sh.c0, sh.c1, (uint32_t *)da.p, (uint32_t *)db.p, nullptr), "d2g_cmp_gtlt_rect_dev");
                    check(ctx, d2g_memcpy_d2h(ctx, cb, db.p, cnt * 4, nullptr), "d2h");
                } else {
                    check(ctx, d2g_cmp_eqcount_rect_dev(ctx, set, r0, r1, sh.c0, sh.c1, (uint32_t *)da.p, nullptr), "d2g_cmp_eqcount_rect_dev");
                }
                check(ctx, d2g_memcpy_d2h(ctx, ca, da.p, cnt * 4, nullptr), "d2h");
                rect_epilogue(o, sh, r0, r1, ca, cb, cards, S, have_lut, lut, multiset, need_gtlt, out);
            }
            t_dev += now() - ta;
            if (sh.symmetric) eq.submit_rows(si, r0, r1, out, [ns](size_t i) { return ns - 1 - i; });
            else { const size_t ncol = sh.ncol; eq.submit_rows(si, r0, r1, out, [ncol](size_t) { return ncol; }); }
            r0 = r1;
            ++nbatches;
        }
        eq.finish();
        emit_busy = eq.t_busy;
        if (o.verbosity) std::fprintf(stderr, "[d2g] cmp: %zu sketches x S=%zu: upload+prepare+buffers %.3fs, %zu batches %.3fs wall (device+D2H+epilogue %.3fs busy, emit %.3fs busy, "
                                              "overlapped) (algo %s)\n", ns, S, t_loop - t0, nbatches, now() - t_loop, t_dev, emit_busy,
                                      d2g_cmp_set_algo(set) == D2G_CMP_BITSLICE ? "bitslice" : "direct");
    }
    if (g_stats.on) {
        const bool bs = d2g_cmp_set_algo(set) == D2G_CMP_BITSLICE;
        g_stats.raw("cmp", std::string("{\"sketches\": ") + std::to_string(ns) + ", \"sketchsize\": " + std::to_string(S) + ", \"values\": " + std::to_string(sh.total_vals) +
                    ", \"shape\": " + (sh.symmetric ? "\"upper triangle\"" : o.ok == PANEL ? "\"panel\"" : "\"square\"") + ", \"algo\": " + (bs ? "\"bitslice\"" : "\"direct\"") +
                    ", \"bit_planes\": " + (bs ? planes_json(ctx, set) : std::string("null")) + ", \"sparse_tiles\": " + (bs ? sparse_json(ctx, set) : std::string("null")) +
                    ", \"algorithmic_bytes\": " + Stats::numstr(8.0 * double(S) * double(ns) + 4.0 * double(sh.total_vals)) + ", \"batches\": " + std::to_string(nbatches) +
                    ", \"slot_values\": " + std::to_string(cap) +
                    ", \"devices\": [{\"index\": " + std::to_string(o.device) + ", \"name\": " + Stats::esc(device_label(o.device)) + ", \"k2\": " + Stats::kernel_json(ctx, "k2") +
                    ", \"k2prep\": " + Stats::kernel_json(ctx, "k2prep") + "}]" +
                    ", \"wall_s\": {\"densify_scan\": " + Stats::numstr(t_densify) + ", \"slots_header_upload_prepare\": " + Stats::numstr(t_set - t0) + ", \"batches\": " + Stats::numstr(now() - t_loop) +
                    ", \"device_d2h_epilogue_busy\": " + Stats::numstr(t_dev) + ", \"emit_busy\": " + Stats::numstr(emit_busy) + "}}");
    }
    if (g_release_at_exit) d2g_cmp_set_destroy(set);
}

d2g_ctx *make_ctx(const Options &o) {
    d2g_ctx *ctx = nullptr;
    const int rc = d2g_ctx_create(o.device, &ctx);
    if (rc) die(std::string("dashing2 (MI355X) needs a gfx950 GPU; d2g_ctx_create: ") + d2g_strerror(rc) + " (there is no CPU fallback)");
    return ctx;
}

// D2_FMT_EXP_UPPER was the round-2 switch for the float text layout (7 = fmt >= 11, 16 = fmt < 11); --fmt-compat replaced it.  It is
// still honoured -- with a warning -- when --fmt-compat is not given, so that scripts written against round 2 keep their output.
void apply_fmt_compat(Options &o) {
    if (!o.fmt_compat)
        if (const char *e = std::getenv("D2_FMT_EXP_UPPER")) {
            const int v = std::atoi(e);
            if (v == 7 || v == 16) {
                o.fmt_compat = v == 7 ? 11 : 10;
                std::fprintf(stderr, "dashing2 (MI355X): D2_FMT_EXP_UPPER=%d is deprecated; use --fmt-compat %d\n", v, o.fmt_compat);
            } else std::fprintf(stderr, "dashing2 (MI355X): D2_FMT_EXP_UPPER=%s ignored (7 or 16; use --fmt-compat 10|11)\n", e);
        }
    if (o.fmt_compat) set_fmt_compat(o.fmt_compat);
}

int sketch_main(int argc, char **argv) {                          // src/sketch_main.cpp:23-152
    Options o;
    if (int rc = parse_options(argc, argv, o)) return rc - 1;
    apply_fmt_compat(o);
    if (o.paths.empty()) { std::fprintf(stderr, "No paths provided. See usage.\n"); sketch_usage(); return 1; }
    o.device = job_devices(o)[0];
    g_stats.on = !o.gpu_stats.empty(); g_stats.path = o.gpu_stats;
    g_stats.str("command", "sketch");
    LazyCtx lctx(o, D2G_WARM_COPY | (o.sspace == SPACE_MULTISET ? D2G_WARM_K3 : D2G_WARM_K1) | (o.cmpout.empty() ? 0 : D2G_WARM_K2));
    Result res;
    if (o.parse_by_seq) sketch_core_byseq(res, o, lctx); else sketch_core(res, o, lctx);
    if (o.verbosity) std::fprintf(stderr, "[d2g] GPU context %.3fs + warm-up %.3fs on a helper thread, under the host ingest\n", lctx.t_create, lctx.t_warm);
    res.nq = o.nq;
    if (!o.cmpout.empty()) cmp_core(o, res, lctx.get());           // sketch_main.cpp:144-148
    g_stats.raw("context", std::string("{\"create_s\": ") + Stats::numstr(lctx.t_create) + ", \"warmup_s\": " + Stats::numstr(lctx.t_warm) + ", \"switches\": " + tuning_json(lctx.get()) + "}");
    return 0;
}

int cmp_main(int argc, char **argv) {                             // src/cmp_main.cpp:200-366
    Options o;
    o.is_cmp = true;
    if (int rc = parse_options(argc, argv, o)) return rc - 1;
    apply_fmt_compat(o);
    const double t_begin = now();
    o.device = job_devices(o)[0];
    g_stats.on = !o.gpu_stats.empty(); g_stats.path = o.gpu_stats;
    g_stats.str("command", "cmp");
    LazyCtx lctx(o, D2G_WARM_COPY | D2G_WARM_K2 | (o.presketched ? 0 : (o.sspace == SPACE_MULTISET ? D2G_WARM_K3 : D2G_WARM_K1)));   // under the reading of the sketch file(s)
    Result res;
    if (o.presketched) {
        // suffix sniffing, cmp_main.cpp:305-351
        const std::string &p0 = o.paths.empty() ? std::string() : o.paths.front();
        const auto dot = p0.find_last_of('.');
        const std::string suf = dot == std::string::npos ? std::string() : p0.substr(dot);
        if (suf == ".bmh" || suf == ".d2gbmh") {
            // stock BagMinHash sketches compare fine among themselves (equality counting does not care how registers were
            // drawn); what must never happen is one matrix over both kinds
            o.sspace = SPACE_MULTISET; o.kmer_result = FULL_SETSKETCH;
            for (const auto &p : o.paths) {
                const auto d2 = p.find_last_of('.');
                const std::string s2 = d2 == std::string::npos ? std::string() : p.substr(d2);
                if ((s2 == ".bmh" || s2 == ".d2gbmh") && s2 != suf)
                    die("cannot compare stock dashing2 BagMinHash sketches (.bmh) with this build's BMH-D2G sketches (.d2gbmh): "
                        "their registers are drawn differently (see README)");
            }
        }
        else if (suf == ".pmh") { o.sspace = SPACE_PSET; o.kmer_result = FULL_SETSKETCH; }
        else if (suf == ".ss") { o.sspace = SPACE_SET; o.kmer_result = FULL_SETSKETCH; }
        else if (suf == ".opss") { o.sspace = SPACE_SET; o.kmer_result = ONE_PERM; }
        else if (suf == ".kmerset64" || suf == ".kmerset128")
            die("k-mer set comparison is outside this build's hot-path scope");
        load_results(o, res);
    } else {
        if (o.paths.empty()) { std::fprintf(stderr, "No paths provided. See usage.\n"); cmp_usage(); return 1; }
        if (o.parse_by_seq) sketch_core_byseq(res, o, lctx); else sketch_core(res, o, lctx);
        res.nq = o.nq;
    }
    const double t_wait = now();
    d2g_ctx *ctx = lctx.get();
    if (o.verbosity) std::fprintf(stderr, "[d2g] inputs loaded in %.3fs; GPU context %.3fs + warm-up (first copy, code objects) %.3fs on a helper thread (%.3fs of it after the inputs were loaded)\n",
                                  t_wait - t_begin, lctx.t_create, lctx.t_warm, now() - t_wait);
    g_stats.raw("context", std::string("{\"create_s\": ") + Stats::numstr(lctx.t_create) + ", \"warmup_s\": " + Stats::numstr(lctx.t_warm) + ", \"inputs_loaded_s\": " + Stats::numstr(t_wait - t_begin) +
                ", \"waited_for_context_s\": " + Stats::numstr(now() - t_wait) + ", \"switches\": " + tuning_json(ctx) + "}");
    const double t_cmp = now();
    cmp_core(o, res, ctx);
    if (o.verbosity) std::fprintf(stderr, "[d2g] cmp_core %.3fs in all (densify, upload, batches, closing the output)\n", now() - t_cmp);
    return 0;
}

int main_usage() {                                                // src/d2.cpp:112-128
    std::fprintf(stderr, "dashing2 has several subcommands: sketch, cmp, wsketch, and contain.\n");
    std::fprintf(stderr, "Usage can be seen in those subcommands. (e.g., `dashing2 sketch -h`)\n\n");
    std::fprintf(stderr, "\tsketch: converts FastX into k-mer sets/sketches; also contains functionality from cmp, for one-step sketch and comparisons\n");
    std::fprintf(stderr, "\tcmp: compares previously sketched/decomposed k-mer sets and emits results. alias: dist\n\n");
    std::fprintf(stderr, "\twsketch: Takes a tuple of [1-3] input binary files [(u32 or u64), (float or double), (u32 or u64)] and performs weighted minhash sketching.\n");
    std::fprintf(stderr, "This MI355X build implements the sketch and cmp hot paths and wsketch's BagMinHash selections (contain/printmin are out of scope).\n");
    return 1;
}

}  // namespace

namespace d2h { int wsketch_main(int argc, char **argv); }        // wsketch_main.cpp

int main(int argc, char **argv) {                                 // src/d2.cpp:133-151
    char cwd[4096];
    std::string cmd = argv[0][0] == '/' ? std::string(argv[0]) : (getcwd(cwd, sizeof cwd) ? std::string(cwd) + "/" + argv[0] : std::string(argv[0]));
    for (char **s = argv + 1; *s; ++s) cmd += std::string(" ") + *s;
    std::fprintf(stderr, "#Calling Dashing2 version %s with command '%s'\n", DASHING2_VERSION, cmd.c_str());
    if (argc > 1) {
        const bool is_sketch = std::strcmp(argv[1], "sketch") == 0, is_cmp = std::strcmp(argv[1], "cmp") == 0 || std::strcmp(argv[1], "dist") == 0;
        if (is_sketch || is_cmp) {
            const double t0 = now();
            const int rc = is_sketch ? sketch_main(argc - 1, argv + 1) : cmp_main(argc - 1, argv + 1);
            // every output file has been flushed and closed by now (Emitter / write_stacked go out of scope inside)
            std::fflush(nullptr);
            g_stats.num("in_process_s", now() - t0);
            g_stats.str("version", DASHING2_VERSION);
            g_stats.write();
            if (std::getenv("D2G_VERBOSE_EXIT")) std::fprintf(stderr, "[d2g] in-process time %.3fs\n", now() - t0);
            if (!std::getenv("D2G_FULL_TEARDOWN")) _exit(rc);
            return rc;
        }
        if (std::strcmp(argv[1], "wsketch") == 0) return d2h::wsketch_main(argc - 1, argv + 1);
        if (std::strcmp(argv[1], "contain") == 0 || std::strcmp(argv[1], "printmin") == 0) {
            std::fprintf(stderr, "dashing2 (MI355X): subcommand %s is outside the hot-path scope of this build.\n", argv[1]);
            return 1;
        }
    }
    return main_usage();
}
