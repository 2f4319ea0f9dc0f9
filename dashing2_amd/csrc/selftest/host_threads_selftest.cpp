// host_threads_selftest.cpp -- the host-side THREADING of the path under ThreadSanitizer (`make tsan` in dashing2_amd/csrc;
// SURVEY 5 lists race detection among the reference's auxiliary tooling).  No GPU, no OpenMP (the OpenMP runtime is not
// instrumented and would only produce false reports): plain std::thread versions of the two pipelines the CLI runs,
//   ingest: parser threads -> d2g_seqpack (one packer per job, recycled through a shared pool) -> bounded queue -> one
//           consumer that reads the packed run streams                      (dashing2_main.cpp sketch_core)
//   emit:   producer filling slots -> SlotQueue -> consumer formatting floats (dashing2_main.cpp cmp_core / slot_queue.h)
// Exit code 0 and no ThreadSanitizer report = pass.
#include "../../../include/d2g.h"
#include "../../host/fmtfloat.h"
#include "../../host/slot_queue.h"
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "threads selftest failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static std::string fasta(unsigned seed, size_t len) {
    std::mt19937_64 rng(seed);
    std::string s = ">g" + std::to_string(seed) + "\n";
    for (size_t i = 0; i < len; ++i) {
        s += "ACGTacgtN"[rng() % (seed % 3 ? 8 : 9)];
        if (i % 70 == 69) s += '\n';
    }
    return s + "\n";
}

int main() {
    // ---------------------------------------------------------------- ingest pipeline
    const int k = 21, njobs = 64, nparsers = 6;
    std::vector<std::string> inputs;
    for (int j = 0; j < njobs; ++j) inputs.push_back(fasta(100 + j, 20000 + 977 * (j % 7)));
    // expected per-job base counts from a single-threaded pass
    std::vector<uint64_t> want(njobs);
    for (int j = 0; j < njobs; ++j) {
        d2g_seqpack *sp = nullptr;
        REQUIRE(d2g_seqpack_create(k, &sp) == D2G_OK);
        REQUIRE(d2g_seqpack_add_fastx(sp, inputs[j].data(), inputs[j].size()) == D2G_OK);
        want[j] = d2g_seqpack_nbases(sp);
        d2g_seqpack_destroy(sp);
    }
    struct Ready { int job; d2g_seqpack *sp; };
    std::deque<Ready> ready;
    std::vector<d2g_seqpack *> pool;
    std::mutex mu;
    std::condition_variable cv_ready, cv_space;
    std::atomic<int> next{0};
    const size_t max_ready = 4;
    std::vector<std::thread> parsers;
    std::atomic<int> failures{0};
    for (int t = 0; t < nparsers; ++t) parsers.emplace_back([&] {
        for (;;) {
            const int j = next.fetch_add(1);
            if (j >= njobs) break;
            d2g_seqpack *sp = nullptr;
            { std::lock_guard<std::mutex> lk(mu); if (!pool.empty()) { sp = pool.back(); pool.pop_back(); } }
            if (!sp && d2g_seqpack_create(k, &sp) != D2G_OK) { ++failures; continue; }
            if (d2g_seqpack_add_fastx(sp, inputs[j].data(), inputs[j].size()) != D2G_OK) ++failures;
            (void)d2g_seqpack_packed_bytes(sp);                        // pad, off the consumer thread
            std::unique_lock<std::mutex> lk(mu);
            cv_space.wait(lk, [&] { return ready.size() < max_ready; });
            ready.push_back({j, sp});
            cv_ready.notify_one();
        }
    });
    uint64_t checksum = 0;
    for (int done = 0; done < njobs; ++done) {
        Ready r;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_ready.wait(lk, [&] { return !ready.empty(); });
            r = ready.front();
            ready.pop_front();
            cv_space.notify_one();
        }
        REQUIRE(d2g_seqpack_nbases(r.sp) == want[r.job]);
        const uint8_t *pk = d2g_seqpack_packed(r.sp);
        for (size_t i = 0; i < d2g_seqpack_packed_bytes(r.sp); i += 97) checksum += pk[i];
        const uint64_t *rs = d2g_seqpack_run_start(r.sp);
        const uint32_t *rl = d2g_seqpack_run_len(r.sp);
        for (size_t i = 0; i < d2g_seqpack_nruns(r.sp); ++i) checksum += rs[i] + rl[i];
        d2g_seqpack_clear(r.sp);
        { std::lock_guard<std::mutex> lk(mu); pool.push_back(r.sp); }
    }
    for (auto &th : parsers) th.join();
    for (d2g_seqpack *p : pool) d2g_seqpack_destroy(p);
    REQUIRE(failures.load() == 0);
    // ---------------------------------------------------------------- emit pipeline
    struct Job { const float *data; size_t n; };
    constexpr int NSLOT = 3;
    std::vector<std::vector<float>> slots(NSLOT, std::vector<float>(4096));
    std::string text;
    size_t nvals = 0;
    {
        d2h::SlotQueue<Job> q(NSLOT, [&](const Job &j) {
            char buf[d2h::FMT_MAX_FLOAT_CHARS + 1];
            for (size_t i = 0; i < j.n; ++i) { const size_t l = d2h::format_float(j.data[i], buf); text.append(buf, l); text += '\t'; }
            nvals += j.n;
        });
        std::mt19937 rng(5);
        for (int b = 0; b < 200; ++b) {
            const int s = q.acquire();
            const size_t n = 1 + rng() % slots[s].size();
            for (size_t i = 0; i < n; ++i) slots[s][i] = float(rng() % 100000) / float(1 + rng() % 1000);   // producer writes the slot ...
            q.submit(s, Job{slots[s].data(), n});                                                           // ... the consumer reads it
        }
        q.finish();
        REQUIRE(q.t_busy >= 0);
    }
    REQUIRE(nvals > 0 && !text.empty());
    std::printf("host threads selftest OK (ingest checksum %llu, %zu values formatted)\n", (unsigned long long)checksum, nvals);
    return 0;
}
