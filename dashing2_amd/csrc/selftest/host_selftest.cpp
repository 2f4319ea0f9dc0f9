// host_selftest.cpp -- exercises the x86 host half of libd2g (d2g_host.cpp: seqpack ingest, x87 finalisation,
// densify, epilogues, partition) under AddressSanitizer + UndefinedBehaviorSanitizer (`make sanitize` in
// dashing2_amd/csrc; the reference has the same kind of target: Makefile:102-103).  No GPU, no HIP: only the
// functions that never touch a device.  Exit code 0 = no finding (the sanitizers abort on any).
#include "../../../include/d2g.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "selftest failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
    std::mt19937_64 rng(7);
    // ---- seqpack: messy FASTA / FASTQ, lower case, N runs, short records, CRLF, no trailing newline, every k
    std::string fa;
    for (int r = 0; r < 40; ++r) {
        fa += (r % 5 == 4) ? "@q" : ">r";
        fa += std::to_string(r) + " desc\n";
        const size_t L = rng() % 700;
        std::string seq;
        for (size_t i = 0; i < L; ++i) seq += "ACGTacgtNn-"[rng() % (r % 3 ? 8 : 11)];
        if (r % 5 == 4) { fa += seq + "\n+\n" + std::string(seq.size(), 'I') + "\n"; continue; }
        for (size_t i = 0; i < seq.size(); i += 61) fa += seq.substr(i, 61) + (r % 7 == 0 ? "\r\n" : "\n");
    }
    fa += ">last\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT";      // no trailing newline
    for (int k = 1; k <= 32; ++k) {
        d2g_seqpack *sp = nullptr;
        REQUIRE(d2g_seqpack_create(k, &sp) == D2G_OK);
        REQUIRE(d2g_seqpack_add_fastx(sp, fa.data(), fa.size()) == D2G_OK);
        REQUIRE(d2g_seqpack_add_sequence(sp, "ACGT", 4) == D2G_OK);
        REQUIRE(d2g_seqpack_add_fastx(sp, "", 0) == D2G_OK);
        REQUIRE(d2g_seqpack_add_fastx_by_record(sp, fa.data(), fa.size()) == D2G_OK);
        const size_t ng = d2g_seqpack_ngenomes(sp), nr = d2g_seqpack_nruns(sp);
        REQUIRE(ng >= 3);
        const uint64_t *rs = d2g_seqpack_run_start(sp);
        const uint32_t *rl = d2g_seqpack_run_len(sp);
        const uint64_t *go = d2g_seqpack_genome_run_off(sp);
        REQUIRE(go[ng] == nr);
        uint64_t bases = 0;
        for (size_t i = 0; i < nr; ++i) { REQUIRE(rl[i] >= (uint32_t)k); REQUIRE(rs[i] == bases); bases += rl[i]; }
        REQUIRE(bases == d2g_seqpack_nbases(sp));
        REQUIRE(d2g_seqpack_packed_bytes(sp) >= (bases + 3) / 4 + 64);
        volatile uint8_t sink = 0;
        const uint8_t *pk = d2g_seqpack_packed(sp);
        for (size_t i = 0; i < d2g_seqpack_packed_bytes(sp); ++i) sink ^= pk[i];       // every byte readable
        uint64_t nk = 0;
        for (size_t g = 0; g < ng; ++g) nk += d2g_seqpack_nkmers(sp, g);
        REQUIRE(nk >= nr);
        d2g_seqpack_clear(sp);
        REQUIRE(d2g_seqpack_ngenomes(sp) == 0);
        REQUIRE(d2g_seqpack_add_path(sp, "/nonexistent/file.fa") == D2G_ERR_IO);
        d2g_seqpack_destroy(sp);
    }
    // ---- finalisation, densify, epilogues
    for (size_t S : {1, 2, 63, 64, 1000, 1024}) {
        const size_t m = d2g_oph_m(S), n = 5;
        std::vector<uint64_t> regs(n * m);
        for (auto &x : regs) x = (rng() % 4 == 0) ? ~0ull : rng() >> (rng() % 40);
        for (size_t i = 0; i < m; ++i) regs[2 * m + i] = ~0ull;                  // an empty sketch
        for (size_t i = 0; i < m; ++i) regs[3 * m + i] = 0;                       // sum == 0
        std::vector<double> sigs(n * S), cards(n);
        REQUIRE(d2g_oph_finalize(regs.data(), n, m, S, sigs.data(), cards.data(), 3) == D2G_OK);
        for (double c : cards) REQUIRE(!(c < 0));
        size_t filled = 0;
        REQUIRE(d2g_densify(sigs.data(), n, S, &filled, 2) == D2G_OK);
        std::vector<float> lut(S + 1);
        for (int meas = 0; meas < 6; ++meas)
            for (int ms = 0; ms < 2; ++ms) {
                const int rc = d2g_epilogue_lut(S, meas, 31, ms, lut.data());
                REQUIRE(rc == D2G_OK || rc == D2G_ERR_UNSUPPORTED);
                for (uint64_t neq : {uint64_t(0), uint64_t(S / 2), uint64_t(S)}) {
                    volatile float a = d2g_epilogue_neq(neq, S, cards[0], cards[1], meas, 31);
                    volatile float b = d2g_epilogue_gtlt(S - neq, 0, S, 1e6, 0.0, meas, 0);
                    (void)a; (void)b;
                }
            }
        const size_t N = 37;
        std::vector<uint32_t> ca(N * (N - 1) / 2), cb(ca.size());
        for (size_t i = 0; i < ca.size(); ++i) { ca[i] = rng() % (S + 1); cb[i] = rng() % (S - ca[i] + 1); }
        std::vector<double> cd(N, 1234.5);
        std::vector<float> out(ca.size());
        for (int meas = 0; meas < 6; ++meas) {
            REQUIRE(d2g_epilogue_ut(ca.data(), cb.data(), cd.data(), N, S, 0, N, meas, 21, 0, 3, out.data()) == D2G_OK);
            REQUIRE(d2g_epilogue_ut(ca.data(), nullptr, cd.data(), N, S, 5, 20, meas, 21, 1, 2, out.data()) == D2G_OK);
        }
    }
    // ---- partition / counts
    for (size_t N : {1, 2, 3, 10, 1000, 50000})
        for (int parts : {1, 2, 3, 8, 13}) {
            std::vector<size_t> b(parts + 1);
            REQUIRE(d2g_ut_partition(N, parts, b.data()) == D2G_OK);
            REQUIRE(b[0] == 0 && b[parts] == N);
            size_t tot = 0;
            for (int p = 0; p < parts; ++p) { REQUIRE(b[p] <= b[p + 1]); tot += d2g_ut_count(N, b[p], b[p + 1]); }
            REQUIRE(tot == N * (N - 1) / 2);
        }
    REQUIRE(d2g_wang_hash(133348) != 0 && d2g_seed_mask(0) == 0 && d2g_seed_mask(5) != 0);
    std::printf("host selftest OK\n");
    return 0;
}
