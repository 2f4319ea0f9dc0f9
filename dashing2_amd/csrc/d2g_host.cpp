// d2g_host.cpp -- x86 host half of libd2g: the O(S) / O(pairs) scalar arithmetic of the
// path that must stay on the host to be bit-exact with the reference (x87 long double),
// plus the FASTX -> packed-run-stream ingest.  Compiled with g++ (not hipcc) so that
// `long double` is the 80-bit x87 type the reference's gcc build uses.
//
// Reference lines restated here are cited per function.
#include "../../include/d2g.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <sys/stat.h>
#include <vector>
#include <zlib.h>
#include <unistd.h>
#include <fcntl.h>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {

int d2g_version(void) { return D2G_VERSION_MAJOR * 1000 + D2G_VERSION_MINOR; }

const char *d2g_strerror(int st) {
    switch (st) {
        case D2G_OK: return "ok";
        case D2G_ERR_INVALID: return "invalid argument";
        case D2G_ERR_NODEVICE: return "no usable HIP (gfx950) device";
        case D2G_ERR_HIP: return "HIP runtime error";
        case D2G_ERR_NOMEM: return "out of memory";
        case D2G_ERR_UNSUPPORTED: return "unsupported configuration for the MI355X hot path";
        case D2G_ERR_IO: return "I/O error";
        case D2G_ERR_INTERNAL: return "internal invariant failed";
        default: return "unknown d2g status";
    }
}

// sketch::hash::WangHash::hash (absent third-party source; Thomas Wang's published 64-bit mix).
// Reference call sites: enums.h:138 (maskfn), oph.h:49 (BHasher).
uint64_t d2g_wang_hash(uint64_t k) {
    k = ~k + (k << 21);
    k ^= k >> 24;
    k = k + (k << 3) + (k << 8);
    k ^= k >> 14;
    k = k + (k << 2) + (k << 4);
    k ^= k >> 28;
    k += k << 31;
    return k;
}

// enums.cpp:133-139
uint64_t d2g_seed_mask(uint64_t seedseed) { return seedseed ? d2g_wang_hash(seedseed) : 0; }

// oph.h:59 seed_ = std::mt19937_64(x)(), oph.h:142 x = 0x321b919a61cb41f7, oph.h:46 CEIXOR constant
uint64_t d2g_oph_xor_const(void) {
    static const uint64_t c = std::mt19937_64(0x321b919a61cb41f7ull)() ^ 0x533f8c2151b20f97ull;
    return c;
}

size_t d2g_oph_m(size_t S) { return S + (S & 1); }   // oph.h:143-146

// oph.h:240-247
double d2g_oph_card(const uint64_t *regs, size_t m) {
    long double sum = 0.L;
    for (size_t i = 0; i < m; ++i) sum += regs[i] * 0x1p-64L;
    if (!sum) return std::numeric_limits<double>::infinity();
    return m * (m / sum);
}

// oph.h:248-263
int d2g_oph_signatures(const uint64_t *regs, size_t m, double *sig) {
    if (!regs || !sig) return D2G_ERR_INVALID;
    constexpr uint64_t MAXV = std::numeric_limits<uint64_t>::max();
    const size_t nmax = std::count(regs, regs + m, MAXV);
    const long double mul = -double(1) / (m - nmax);          // double division, then widened
    for (size_t i = 0; i < m; ++i) {
        const uint64_t x = regs[i];
        sig[i] = (x == MAXV || x == 0) ? 0. : double(mul * std::log(0x1p-64L * (MAXV - x + 1)));
    }
    return D2G_OK;
}

int d2g_oph_finalize(const uint64_t *regs, size_t n, size_t m, size_t S, double *sigs, double *cards, int nthreads) {
    if (!regs || !sigs || !cards || S > m) return D2G_ERR_INVALID;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
    #pragma omp parallel num_threads(nthreads)
#endif
    {
        std::vector<double> tmp(m);
#ifdef _OPENMP
        #pragma omp for schedule(static)
#endif
        for (size_t g = 0; g < n; ++g) {
            const uint64_t *r = regs + g * m;
            cards[g] = d2g_oph_card(r, m);                      // fastxsketch.cpp:567
            d2g_oph_signatures(r, m, tmp.data());               // fastxsketch.cpp:586
            std::memcpy(sigs + g * S, tmp.data(), S * sizeof(double));  // :605,:610 (first S only)
        }
    }
    return D2G_OK;
}

// ssi.h:26-36 (in-tree twin of wy::wyhash64_stateless used at cmp_core.cpp:597)
static inline uint64_t wyhash64_stateless(uint64_t *seed) {
    *seed += 0x60bee2bee120fc15ull;
    __uint128_t l = *seed ^ 0xe7037ed1a0b428dbull;
    l *= *seed;
    return uint64_t(l ^ (l >> 64));
}

// cmp_core.cpp:577-613
static size_t densify_one(double *sig, size_t S, std::vector<double> &tmp) {
    if (size_t(std::count(sig, sig + S, 0.)) == S) return S;
    size_t ne = 0;
    tmp.assign(sig, sig + S);
    for (size_t i = 0; i < S; ++i) {
        if (sig[i] != 0.) continue;
        ++ne;
        uint64_t rng = i + 0x5bf2b8bdf07c06cull, j;
        do j = wyhash64_stateless(&rng) % S; while (sig[j] == 0.);
        tmp[i] = sig[j];
    }
    std::copy(tmp.begin(), tmp.end(), sig);
    return ne;
}

int d2g_densify(double *sigs, size_t n, size_t S, size_t *nfilled, int nthreads) {
    if (!sigs || !S) return D2G_ERR_INVALID;
    if (nthreads < 1) nthreads = 1;
    size_t total = 0;
#ifdef _OPENMP
    #pragma omp parallel num_threads(nthreads) reduction(+:total)
#endif
    {
        std::vector<double> tmp;
#ifdef _OPENMP
        #pragma omp for schedule(dynamic, 32)                   // cmp_core.cpp:707
#endif
        for (size_t i = 0; i < n; ++i) total += densify_one(sigs + i * S, S, tmp);
    }
    if (nfilled) *nfilled = total;
    return D2G_OK;
}

// cmp_core.cpp:361
static inline double sim2dist_f(float x, int k) {
    const double pm = -1. / std::max(1, k);
    return x ? std::log(2. * x / (1. + x)) * pm : std::numeric_limits<double>::infinity();
}
// cmp_core.cpp:573-575
static inline float finish(long double ret) {
    if (std::isnan(ret) || std::isinf(ret)) ret = std::numeric_limits<long double>::max();
    return float(ret);
}

// cmp_core.cpp:458-494
float d2g_epilogue_gtlt(uint64_t gt, uint64_t lt, size_t S, double lhc, double rhc, int measure, int k) {
    long double ret = std::numeric_limits<float>::max();
    const long double invdenom = 1.L / S;
    const long double alpha = gt * invdenom, beta = lt * invdenom;
    const long double lhcard = lhc, rhcard = rhc;
    long double eq = (1. - alpha - beta);
    const long double ucard = std::max((lhcard + rhcard) / (2.L - alpha - beta), 0.L);
    if (eq <= 0.)
        return measure != D2G_POISSON_LLR ? 0.f : std::numeric_limits<float>::infinity();
    static constexpr long double EPS = 1e-15;
    if (eq <= EPS) eq = 0;
    const float isz = ucard * eq, sim = eq;
    switch (measure) {
        case D2G_SIMILARITY: ret = sim; break;
        case D2G_INTERSECTION: ret = isz; break;
        case D2G_CONTAINMENT: ret = isz / rhcard; break;
        case D2G_SYMMETRIC_CONTAINMENT: ret = isz / std::min(lhcard, rhcard); break;
        case D2G_POISSON_LLR: ret = sim2dist_f(sim, k); break;
        case D2G_UNION_SIZE: ret = lhcard + rhcard - isz; break;
        default: ret = -1.f; break;
    }
    return finish(ret);
}

// cmp_core.cpp:495-517
float d2g_epilogue_neq(uint64_t neq, size_t S, double lhc, double rhc, int measure, int k) {
    const long double lhcard = lhc, rhcard = rhc, invdenom = 1.L / S;
    long double ret = invdenom * neq;
    auto ucard = [&]() { return std::max((lhcard + rhcard) / (1.L + ret), 0.L); };
    if (measure == D2G_INTERSECTION) ret *= ucard();
    else if (measure == D2G_SYMMETRIC_CONTAINMENT) ret *= ucard() / std::min(lhcard, rhcard);
    else if (measure == D2G_CONTAINMENT) ret *= ucard() / lhcard;
    else if (measure == D2G_POISSON_LLR) {
        const double pm = -1. / std::max(1, k);
        // sim2dist(auto x) with x = long double: logl and the product in x87, lambda returns double
        ret = ret ? double(std::log(2. * ret / (1. + ret)) * pm) : std::numeric_limits<double>::infinity();
    } else if (measure == D2G_UNION_SIZE) {
        const long double isz = ret * ucard();
        ret = lhcard + rhcard - isz;
    }
    return finish(ret);
}

int d2g_epilogue_lut(size_t S, int measure, int k, int multiset_space, float *lut) {
    if (!lut || !S) return D2G_ERR_INVALID;
    if (measure != D2G_SIMILARITY && measure != D2G_POISSON_LLR) return D2G_ERR_UNSUPPORTED;
    if (multiset_space) {
        for (size_t e = 0; e <= S; ++e) lut[e] = d2g_epilogue_neq(e, S, 1., 1., measure, k);
        return D2G_OK;
    }
    // set space: the value is a function of gt and lt; it depends on gt+lt only when every
    // product gt * (1.L/S) is exact, i.e. S is a power of two.
    if (S & (S - 1)) return D2G_ERR_UNSUPPORTED;
    for (size_t e = 0; e <= S; ++e) lut[e] = d2g_epilogue_gtlt(S - e, 0, S, 1., 1., measure, k);
    return D2G_OK;
}

// epilogue over rows [r0,r1) of the condensed upper triangle from the device's integer counts.
// ca = neq (or gt when cb != null), cb = lt.  Same arithmetic as compare(): cmp_core.cpp:458-517.
int d2g_epilogue_ut(const uint32_t *ca, const uint32_t *cb, const double *cards, size_t N, size_t S,
                    size_t r0, size_t r1, int measure, int k, int multiset_space, int nthreads, float *out) {
    if (r0 > r1 || r1 > N || !S) return D2G_ERR_INVALID;
    if (d2g_ut_count(N, r0, r1) == 0) return D2G_OK;
    if (!ca || !cards || !out) return D2G_ERR_INVALID;
    std::vector<size_t> off(r1 - r0 + 1, 0);
    for (size_t i = r0; i < r1; ++i) off[i - r0 + 1] = off[i - r0] + (N - 1 - i);
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
    #pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
    for (size_t i = r0; i < r1; ++i) {
        const size_t base = off[i - r0];
        for (size_t j = i + 1; j < N; ++j) {
            const size_t p = base + (j - i - 1);
            if (multiset_space) out[p] = d2g_epilogue_neq(ca[p], S, cards[i], cards[j], measure, k);
            else if (cb) out[p] = d2g_epilogue_gtlt(ca[p], cb[p], S, cards[i], cards[j], measure, k);
            // power-of-two S: every multiple of 1/S is exact, so only gt+lt = S-neq matters
            else out[p] = d2g_epilogue_gtlt(S - ca[p], 0, S, cards[i], cards[j], measure, k);
        }
    }
    return D2G_OK;
}

size_t d2g_ut_count(size_t N, size_t r0, size_t r1) {
    if (r1 > N) r1 = N;
    if (r0 >= r1) return 0;
    // sum_{r=r0}^{r1-1} (N-1-r)
    const size_t n = r1 - r0;
    const size_t tri1 = r1 * (r1 - 1) / 2, tri0 = r0 ? r0 * (r0 - 1) / 2 : 0;
    return n * (N - 1) - (tri1 - tri0);
}

int d2g_ut_partition(size_t N, int nparts, size_t *bounds) {
    if (nparts < 1 || !bounds) return D2G_ERR_INVALID;
    const long double total = N ? (long double)N * (N - 1) / 2 : 0;
    bounds[0] = 0;
    size_t r = 0;
    for (int p = 1; p < nparts; ++p) {
        const long double target = total * p / nparts;
        // pairs in rows [0,r) = r*(N-1) - r(r-1)/2 ; advance to the first r reaching the target
        while (r < N && (long double)r * (N - 1) - (long double)r * (r - 1) / 2 < target) ++r;
        bounds[p] = r;
    }
    bounds[nparts] = N;
    return D2G_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// FASTX ingest -> packed run stream
// ---------------------------------------------------------------------------
// 32 ASCII bases -> 64 bits (2 bits/base, base i at bits [2i, 2i+2)); false if any byte is not ACGTacgt.
// code = ((c >> 1) & 3) ^ ((c >> 2) & 1): A 0, C 1, G 2, T 3.
__attribute__((target("avx2"))) static inline bool pack32_avx2(const char *s, uint64_t *out) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s));
    const __m256i u = _mm256_and_si256(v, _mm256_set1_epi8((char)0xDF));
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('C'))),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(u, _mm256_set1_epi8('T'))));
    if ((uint32_t)_mm256_movemask_epi8(ok) != 0xFFFFFFFFu) return false;
    const __m256i c1 = _mm256_and_si256(_mm256_srli_epi16(u, 1), _mm256_set1_epi8(3));
    const __m256i c2 = _mm256_and_si256(_mm256_srli_epi16(u, 2), _mm256_set1_epi8(1));
    const __m256i code = _mm256_xor_si256(c1, c2);
    const __m256i p16 = _mm256_maddubs_epi16(code, _mm256_set1_epi16(0x0401));       // c0 + 4 c1 per 16-bit lane
    const __m256i p32 = _mm256_madd_epi16(p16, _mm256_set1_epi32(0x00100001));        // 4 bases per 32-bit lane (one byte)
    const __m256i sh = _mm256_shuffle_epi8(p32, _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                                                  0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
    const uint32_t lo = (uint32_t)_mm256_extract_epi32(sh, 0), hi = (uint32_t)_mm256_extract_epi32(sh, 4);
    *out = (uint64_t)lo | ((uint64_t)hi << 32);
    return true;
}

// 16-base variant (the tail of an 80-column line): 32 bits out
__attribute__((target("avx2"))) static inline bool pack16_avx2(const char *s, uint32_t *out) {
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s));
    const __m128i u = _mm_and_si128(v, _mm_set1_epi8((char)0xDF));
    const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(u, _mm_set1_epi8('A')), _mm_cmpeq_epi8(u, _mm_set1_epi8('C'))),
                                    _mm_or_si128(_mm_cmpeq_epi8(u, _mm_set1_epi8('G')), _mm_cmpeq_epi8(u, _mm_set1_epi8('T'))));
    if (_mm_movemask_epi8(ok) != 0xFFFF) return false;
    const __m128i code = _mm_xor_si128(_mm_and_si128(_mm_srli_epi16(u, 1), _mm_set1_epi8(3)), _mm_and_si128(_mm_srli_epi16(u, 2), _mm_set1_epi8(1)));
    const __m128i p32 = _mm_madd_epi16(_mm_maddubs_epi16(code, _mm_set1_epi16(0x0401)), _mm_set1_epi32(0x00100001));
    const __m128i sh = _mm_shuffle_epi8(p32, _mm_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
    *out = (uint32_t)_mm_cvtsi128_si32(sh);
    return true;
}

// AVX-512 (BW + VL + VBMI2): 64 ASCII bytes that are all bases or '\n' -> the bases, line feeds squeezed out
// (vpcompressb), 2 bits each in a 128-bit little-endian bit string; returns the number of bases, or -1 when
// the block holds anything else (header, 'N', '\r', quality, ...), which the per-line code then takes.
__attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi2"))) static inline int pack64_skip_newlines(const char *s, uint64_t out[2]) {
    const __m512i v = _mm512_loadu_si512(s);
    const __m512i u = _mm512_and_si512(v, _mm512_set1_epi8((char)0xDF));
    const __mmask64 base = _mm512_cmpeq_epi8_mask(u, _mm512_set1_epi8('A')) | _mm512_cmpeq_epi8_mask(u, _mm512_set1_epi8('C')) |
                           _mm512_cmpeq_epi8_mask(u, _mm512_set1_epi8('G')) | _mm512_cmpeq_epi8_mask(u, _mm512_set1_epi8('T'));
    const __mmask64 nl = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('\n'));
    if ((base | nl) != ~(__mmask64)0) return -1;
    const __m512i c1 = _mm512_and_si512(_mm512_srli_epi16(u, 1), _mm512_set1_epi8(3));
    const __m512i c2 = _mm512_and_si512(_mm512_srli_epi16(u, 2), _mm512_set1_epi8(1));
    const __m512i code = _mm512_maskz_compress_epi8(base, _mm512_xor_si512(c1, c2));
    const __m512i p16 = _mm512_maddubs_epi16(code, _mm512_set1_epi16(0x0401));        // c0 + 4 c1 per 16-bit lane
    const __m512i p32 = _mm512_madd_epi16(p16, _mm512_set1_epi32(0x00100001));         // 4 bases per 32-bit lane (its low byte)
    const __m128i bytes = _mm512_cvtepi32_epi8(p32);                                    // 16 bytes = 64 bases
    out[0] = (uint64_t)_mm_cvtsi128_si64(bytes);
    out[1] = (uint64_t)_mm_extract_epi64(bytes, 1);
    return (int)_mm_popcnt_u64(base);
}

struct d2g_seqpack {
    int k;
    std::vector<uint8_t> packed;       // 4 bases / byte
    uint64_t nbases = 0;               // bases appended to the stream
    std::vector<uint64_t> run_start;
    std::vector<uint32_t> run_len;
    std::vector<uint64_t> genome_run_off{0};
    std::vector<uint64_t> genome_nkmers;
    bool by_record = false;            // --parse-by-seq: every FASTX record is its own genome ...
    std::vector<std::string> names;    // ... named by its header up to the first whitespace (kseq name)
    // open run state
    uint64_t cur_start = 0, cur_len = 0;
    uint64_t cur_kmers = 0;
    uint8_t  accum = 0;                // partially filled byte lives in packed.back()

    // runs longer than this are split into pieces that overlap by k-1 bases in place (run_len is u32);
    // D2G_MAX_RUN lowers it so that tests can exercise the split
    uint32_t MAX_RUN = [] { const char *e = std::getenv("D2G_MAX_RUN"); const long v = e ? std::atol(e) : 0; return v >= 64 ? uint32_t(v) : (1u << 30); }();

    // bases of a run shorter than k are useless: rewind the stream over them
    // (packed.size() may exceed the stream: the logical length is nbases)
    inline void rewind_to(uint64_t nb) {
        nbases = nb;
        if (nb & 3) packed[nb >> 2] &= uint8_t((1u << ((nb & 3) * 2)) - 1);
    }
    void close_run_raw() {
        if (cur_len >= uint64_t(k)) {
            // split very long runs; pieces overlap by k-1 bases in place
            uint64_t s = cur_start, left = cur_len;
            while (left > MAX_RUN) {
                run_start.push_back(s); run_len.push_back(MAX_RUN);
                const uint64_t adv = MAX_RUN - (k - 1);
                s += adv; left -= adv;
            }
            run_start.push_back(s); run_len.push_back(uint32_t(left));
            cur_kmers += cur_len - k + 1;
        } else if (cur_len) {
            rewind_to(cur_start);
        }
        cur_len = 0;
    }
    void close_run() { close_run_raw(); }
    inline void feed(const char *s, size_t n) {
        static const int8_t *lut = code_lut();
        // worst case every byte is a base: make room once, then write through a raw pointer
        const size_t need = (nbases + n + 3) / 4 + 1;
        if (packed.size() < need) packed.resize(std::max(need, packed.size() + packed.size() / 2));
        uint8_t *pk = packed.data();
        uint64_t nb = nbases, clen = cur_len;
        size_t i = 0;
        static const bool have_avx2 = __builtin_cpu_supports("avx2");
        if (have_avx2) {
            // 32 bases at a time while the bytes are all ACGT (the common case inside a sequence line);
            // anything else drops to the scalar loop below for the rest of this call
            uint64_t w;
            while (i + 32 <= n && pack32_avx2(s + i, &w)) {
                if (!clen) cur_start = nb;
                const unsigned sh = (unsigned)(nb & 3) * 2;
                uint8_t *dst = pk + (nb >> 2);
                if (sh == 0) {
                    std::memcpy(dst, &w, 8);
                } else {
                    const uint64_t lowpart = (w << sh) | dst[0];        // dst[0] holds the stream's last (4 - sh/2 ... ) bases in its low sh bits
                    std::memcpy(dst, &lowpart, 8);
                    dst[8] = uint8_t(w >> (64 - sh));
                }
                nb += 32; clen += 32; i += 32;
            }
            uint32_t w32;
            if (i + 16 <= n && n - i < 32 && pack16_avx2(s + i, &w32)) {
                if (!clen) cur_start = nb;
                const unsigned sh = (unsigned)(nb & 3) * 2;
                uint8_t *dst = pk + (nb >> 2);
                if (sh == 0) {
                    std::memcpy(dst, &w32, 4);
                } else {
                    const uint32_t lowpart = (w32 << sh) | dst[0];
                    std::memcpy(dst, &lowpart, 4);
                    dst[4] = uint8_t(w32 >> (32 - sh));
                }
                nb += 16; clen += 16; i += 16;
            }
        }
        for (; i < n; ++i) {
            const int c = lut[(unsigned char)s[i]];
            if (c < 0) {
                if (clen) { nbases = nb; cur_len = clen; close_run_raw(); nb = nbases; clen = 0; pk = packed.data(); }
                continue;
            }
            if (!clen) cur_start = nb;
            const unsigned sh = (unsigned)(nb & 3) * 2;
            if (sh == 0) pk[nb >> 2] = uint8_t(c); else pk[nb >> 2] |= uint8_t(c << sh);
            ++nb; ++clen;
        }
        nbases = nb; cur_len = clen;
    }
    // Sequence bytes from `s` in 64-byte blocks while every block is bases and line feeds only (the body of a FASTA record
    // with any line width); returns the bytes consumed and adds the bases taken to *nseq.  Runs are not broken by line
    // feeds, exactly as with one feed() per line.
    __attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi2,popcnt"))) size_t feed_blocks(const char *s, size_t n, size_t *nseq) {
        const size_t need = (nbases + n + 3) / 4 + 24;
        if (packed.size() < need) packed.resize(std::max(need, packed.size() + packed.size() / 2));
        uint8_t *pk = packed.data();
        uint64_t nb = nbases, clen = cur_len;
        size_t i = 0;
        uint64_t w[2];
        while (i + 64 <= n) {
            const int cnt = pack64_skip_newlines(s + i, w);
            if (cnt < 0) break;
            if (cnt) {
                if (!clen) cur_start = nb;
                const unsigned sh = (unsigned)(nb & 3) * 2;
                uint8_t *dst = pk + (nb >> 2);
                if (sh == 0) {
                    std::memcpy(dst, w, 16);
                } else {
                    const uint64_t lo = (w[0] << sh) | dst[0];           // dst[0]: the stream's partially filled last byte
                    const uint64_t hi = (w[1] << sh) | (w[0] >> (64 - sh));
                    std::memcpy(dst, &lo, 8);
                    std::memcpy(dst + 8, &hi, 8);
                    dst[16] = uint8_t(w[1] >> (64 - sh));
                }
                nb += (uint64_t)cnt; clen += (uint64_t)cnt;
            }
            i += 64;
        }
        *nseq += (size_t)(nb - nbases);
        nbases = nb; cur_len = clen;
        // bytes written past the stream's end are zero except where codes of THIS block landed: the next write ORs its
        // low bits into the last partial byte, like feed() does
        return i;
    }
    void end_record() { if (cur_len) close_run(); }
    void end_genome() {
        end_record();
        genome_run_off.push_back(run_start.size());
        genome_nkmers.push_back(cur_kmers);
        cur_kmers = 0;
    }
    static const int8_t *code_lut() {
        static int8_t t[256];
        static bool init = false;
        if (!init) {
            std::memset(t, -1, sizeof(t));
            t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
            init = true;
        }
        return t;
    }
    // kseq_read()'s record walk over an in-memory FASTA/FASTQ buffer (klib kseq.h as published -- the copy bonsai vendors is
    // absent: UNVERIFIED-AGAINST-SOURCE; reference call site src/fastxsketch.cpp:416-417, holder src/d2.h:273-305; the oracle's
    // d2o_walk_fastx_records restates the same rules):
    //  * at the start of the input and after every FASTQ record (kseq's last_char == 0) the next header is searched BYTE BY BYTE:
    //    "junk>name" starts a record in the middle of a line; after a FASTA record the header character was consumed by the
    //    sequence loop, which only looks at the first character of each line;
    //  * a trailing '\r' of a sequence / quality line is dropped while the accumulated string is longer than one character;
    //  * after '+': the rest of that line, then AT LEAST ONE quality line, until qual.l >= seq.l; a record whose input ends inside
    //    the '+' line or whose quality length differs is an error (-2): `while (kseq_read(ks) >= 0)` stops there, so the record is
    //    not sketched (its bases are rewound out of the stream) and the rest of the file is ignored.
    void feed_fastx(const char *buf, size_t len) {
        size_t pos = 0;
        auto skip_line = [&]() {
            const char *nl = (const char *)std::memchr(buf + pos, '\n', len - pos);
            pos = nl ? size_t(nl - buf) + 1 : len;
        };
        static const bool have_vbmi2 = __builtin_cpu_supports("avx512vbmi2") && __builtin_cpu_supports("avx512bw") &&
                                       __builtin_cpu_supports("avx512vl") && !std::getenv("D2G_NO_AVX512");
        int last_char = 0;
        for (;;) {
            if (!last_char) {
                while (pos < len && buf[pos] != '>' && buf[pos] != '@') ++pos;     // usually the very next byte
                if (pos >= len) break;
                last_char = (unsigned char)buf[pos++];
            }
            if (pos >= len) break;                         // a header character at the very end: kseq returns -1, no record
            // state to return to if this record turns out to be an error
            const uint64_t snap_bases = nbases, snap_kmers = cur_kmers;
            const size_t snap_runs = run_start.size();
            size_t name_end = pos;
            while (name_end < len && !std::isspace((unsigned char)buf[name_end])) ++name_end;
            const size_t name_pos = pos;
            skip_line();                                   // header
            size_t seqlen = 0;
            int term = -1;                                 // the line-start character that ended the sequence, -1 = end of input
            bool at_line_start = true;
            while (pos < len) {
                const int c = (unsigned char)buf[pos];
                // a record boundary is only looked for at the start of a line (the block path may stop inside one)
                if (at_line_start && (c == '>' || c == '+' || c == '@')) { term = c; ++pos; break; }
                if (have_vbmi2 && len - pos >= 64) {
                    const size_t took = feed_blocks(buf + pos, len - pos, &seqlen);
                    if (took) {
                        pos += took;
                        at_line_start = buf[pos - 1] == '\n';
                        continue;
                    }
                }
                const char *nl = (const char *)std::memchr(buf + pos, '\n', len - pos);
                const size_t e = nl ? size_t(nl - buf) : len;
                size_t ll = e - pos;
                if (ll && buf[pos + ll - 1] == '\r' && seqlen + ll > 1) --ll;
                feed(buf + pos, ll);
                seqlen += ll;
                pos = nl ? e + 1 : len;
                at_line_start = true;
            }
            bool ok = true;
            if (term == '>' || term == '@') last_char = term;
            else if (term == '+') {
                const char *nl = (const char *)std::memchr(buf + pos, '\n', len - pos);
                if (!nl) ok = false;                       // the input ends inside the '+' line
                else {
                    pos = size_t(nl - buf) + 1;
                    size_t ql = 0;
                    do {
                        if (pos >= len) break;
                        nl = (const char *)std::memchr(buf + pos, '\n', len - pos);
                        const size_t e = nl ? size_t(nl - buf) : len;
                        const size_t ll = e - pos;
                        ql += ll;
                        if (ql > 1 && ll && buf[pos + ll - 1] == '\r') --ql;
                        pos = nl ? e + 1 : len;
                    } while (ql < seqlen);
                    last_char = 0;
                    ok = ql == seqlen;
                }
            }
            if (!ok) {                                     // kseq_read() == -2: drop the record, stop reading this input
                if (cur_len) { cur_len = 0; }
                run_start.resize(snap_runs); run_len.resize(snap_runs);
                cur_kmers = snap_kmers;
                rewind_to(snap_bases);
                break;
            }
            if (by_record) {                               // fastxsketchbyseq.cpp:243-244: names_ = kseq name
                names.emplace_back(buf + name_pos, name_end - name_pos);
                end_genome();
            } else end_record();
            if (term == -1) break;
        }
    }
    void finalize_pad() {
        // logical stream = ceil(nbases/4) bytes; everything after it must read as zero-initialised pad
        const size_t nbytes = (nbases + 3) / 4;
        if (packed.size() < nbytes + 64) packed.resize(nbytes + 64, 0);
        std::memset(packed.data() + nbytes, 0, 64);
        padded_bytes = nbytes + 64;
    }
    void unpad() {}
    void reserve_bases(uint64_t more) {
        const size_t need = (nbases + more + 3) / 4 + 65;
        if (packed.size() < need) packed.resize(need);
    }
    size_t padded_bytes = 0;
};

// whole file into a reusable per-thread buffer; gzip members go through zlib, plain files are read directly
static bool slurp(const char *path, std::vector<char> &out, size_t &len) {
    len = 0;
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    unsigned char magic[2] = {0, 0};
    size_t got = 0;
    while (got < 2) {                                         // raw reads: stdio would buffer past the magic
        const ssize_t n = ::read(fd, magic + got, 2 - got);
        if (n <= 0) break;
        got += size_t(n);
    }
    const bool gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    struct stat st;
    const bool regular = ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
    if (!gz && regular) {
        const size_t sz = size_t(st.st_size);
        if (out.size() < sz + 1) out.resize(sz + 1);
        size_t off = 0;
        while (off < sz) {
            const ssize_t n = ::pread(fd, out.data() + off, sz - off, off_t(off));
            if (n <= 0) break;
            off += size_t(n);
        }
        ::close(fd);
        len = off;
        return off == sz;
    }
    // gzip member, or a non-regular plain input (FIFO, process substitution): through zlib on the descriptor
    // that is ALREADY open -- reopening by name would lose the bytes read above.  zlib must see a gzip stream
    // from its first byte: rewind when the file is seekable; a compressed pipe cannot be re-fed, so refuse it.
    const bool seekable = ::lseek(fd, 0, SEEK_SET) == 0;
    if (!seekable && gz) { ::close(fd); return false; }
    gzFile gp = gzdopen(fd, "rb");
    if (!gp) { ::close(fd); return false; }
    gzbuffer(gp, 1 << 20);
    if (!seekable) {                                          // plain pipe: the bytes already consumed come first
        if (out.size() < (1u << 22)) out.resize(1u << 22);
        std::memcpy(out.data(), magic, got);
        len = got;
    }
    bool ok = true;
    for (;;) {
        if (out.size() - len < (1u << 20)) out.resize(std::max<size_t>(out.size() * 2, 1u << 22));
        const int n = gzread(gp, out.data() + len, unsigned(std::min<size_t>(out.size() - len, 1u << 30)));
        if (n < 0) { ok = false; break; }
        if (n == 0) break;
        len += size_t(n);
    }
    if (ok) {
        // a truncated or corrupt member must not be sketched from its readable prefix: gzread
        // returns 0 at a premature end of file too; gzerror / gzeof / gzclose tell the cases apart
        int errnum = Z_OK;
        (void)gzerror(gp, &errnum);
        if ((errnum != Z_OK && errnum != Z_STREAM_END) || !gzeof(gp)) ok = false;
    }
    if (gzclose(gp) != Z_OK) ok = false;                      // Z_BUF_ERROR: input ended inside a deflate stream
    return ok;
}

extern "C" {

int d2g_seqpack_create(int k, d2g_seqpack **out) {
    if (!out) return D2G_ERR_INVALID;
    if (k < 1 || k > 32) return D2G_ERR_UNSUPPORTED;   // k > 32 is the reference's rolling-hash path (fastxsketch.cpp:420)
    auto *sp = new (std::nothrow) d2g_seqpack();
    if (!sp) return D2G_ERR_NOMEM;
    sp->k = k;
    *out = sp;
    return D2G_OK;
}
void d2g_seqpack_destroy(d2g_seqpack *sp) { delete sp; }
void d2g_seqpack_clear(d2g_seqpack *sp) {          // keep every allocation, forget the content
    if (!sp) return;
    sp->nbases = 0; sp->cur_start = sp->cur_len = sp->cur_kmers = 0; sp->padded_bytes = 0;
    sp->run_start.clear(); sp->run_len.clear(); sp->genome_nkmers.clear(); sp->names.clear();
    sp->genome_run_off.assign(1, 0);
}

int d2g_seqpack_add_path(d2g_seqpack *sp, const char *line) {
    if (!sp || !line) return D2G_ERR_INVALID;
    sp->unpad();
    // d2.h:52-71 for_each_substr: space-separated sub-paths feed one sketch
    std::string s(line);
    size_t b = 0;
    static thread_local std::vector<char> buf;        // reused across files: no per-file allocation / zero fill
    int rc = D2G_OK;
    while (b <= s.size()) {
        size_t e = s.find(' ', b);
        if (e == std::string::npos) e = s.size();
        if (e > b) {
            const std::string sub = s.substr(b, e - b);
            size_t len = 0;
            if (!slurp(sub.c_str(), buf, len)) { rc = D2G_ERR_IO; break; }
            sp->reserve_bases(len);
            sp->feed_fastx(buf.data(), len);
        }
        b = e + 1;
    }
    sp->end_genome();
    return rc;
}
// --parse-by-seq: one genome per record of the (space-separated) files of `line`
int d2g_seqpack_add_path_by_record(d2g_seqpack *sp, const char *line) {
    if (!sp || !line) return D2G_ERR_INVALID;
    std::string s(line);
    size_t b = 0;
    static thread_local std::vector<char> buf;
    int rc = D2G_OK;
    sp->by_record = true;
    while (b <= s.size()) {
        size_t e = s.find(' ', b);
        if (e == std::string::npos) e = s.size();
        if (e > b) {
            const std::string sub = s.substr(b, e - b);
            size_t len = 0;
            if (!slurp(sub.c_str(), buf, len)) { rc = D2G_ERR_IO; break; }
            sp->reserve_bases(len);
            sp->feed_fastx(buf.data(), len);
        }
        b = e + 1;
    }
    sp->by_record = false;
    return rc;
}
int d2g_seqpack_add_fastx_by_record(d2g_seqpack *sp, const char *buf, size_t len) {
    if (!sp || (!buf && len)) return D2G_ERR_INVALID;
    sp->by_record = true;
    sp->feed_fastx(buf, len);
    sp->by_record = false;
    return D2G_OK;
}
const char *d2g_seqpack_name(const d2g_seqpack *sp, size_t g) { return sp && g < sp->names.size() ? sp->names[g].c_str() : ""; }
int d2g_seqpack_add_fastx(d2g_seqpack *sp, const char *buf, size_t len) {
    if (!sp || (!buf && len)) return D2G_ERR_INVALID;
    sp->unpad();
    sp->feed_fastx(buf, len);
    sp->end_genome();
    return D2G_OK;
}
int d2g_seqpack_add_sequence(d2g_seqpack *sp, const char *seq, size_t len) {
    if (!sp || (!seq && len)) return D2G_ERR_INVALID;
    sp->unpad();
    sp->feed(seq, len);
    sp->end_genome();
    return D2G_OK;
}
size_t d2g_seqpack_ngenomes(const d2g_seqpack *sp) { return sp->genome_run_off.size() - 1; }
size_t d2g_seqpack_nruns(const d2g_seqpack *sp) { return sp->run_start.size(); }
size_t d2g_seqpack_packed_bytes(const d2g_seqpack *sp) { const_cast<d2g_seqpack *>(sp)->finalize_pad(); return sp->padded_bytes; }
const uint8_t *d2g_seqpack_packed(const d2g_seqpack *sp) { const_cast<d2g_seqpack *>(sp)->finalize_pad(); return sp->packed.data(); }
const uint64_t *d2g_seqpack_run_start(const d2g_seqpack *sp) { return sp->run_start.data(); }
const uint32_t *d2g_seqpack_run_len(const d2g_seqpack *sp) { return sp->run_len.data(); }
const uint64_t *d2g_seqpack_genome_run_off(const d2g_seqpack *sp) { return sp->genome_run_off.data(); }
uint64_t d2g_seqpack_nkmers(const d2g_seqpack *sp, size_t g) { return g < sp->genome_nkmers.size() ? sp->genome_nkmers[g] : 0; }
uint64_t d2g_seqpack_nbases(const d2g_seqpack *sp) { return sp->nbases; }

}  // extern "C"
