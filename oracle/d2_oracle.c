/*
 * d2_oracle.c -- CPU restatement of dashing2's sketch + all-pairs cmp hot paths.
 * TEST INFRASTRUCTURE ONLY (see d2_oracle.h header for the pin status).
 * Plain C11 + OpenMP + zlib.  x86-64 only: relies on 80-bit x87 `long double`
 * exactly where the reference does (oph.h:240-263, cmp_core.cpp:355-489).
 */
#define _GNU_SOURCE
#include "d2_oracle.h"
#include <float.h>
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* primitives                                                          */
/* ------------------------------------------------------------------ */

/* UNVERIFIED-AGAINST-SOURCE: sketch::hash::WangHash::hash (dnbaker/sketch hash.h, absent).
 * Thomas Wang's 64-bit integer mix as published; bijective, inverse below. */
uint64_t d2o_wang_hash(uint64_t key) {
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

uint64_t d2o_wang_inverse(uint64_t key) {
    uint64_t tmp;
    /* invert key = key + (key << 31) */
    tmp = key - (key << 31);
    key = key - (tmp << 31);
    /* invert key = key ^ (key >> 28) */
    tmp = key ^ key >> 28;
    key = key ^ tmp >> 28;
    /* invert key *= 21 */
    key *= 14933078535860113213ull;
    /* invert key = key ^ (key >> 14) */
    tmp = key ^ key >> 14;
    tmp = key ^ tmp >> 14;
    tmp = key ^ tmp >> 14;
    key = key ^ tmp >> 14;
    /* invert key *= 265 */
    key *= 15244667743933553977ull;
    /* invert key = key ^ (key >> 24) */
    tmp = key ^ key >> 24;
    key = key ^ tmp >> 24;
    /* invert key = (~key) + (key << 21) */
    tmp = ~key;
    tmp = ~(key - (tmp << 21));
    tmp = ~(key - (tmp << 21));
    key = ~(key - (tmp << 21));
    return key;
}

/* std::mt19937_64 (ISO C++ [rand.predef]); oph.h:59 uses its first output as seed_. */
uint64_t d2o_mt19937_64_first(uint64_t seed) {
    enum { NN = 312, MM = 156 };
    static const uint64_t MATRIX_A = 0xB5026F5AA96619E9ull, UM = 0xFFFFFFFF80000000ull, LM = 0x7FFFFFFFull;
    uint64_t mt[NN];
    mt[0] = seed;
    for (int i = 1; i < NN; ++i) mt[i] = 6364136223846793005ull * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i;
    /* one full twist */
    for (int i = 0; i < NN; ++i) {
        uint64_t x = (mt[i] & UM) | (mt[(i + 1) % NN] & LM);
        uint64_t xa = x >> 1;
        if (x & 1ull) xa ^= MATRIX_A;
        mt[i] = mt[(i + MM) % NN] ^ xa;
    }
    uint64_t x = mt[0];
    x ^= (x >> 29) & 0x5555555555555555ull;
    x ^= (x << 17) & 0x71D67FFFEDA60000ull;
    x ^= (x << 37) & 0xFFF7EEE000000000ull;
    x ^= (x >> 43);
    return x;
}

/* enums.cpp:131-140 */
uint64_t d2o_seed_mask(uint64_t seedseed) { return seedseed == 0 ? 0 : d2o_wang_hash(seedseed); }
uint64_t d2o_default_xormask(void) { return 0x724526e320f9967dull; }

/* oph.h:44-53 BHasher = Wang(x ^ 0x533f8c2151b20f97); oph.h:59 DHasher xors seed_ first;
 * oph.h:142 seed = 0x321b919a61cb41f7. */
uint64_t d2o_oph_xor_const(void) {
    static uint64_t c = 0;
    if (!c) c = d2o_mt19937_64_first(0x321b919a61cb41f7ull) ^ 0x533f8c2151b20f97ull;
    return c;
}

uint64_t d2o_maskfn(uint64_t x, uint64_t xormask) { return d2o_wang_hash(x ^ xormask); }
uint64_t d2o_oph_id(uint64_t masked) { return d2o_wang_hash(masked ^ d2o_oph_xor_const()); }

/* ssi.h:26-36 (in-tree twin of wy::wyhash64_stateless) */
static inline uint64_t wymum(uint64_t x, uint64_t y) {
    __uint128_t l = x;
    l *= y;
    return (uint64_t)(l ^ (l >> 64));
}
uint64_t d2o_wyhash64_stateless(uint64_t *seed) {
    *seed += 0x60bee2bee120fc15ull;
    return wymum(*seed ^ 0xe7037ed1a0b428dbull, *seed);
}

/* ------------------------------------------------------------------ */
/* OPH sketch                                                          */
/* ------------------------------------------------------------------ */

int d2o_oph_init(d2o_oph *s, size_t sketchsize) {
    size_t m = sketchsize;
    if (m & 1) ++m;                                  /* oph.h:145 (pow2=false) */
    s->m = m;
    s->regs = (uint64_t *)malloc(m * sizeof(uint64_t));
    s->counts = (double *)malloc(m * sizeof(double));
    if (!s->regs || !s->counts) return -1;
    d2o_oph_reset(s);
    return 0;
}
void d2o_oph_free(d2o_oph *s) { free(s->regs); free(s->counts); s->regs = 0; s->counts = 0; }
void d2o_oph_reset(d2o_oph *s) {                     /* oph.h:232-239 */
    for (size_t i = 0; i < s->m; ++i) s->regs[i] = ~0ull;
    memset(s->counts, 0, s->m * sizeof(double));
    s->total_updates = 0;
}

/* UNVERIFIED-AGAINST-SOURCE: schism::Schismatic<uint32_t>::mod(size_t) narrows its argument
 * to uint32_t, then reduces modulo m (fastmod == exact %).  For m | 2^32 (every power of two,
 * i.e. every BASELINE config) this equals id % m under either reading. */
uint32_t d2o_oph_bucket(uint64_t id, size_t m) { return (uint32_t)id % (uint32_t)m; }

void d2o_oph_update(d2o_oph *s, uint64_t oid) {      /* oph.h:176-211, mincount_ <= 1 branch */
    ++s->total_updates;
    const uint64_t id = d2o_oph_id(oid);
    const size_t idx = d2o_oph_bucket(id, s->m);
    if (s->regs[idx] > id) { s->regs[idx] = id; s->counts[idx] = 1.; }
    else s->counts[idx] += (s->regs[idx] == id);
}

/* test/oph.cpp:12: for(i < exp) l.update(i) */
void d2o_oph_update_range(d2o_oph *s, uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) d2o_oph_update(s, i);
}

double d2o_regs_getcard(const uint64_t *regs, size_t m) {  /* oph.h:240-247 */
    long double sum = 0.L;
    for (size_t i = 0; i < m; ++i) sum = sum + regs[i] * 0x1p-64L;
    if (!sum) return INFINITY;
    return m * (m / sum);
}

void d2o_regs_data(const uint64_t *regs, size_t m, double *sig) {  /* oph.h:248-263 */
    size_t nmax = 0;
    for (size_t i = 0; i < m; ++i) nmax += (regs[i] == UINT64_MAX);
    /* `-SigT(1) / (m_ - count)` is a double division, then widened */
    const long double mul = -1.0 / (double)(m - nmax);
    for (size_t i = 0; i < m; ++i) {
        const uint64_t x = regs[i];
        if (x == UINT64_MAX || x == 0) sig[i] = 0.;
        else sig[i] = (double)(mul * logl(0x1p-64L * (UINT64_MAX - x + 1)));
    }
}
double d2o_oph_getcard(const d2o_oph *s) { return d2o_regs_getcard(s->regs, s->m); }
void   d2o_oph_data(const d2o_oph *s, double *sig) { d2o_regs_data(s->regs, s->m, sig); }

/* ------------------------------------------------------------------ */
/* encoder + FASTX parsing                                             */
/* ------------------------------------------------------------------ */

/* UNVERIFIED-AGAINST-SOURCE: bns::Encoder<>::for_each for unspaced k <= 32 DNA
 * (fastxsketch.cpp:411-421): A0 C1 G2 T3 case-insensitive, newest base in the low bits,
 * revcomp = reversed complement, canonical = numeric min, any other byte restarts the window. */
static inline int base_code(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

size_t d2o_encode_seq(const char *seq, size_t len, int k, int canon, d2o_kmer_cb cb, void *ud) {
    if (k < 1 || k > 32) return 0;
    const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const int rcshift = 2 * (k - 1);
    uint64_t fwd = 0, rc = 0;
    int filled = 0;
    size_t n = 0;
    for (size_t i = 0; i < len; ++i) {
        const int c = base_code((unsigned char)seq[i]);
        if (c < 0) { filled = 0; fwd = rc = 0; continue; }
        fwd = ((fwd << 2) | (uint64_t)c) & mask;
        rc = (rc >> 2) | ((uint64_t)(3 - c) << rcshift);
        if (++filled >= k) {
            cb(canon ? (fwd < rc ? fwd : rc) : fwd, ud);
            ++n;
        }
    }
    return n;
}

/* UNVERIFIED-AGAINST-SOURCE: kseq.h (klib, vendored by the absent bonsai submodule; call site fastxsketch.cpp:416-417 through
 * bns::Encoder::for_each, holder d2.h:273-305).  Restated from the published klib kseq_read():
 *  - with last_char == 0 (file start, and after every FASTQ record) the reader scans BYTE BY BYTE for the next '>' or '@' -- a
 *    header may start in the middle of a line ("junk>name") --; otherwise the header character was already consumed by the
 *    previous record's sequence loop, which only looks at the FIRST character of each line;
 *  - name = up to the first whitespace, the rest of the header line is the comment; a header character that is the last byte of
 *    the input yields no record;
 *  - the sequence is every following line (line feed stripped; a trailing '\r' stripped while the accumulated sequence is longer
 *    than one character -- ks_getuntil2's `str->l > 1`) until a line starts with '>', '@' or '+'; empty lines are skipped;
 *  - after '+': the rest of that line is skipped, then quality lines are read -- at least one -- until qual.l >= seq.l; the record
 *    is an ERROR (-2) when the input ends inside the '+' line or when qual.l != seq.l, and the encoder's loop
 *    `while (kseq_read(ks) >= 0)` then stops: the bad record is not sketched and the rest of the file is ignored. */
size_t d2o_walk_fastx_records(const char *buf, size_t len, d2o_record_cb rcb, void *ud) {
    size_t pos = 0, nrec = 0;
    char *seq = NULL; size_t cap = 0;
    int last_char = 0;
    for (;;) {
        if (last_char == 0) {
            while (pos < len && buf[pos] != '>' && buf[pos] != '@') ++pos;
            if (pos >= len) break;
            last_char = (unsigned char)buf[pos++];
        }
        if (pos >= len) break;                              /* ks_getuntil(name) meets the end of the input: -1 */
        /* header line: name = up to the first whitespace (kseq name/comment split) */
        const char *nl = memchr(buf + pos, '\n', len - pos);
        const size_t hend = nl ? (size_t)(nl - buf) : len;
        const char *name = buf + pos;
        size_t name_len = 0;
        while (pos + name_len < hend && !isspace((unsigned char)name[name_len])) ++name_len;
        pos = nl ? hend + 1 : len;
        size_t sl = 0;
        int c = -1;
        while (pos < len) {
            c = (unsigned char)buf[pos];
            if (c == '>' || c == '+' || c == '@') { ++pos; break; }
            if (c == '\n') { ++pos; c = -1; continue; }
            nl = memchr(buf + pos, '\n', len - pos);
            size_t e = nl ? (size_t)(nl - buf) : len;
            size_t ll = e - pos;
            if (sl + ll + 1 > cap) { cap = (sl + ll + 1) * 2; seq = (char *)realloc(seq, cap); }
            memcpy(seq + sl, buf + pos, ll);
            sl += ll;
            if (sl > 1 && seq[sl - 1] == '\r') --sl;
            pos = nl ? e + 1 : len;
            c = -1;
        }
        if (c == '>' || c == '@') last_char = c;           /* the next record's header character has been read */
        if (c != '+') {                                    /* FASTA record (or the input ended) */
            rcb(name, name_len, seq, sl, ud);
            ++nrec;
            if (c == -1) break;
            continue;
        }
        nl = memchr(buf + pos, '\n', len - pos);           /* rest of the '+' line */
        if (!nl) break;                                    /* -2: no quality string */
        pos = (size_t)(nl - buf) + 1;
        size_t ql = 0;
        do {
            if (pos >= len) break;                         /* ks_getuntil2 at the end of the input: -1, the loop ends */
            nl = memchr(buf + pos, '\n', len - pos);
            size_t e = nl ? (size_t)(nl - buf) : len;
            size_t ll = e - pos;
            ql += ll;
            if (ql > 1 && ll && buf[pos + ll - 1] == '\r') --ql;
            pos = nl ? e + 1 : len;
        } while (ql < sl);
        last_char = 0;
        if (ql != sl) break;                               /* -2: quality string of a different length */
        rcb(name, name_len, seq, sl, ud);
        ++nrec;
    }
    free(seq);
    return nrec;
}

typedef struct { int k, canon; d2o_kmer_cb cb; void *ud; size_t total; } enc_rec_ctx;
static void enc_rec_cb(const char *name, size_t name_len, const char *seq, size_t seq_len, void *ud) {
    enc_rec_ctx *e = (enc_rec_ctx *)ud;
    (void)name; (void)name_len;
    e->total += d2o_encode_seq(seq, seq_len, e->k, e->canon, e->cb, e->ud);
}
size_t d2o_encode_fastx_buffer(const char *buf, size_t len, int k, int canon, d2o_kmer_cb cb, void *ud) {
    enc_rec_ctx e = { k, canon, cb, ud, 0 };
    d2o_walk_fastx_records(buf, len, enc_rec_cb, &e);
    return e.total;
}

typedef struct { d2o_oph *s; uint64_t xormask; } upd_ctx;
static void upd_cb(uint64_t kmer, void *ud) {        /* fastxsketch.cpp:389 lfunc2 -> :565 update */
    upd_ctx *u = (upd_ctx *)ud;
    d2o_oph_update(u->s, d2o_maskfn(kmer, u->xormask));
}

static int finish_sketch(d2o_oph *s, size_t sketchsize, uint64_t *regs_out, double *sig_out,
                         double *card_out, uint64_t *nkmers_out) {
    if (card_out) *card_out = d2o_oph_getcard(s);            /* fastxsketch.cpp:567 */
    if (sig_out) {
        double *tmp = (double *)malloc(s->m * sizeof(double));
        if (!tmp) return -1;
        d2o_oph_data(s, tmp);                                /* fastxsketch.cpp:586 */
        memcpy(sig_out, tmp, sketchsize * sizeof(double));   /* :605,:610 copy first S only */
        free(tmp);
    }
    if (regs_out) memcpy(regs_out, s->regs, s->m * sizeof(uint64_t));
    if (nkmers_out) *nkmers_out = s->total_updates;
    return 0;
}

int d2o_sketch_buffer(const char *buf, size_t len, int k, int canon, uint64_t xormask,
                      size_t sketchsize, uint64_t *regs_out, double *sig_out, double *card_out,
                      uint64_t *nkmers_out) {
    d2o_oph s;
    if (d2o_oph_init(&s, sketchsize)) return -1;
    upd_ctx u = { &s, xormask };
    d2o_encode_fastx_buffer(buf, len, k, canon, upd_cb, &u);
    int rc = finish_sketch(&s, sketchsize, regs_out, sig_out, card_out, nkmers_out);
    d2o_oph_free(&s);
    return rc;
}

char *d2o_slurp(const char *path, size_t *len_out) {
    gzFile fp = gzopen(path, "rb");
    if (!fp) return NULL;
    size_t cap = 1 << 20, len = 0;
    char *buf = (char *)malloc(cap);
    for (;;) {
        if (cap - len < (1 << 19)) { cap <<= 1; buf = (char *)realloc(buf, cap); }
        int n = gzread(fp, buf + len, (unsigned)(cap - len > (1u << 30) ? (1u << 30) : cap - len));
        if (n <= 0) break;
        len += (size_t)n;
    }
    gzclose(fp);
    *len_out = len;
    return buf;
}

/* one "line" of paths: space-separated sub-paths feed the same sketch (d2.h:52-71 for_each_substr) */
int d2o_sketch_file(const char *path, int k, int canon, uint64_t xormask, size_t sketchsize,
                    uint64_t *regs_out, double *sig_out, double *card_out, uint64_t *nkmers_out) {
    d2o_oph s;
    if (d2o_oph_init(&s, sketchsize)) return -1;
    upd_ctx u = { &s, xormask };
    char *line = strdup(path);
    int rc = 0;
    for (char *save = NULL, *tok = strtok_r(line, " ", &save); tok; tok = strtok_r(NULL, " ", &save)) {
        size_t len = 0;
        char *buf = d2o_slurp(tok, &len);
        if (!buf) { rc = -2; break; }
        d2o_encode_fastx_buffer(buf, len, k, canon, upd_cb, &u);
        free(buf);
    }
    free(line);
    if (!rc) rc = finish_sketch(&s, sketchsize, regs_out, sig_out, card_out, nkmers_out);
    d2o_oph_free(&s);
    return rc;
}

typedef struct { size_t size; size_t idx; } fsz_t;
static int fsz_desc(const void *a, const void *b) {   /* std::greater<pair<size,idx>> */
    const fsz_t *x = (const fsz_t *)a, *y = (const fsz_t *)b;
    if (x->size != y->size) return x->size > y->size ? -1 : 1;
    if (x->idx != y->idx) return x->idx > y->idx ? -1 : 1;
    return 0;
}

int d2o_sketch_files(const char *const *paths, size_t n, int k, int canon, uint64_t xormask,
                     size_t sketchsize, double *sigs_out, double *cards_out, int nthreads) {
    fsz_t *fs = (fsz_t *)malloc(n * sizeof(fsz_t));
    for (size_t i = 0; i < n; ++i) {                 /* sketch_core.cpp:175-184 */
        struct stat st;
        fs[i].size = stat(paths[i], &st) == 0 ? (size_t)st.st_size : 0;
        fs[i].idx = i;
    }
    qsort(fs, n, sizeof(fsz_t), fsz_desc);
    int err = 0;
    (void)nthreads;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    #pragma omp parallel for schedule(dynamic)
#endif
    for (size_t i = 0; i < n; ++i) {                 /* fastxsketch.cpp:302 */
        const size_t myind = fs[i].idx;
        int rc = d2o_sketch_file(paths[myind], k, canon, xormask, sketchsize, NULL,
                                 sigs_out + myind * sketchsize, cards_out + myind, NULL);
        if (rc) {
#ifdef _OPENMP
            #pragma omp atomic write
#endif
            err = rc;
        }
    }
    free(fs);
    return err;
}

/* ------------------------------------------------------------------ */
/* cmp                                                                 */
/* ------------------------------------------------------------------ */

size_t d2o_densify(double *sig, size_t S) {          /* cmp_core.cpp:577-613, empty = 0 */
    size_t nz = 0;
    for (size_t i = 0; i < S; ++i) nz += (sig[i] == 0.);
    if (nz == S) return S;
    size_t ne = 0;
    double *tmp = (double *)malloc(S * sizeof(double));
    memcpy(tmp, sig, S * sizeof(double));
    for (size_t i = 0; i < S; ++i) {
        if (sig[i] != 0.) continue;
        ++ne;
        uint64_t rng_i = i + 0x5bf2b8bdf07c06cull;
        uint64_t j;
        do {
            j = d2o_wyhash64_stateless(&rng_i) % (uint64_t)S;   /* Schismatic<uint64_t>::mod */
        } while (sig[j] == 0.);
        tmp[i] = sig[j];
    }
    memcpy(sig, tmp, S * sizeof(double));
    free(tmp);
    return ne;
}

/* UNVERIFIED-AGAINST-SOURCE: sketch::eq::count_gtlt returns {#(a>b), #(a<b)}; the identity
 * S - gt - lt == #equal is pinned by the assert at cmp_core.cpp:465. */
void d2o_count_gtlt(const double *a, const double *b, size_t n, uint64_t *gt, uint64_t *lt) {
    uint64_t g = 0, l = 0;
    for (size_t i = 0; i < n; ++i) { g += a[i] > b[i]; l += a[i] < b[i]; }
    *gt = g; *lt = l;
}
uint64_t d2o_count_eq(const double *a, const double *b, size_t n) {
    uint64_t e = 0;
    for (size_t i = 0; i < n; ++i) e += a[i] == b[i];
    return e;
}

static inline double sim2dist(float x, int k) {       /* cmp_core.cpp:361 */
    const double poisson_mult = -1. / (k > 1 ? k : 1);
    if (x) return log(2. * x / (1. + x)) * poisson_mult;
    return INFINITY;
}

static inline float finish_ret(long double ret) {     /* cmp_core.cpp:573-575 */
    if (isnan(ret) || isinf(ret)) ret = LDBL_MAX;
    return (float)ret;                                 /* LDBL_MAX -> +inf in float */
}

float d2o_compare_from_gtlt(uint64_t gt, uint64_t lt, size_t S, double lhc, double rhc, int measure, int k) {
    /* cmp_core.cpp:355-356,461-489 */
    long double ret = FLT_MAX;
    const long double invdenom = 1.L / S;
    long double alpha, beta, eq, lhcard, ucard, rhcard;
    alpha = gt * invdenom;
    beta = lt * invdenom;
    lhcard = lhc; rhcard = rhc;
    eq = (1. - alpha - beta);
    {   /* std::max(a, 0.L) == (a < 0.L) ? 0.L : a  (a NaN stays NaN) */
        const long double a = (lhcard + rhcard) / (2.L - alpha - beta);
        ucard = (a < 0.L) ? 0.L : a;
    }
    if (eq <= 0.) return measure != D2O_POISSON_LLR ? 0.f : INFINITY; /* (float)DBL_MAX */
    static const long double EPS = 1e-15;             /* double literal widened, as in the reference */
    if (eq <= EPS) eq = 0;
    const float isz = (float)(ucard * eq), sim = (float)eq;
    switch (measure) {
        case D2O_SIMILARITY: ret = sim; break;
        case D2O_INTERSECTION: ret = isz; break;
        case D2O_CONTAINMENT: ret = isz / rhcard; break;
        case D2O_SYMMETRIC_CONTAINMENT: ret = isz / (lhcard < rhcard ? lhcard : rhcard); break;
        case D2O_POISSON_LLR: ret = sim2dist(sim, k); break;
        case D2O_UNION_SIZE: ret = lhcard + rhcard - isz; break;
        default: ret = -1.f; break;
    }
    return finish_ret(ret);
}

float d2o_compare_from_neq(uint64_t neq, size_t S, double lhc, double rhc, int measure, int k) {
    /* cmp_core.cpp:495-517 */
    const long double lhcard = lhc, rhcard = rhc;
    const long double invdenom = 1.L / S;
    long double ret = invdenom * neq;
#define UC() ({ long double a_ = (lhcard + rhcard) / (1.L + ret); (a_ < 0.L) ? 0.L : a_; })
    if (measure == D2O_INTERSECTION) ret *= UC();
    else if (measure == D2O_SYMMETRIC_CONTAINMENT) ret *= UC() / (lhcard < rhcard ? lhcard : rhcard);
    else if (measure == D2O_CONTAINMENT) ret *= UC() / lhcard;
    else if (measure == D2O_POISSON_LLR) {
        /* sim2dist takes `auto x` = long double here */
        const double poisson_mult = -1. / (k > 1 ? k : 1);
        if (ret) ret = (double)(logl(2. * ret / (1. + ret)) * poisson_mult);   /* -> double return of the lambda */
        else ret = INFINITY;
    } else if (measure == D2O_UNION_SIZE) {
        const long double isz = ret * UC();
        ret = (lhcard + rhcard - isz);
    }
#undef UC
    return finish_ret(ret);
}

float d2o_compare(const double *sigs, const double *cards, size_t S, size_t i, size_t j, int measure, int k) {
    uint64_t gt, lt;
    d2o_count_gtlt(sigs + S * i, sigs + S * j, S, &gt, &lt);
    return d2o_compare_from_gtlt(gt, lt, S, cards[i], cards[j], measure, k);
}

size_t d2o_default_batchsize(size_t batch_size, size_t S, unsigned nthreads) {  /* cmp_main.cpp:370-388 */
    if (batch_size == 0) {
        size_t b = (size_t)(0x400000 / S / 8.);
        batch_size = b > 1 ? b : 1;
    }
    unsigned nt = nthreads > 1 ? nthreads : 1;
    if (batch_size > nt) batch_size = nthreads;
    return batch_size;
}

void d2o_allpairs_ut_rows(const double *sigs, const double *cards, size_t N, size_t S, int measure, int k,
                          size_t r0, size_t r1, float *out, int nthreads, size_t batch) {
    /* emitrect.cpp:198,290-323: batches of `batch_size` rows, omp dynamic over the rows of a batch */
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
    omp_set_num_threads(nthreads);
#endif
    size_t bs = batch < (size_t)nthreads ? batch : (size_t)nthreads;
    if (bs < 1) bs = 1;
    size_t base = 0;
    for (size_t firstrow = r0; firstrow < r1; firstrow += bs) {
        const size_t erow = firstrow + bs < r1 ? firstrow + bs : r1;
        size_t *offsets = (size_t *)malloc((erow - firstrow + 1) * sizeof(size_t));
        offsets[0] = 0;
        for (size_t fs = firstrow; fs < erow; ++fs) offsets[fs - firstrow + 1] = offsets[fs - firstrow] + (N - fs - 1);
#ifdef _OPENMP
        #pragma omp parallel for schedule(dynamic)
#endif
        for (size_t fs = firstrow; fs < erow; ++fs) {
            float *datp = out + base + offsets[fs - firstrow] - fs - 1;
            for (size_t j = fs + 1; j < N; ++j) datp[j] = d2o_compare(sigs, cards, S, fs, j, measure, k);
        }
        base += offsets[erow - firstrow];
        free(offsets);
    }
}

void d2o_allpairs_ut(const double *sigs, const double *cards, size_t N, size_t S, int measure, int k,
                     float *out, int nthreads, size_t batch) {
    d2o_allpairs_ut_rows(sigs, cards, N, S, measure, k, 0, N, out, nthreads, batch);
}

/* rows [r0, r1) of the condensed upper triangle (test infrastructure for the full-size parity tests: a few hundred rows of a
 * 10 000- or 50 000-sketch matrix instead of all of them) */
void d2o_eqcounts_ut_rows(const double *sigs, size_t N, size_t S, size_t r0, size_t r1, uint32_t *neq_out) {
    size_t idx = 0;
    for (size_t i = r0; i < r1 && i < N; ++i)
        for (size_t j = i + 1; j < N; ++j)
            neq_out[idx++] = (uint32_t)d2o_count_eq(sigs + S * i, sigs + S * j, S);
}

void d2o_eqcounts_ut(const double *sigs, size_t N, size_t S, uint32_t *neq_out) {
    size_t idx = 0;
    for (size_t i = 0; i < N; ++i)
        for (size_t j = i + 1; j < N; ++j)
            neq_out[idx++] = (uint32_t)d2o_count_eq(sigs + S * i, sigs + S * j, S);
}
