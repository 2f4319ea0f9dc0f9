/*
 * d2_bmh_oracle.c -- CPU oracle for the --multiset path: exact k-mer counting (R11) and
 * BagMinHash (R12).  TEST INFRASTRUCTURE ONLY (see d2_oracle.h).
 *
 * R12 is PARITY UNPINNED against a real dashing2 binary: sketch/bmh.h is absent from
 * /root/reference.  This file restates the published algorithm (Ertl, KDD 2018) under the
 * "BMH-D2G" spec of DESIGN.md; processes are explored in time order with a binary heap, as
 * published.
 */
#include "d2_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* deterministic log                                                    */
/* ------------------------------------------------------------------ */
static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

double d2o_dlog(double u) {
    /* constants of the classic fdlibm e_log.c kernel; accuracy (< 1 ulp) is checked against libm
     * in tests/test_oracle_bmh.py, which is what pins them */
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    const uint64_t b = d2u(u);
    int e = (int)(b >> 52) - 1023;
    double m = u2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);    /* [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }                     /* (0.707, 1.414] */
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double R = z * (Lg1 + z * (Lg2 + z * (Lg3 + z * (Lg4 + z * (Lg5 + z * (Lg6 + z * Lg7))))));
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* ------------------------------------------------------------------ */
/* BMH-D2G                                                              */
/* ------------------------------------------------------------------ */
typedef struct {
    uint64_t p, q;       /* weight-level range [V(p), V(q)) as double bit patterns */
    double x;            /* time of the process's current point */
    uint64_t rng;        /* wyhash64_stateless state */
    uint32_t i;          /* register the current point belongs to */
} proc_t;

struct d2o_bmh {
    size_t m;            /* registers */
    size_t leaves;       /* power of two >= m */
    double *tree;        /* max-tree: tree[leaves + i] = h_i, tree[n] = max(tree[2n], tree[2n+1]) */
    double total_weight;
    proc_t *heap; size_t nheap, capheap;
    uint64_t *owner;     /* BagMinHash2::ids(): the update() call (its `tag`) that set each register; ~0 = untouched */
    uint64_t cur_tag;    /* tag of the update in progress */
};

static inline double V(uint64_t l) { return u2d(l); }
static inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((__uint128_t)a * b) >> 64); }

static void proc_next(proc_t *P, size_t m) {
    const uint64_t r1 = d2o_wyhash64_stateless(&P->rng);
    const double uu = (double)((r1 >> 11) + 1) * 0x1p-53;        /* (0, 1] */
    const double E = -d2o_dlog(uu);
    P->x = P->x + E / (V(P->q) - V(P->p));
    const uint64_t r2 = d2o_wyhash64_stateless(&P->rng);
    P->i = (uint32_t)mulhi64(r2, (uint64_t)m);
}

static void heap_push(d2o_bmh *b, const proc_t *P) {
    if (b->nheap == b->capheap) {
        b->capheap = b->capheap ? b->capheap * 2 : 128;
        b->heap = (proc_t *)realloc(b->heap, b->capheap * sizeof(proc_t));
    }
    size_t c = b->nheap++;
    while (c > 0) {
        size_t par = (c - 1) >> 1;
        if (b->heap[par].x <= P->x) break;
        b->heap[c] = b->heap[par];
        c = par;
    }
    b->heap[c] = *P;
}
static proc_t heap_pop(d2o_bmh *b) {
    proc_t top = b->heap[0];
    proc_t last = b->heap[--b->nheap];
    size_t c = 0;
    for (;;) {
        size_t l = 2 * c + 1, r = l + 1, s = c;
        double best = last.x;
        if (l < b->nheap && b->heap[l].x < best) { s = l; best = b->heap[l].x; }
        if (r < b->nheap && b->heap[r].x < best) { s = r; }
        if (s == c) break;
        b->heap[c] = b->heap[s];
        c = s;
    }
    if (b->nheap) b->heap[c] = last;
    return top;
}

static inline double hmax(const d2o_bmh *b) { return b->tree[1]; }
static void reg_update(d2o_bmh *b, uint32_t i, double x) {
    size_t n = b->leaves + i;
    if (!(x < b->tree[n])) return;              /* strict: on an exact tie the earlier update keeps the register */
    b->tree[n] = x;
    b->owner[i] = b->cur_tag;
    for (n >>= 1; n >= 1; n >>= 1) {
        const double a = b->tree[2 * n], c = b->tree[2 * n + 1];
        const double mx = a > c ? a : c;
        if (b->tree[n] == mx) break;
        b->tree[n] = mx;
    }
}

d2o_bmh *d2o_bmh_create(size_t sketchsize) {
    if (sketchsize == 0) return NULL;
    d2o_bmh *b = (d2o_bmh *)calloc(1, sizeof(d2o_bmh));
    b->m = sketchsize;
    b->leaves = 1;
    while (b->leaves < sketchsize) b->leaves <<= 1;
    b->tree = (double *)malloc(2 * b->leaves * sizeof(double));
    b->owner = (uint64_t *)malloc(sketchsize * sizeof(uint64_t));
    d2o_bmh_reset(b);
    return b;
}
void d2o_bmh_destroy(d2o_bmh *b) {
    if (!b) return;
    free(b->tree); free(b->heap); free(b->owner); free(b);
}
void d2o_bmh_reset(d2o_bmh *b) {
    for (size_t i = 0; i < b->leaves; ++i) b->tree[b->leaves + i] = i < b->m ? INFINITY : 0.0;
    for (size_t n = b->leaves - 1; n >= 1; --n) {
        const double a = b->tree[2 * n], c = b->tree[2 * n + 1];
        b->tree[n] = a > c ? a : c;
    }
    b->total_weight = 0.;
    b->nheap = 0;
    memset(b->owner, 0xFF, b->m * sizeof(uint64_t));
    b->cur_tag = 0;
}
double d2o_bmh_total_weight(const d2o_bmh *b) { return b->total_weight; }
void d2o_bmh_data(const d2o_bmh *b, double *sig) { memcpy(sig, b->tree + b->leaves, b->m * sizeof(double)); }

/* BMH-D2G top level: [0, 2^53) is cut into 65 fixed strips -- 16 unit strips [t, t+1) for the
 * small integer weights k-mer counts mostly are, then the octaves [2^j, 2^(j+1)) up to 2^53 -- and
 * each strip is an independent Poisson process from time 0 with its own generator (a Poisson
 * process over a union of disjoint strips IS the superposition of independent ones, so no root
 * point is needed).  Inside a strip the published lazy bisection applies, over double bit patterns. */
#define D2O_BMH_NTOP 65
static inline double top_edge(int t) { return t <= 16 ? (double)t : ldexp(1.0, t - 12); }

uint64_t d2o_bmh_update(d2o_bmh *b, uint64_t id, double w) {
    if (!(w > 0.) || !(w <= 0x1p53)) return 0;
    b->total_weight += w;
    uint64_t steps = 0;
    b->nheap = 0;
    for (int t = 0; t < D2O_BMH_NTOP && top_edge(t) < w; ++t) {
        proc_t P = { d2u(top_edge(t)), d2u(top_edge(t + 1)), 0.,
                     id ^ ((uint64_t)(t + 1) * 0xA0761D6478BD642Full) ^ 0x8EBC6AF09C88C6E3ull, 0 };
        proc_next(&P, b->m);
        ++steps;
        if (P.x <= hmax(b)) heap_push(b, &P);
    }
    while (b->nheap) {
        proc_t P = heap_pop(b);
        if (P.x > hmax(b)) break;                  /* time-ordered: everything pending is later still */
        /* locate the point (P.x, P.i): narrow P to the half that holds it, down to one level */
        int counted = 0, relevant = 1;
        for (;;) {
            ++steps;
            if (!counted && V(P.q) <= w) { reg_update(b, P.i, P.x); counted = 1; }
            if (P.q - P.p <= 1) break;
            const uint64_t r = P.p + ((P.q - P.p) >> 1);
            const uint64_t rb = d2o_wyhash64_stateless(&P.rng);
            const double ub = (double)(rb >> 11) * 0x1p-53;                   /* [0, 1) */
            const int left = ub * (V(P.q) - V(P.p)) < (V(r) - V(P.p));
            proc_t S;                              /* the half WITHOUT the point: fresh process from P.x */
            S.x = P.x; S.i = 0;
            S.rng = id ^ (r * 0x9E3779B97F4A7C15ull) ^ 0xD6E8FEB86659FD93ull;
            if (left) { S.p = r; S.q = P.q; P.q = r; }
            else      { S.p = P.p; S.q = r; P.p = r; }
            if (V(S.p) < w) {
                proc_next(&S, b->m);
                if (S.x <= hmax(b)) heap_push(b, &S);
            }
            if (!(V(P.p) < w)) { relevant = 0; break; }
        }
        if (relevant) {                            /* single-level process: next point of the same strip */
            proc_next(&P, b->m);
            if (P.x <= hmax(b)) heap_push(b, &P);
        }
    }
    return steps;
}

int d2o_bmh_from_weighted(const uint64_t *ids, const double *w, size_t n, size_t sketchsize,
                          double *sig_out, double *total_weight_out) {
    return d2o_bmh_from_weighted_ids(ids, w, n, sketchsize, sig_out, total_weight_out, NULL);
}

/* + BagMinHash2::ids() as wsketch.cpp:66-67 uses it: owner_out[r] = position i of the element whose point
 * register r holds (elements are fed in position order, so a strict `<` keeps the smaller position on a tie) */
int d2o_bmh_from_weighted_ids(const uint64_t *ids, const double *w, size_t n, size_t sketchsize,
                              double *sig_out, double *total_weight_out, uint64_t *owner_out) {
    d2o_bmh *b = d2o_bmh_create(sketchsize);
    if (!b) return -1;
    for (size_t i = 0; i < n; ++i) { b->cur_tag = i; d2o_bmh_update(b, ids[i], w ? w[i] : 1.0); }
    if (owner_out) memcpy(owner_out, b->owner, sketchsize * sizeof(uint64_t));
    if (sig_out) d2o_bmh_data(b, sig_out);
    if (total_weight_out) *total_weight_out = d2o_bmh_total_weight(b);
    d2o_bmh_destroy(b);
    return 0;
}

/* ------------------------------------------------------------------ */
/* R11: exact counts of maskfn'd k-mers                                 */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t *v; size_t n, cap; uint64_t xormask; } kvec_t;
static void kvec_cb(uint64_t kmer, void *ud) {
    kvec_t *kv = (kvec_t *)ud;
    if (kv->n == kv->cap) {
        kv->cap = kv->cap ? kv->cap * 2 : (1u << 16);
        kv->v = (uint64_t *)realloc(kv->v, kv->cap * sizeof(uint64_t));
    }
    kv->v[kv->n++] = d2o_maskfn(kmer, kv->xormask);      /* fastxsketch.cpp:386 lfunc2: func(maskfn(x)) */
}
static int cmp_u64(const void *a, const void *b) {
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
/* sort + run-length == the content of Counter::c64_ after the adds (counter.h:68-77); the
 * reference's hash-map iteration order is irrelevant to BagMinHash (min is order-free) */
static void kvec_finish(kvec_t *kv, uint64_t **keys, uint32_t **counts, size_t *nd) {
    qsort(kv->v, kv->n, sizeof(uint64_t), cmp_u64);
    size_t d = 0;
    uint32_t *c = (uint32_t *)malloc((kv->n ? kv->n : 1) * sizeof(uint32_t));
    for (size_t i = 0; i < kv->n;) {
        size_t j = i;
        while (j < kv->n && kv->v[j] == kv->v[i]) ++j;
        kv->v[d] = kv->v[i];
        c[d] = (uint32_t)(j - i);
        ++d;
        i = j;
    }
    *keys = kv->v; *counts = c; *nd = d;
}

void d2o_free(void *p) { free(p); }

int d2o_kmer_count_buffer(const char *buf, size_t len, int k, int canon, uint64_t xormask,
                          uint64_t **keys_out, uint32_t **counts_out, size_t *ndistinct_out,
                          uint64_t *nkmers_out) {
    kvec_t kv = { NULL, 0, 0, xormask };
    d2o_encode_fastx_buffer(buf, len, k, canon, kvec_cb, &kv);
    if (nkmers_out) *nkmers_out = kv.n;
    if (!kv.v) kv.v = (uint64_t *)malloc(8);
    kvec_finish(&kv, keys_out, counts_out, ndistinct_out);
    return 0;
}

static int bmh_from_kvec(kvec_t *kv, size_t sketchsize, double thr, double *sig_out, double *tw_out,
                         uint64_t *nkmers_out) {
    if (nkmers_out) *nkmers_out = kv->n;
    if (!kv->v) kv->v = (uint64_t *)malloc(8);
    uint64_t *keys; uint32_t *counts; size_t nd;
    kvec_finish(kv, &keys, &counts, &nd);
    d2o_bmh *b = d2o_bmh_create(sketchsize);
    if (!b) { free(keys); free(counts); return -1; }
    for (size_t i = 0; i < nd; ++i)
        if ((double)counts[i] > thr) d2o_bmh_update(b, keys[i], (double)counts[i]);   /* counter.h:123-125 */
    d2o_bmh_data(b, sig_out);
    if (tw_out) *tw_out = d2o_bmh_total_weight(b);                                    /* fastxsketch.cpp:444 */
    d2o_bmh_destroy(b);
    free(keys); free(counts);
    return 0;
}

int d2o_bmh_sketch_buffer(const char *buf, size_t len, int k, int canon, uint64_t xormask,
                          size_t sketchsize, double count_threshold, double *sig_out,
                          double *total_weight_out, uint64_t *nkmers_out) {
    kvec_t kv = { NULL, 0, 0, xormask };
    d2o_encode_fastx_buffer(buf, len, k, canon, kvec_cb, &kv);
    return bmh_from_kvec(&kv, sketchsize, count_threshold, sig_out, total_weight_out, nkmers_out);
}

int d2o_bmh_sketch_file(const char *path, int k, int canon, uint64_t xormask, size_t sketchsize,
                        double count_threshold, double *sig_out, double *total_weight_out,
                        uint64_t *nkmers_out) {
    kvec_t kv = { NULL, 0, 0, xormask };
    char *line = strdup(path);
    int rc = 0;
    for (char *save = NULL, *tok = strtok_r(line, " ", &save); tok; tok = strtok_r(NULL, " ", &save)) {
        size_t len = 0;
        char *buf = d2o_slurp(tok, &len);
        if (!buf) { rc = -2; break; }
        d2o_encode_fastx_buffer(buf, len, k, canon, kvec_cb, &kv);
        free(buf);
    }
    free(line);
    if (rc) { free(kv.v); return rc; }
    return bmh_from_kvec(&kv, sketchsize, count_threshold, sig_out, total_weight_out, nkmers_out);
}

int d2o_bmh_sketch_files(const char *const *paths, size_t n, int k, int canon, uint64_t xormask,
                         size_t sketchsize, double count_threshold, double *sig_out,
                         double *total_weight_out, uint64_t *nkmers_out) {
    int rc = 0;
#pragma omp parallel for schedule(dynamic)
    for (size_t i = 0; i < n; ++i) {
        uint64_t nk = 0;
        double tw = 0.;
        const int r = d2o_bmh_sketch_file(paths[i], k, canon, xormask, sketchsize, count_threshold,
                                          sig_out + i * sketchsize, &tw, &nk);
        if (total_weight_out) total_weight_out[i] = tw;
        if (nkmers_out) nkmers_out[i] = nk;
        if (r) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

/* ------------------------------------------------------------------ */
/* --parse-by-seq: one sketch per record                                */
/* ------------------------------------------------------------------ */
typedef struct {
    int k, canon, multiset; uint64_t xormask; size_t S; double thr;
    size_t n, cap; double *sigs, *cards; char *names; size_t names_len, names_cap;
} byseq_t;

static void byseq_cb(const char *name, size_t name_len, const char *seq, size_t seq_len, void *ud) {
    byseq_t *b = (byseq_t *)ud;
    if (b->n == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 64;
        b->sigs = (double *)realloc(b->sigs, b->cap * b->S * sizeof(double));
        b->cards = (double *)realloc(b->cards, b->cap * sizeof(double));
    }
    if (b->names_len + name_len + 2 > b->names_cap) {
        b->names_cap = (b->names_len + name_len + 2) * 2;
        b->names = (char *)realloc(b->names, b->names_cap);
    }
    memcpy(b->names + b->names_len, name, name_len);                  /* fastxsketchbyseq.cpp:243-244 */
    b->names_len += name_len;
    b->names[b->names_len++] = '\n';
    double *sig = b->sigs + b->n * b->S;
    kvec_t kv = { NULL, 0, 0, b->xormask };
    d2o_encode_seq(seq, seq_len, b->k, b->canon, kvec_cb, &kv);        /* masked k-mers of this record */
    if (b->multiset) {                                                  /* lines 443-451 */
        double tw = 0.;
        bmh_from_kvec(&kv, b->S, b->thr, sig, &tw, NULL);
        b->cards[b->n] = tw;
    } else {                                                            /* lines 366-442, OPH */
        d2o_oph s;
        d2o_oph_init(&s, b->S);
        for (size_t i = 0; i < kv.n; ++i) d2o_oph_update(&s, kv.v[i]);
        double card = d2o_oph_getcard(&s);
        double *tmp = (double *)malloc(s.m * sizeof(double));
        d2o_oph_data(&s, tmp);
        memcpy(sig, tmp, b->S * sizeof(double));
        free(tmp);
        d2o_oph_free(&s);
        if (card != card) card = 0.;                                    /* 410-414 */
        if (card < 10. * (double)b->S) {                                /* 415-430: exact distinct count */
            if (!kv.v) kv.v = (uint64_t *)malloc(8);
            uint64_t *keys; uint32_t *counts; size_t nd;
            kvec_finish(&kv, &keys, &counts, &nd);
            card = (double)nd;
            free(counts);
            kv.v = keys;
        }
        b->cards[b->n] = card;
        free(kv.v);
    }
    ++b->n;
}

int d2o_sketch_buffer_byseq(const char *buf, size_t len, int k, int canon, uint64_t xormask, size_t sketchsize,
                            int multiset, double count_threshold, size_t *nrec_out, double **sigs_out,
                            double **cards_out, char **names_out) {
    byseq_t b;
    memset(&b, 0, sizeof(b));
    b.k = k; b.canon = canon; b.multiset = multiset; b.xormask = xormask; b.S = sketchsize; b.thr = count_threshold;
    d2o_walk_fastx_records(buf, len, byseq_cb, &b);
    if (!b.names) b.names = (char *)calloc(1, 1);
    else b.names[b.names_len] = 0;
    *nrec_out = b.n;
    *sigs_out = b.sigs ? b.sigs : (double *)malloc(8);
    *cards_out = b.cards ? b.cards : (double *)malloc(8);
    *names_out = b.names;
    return 0;
}
