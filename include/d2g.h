/*
 * d2g.h -- C ABI of libd2g.so: the MI355X (gfx950) implementation of dashing2's two
 * data-parallel hot paths.  This is the drop-in boundary: plain pointers and sizes,
 * no C++ / torch types, no exceptions across the ABI.
 *
 * dashing2 has no FFI of its own; the seams this library replaces are the C++ calls
 *   fastx2sketch()      reference src/fastxsketch.h:66   (called at sketch_core.cpp:31)
 *   cmp_core()/compare()/emit_rectangular()   reference src/cmp_main.h:130-132
 * INTEGRATION.md shows the few lines a dashing2 maintainer would add at those seams.
 *
 * Conventions
 *   - every call returns 0 on success or a negative d2g_status; d2g_strerror() names it and
 *     d2g_last_error(ctx) carries the HIP/runtime detail.
 *   - `_dev` entry points take DEVICE pointers (caller-owned, e.g. from d2g_malloc or a
 *     torch tensor's data_ptr) and a hipStream_t passed as void* (NULL = default stream);
 *     they enqueue work and do not synchronise.  Entry points without `_dev` take HOST
 *     pointers, move the data, run, synchronise and copy results back.
 *   - a d2g_ctx is bound to one GPU and is used by one host thread at a time
 *     (multi-GPU = one process / one ctx per device).
 *   - there is NO CPU fallback: without a usable gfx950 device d2g_ctx_create fails.
 */
#ifndef D2G_H
#define D2G_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D2G_VERSION_MAJOR 0
#define D2G_VERSION_MINOR 1

typedef struct d2g_ctx d2g_ctx;

enum d2g_status {
    D2G_OK = 0,
    D2G_ERR_INVALID = -1,      /* bad argument (the reference would throw std::invalid_argument) */
    D2G_ERR_NODEVICE = -2,     /* no usable HIP device */
    D2G_ERR_HIP = -3,          /* HIP runtime error; see d2g_last_error */
    D2G_ERR_NOMEM = -4,
    D2G_ERR_UNSUPPORTED = -5,  /* outside the hot-path scope (e.g. k > 32) */
    D2G_ERR_IO = -6,
    D2G_ERR_INTERNAL = -7      /* an internal invariant failed (reported, never silent) */
};

/* measures: order of enum Measure, reference src/cmp_main.h:8-17 */
enum d2g_measure {
    D2G_SIMILARITY = 0, D2G_CONTAINMENT = 1, D2G_SYMMETRIC_CONTAINMENT = 2,
    D2G_POISSON_LLR = 3, D2G_INTERSECTION = 4, D2G_UNION_SIZE = 5
};

/* which all-pairs kernel family to use (d2g_cmp_* `algo` argument) */
enum d2g_cmp_algo {
    D2G_CMP_AUTO = 0,       /* bit-sliced when it applies, else direct */
    D2G_CMP_DIRECT = 1,     /* 64-bit register compare, LDS-tiled */
    D2G_CMP_BITSLICE = 2    /* per-column dense ids -> bit planes -> v_bitop3/v_bcnt */
};

/* ---- runtime ------------------------------------------------------------ */
int         d2g_version(void);                       /* major*1000 + minor */
const char *d2g_strerror(int status);
int         d2g_device_count(void);
int         d2g_ctx_create(int device, d2g_ctx **out);
void        d2g_ctx_destroy(d2g_ctx *ctx);
const char *d2g_last_error(const d2g_ctx *ctx);
int         d2g_ctx_device(const d2g_ctx *ctx);
/* The library's D2G_* tuning switches and test hooks are read from the environment ONCE, when the context is
 * created; d2g_ctx_reload_tuning reads them again (tests that change a switch between calls).  d2g_ctx_tuning writes the resolved
 * set as a JSON object {"D2G_X": "value", ...} (only the switches that were set) into buf and returns the length it needs. */
int         d2g_ctx_reload_tuning(d2g_ctx *ctx);
int         d2g_ctx_tuning(const d2g_ctx *ctx, char *buf, size_t cap);
int         d2g_sync(d2g_ctx *ctx, void *stream);
int         d2g_malloc(d2g_ctx *ctx, size_t nbytes, void **dptr);
int         d2g_free(d2g_ctx *ctx, void *dptr);
/* page-locked host memory: D2H/H2D at PCIe rate, no per-batch page faults */
int         d2g_malloc_host(d2g_ctx *ctx, size_t nbytes, void **hptr);
int         d2g_free_host(d2g_ctx *ctx, void *hptr);
/* page-lock / release memory the caller owns (an ingest pipeline can start filling its staging buffers before the context
 * exists and register them once it does) */
int         d2g_host_register(d2g_ctx *ctx, void *hptr, size_t nbytes);
int         d2g_host_unregister(d2g_ctx *ctx, void *hptr);
/* Pays one-time costs NOW -- on whatever thread calls it, e.g. a helper thread that has just created the context while the main
 * thread is still reading its inputs -- instead of inside the first real operation: D2G_WARM_COPY = the runtime's copy machinery
 * (the first host<->device copy of a process costs ~30 ms, whatever its size), D2G_WARM_K* = the code objects of a kernel family. */
#define D2G_WARM_COPY 1
#define D2G_WARM_K0   2
#define D2G_WARM_K1   4
#define D2G_WARM_K2   8
#define D2G_WARM_K3   16
int         d2g_warmup(d2g_ctx *ctx, int what);
/* "<marketing name> (<gcn arch>, <n> CUs)" of a device, for logs and --gpu-stats */
int         d2g_device_name(int device, char *buf, size_t cap);
int         d2g_memcpy_h2d(d2g_ctx *ctx, void *dst_dev, const void *src_host, size_t nbytes, void *stream);
int         d2g_memcpy_d2h(d2g_ctx *ctx, void *dst_host, const void *src_dev, size_t nbytes, void *stream);
/* Per-launch HIP-event timing of the dominant kernels.  With timing enabled every launch of
 * the K1 kernel ("k1"), the K0 ingest chain ("k0"), the K3 chain ("k3"), the K2 pair kernel ("k2") and the K2 prepare chain ("k2prep") is
 * bracketed by events recorded on the launch stream; nothing synchronises until d2g_kernel_ms,
 * which reports the number of logged launches, their average and the last duration (ms) and
 * optionally clears the log. */
#define D2G_TIME_K1     2
#define D2G_TIME_K2     4
#define D2G_TIME_K2PREP 8
#define D2G_TIME_K3     16
#define D2G_TIME_K0     32
/* enabled: 0 = off, 1 = every kernel above, or an OR of D2G_TIME_* (an event pair in the stream costs a few microseconds of
 * device time per launch: time only what is being reported) */
int         d2g_set_timing(d2g_ctx *ctx, int enabled);
int         d2g_kernel_ms(d2g_ctx *ctx, const char *which, int reset, int *count, float *avg_ms, float *last_ms);

/* ---- host-side primitives of the path (x86, x87 long double where the reference uses it) ---- */
/* sketch::hash::WangHash::hash -- reference call sites src/enums.h:136-140, src/oph.h:44-53 */
uint64_t d2g_wang_hash(uint64_t x);
/* seed_mask(): reference src/enums.cpp:131-140.  0 -> 0 (CLI default, sketch_main.cpp:112) */
uint64_t d2g_seed_mask(uint64_t seedseed);
/* DHasher xor constant seed_ ^ 0x533f8c2151b20f97: reference src/oph.h:44-53,59,142 */
uint64_t d2g_oph_xor_const(void);
/* LazyOnePermSetSketch ctor: m = S rounded up to even. reference src/oph.h:143-146 */
size_t   d2g_oph_m(size_t sketchsize);
/* getcard(): reference src/oph.h:240-247 */
double   d2g_oph_card(const uint64_t *regs, size_t m);
/* data(): u64 registers -> double signatures. reference src/oph.h:248-263 */
int      d2g_oph_signatures(const uint64_t *regs, size_t m, double *sig_out /* [m] */);
/* both, for n sketches; writes the first S of each m (reference src/fastxsketch.cpp:605,610) */
int      d2g_oph_finalize(const uint64_t *regs /* [n][m] */, size_t n, size_t m, size_t sketchsize,
                          double *sigs_out /* [n][S] */, double *cards_out /* [n] */, int nthreads);
/* densify(): reference src/cmp_core.cpp:577-613 (caller 686-718). In place. Returns #filled via *nfilled. */
int      d2g_densify(double *sigs /* [n][S] */, size_t n, size_t sketchsize, size_t *nfilled, int nthreads);
/* compare() epilogues: reference src/cmp_core.cpp:458-494 (set space, from gt/lt) and
 * 495-517 (multiset space, from neq), both followed by 573-575. */
float    d2g_epilogue_gtlt(uint64_t gt, uint64_t lt, size_t sketchsize, double lhcard, double rhcard,
                           int measure, int k);
float    d2g_epilogue_neq(uint64_t neq, size_t sketchsize, double lhcard, double rhcard, int measure, int k);
/* the same epilogues over rows [r0,r1) of the condensed upper triangle from the device's integer
 * counts (OpenMP): ca = neq, or gt when cb (= lt) is non-null.  With cb == NULL in set space the
 * sketch size must be a power of two (then only gt+lt = S-neq matters: every multiple of 1/S is exact). */
int      d2g_epilogue_ut(const uint32_t *ca, const uint32_t *cb, const double *cards, size_t N, size_t sketchsize,
                         size_t r0, size_t r1, int measure, int k, int multiset_space, int nthreads, float *out);
/* table t[neq] = epilogue for card-independent measures (SIMILARITY, POISSON_LLR) when the
 * value depends on neq only (power-of-two S in set space; any S in multiset space).
 * Returns D2G_ERR_UNSUPPORTED otherwise. lut_out has S+1 floats. */
int      d2g_epilogue_lut(size_t sketchsize, int measure, int k, int multiset_space, float *lut_out);

/* ---- host ingest: FASTA/FASTQ(.gz) -> packed run stream (input of K1) -------------
 * Replaces the parsing half of bns::Encoder::for_each + kseq (reference call sites
 * src/fastxsketch.cpp:383-424 ; src/d2.h:52-71 for_each_substr ; src/d2.h:273-305 KSeqHolder).
 * One "genome" = one input line of the reference (a path, or several space-separated paths
 * that feed the same sketch).  Records and non-ACGT bytes end a run; runs shorter than k
 * are dropped (they contain no k-mer). */
typedef struct d2g_seqpack d2g_seqpack;
int  d2g_seqpack_create(int k, d2g_seqpack **out);
void d2g_seqpack_destroy(d2g_seqpack *sp);
/* forget the content but keep the allocations (pooling packers avoids page-fault storms) */
void d2g_seqpack_clear(d2g_seqpack *sp);
/* appends one genome from a file "line" (gz transparently via zlib). */
int  d2g_seqpack_add_path(d2g_seqpack *sp, const char *path_line);
/* appends one genome from an in-memory FASTA/FASTQ buffer */
int  d2g_seqpack_add_fastx(d2g_seqpack *sp, const char *buf, size_t len);
/* appends one genome consisting of a single raw sequence (no header) */
int  d2g_seqpack_add_sequence(d2g_seqpack *sp, const char *seq, size_t len);
/* --parse-by-seq (reference src/fastxsketchbyseq.cpp:233-252): every FASTA/FASTQ record of the file(s)
 * becomes its own genome, named by its header up to the first whitespace (kseq's name); records
 * without a k-mer still get an (empty) genome so that indices line up with the reference's names_. */
int             d2g_seqpack_add_path_by_record(d2g_seqpack *sp, const char *path_line);
int             d2g_seqpack_add_fastx_by_record(d2g_seqpack *sp, const char *buf, size_t len);
const char     *d2g_seqpack_name(const d2g_seqpack *sp, size_t genome);   /* packs filled by *_by_record only */
size_t          d2g_seqpack_ngenomes(const d2g_seqpack *sp);
size_t          d2g_seqpack_nruns(const d2g_seqpack *sp);
size_t          d2g_seqpack_packed_bytes(const d2g_seqpack *sp);   /* including the 64-byte pad */
const uint8_t  *d2g_seqpack_packed(const d2g_seqpack *sp);
const uint64_t *d2g_seqpack_run_start(const d2g_seqpack *sp);
const uint32_t *d2g_seqpack_run_len(const d2g_seqpack *sp);
const uint64_t *d2g_seqpack_genome_run_off(const d2g_seqpack *sp);  /* [ngenomes+1] */
uint64_t        d2g_seqpack_nkmers(const d2g_seqpack *sp, size_t genome); /* = total_updates */
uint64_t        d2g_seqpack_nbases(const d2g_seqpack *sp);           /* bases stored */

/* ---- K1: 2-bit packed bases -> OPH registers ---------------------------------
 * Replaces the per-k-mer chain bns::Encoder::for_each -> maskfn -> OPSetSketch::update
 * (reference src/fastxsketch.cpp:383-424,565 ; src/enums.h:136-140 ; src/oph.h:176-211).
 *
 * Input layout ("packed run stream"):
 *   packed      2 bits per base, A0 C1 G2 T3, base p at bits [2(p%4), 2(p%4)+2) of byte p/4.
 *               Only the maximal ACGT runs of length >= k are stored, back to back; the
 *               buffer must be padded with >= 64 readable bytes after the last base.
 *   run_start   [nrun]   first base (index into the packed stream) of each run
 *   run_len     [nrun]   run length in bases (>= k)
 *   genome_run_off [n+1] runs of genome g are [genome_run_off[g], genome_run_off[g+1])
 * k-mers never span runs (window resets at non-ACGT bytes and record boundaries).
 * Output: regs_out[n][m] (m = d2g_oph_m(S)), each register = min OPH id of its bucket or ~0.
 */
int d2g_oph_sketch(d2g_ctx *ctx, const uint8_t *packed, size_t packed_bytes,
                   const uint64_t *run_start, const uint32_t *run_len, size_t nrun,
                   const uint64_t *genome_run_off, size_t n,
                   int k, int canon, uint64_t xormask, size_t sketchsize,
                   uint64_t *regs_out /* host [n][m] */);
/* device-resident form. plan = host-built launch plan (d2g_oph_plan_*); all other pointers device. */
typedef struct d2g_oph_plan d2g_oph_plan;
int  d2g_oph_plan_create(d2g_ctx *ctx, const uint64_t *run_start, const uint32_t *run_len, size_t nrun,
                         const uint64_t *genome_run_off, size_t n, int k, d2g_oph_plan **out);
void d2g_oph_plan_destroy(d2g_oph_plan *plan);
uint64_t d2g_oph_plan_nkmers(const d2g_oph_plan *plan);   /* total k-mers the plan covers */
uint64_t d2g_oph_plan_nbases(const d2g_oph_plan *plan);   /* total bases in the runs */
int  d2g_oph_sketch_dev(d2g_ctx *ctx, const d2g_oph_plan *plan, const uint8_t *packed_dev,
                        int canon, uint64_t xormask, size_t sketchsize,
                        uint64_t *regs_out_dev /* [n][m] */, void *stream);

/* persistent form for host ingest pipelines: grow-only device buffers, one pinned-arena upload of
 * the launch tables per call, own stream.  Same arguments and result as d2g_oph_sketch. */
typedef struct d2g_sketcher d2g_sketcher;
int  d2g_sketcher_create(d2g_ctx *ctx, d2g_sketcher **out);
void d2g_sketcher_destroy(d2g_sketcher *sk);
int  d2g_sketcher_run(d2g_sketcher *sk, const uint8_t *packed, size_t packed_bytes,
                      const uint64_t *run_start, const uint32_t *run_len, size_t nrun,
                      const uint64_t *genome_run_off, size_t n, int k, int canon, uint64_t xormask,
                      size_t sketchsize, uint64_t *regs_out /* host [n][m] */);

/* ---- K0: FASTA bytes -> packed run stream on the GPU (host ingest pipelines) ------
 * Replaces, for plain FASTA inputs, the host parser + 2-bit packer (d2g_seqpack_*; reference call sites
 * src/fastxsketch.cpp:383-424, src/d2.h:273-305): the caller read()s the files into ONE host buffer (page-locked memory from
 * d2g_malloc_host makes the upload a single DMA), file f at raw + file_off[f] (16-byte aligned), file_len[f] bytes; genome g =
 * files [genome_file_off[g], genome_file_off[g+1]) (several files can feed one sketch, like a reference input line with
 * spaces).  The packed stream stays in the sketcher's device buffer; the run table comes back to the host.  Then
 * d2g_sketcher_run / d2g_sketcher_run_bmh / d2g_sketcher_run_distinct with packed == NULL and the table of
 * d2g_sketcher_ingested_runs sketch it: registers bit-identical to the host-parsed path.
 * Returns D2G_ERR_UNSUPPORTED for what only the host parser handles: inputs that do not begin with '>' (gzip members, FASTQ,
 * leading junk), lines that begin with '+' (FASTQ quality sections, found only after the device passes have run), files of 2 GiB
 * and more.  After ANY failed ingest the sketcher's device stream is INVALID -- whatever an earlier ingest left there has been
 * overwritten or discarded -- and d2g_sketcher_run* with packed == NULL fails until the next successful ingest: re-stage.
 * The pointers of d2g_sketcher_ingested_runs stay valid until the next ingest on this sketcher. */
int d2g_sketcher_ingest_fasta(d2g_sketcher *sk, const uint8_t *raw, size_t raw_bytes, const uint64_t *file_off,
                              const uint64_t *file_len, size_t nfiles, const uint64_t *genome_file_off /* [n+1] */, size_t n, int k);
int d2g_sketcher_ingested_runs(const d2g_sketcher *sk, const uint64_t **run_start, const uint32_t **run_len, size_t *nrun,
                               const uint64_t **genome_run_off /* [n+1] */, const uint64_t **genome_nkmers /* [n] */,
                               uint64_t *nbases /* bases in the stream */);

/* ---- K3: --multiset sketches: exact k-mer counts (R11) -> BagMinHash (R12) ------
 * Replaces, per input, the reference chain
 *   Counter::add(maskfn(kmer))          src/counter.h:68-77 ; src/fastxsketch.cpp:386,430
 *   Counter::finalize(bmh, threshold)   src/counter.h:118-138 (update(key, count) for count > threshold)
 *   BagMinHash2<double>::update/data/total_weight   (ABSENT dnbaker/sketch bmh.h; src/d2.h:247,
 *                                                    src/fastxsketch.cpp:443-445,477-487)
 * Inputs are the packed run stream of K1.  Outputs per genome: S doubles (the register minima)
 * and the total weight (= number of counted k-mers with count > threshold), which the reference
 * stores as the cardinality.  BagMinHash arithmetic follows the published algorithm under the
 * "BMH-D2G" spec in DESIGN.md: PARITY UNPINNED against a real dashing2 binary (its source is
 * absent), bit-exact against oracle/d2_bmh_oracle.c.  An input without k-mers yields +inf registers.
 */
int d2g_bmh_sketch(d2g_ctx *ctx, const uint8_t *packed, size_t packed_bytes,
                   const uint64_t *run_start, const uint32_t *run_len, size_t nrun,
                   const uint64_t *genome_run_off, size_t n,
                   int k, int canon, uint64_t xormask, size_t sketchsize, double count_threshold,
                   double *sig_out /* host [n][S] */, double *total_weight_out /* host [n] */);
/* device-resident form (plan and packed stream as for d2g_oph_sketch_dev); synchronises `stream`
 * before returning (the status word of the kernels is checked) */
int d2g_bmh_sketch_dev(d2g_ctx *ctx, const d2g_oph_plan *plan, const uint8_t *packed_dev,
                       int canon, uint64_t xormask, size_t sketchsize, double count_threshold,
                       double *sig_out_dev /* [n][S] */, double *total_weight_out_dev /* [n] */, void *stream);
/* persistent form (same buffers/stream as d2g_sketcher_run) */
int d2g_sketcher_run_bmh(d2g_sketcher *sk, const uint8_t *packed, size_t packed_bytes,
                         const uint64_t *run_start, const uint32_t *run_len, size_t nrun,
                         const uint64_t *genome_run_off, size_t n, int k, int canon, uint64_t xormask,
                         size_t sketchsize, double count_threshold,
                         double *sig_out /* host [n][S] */, double *total_weight_out /* host [n] */);
/* R11 alone (Counter::finalize(vector&, vector&, threshold), src/counter.h:78-117, minus the
 * sort): the distinct masked k-mers of genome g with count > threshold and their counts land in
 * keys_out/counts_out[genome_off_out[g] .. genome_off_out[g+1]) in unspecified order.
 * cap = capacity of the two output arrays (the total k-mer count always suffices). */
int d2g_kmer_count(d2g_ctx *ctx, const uint8_t *packed, size_t packed_bytes,
                   const uint64_t *run_start, const uint32_t *run_len, size_t nrun,
                   const uint64_t *genome_run_off, size_t n, int k, int canon, uint64_t xormask,
                   double count_threshold, uint64_t *keys_out, uint32_t *counts_out, size_t cap,
                   uint64_t *genome_off_out /* [n+1] */);
/* number of DISTINCT masked k-mers per genome (exact): the cardinality the reference substitutes for
 * small --parse-by-seq sketches (src/fastxsketchbyseq.cpp:415-430, `ids.size()`); same bucketed LDS
 * counting as d2g_kmer_count without materialising the keys on the host. */
int d2g_kmer_distinct(d2g_ctx *ctx, const uint8_t *packed, size_t packed_bytes,
                      const uint64_t *run_start, const uint32_t *run_len, size_t nrun,
                      const uint64_t *genome_run_off, size_t n, int k, int canon, uint64_t xormask,
                      uint64_t *ndistinct_out /* host [n] */);
int d2g_sketcher_run_distinct(d2g_sketcher *sk, const uint8_t *packed, size_t packed_bytes,
                              const uint64_t *run_start, const uint32_t *run_len, size_t nrun,
                              const uint64_t *genome_run_off, size_t n, int k, int canon, uint64_t xormask,
                              uint64_t *ndistinct_out /* host [n] */);
/* BagMinHash of explicit weighted sets (reference src/wsketch.cpp:54-73 minwise_det and 17-51
 * minhash_rowwise_csr: h.update(id, weight) per element): set i = elements
 * [set_off[i], set_off[i+1]); weights == NULL means 1.0; weights <= 0 are ignored (as
 * BagMinHash2::update does), weights above 2^53 or NaN are rejected. */
int d2g_bmh_from_weighted(d2g_ctx *ctx, const uint64_t *ids, const double *weights,
                          const uint64_t *set_off /* [nsets+1] */, size_t nsets, size_t sketchsize,
                          double *sig_out /* host [nsets][S] */, double *total_weight_out /* host [nsets] */);
/* the same + BagMinHash2::ids() (reference src/wsketch.cpp:36-37,66-67): owner_out[i][r] = position, within
 * set i, of the element whose point register r holds (the smaller position on an exact tie); ~0 for a
 * register no element reached (empty set).  owner_out may be NULL. */
int d2g_bmh_from_weighted_ids(d2g_ctx *ctx, const uint64_t *ids, const double *weights,
                              const uint64_t *set_off /* [nsets+1] */, size_t nsets, size_t sketchsize,
                              double *sig_out /* host [nsets][S] */, double *total_weight_out /* host [nsets] */,
                              uint64_t *owner_out /* host [nsets][S] or NULL */);

/* ---- K2: dense all-pairs comparison -------------------------------------------
 * Replaces HOT LOOP B: emit_rectangular's row loops calling compare()
 * (reference src/emitrect.cpp:211-323 ; src/cmp_core.cpp:349-361,458-517).
 * sig_bits = the N x S signature matrix as raw 64-bit patterns (the doubles of
 * SketchingResult::signatures_, or u64 k-mer ids), row-major.  Equality of the bit
 * patterns == equality of the doubles (all signatures are >= +0.0 and never NaN).
 *
 * "ut" = condensed upper triangle in the reference's order (0,1),(0,2)...(0,N-1),(1,2)...
 * restricted to rows [r0, r1): out has sum_{r=r0}^{r1-1} (N-r-1) entries.
 */
size_t d2g_ut_count(size_t N, size_t r0, size_t r1);
/* prepared operand for repeated / sharded comparisons (device resident) */
typedef struct d2g_cmp_set d2g_cmp_set;
int  d2g_cmp_set_create_dev(d2g_ctx *ctx, const uint64_t *sig_bits_dev, size_t N, size_t sketchsize,
                            int algo, void *stream, d2g_cmp_set **out);
int  d2g_cmp_set_create(d2g_ctx *ctx, const uint64_t *sig_bits_host, size_t N, size_t sketchsize,
                        int algo, d2g_cmp_set **out);
/* re-load an existing set with a new N x S matrix of the same shape, reusing every device
 * buffer (no allocation; the whole prepare chain is enqueued on `stream`.  The FIRST prepare of a BITSLICE set of 8192 sketches or more -- and the
 * first after d2g_cmp_set_forget -- waits a few tens of microseconds for two small kernels that look at the matrix: INTEGRATION.md section 2) */
int  d2g_cmp_set_update_dev(d2g_ctx *ctx, d2g_cmp_set *set, const uint64_t *sig_bits_dev, void *stream);
/* The `_dev` prepare chain is asynchronous and cannot report a data-dependent failure when it is
 * enqueued.  The one such failure: a register column that puts more distinct values into one hash
 * partition of the bit-sliced prepare than its LDS table holds (N > 21 845 and an adversarial or
 * extremely skewed column; impossible for hashed registers).  d2g_cmp_set_status synchronises `stream`
 * and returns D2G_ERR_INTERNAL in that case (results computed from the set are then invalid; re-create
 * it with D2G_CMP_DIRECT).  The host-pointer entry points (d2g_cmp_set_create, d2g_cmp_eqcount_ut,
 * d2g_cmp_dist_ut) check it themselves and fall back to the direct algorithm under D2G_CMP_AUTO. */
int  d2g_cmp_set_status(d2g_ctx *ctx, const d2g_cmp_set *set, void *stream);
/* bit-sliced sets: max over register columns of (#values occurring >= 2 times) + 1 (values that
 * occur once are coded 0 in the row operand and all-ones in the column operand), the largest id-plane
 * count of any 32-register group and the mean over groups (each group only walks its own planes).
 * Synchronises `stream` and returns d2g_cmp_set_status's error, if any; all 0 for a DIRECT set. */
int  d2g_cmp_set_planes(d2g_ctx *ctx, const d2g_cmp_set *set, void *stream, unsigned *max_distinct, int *nbits,
                        float *mean_nbits);
/* Sparse tiles + pair list (bit-sliced sets of N >= 8192 sketches, owning sets and the multi-GPU engine's gathered operand;
 * D2G_BS_SPARSE=0 switches it off, D2G_BS_SPARSE_MIN_N moves the threshold): an equality count is 0 unless the two sketches share a
 * value in some register column.  The prepare finds the families (sketches that agree in MANY registers; a single chance collision
 * does not weld two families together), puts every family on adjacent positions and lists every pair of DIFFERENT families that
 * shares a value; an upper-triangle launch pre-fills the output with the value of "0 equal", walks only the 32 x 256 tiles a family's
 * rows and columns meet in and adds the listed pairs -- or, when that would not pay (one family holds most sketches, the families'
 * tiles are more than a third of all tiles, the list outgrows pairs / 4 entries), the plain kernel over every tile; decided on the
 * device, no host round trip.  Results are identical either way.
 * A sparse launch uses scratch of the set (work list, launch rows, control words): upper-triangle launches on ONE set must be issued
 * one after the other on ONE stream (rectangular launches and launches on different sets are independent).
 * info4 of the LAST upper-triangle launch on the set (synchronises `stream`): [0] 1 = the set has a sorted operand, [1] tiles
 * listed, [2] flags (bit 0: the prepare decided for the dense walk, bit 1: the dense kernel ran, bit 2: tiles + pair list were used,
 * bit 3: the caller's order was kept, bit 4: the prepare skipped the ordering because the previous prepare of this set had given up --
 * remembered per set, retried every 16th prepare, D2G_SP_REMEMBER=0 switches it off), [3] entries of the pair list (one per pair of
 * different families and shared value). */
int  d2g_cmp_set_sparse_info(d2g_ctx *ctx, const d2g_cmp_set *set, void *stream, uint32_t *info4);
/* diagnostics of the last prepare (synchronises `stream`): up to `cap` entries of the pair list (i | j << 32, i < j) and, if root_out is not null, the family
 * root of every sketch (sketches with equal roots sit in one segment).  Nothing in the product reads these back; tools/plist_stats.py does. */
int  d2g_cmp_set_debug_pairs(d2g_ctx *ctx, const d2g_cmp_set *set, void *stream, uint64_t *pairs_out, size_t cap, size_t *npairs, uint32_t *root_out);
/* ---- sharded prepare (multi-GPU).  The bit-sliced operand is an array of independent
 * 32-register groups of `group_words` u32 each (+ one u32 of meta per group) whose geometry depends
 * on N only, so ranks can each build the groups of their own column slice and all-gather them:
 *   rows held by rank r --d2g_pack_column_slices_dev--> all-to-all --> column slice [N][S/W]
 *   --d2g_cmp_set_create_dev(N, S/W, BITSLICE)--> d2g_cmp_set_export_operand_dev --> all-gather
 *   --d2g_cmp_set_from_planes_dev(N, S, gathered)--> d2g_cmp_*_ut_dev on this rank's row range. */
int  d2g_operand_layout(size_t N, size_t sketchsize, size_t *group_words, size_t *ngroups);
int  d2g_cmp_set_export_operand_dev(d2g_ctx *ctx, const d2g_cmp_set *set, uint32_t *planes_out_dev,
                                    uint32_t *meta_out_dev, void *stream);
/* wraps a caller-owned (gathered) operand; only equality-count entry points work on it.  The operand may be
 * re-gathered into the same buffers between launches (its column coding is re-derived before every launch). */
int  d2g_cmp_set_from_planes_dev(d2g_ctx *ctx, size_t N, size_t sketchsize, const uint32_t *planes_dev,
                                 const uint32_t *meta_dev, d2g_cmp_set **out);
int  d2g_pack_column_slices_dev(d2g_ctx *ctx, const uint64_t *rows_dev, size_t n, size_t sketchsize, int nslices,
                                uint64_t *out_dev, void *stream);
void d2g_cmp_set_destroy(d2g_cmp_set *set);
int  d2g_cmp_set_algo(const d2g_cmp_set *set);      /* the algorithm actually selected */
/* equality counts (u32) for rows [r0,r1) of the upper triangle */
int  d2g_cmp_eqcount_ut_dev(d2g_ctx *ctx, const d2g_cmp_set *set, size_t r0, size_t r1,
                            uint32_t *neq_out_dev, void *stream);
/* fused table epilogue: out = lut[neq] (lut_dev: S+1 floats, see d2g_epilogue_lut) */
int  d2g_cmp_lut_ut_dev(d2g_ctx *ctx, const d2g_cmp_set *set, size_t r0, size_t r1,
                        const float *lut_dev, float *out_dev, void *stream);
/* The FILL of an upper-triangle launch, ahead of the launch.  From 8192 sketches on, a bit-sliced set's launch first fills its output
 * with the value of "no register equal" (200 MB at 10 000 sketches: 30 us at the HBM rate) and then writes the pairs that share
 * something.  The fill depends on nothing: a caller that knows the output before the operand is ready --
 * d2g_cmp_set_update_dev still to come, an exchange in flight -- enqueues it here, on ANY stream, and the NEXT upper-triangle launch
 * on the set into the same output and rows skips its own.  The caller orders the two: the same stream, or an event the launch's stream
 * waits for.  Exactly one of neq_out_dev / (lut_dev, out_dev) is given, as in the launch that follows.  A no-op for sets that would not
 * fill (small sets, the direct kernel); harmless when the launch decides for the dense walk (every output is written again). */
int  d2g_cmp_ut_prefill_dev(d2g_ctx *ctx, d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *neq_out_dev,
                            const float *lut_dev, float *out_dev, void *stream);
/* The same fill with NO launch and NO stream of its own: the caller ANNOUNCES the output of the next upper-triangle launch before the
 * d2g_cmp_set_update_dev that precedes it, and that prepare's latency-bound kernels (one to forty workgroups each on an idle chip: column
 * plan, the counting sort of the families) carry the fill as extra workgroups behind their own.  Sequence per step:
 *     d2g_cmp_ut_announce_dev(ctx, set, r0, r1, ...);  d2g_cmp_set_update_dev(ctx, set, sigs, stream);  d2g_cmp_lut_ut_dev(..., stream);
 * The announcement serves exactly one prepare; the launch must follow on the prepare's stream with the same rows and output (any other
 * launch simply fills for itself).  lut_dev must hold the table by the time the prepare runs on its stream.  Nothing is enqueued here.
 * A no-op for sets that would not fill; D2G_SP_RIDE=0 turns the riding off (the launch fills as before).
 * The set keeps the announced pointer until its next prepare: a caller that frees the output before that prepare CANCELS the announcement
 * with neq_out_dev = out_dev = NULL.  A launch skips its own fill only where the announced (or pre-filled) output, rows AND fill value --
 * the count 0, or the table pointer -- are its own; a table whose first entry changes between the prepare and the launch is the caller's
 * business, as is a buffer written between the two. */
int  d2g_cmp_ut_announce_dev(d2g_ctx *ctx, d2g_cmp_set *set, size_t r0, size_t r1, uint32_t *neq_out_dev,
                             const float *lut_dev, float *out_dev);
/* measurements: the set forgets what its earlier prepares learnt about its matrix (which path pays); its next prepare decides as the first
 * prepare of a new set does */
int  d2g_cmp_set_forget(d2g_ctx *ctx, d2g_cmp_set *set);
/* (#a>b, #a<b) counts per pair: needs a set created with D2G_CMP_DIRECT (the raw patterns);
 * required when S is not a power of two in set space */
int  d2g_cmp_gtlt_ut_dev(d2g_ctx *ctx, const d2g_cmp_set *set, size_t r0, size_t r1,
                         uint32_t *gt_out_dev, uint32_t *lt_out_dev, void *stream);
/* rectangular block: rows [a0,a1) x cols [b0,b1) of the full N x N equality-count matrix,
 * row-major (asymmetric all-pairs = [0,N)x[0,N); panel = refs x queries). emitrect.cpp:211-268 */
int  d2g_cmp_eqcount_rect_dev(d2g_ctx *ctx, const d2g_cmp_set *set, size_t a0, size_t a1,
                              size_t b0, size_t b1, uint32_t *neq_out_dev, void *stream);

/* (#a>b, #a<b) for a rectangular block, a = row sketch, b = column sketch (compare(i,j) order) */
int  d2g_cmp_gtlt_rect_dev(d2g_ctx *ctx, const d2g_cmp_set *set, size_t a0, size_t a1, size_t b0, size_t b1,
                           uint32_t *gt_out_dev, uint32_t *lt_out_dev, void *stream);

/* host-pointer conveniences (H2D, run, D2H, sync) */
int  d2g_cmp_eqcount_ut(d2g_ctx *ctx, const uint64_t *sig_bits, size_t N, size_t sketchsize,
                        size_t r0, size_t r1, int algo, uint32_t *neq_out);
/* full compare(): float distances for rows [r0,r1) with the reference's epilogue arithmetic.
 * cards = SketchingResult::cardinalities_; multiset_space selects cmp_core.cpp:495-517. */
int  d2g_cmp_dist_ut(d2g_ctx *ctx, const uint64_t *sig_bits, const double *cards, size_t N,
                     size_t sketchsize, size_t r0, size_t r1, int measure, int k, int multiset_space,
                     int algo, int nthreads, float *out);

/* ---- multi-GPU: RCCL communicator + row-sharded all-pairs engine ------------------------------------
 * Replaces nothing in the reference (it has no multi-process code); it is how the all-pairs seam
 * (cmp_core -> emit_rectangular, src/cmp_core.cpp:746-751) spreads over the GPUs of a node: rows of the upper
 * triangle are independent units, one exchange of the compact bit-plane operand, no reduction.
 * One d2g_ctx + one d2g_comm + one d2g_allpairs per GPU.  Two deployments share all of it:
 *   - one process (or host thread) per GPU: rank 0 calls d2g_comm_unique_id, the 128 bytes reach the other
 *     ranks by any means (file, socket, MPI, a torch.distributed store), every rank calls d2g_comm_create;
 *   - one process driving several GPUs (the `dashing2 cmp` CLI): d2g_comm_create_all + the `_all` entry
 *     points, which issue each collective phase for all ranks inside one RCCL group.
 * RCCL is loaded at first use (dlopen); contexts that share a device get a loopback transport instead
 * (device copies ordered by events) so that the whole path can be exercised on one GPU. */
typedef struct d2g_comm d2g_comm;
#define D2G_COMM_ID_BYTES 128
int  d2g_comm_unique_id(void *id_out /* D2G_COMM_ID_BYTES */);                       /* ncclGetUniqueId */
int  d2g_comm_create(d2g_ctx *ctx, const void *id, int rank, int world, d2g_comm **out);   /* ncclCommInitRank (collective) */
int  d2g_comm_create_all(d2g_ctx **ctxs, int nctx, d2g_comm **comms_out /* [nctx] */);     /* ncclCommInitAll, or loopback */
void d2g_comm_destroy(d2g_comm *comm);
int  d2g_comm_rank(const d2g_comm *comm);
int  d2g_comm_world(const d2g_comm *comm);
int  d2g_comm_is_rccl(const d2g_comm *comm);                                          /* 0: loopback / single rank without id */
/* the host matrix to every GPU of this process -- one H2D to ctxs[0], then one RCCL
 * broadcast over xGMI.  sig_dev_out[i] is allocated on ctxs[i] (release with d2g_free).  Synchronises. */
int  d2g_bcast_sigs(d2g_ctx **ctxs, d2g_comm **comms, int nctx, const uint64_t *host_sig, size_t N, size_t sketchsize,
                    uint64_t **sig_dev_out /* [nctx] */);

typedef struct d2g_allpairs d2g_allpairs;
/* rank `d2g_comm_rank(comm)` of an all-pairs job over an N x S matrix.  No divisibility requirement on N or S. */
int  d2g_allpairs_create(d2g_ctx *ctx, d2g_comm *comm, size_t N, size_t sketchsize, d2g_allpairs **out);
void d2g_allpairs_destroy(d2g_allpairs *eng);
/* rows [lo, hi) of the matrix this rank must HOLD as its input block (contiguous, sizes differ by <= 1) */
int  d2g_allpairs_rows_held(const d2g_allpairs *eng, size_t *lo, size_t *hi);
/* rows [r0, r1) of the upper triangle this rank COMPUTES (pair-balanced, d2g_ut_partition) */
int  d2g_allpairs_rows_computed(const d2g_allpairs *eng, size_t *r0, size_t *r1);
/* exchange + sharded prepare: afterwards d2g_allpairs_operand() is the whole N x S bit-sliced operand on this
 * rank's GPU, usable with d2g_cmp_eqcount_ut_dev / d2g_cmp_lut_ut_dev / d2g_cmp_eqcount_rect_dev on ANY rows
 * (the CLI deals row batches to the GPUs round-robin).  Collective; enqueues on `stream`, no synchronisation. */
int  d2g_allpairs_prepare_dev(d2g_allpairs *eng, const uint64_t *my_rows_dev /* [hi-lo][S] */, void *stream);
int  d2g_allpairs_prepare_all(d2g_allpairs **engs, int n, const uint64_t *const *rows_dev, void *const *streams);
const d2g_cmp_set *d2g_allpairs_operand(const d2g_allpairs *eng);
/* The sharded prepare can fail for the reason d2g_cmp_set_status documents (rank-table overflow on an adversarial column), on
 * ANY rank; every rank's status word travels with its groups, so each rank learns it.  Synchronises `stream`; D2G_ERR_INTERNAL
 * means the results of the last prepare/step are invalid on every rank (fall back to one GPU with D2G_CMP_DIRECT).
 * d2g_cmp_set_status(d2g_allpairs_operand(eng)) returns the same. */
int  d2g_allpairs_status(d2g_allpairs *eng, void *stream);
/* chunks the engine cuts a rank's column slice into: the exchange of chunk c+1 overlaps the prepare of chunk c inside ONE step
 * (a function of N, S and the world size; D2G_MGPU_CHUNKS overrides it -- identically on every rank: with one process per GPU
 * d2g_allpairs_create exchanges (N, S, world, chunks) over the communicator once and fails with D2G_ERR_INVALID on every rank
 * when any two ranks disagree, instead of posting mismatched transfers later) */
int  d2g_allpairs_chunks(const d2g_allpairs *eng);
/* one whole step: prepare + this rank's slab (rows_computed) of the condensed triangle; out has
 * d2g_ut_count(N, r0, r1) entries.  lut_dev == NULL (or lut_dev[i] == NULL): u32 equality counts. */
int  d2g_allpairs_step_lut_dev(d2g_allpairs *eng, const uint64_t *my_rows_dev, const float *lut_dev, float *out_dev, void *stream);
int  d2g_allpairs_step_eqcount_dev(d2g_allpairs *eng, const uint64_t *my_rows_dev, uint32_t *out_dev, void *stream);
int  d2g_allpairs_step_all(d2g_allpairs **engs, int n, const uint64_t *const *rows_dev, const float *const *lut_dev,
                           void *const *out_dev, void *const *streams);
/* Per-phase times of ONE step (what a scaling run needs to check a cost model term by term).  With phase timing on, every phase
 * the next prepare/step enqueues for this engine is bracketed by timing events on the stream it runs on (a few microseconds
 * each: switch it off for timed runs); d2g_allpairs_phase_times synchronises the device and returns one record per phase in
 * enqueue order: kind (D2G_PHASE_*), chunk, start (ms after the step's first enqueue reached the GPU) and duration (ms).
 * An exchange phase's duration includes waiting for the slowest peer. */
#define D2G_PHASE_PACK    0
#define D2G_PHASE_X1      1   /* rows -> column slices ("all-to-all-v"), per chunk */
#define D2G_PHASE_PREPARE 2   /* transpose + rank + plan + planes of this rank's column chunk */
#define D2G_PHASE_X2      3   /* bit-plane groups to everyone ("all-gather-v"), per chunk */
#define D2G_PHASE_DERIVE  4   /* plane stream of a gathered chunk */
#define D2G_PHASE_PAIR    5   /* the pair kernel over this rank's rows (d2g_allpairs_step_* only) */
#define D2G_PHASE_ORDER   6   /* sparse tiles on the gathered operand (N >= 8192): ids from the planes, sketch order, sorted stream */
#define D2G_PHASE_FILL    7   /* the slab pre-filled with the value of "0 equal" at the start of the step, under the exchanges (sparse path) */
int  d2g_allpairs_set_phase_timing(d2g_allpairs *eng, int on);
int  d2g_allpairs_phase_times(d2g_allpairs *eng, int cap, int *n_out, int *kind /* [cap] */, int *chunk /* [cap] */,
                              float *start_ms /* [cap] */, float *dur_ms /* [cap] */);
/* what the sparse-tile path did in this engine's LAST pair phase (device-synchronising): info4 as d2g_cmp_set_sparse_info --
 * [0] the gathered operand was ordered (N >= 8192), [1] tiles listed, [2] bit 0 dense walk decided by the ordering / bit 1 the dense kernel ran /
 * bit 2 tiles + pair list used / bit 3 the gathered order was kept, [3] entries of the pair list. */
int  d2g_allpairs_sparse_info(d2g_allpairs *eng, uint32_t *info4);
/* software-pipelined step for a stream of matrices: the exchange + prepare of this call overlap the pair kernel
 * of the previous call (own stream, two operand buffers); results land in out_dev in call order on `stream`.
 * input_ready != 0: my_rows_dev is already complete (no dependency on work queued on `stream`).
 * Plain (prepare/step) and pipelined calls may be mixed on one engine: each form waits, through events, for what the
 * other still has in flight on the buffers they share. */
int  d2g_allpairs_enqueue_lut_dev(d2g_allpairs *eng, const uint64_t *my_rows_dev, const float *lut_dev, float *out_dev,
                                  void *stream, int input_ready);

/* ---- multi-GPU row partition (host arithmetic; emitrect.cpp:290-323 row order) ----
 * Splits rows [0,N) into `nparts` contiguous ranges with (near-)equal pair counts.
 * bounds_out has nparts+1 entries. */
int  d2g_ut_partition(size_t N, int nparts, size_t *bounds_out);

#ifdef __cplusplus
}
#endif
#endif
