// d2g_k1.h -- host-side pieces shared by K1 (OPH) and K3 (k-mer counting + BagMinHash): the
// launch plan over a packed run stream and the persistent sketcher's staging buffers.
#pragma once
#include "d2g_kmers.h"
#include <vector>

// host-side launch plan: 64-k-mer chunks per run, <= K1_BLOCK_CHUNKS chunks of one genome per workgroup
struct PlanHost {
    std::vector<uint64_t> chunk_off, bc0;
    std::vector<uint32_t> bg, bn, blo, bhi;
    uint64_t nkmers = 0, nbases = 0;
};
int d2g_build_plan_host(d2g_ctx *ctx, const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n,
                        int k, PlanHost &p);

struct d2g_oph_plan {
    d2g_ctx *ctx = nullptr;
    int k = 0;
    size_t n = 0, nrun = 0, nblk = 0;
    uint64_t nkmers = 0, nbases = 0;
    std::vector<uint32_t> h_run_len;           // host copies kept for K3's bucket layout
    std::vector<uint64_t> h_genome_run_off;
    uint64_t *d_run_start = nullptr;
    uint32_t *d_run_len = nullptr;
    uint64_t *d_run_chunk_off = nullptr;
    uint32_t *d_blk_genome = nullptr;
    uint64_t *d_blk_chunk0 = nullptr;
    uint32_t *d_blk_nchunks = nullptr;
    uint32_t *d_blk_run_lo = nullptr;
    uint32_t *d_blk_run_hi = nullptr;
};


inline KmerArgs d2g_plan_args(const d2g_oph_plan *plan, const uint8_t *packed_dev, int canon) {
    KmerArgs a;
    a.packed = reinterpret_cast<const uint32_t *>(packed_dev);
    a.run_start = plan->d_run_start; a.run_len = plan->d_run_len; a.run_chunk_off = plan->d_run_chunk_off;
    a.blk_genome = plan->d_blk_genome; a.blk_chunk0 = plan->d_blk_chunk0; a.blk_nchunks = plan->d_blk_nchunks;
    a.blk_run_lo = plan->d_blk_run_lo; a.blk_run_hi = plan->d_blk_run_hi;
    a.k = plan->k; a.canon = canon; a.blk0 = 0;
    return a;
}

template <class T> inline int d2g_grow(d2g_ctx *ctx, T **p, size_t *cap, size_t need) {
    if (need <= *cap) return D2G_OK;
    (void)hipFree(*p); *p = nullptr; *cap = 0;
    const size_t ncap = need + need / 4 + 4096;
    D2G_HIP(ctx, hipMalloc((void **)p, ncap * sizeof(T)));
    *cap = ncap;
    return D2G_OK;
}

// Host ingest feeds K1 in groups of inputs; re-allocating device buffers and uploading eight
// small tables per group costs ~20 ms, the kernel ~0.1 ms.  The sketcher keeps grow-only device
// buffers and ships all launch tables in ONE copy from a pinned arena.
struct d2g_k3_state;
struct d2g_k0_state;
struct d2g_sketcher {
    d2g_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    uint8_t *d_packed = nullptr; size_t cap_packed = 0;
    uint64_t *d_regs = nullptr;  size_t cap_regs = 0;      // in u64
    uint8_t *d_arena = nullptr, *h_arena = nullptr; size_t cap_arena = 0;
    uint8_t *h_stage = nullptr; size_t cap_stage = 0;      // pinned staging of the packed stream
    d2g_k3_state *k3 = nullptr;                            // --multiset work buffers (d2g_k3_bmh.hip)
    d2g_k0_state *k0 = nullptr;                            // device FASTA ingest (d2g_k0.hip): raw bytes, tile tables, the last run table
};
void d2g_k0_state_destroy(d2g_k0_state *st);
bool d2g_k0_ingested(const d2g_sketcher *sk, uint64_t *nbases);   // d_packed holds a stream ingested by K0 (and how many bases)
void d2g_k0_invalidate(d2g_sketcher *sk);


// validate + upload one batch (launch tables in one pinned-arena copy, packed stream through the
// pinned stage) on sk->stream; fills `out` with device pointers, `nblk` with the grid size.
// packed == nullptr: the stream d2g_sketcher_ingest_fasta left in the device buffer is used as it is.
int d2g_sketcher_stage(d2g_sketcher *sk, const uint8_t *packed, size_t packed_bytes, const uint64_t *run_start,
                       const uint32_t *run_len, size_t nrun, const uint64_t *genome_run_off, size_t n, int k, int canon,
                       KmerArgs *out, size_t *nblk, PlanHost *ph_out);
