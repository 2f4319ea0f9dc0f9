// d2g_k0.hip -- K0: FASTA bytes -> packed run stream ON THE GPU.
//
// Replaces, for plain FASTA inputs, the host half of the ingest: kseq's record walk + bns::Encoder's window resets
// (reference call sites src/fastxsketch.cpp:383-424, src/d2.h:273-305) as restated by d2g_seqpack (d2g_host.cpp).  The
// host only read()s the files into page-locked memory; the raw bytes cross PCIe once (1 byte per base instead of the
// host packer's 2 CPU-seconds per 5 GB) and three streaming passes turn them into exactly what K1/K3 consume:
//   packed stream   every ACGT/acgt byte of a sequence line, 2 bits each, in file order, line feeds squeezed out
//   run starts      the stream index of every base that begins a maximal ACGT run: the first base of a record, and the first
//                   base after any byte that is not a base, a line feed, or a carriage return right before a line feed
// Runs shorter than k stay in the stream as dead bases (the host packer rewinds over them; registers only depend on
// the runs listed, so they are bit-identical).  The host finishes the run table (lengths, the >= k filter, the split of
// runs longer than 2^30, per-genome offsets): a few entries per record.
//
// A byte's meaning depends on state that runs along the file -- is this line a header? did a break occur since the last
// base? how many bases came before? -- but each of the three is a PREFIX SCAN with an associative operator:
//   header      = the first character of the current line is '>' or '@'      -> max-scan of line-feed positions
//   pending break= class of the nearest earlier byte that is a base or a break -> max-scan of (position, class)
//   output index = number of bases before                                      -> sum-scan
// Pass A reduces line-feed positions per 4 KiB tile, a per-file wave scans the tiles; pass B classifies with the header state
// known and reduces base counts / last classes, scanned the same way; pass C classifies again and emits.  Lines starting
// with '+' (FASTQ quality sections are skipped BY LENGTH in kseq: not a scan) raise a flag and the caller uses the host
// parser; so do gz members and anything that does not begin with '>' (checked on the host before the upload).
#include "d2g_internal.h"
#include "d2g_k1.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

constexpr int K0_THREADS = 256, K0_PER = 16, K0_TILE = K0_THREADS * K0_PER;   // 4096 bytes per workgroup
constexpr uint32_t K0_NONE = 0, K0_BASE = 1, K0_BREAK = 2;
constexpr uint32_t K0_ST_PLUS = 1, K0_ST_RUNLIST = 2;

struct K0File { uint64_t off; uint64_t len; uint64_t tile0; };          // byte offset in raw (16-aligned), length, first tile

__device__ __forceinline__ uint32_t k0_file_of_tile(const K0File *files, uint32_t nf, uint64_t tile) {
    uint32_t lo = 0, hi = nf;                                           // files[lo].tile0 <= tile < files[hi].tile0 (files[nf] = sentinel)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (files[mid].tile0 <= tile) lo = mid; else hi = mid; }
    return lo;
}

// block-wide EXCLUSIVE scans over one value per thread (256 threads = 4 waves)
__device__ __forceinline__ uint32_t k0_excl_max(uint32_t v, uint32_t *wave_tmp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl = max(incl, x); }
    if (lane == 63) wave_tmp[wave] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (int w = 0; w < wave; ++w) pre = max(pre, wave_tmp[w]);
    uint32_t ex = __shfl_up(incl, 1);
    if (lane == 0) ex = 0;
    __syncthreads();
    return max(pre, ex);
}
__device__ __forceinline__ uint32_t k0_excl_sum(uint32_t v, uint32_t *wave_tmp, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
    if (lane == 63) wave_tmp[wave] = incl;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (int w = 0; w < K0_THREADS / 64; ++w) { if (w < wave) pre += wave_tmp[w]; tot += wave_tmp[w]; }
    __syncthreads();
    *total = tot;
    return pre + incl - v;
}

struct K0Chunk {                       // a thread's 16 bytes and where they sit
    uint8_t b[K0_PER];
    uint64_t fpos;                     // file-relative position of b[0]
    uint32_t nvalid;                   // bytes of the chunk inside the file
};

__device__ __forceinline__ K0Chunk k0_load(const uint8_t *raw, const K0File &f, uint64_t tile_in_file) {
    K0Chunk c;
    c.fpos = tile_in_file * K0_TILE + (uint64_t)threadIdx.x * K0_PER;
    c.nvalid = c.fpos >= f.len ? 0u : (uint32_t)min<uint64_t>(K0_PER, f.len - c.fpos);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c.nvalid) v = *reinterpret_cast<const uint4 *>(raw + f.off + c.fpos);      // 16-byte aligned; the raw buffer is padded past its end
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < K0_PER; ++i) c.b[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    return c;
}

// Scan key of the last line feed in the chunk: ((file-relative position + 1) << 1) | (the line that FOLLOWS it is a header:
// its first byte is '>' or '@'); 0 = no line feed.  Files stay below 2 GiB so that the key fits 32 bits; carrying the flag in
// the key spares every thread a dependent gather of its line's first byte.
__device__ __forceinline__ uint32_t k0_last_nl(const uint8_t *raw, const K0File &f, const K0Chunk &c) {
    int last = -1;
#pragma unroll
    for (int i = 0; i < K0_PER; ++i) if (i < (int)c.nvalid && c.b[i] == '\n') last = i;
    if (last < 0) return 0u;
    const uint64_t p1 = c.fpos + (uint64_t)last + 1;                     // position of the byte after the line feed
    uint8_t nx = 0;
    if (p1 < f.len) {
        nx = raw[f.off + p1];                                            // (a register select over b[] costs more than this cached byte)
    }
    return ((uint32_t)p1 << 1) | (uint32_t)(nx == '>' || nx == '@');
}

__device__ __forceinline__ int k0_code(uint8_t ch) {                    // A0 C1 G2 T3 (either case), -1 otherwise
    const uint8_t u = ch & 0xDF;
    if (u == 'A' || u == 'C' || u == 'G' || u == 'T') return ((u >> 1) & 3) ^ ((u >> 2) & 1);
    return -1;
}

// One walk over the chunk with the line state known at its start: the chunk's bytes as three masks.  key = scan key of the
// last line feed before the chunk (0 = none: the current line is the file's first).
struct K0Masks { uint32_t base, brk, codes; };                          // bit i: byte i is a base / a break; codes: 2 bits per byte position
__device__ __forceinline__ K0Masks k0_classify(const uint8_t *raw, const K0File &f, const K0Chunk &c, uint32_t key, uint32_t *plus) {
    K0Masks m{0u, 0u, 0u};
    if (!c.nvalid) return m;
    bool at_ls = (uint64_t)(key >> 1) == c.fpos;                        // the chunk starts a line (key 0 and fpos 0: start of file)
    bool hdr = false;
    if (!at_ls) {
        if (key) hdr = key & 1u;
        else { const uint8_t first = raw[f.off]; hdr = first == '>' || first == '@'; }   // still on the file's first line
    }
    // branch-free per byte (neighbouring lanes see different bytes: every branch here would diverge); only the rare carriage
    // return takes one
    uint32_t hd = hdr, ls = at_ls, pl = 0;
#pragma unroll
    for (int i = 0; i < K0_PER; ++i) {
        const uint32_t ch = c.b[i], u = ch & 0xDFu;
        const uint32_t valid = i < (int)c.nvalid;
        const uint32_t gt = (ch == '>') | (ch == '@');
        hd = ls ? gt : hd;                                               // a line start decides whether the line is a header
        pl |= ls & (uint32_t)(ch == '+') & valid;
        const uint32_t nl = ch == '\n';
        const uint32_t isb = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
        uint32_t other = (nl | isb) ^ 1u;                                // neither a line feed nor a base
        if (ch == '\r' && valid) {
            // kseq-style line ends: ONE carriage return right before the line feed (or as the last byte of a file without a
            // final line feed) belongs to the line end; anywhere else it is just a byte that is not a base
            const uint64_t p = c.fpos + i;
            const bool at_end = p + 1 == f.len || (i + 1 < K0_PER ? c.b[i + 1 < K0_PER ? i + 1 : i] == '\n' : raw[f.off + p + 1] == '\n');
            other = at_end ? 0u : 1u;
        }
        const uint32_t base = isb & (hd ^ 1u) & valid;
        const uint32_t brk = (hd | other) & valid;                       // every byte of a header line, and what is neither base nor line end
        m.base |= base << i;
        m.brk |= (brk & (base ^ 1u)) << i;
        m.codes |= ((((u >> 1) & 3u) ^ ((u >> 2) & 1u)) & (0u - base)) << (2 * i);
        ls = nl & valid ? 1u : (valid ? 0u : ls);
    }
    if (pl) *plus = 1;
    return m;
}
// class of the chunk's LAST base-or-break byte (K0_NONE if it has neither)
__device__ __forceinline__ uint32_t k0_last_class(const K0Masks &m) {
    const uint32_t e = m.base | m.brk;
    if (!e) return K0_NONE;
    return (m.brk >> (31 - __clz(e))) & 1u ? K0_BREAK : K0_BASE;
}

// ---------------------------------------------------------------------------------------------- pass A
__global__ __launch_bounds__(K0_THREADS) void k0_newline_kernel(const uint8_t *__restrict__ raw, const K0File *__restrict__ files, uint32_t nf,
                                                                uint32_t *__restrict__ tile_nl) {
    __shared__ uint32_t wt[K0_THREADS / 64];
    const uint64_t tile = blockIdx.x;
    const uint32_t fi = k0_file_of_tile(files, nf, tile);
    const K0File f = files[fi];
    const K0Chunk c = k0_load(raw, f, tile - f.tile0);
    uint32_t v = k0_last_nl(raw, f, c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    if (lane == 0) wt[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_nl[tile] = max(max(wt[0], wt[1]), max(wt[2], wt[3]));
}

// one wave per file: exclusive running maximum over its tiles (in place: tile_nl[t] becomes the carry INTO tile t)
__global__ __launch_bounds__(64) void k0_carry_nl_kernel(const K0File *__restrict__ files, uint32_t nf, uint32_t *__restrict__ tile_nl) {
    const uint32_t fi = blockIdx.x;
    if (fi >= nf) return;
    const uint64_t t0 = files[fi].tile0, t1 = files[fi + 1].tile0;
    const int lane = threadIdx.x;
    uint32_t carry = 0;
    for (uint64_t b = t0; b < t1; b += 64) {
        const uint64_t t = b + lane;
        const uint32_t v = t < t1 ? tile_nl[t] : 0u;
        uint32_t incl = v;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl = max(incl, x); }
        uint32_t ex = __shfl_up(incl, 1);
        if (lane == 0) ex = 0;
        if (t < t1) tile_nl[t] = max(carry, ex);
        carry = max(carry, __shfl(incl, 63));
    }
}

// ---------------------------------------------------------------------------------------------- pass B
__global__ __launch_bounds__(K0_THREADS) void k0_count_kernel(const uint8_t *__restrict__ raw, const K0File *__restrict__ files, uint32_t nf,
                                                              const uint32_t *__restrict__ tile_nl, uint32_t *__restrict__ tile_nbase,
                                                              uint32_t *__restrict__ tile_cls, uint32_t *__restrict__ status) {
    __shared__ uint32_t wt[K0_THREADS / 64];
    const uint64_t tile = blockIdx.x;
    const uint32_t fi = k0_file_of_tile(files, nf, tile);
    const K0File f = files[fi];
    const K0Chunk c = k0_load(raw, f, tile - f.tile0);
    const uint32_t key = max(tile_nl[tile], k0_excl_max(k0_last_nl(raw, f, c), wt));
    uint32_t plus = 0;
    const K0Masks mk = k0_classify(raw, f, c, key, &plus);
    uint32_t nbase = __popc(mk.base);
    const uint32_t last = k0_last_class(mk);
    if (plus) atomicOr(status, K0_ST_PLUS);
    // tile totals: base count; the class of the LAST base-or-break byte of the tile (key = thread order)
    uint32_t ckey = last ? ((threadIdx.x + 1u) << 2 | last) : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) { nbase += __shfl_xor(nbase, o); ckey = max(ckey, __shfl_xor(ckey, o)); }
    __shared__ uint32_t wk[K0_THREADS / 64];
    if (lane == 0) { wt[wave] = nbase; wk[wave] = ckey; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_nbase[tile] = wt[0] + wt[1] + wt[2] + wt[3];
        tile_cls[tile] = max(max(wk[0], wk[1]), max(wk[2], wk[3])) & 3u;
    }
}

// one wave per file: exclusive prefix of the tiles' base counts (file-relative, in place), the pending-break state INTO
// every tile (in place: 1 = a break -- or the start of the file -- lies between the last base before the tile and the
// tile), and the file's total
__global__ __launch_bounds__(64) void k0_carry_base_kernel(const K0File *__restrict__ files, uint32_t nf, uint32_t *__restrict__ tile_nbase,
                                                           uint32_t *__restrict__ tile_cls, uint64_t *__restrict__ file_total) {
    const uint32_t fi = blockIdx.x;
    if (fi >= nf) return;
    const uint64_t t0 = files[fi].tile0, t1 = files[fi + 1].tile0;
    const int lane = threadIdx.x;
    uint64_t run = 0;
    uint32_t pend = 1;                                                  // the first base of a file starts a run
    for (uint64_t b = t0; b < t1; b += 64) {
        const uint64_t t = b + lane;
        const uint32_t n = t < t1 ? tile_nbase[t] : 0u, cls = t < t1 ? tile_cls[t] : 0u;
        uint32_t incl = n;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
        // nearest earlier tile (of this round) with a class; lanes without one inherit the carried state
        uint32_t key = cls ? ((uint32_t)(lane + 1) << 2 | cls) : 0u, kin = key;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(kin, o); if (lane >= o) kin = max(kin, x); }
        uint32_t kex = __shfl_up(kin, 1);
        if (lane == 0) kex = 0;
        if (t < t1) {
            // file-relative base offsets fit 32 bits (a file is below 2 GiB); the file's own offset is added in pass C
            tile_nbase[t] = (uint32_t)(run + incl - n);
            tile_cls[t] = kex ? ((kex & 3u) == K0_BREAK ? 1u : 0u) : pend;
        }
        run += __shfl(incl, 63);
        const uint32_t klast = __shfl(kin, 63);
        if (klast) pend = (klast & 3u) == K0_BREAK ? 1u : 0u;
    }
    if (lane == 0) file_total[fi] = run;
}

// exclusive prefix over the files' totals (one workgroup; nf is small); out[nf] = the batch's total
__global__ __launch_bounds__(1024) void k0_file_prefix_kernel(const uint64_t *__restrict__ file_total, uint32_t nf, uint64_t *__restrict__ file_base) {
    __shared__ uint64_t part[1024];
    const uint32_t per = (nf + 1023) / 1024, lo = min(nf, threadIdx.x * per), hi = min(nf, lo + per);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += file_total[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint64_t run = 0; for (int i = 0; i < 1024; ++i) { const uint64_t x = part[i]; part[i] = run; run += x; } file_base[nf] = run; }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) { file_base[i] = run; run += file_total[i]; }
}

// ---------------------------------------------------------------------------------------------- pass C
__global__ __launch_bounds__(K0_THREADS) void k0_emit_kernel(const uint8_t *__restrict__ raw, const K0File *__restrict__ files, uint32_t nf,
                                                             const uint32_t *__restrict__ tile_nl, const uint32_t *__restrict__ tile_base,
                                                             const uint32_t *__restrict__ tile_pend, const uint64_t *__restrict__ file_base,
                                                             uint32_t *__restrict__ packed, uint64_t *__restrict__ run_list, uint32_t run_cap,
                                                             uint32_t *__restrict__ run_count, uint32_t *__restrict__ status) {
    __shared__ uint32_t wt[K0_THREADS / 64];
    __shared__ uint32_t words[K0_TILE / 16 + 2];                         // the tile's bases, 16 per word, from the word its first base falls into
    const uint64_t tile = blockIdx.x;
    const uint32_t fi = k0_file_of_tile(files, nf, tile);
    const K0File f = files[fi];
    const K0Chunk c = k0_load(raw, f, tile - f.tile0);
    for (int i = threadIdx.x; i < K0_TILE / 16 + 2; i += K0_THREADS) words[i] = 0;
    const uint32_t key = max(tile_nl[tile], k0_excl_max(k0_last_nl(raw, f, c), wt));
    uint32_t plus = 0;
    const K0Masks mk = k0_classify(raw, f, c, key, &plus);               // ONE walk; everything below works on its masks
    const uint32_t nbase = __popc(mk.base), last = k0_last_class(mk);
    uint32_t tile_total = 0;
    const uint32_t before = k0_excl_sum(nbase, wt, &tile_total);
    const uint32_t kex = k0_excl_max(last ? ((threadIdx.x + 1u) << 2 | last) : 0u, wt);
    bool pend = kex ? (kex & 3u) == K0_BREAK : tile_pend[tile] != 0;
    const uint64_t o_tile = file_base[fi] + tile_base[tile];            // stream index of the tile's first base
    const uint32_t shift0 = (uint32_t)(o_tile & 15u);                   // its place in the first word
    // the chunk's bases in order: codes squeezed together into one word, run starts into the list.  A base starts a run if a
    // break lies between it and the base before it (or, for the chunk's first base, if one was pending)
    uint32_t w = 0, nw = 0, todo = mk.base, below = 0;                  // below: mask of the byte positions up to the previous base
    const uint64_t o = o_tile + before;
    while (todo) {
        const int i = __ffs(todo) - 1;
        todo &= todo - 1;
        const uint32_t upto = (1u << i) - 1u;
        if (mk.brk & upto & ~below) pend = true;
        if (pend) {
            const uint32_t slot = atomicAdd(run_count, 1u);
            if (slot < run_cap) run_list[slot] = o + nw; else atomicOr(status, K0_ST_RUNLIST);
            pend = false;
        }
        w |= ((mk.codes >> (2 * i)) & 3u) << (2 * nw);
        ++nw;
        below = upto | (1u << i);
    }
    if (nw) {
        const uint32_t bit = (shift0 + before) * 2, d = bit >> 5, sh = bit & 31;
        atomicOr(&words[d], w << sh);
        if (sh && 2 * nw + sh > 32) atomicOr(&words[d + 1], w >> (32 - sh));
    }
    __syncthreads();
    if (!tile_total) return;
    const uint64_t w0 = o_tile >> 4;
    const uint32_t nwords = (shift0 + tile_total + 15) >> 4;
    for (uint32_t i = threadIdx.x; i < nwords; i += K0_THREADS) {
        const uint32_t v = words[i];
        if (i == 0 || i + 1 == nwords) { if (v) atomicOr(&packed[w0 + i], v); }   // shared with the neighbouring tile
        else packed[w0 + i] = v;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ host side
struct d2g_k0_state {
    uint8_t *d_raw = nullptr; size_t cap_raw = 0;
    K0File *d_files = nullptr; size_t cap_files = 0;
    uint32_t *d_tile_nl = nullptr, *d_tile_nbase = nullptr, *d_tile_cls = nullptr; size_t cap_tiles = 0;
    uint64_t *d_file_total = nullptr, *d_file_base = nullptr; size_t cap_ftot = 0;
    uint64_t *d_run_list = nullptr; size_t cap_runs = 0;
    uint32_t *d_ctl = nullptr;                      // [0] run count, [1] status
    // the run table of the stream ingested last (host)
    std::vector<uint64_t> run_start, genome_run_off, genome_nkmers;
    std::vector<uint32_t> run_len;
    uint64_t nbases = 0;                            // bases in the ingested stream (live and dead)
    bool valid = false;
    int k = 0;
};

void d2g_k0_state_destroy(d2g_k0_state *st) {
    if (!st) return;
    (void)hipFree(st->d_raw); (void)hipFree(st->d_files); (void)hipFree(st->d_tile_nl); (void)hipFree(st->d_tile_nbase);
    (void)hipFree(st->d_tile_cls); (void)hipFree(st->d_file_total); (void)hipFree(st->d_file_base); (void)hipFree(st->d_run_list);
    (void)hipFree(st->d_ctl);
    delete st;
}

bool d2g_k0_ingested(const d2g_sketcher *sk, uint64_t *nbases) {
    if (!sk->k0 || !sk->k0->valid) return false;
    if (nbases) *nbases = sk->k0->nbases;
    return true;
}
void d2g_k0_invalidate(d2g_sketcher *sk) { if (sk->k0) sk->k0->valid = false; }

extern "C" {

int d2g_sketcher_ingest_fasta(d2g_sketcher *sk, const uint8_t *raw, size_t raw_bytes, const uint64_t *file_off, const uint64_t *file_len,
                              size_t nfiles, const uint64_t *genome_file_off, size_t n, int k) {
    if (!sk) return D2G_ERR_INVALID;
    d2g_ctx *ctx = sk->ctx;
    D2G_CHECK(ctx, k >= 1 && k <= 32, "k out of range (1..32)");
    D2G_CHECK(ctx, nfiles < (1u << 30) && (nfiles == 0 || (raw && file_off && file_len)) && genome_file_off && genome_file_off[n] == nfiles, "ingest: bad file table");
    if (!sk->k0) { sk->k0 = new (std::nothrow) d2g_k0_state(); if (!sk->k0) return D2G_ERR_NOMEM; }
    d2g_k0_state *st = sk->k0;
    st->valid = false;
    // what the device parser does not do: anything that does not start like a FASTA record (gz members, FASTQ, leading junk)
    std::vector<K0File> files(nfiles + 1);
    uint64_t ntiles = 0;
    for (size_t f = 0; f < nfiles; ++f) {
        D2G_CHECK(ctx, (file_off[f] & 15) == 0 && file_off[f] + file_len[f] <= raw_bytes, "ingest: file offsets must be 16-byte aligned and inside the buffer");
        if (file_len[f] >= (1ull << 31)) { ctx->last_error = "ingest: inputs of 2 GiB and more go through the host parser"; return D2G_ERR_UNSUPPORTED; }
        if (file_len[f] && raw[file_off[f]] != '>') { ctx->last_error = "ingest: input does not start with '>' (gz / FASTQ / other): host parser"; return D2G_ERR_UNSUPPORTED; }
        files[f] = {file_off[f], file_len[f], ntiles};
        ntiles += (file_len[f] + K0_TILE - 1) / K0_TILE;
    }
    files[nfiles] = {raw_bytes, 0, ntiles};
    D2G_CHECK(ctx, ntiles < (1ull << 31), "ingest: batch too large (split it)");
    D2G_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = sk->stream;
    const size_t out_bytes = raw_bytes / 4 + 128;                        // every byte a base at worst; + the 64-byte pad K1 reads
    if (int rc = d2g_grow(ctx, &st->d_raw, &st->cap_raw, raw_bytes + 64)) return rc;
    if (int rc = d2g_grow(ctx, &sk->d_packed, &sk->cap_packed, out_bytes + 64)) return rc;
    if (int rc = d2g_grow(ctx, &st->d_files, &st->cap_files, nfiles + 1)) return rc;
    if (ntiles > st->cap_tiles) {
        (void)hipFree(st->d_tile_nl); (void)hipFree(st->d_tile_nbase); (void)hipFree(st->d_tile_cls);
        st->d_tile_nl = st->d_tile_nbase = st->d_tile_cls = nullptr; st->cap_tiles = 0;
        const size_t ncap = ntiles + ntiles / 4 + 1024;
        D2G_HIP(ctx, hipMalloc((void **)&st->d_tile_nl, ncap * 4));
        D2G_HIP(ctx, hipMalloc((void **)&st->d_tile_nbase, ncap * 4));
        D2G_HIP(ctx, hipMalloc((void **)&st->d_tile_cls, ncap * 4));
        st->cap_tiles = ncap;
    }
    if (nfiles + 1 > st->cap_ftot) {
        (void)hipFree(st->d_file_total); (void)hipFree(st->d_file_base);
        st->d_file_total = st->d_file_base = nullptr; st->cap_ftot = 0;
        const size_t ncap = nfiles + 1 + nfiles / 4 + 64;
        D2G_HIP(ctx, hipMalloc((void **)&st->d_file_total, ncap * 8));
        D2G_HIP(ctx, hipMalloc((void **)&st->d_file_base, ncap * 8));
        st->cap_ftot = ncap;
    }
    if (!st->d_ctl) D2G_HIP(ctx, hipMalloc((void **)&st->d_ctl, 16));
    if (!st->cap_runs) { D2G_HIP(ctx, hipMalloc((void **)&st->d_run_list, (size_t(1) << 20) * 8)); st->cap_runs = size_t(1) << 20; }
    st->nbases = 0; st->k = k;
    st->run_start.clear(); st->run_len.clear();
    st->genome_run_off.assign(n + 1, 0); st->genome_nkmers.assign(n, 0);
    std::vector<uint64_t> fbase(nfiles + 1, 0);
    std::vector<uint64_t> starts;
    if (ntiles) {
        D2G_HIP(ctx, hipMemcpyAsync(st->d_raw, raw, raw_bytes, hipMemcpyHostToDevice, s));      // page-locked source: one DMA
        D2G_HIP(ctx, hipMemsetAsync(st->d_raw + raw_bytes, 0, 64, s));
        D2G_HIP(ctx, hipMemcpyAsync(st->d_files, files.data(), (nfiles + 1) * sizeof(K0File), hipMemcpyHostToDevice, s));
        D2G_HIP(ctx, hipMemsetAsync(st->d_ctl, 0, 16, s));
        D2G_HIP(ctx, hipMemsetAsync(sk->d_packed, 0, out_bytes + 64, s));
        d2g_timer tm(ctx, &ctx->ev_k0, s);
        hipLaunchKernelGGL(k0_newline_kernel, dim3((unsigned)ntiles), dim3(K0_THREADS), 0, s, st->d_raw, st->d_files, (uint32_t)nfiles, st->d_tile_nl);
        hipLaunchKernelGGL(k0_carry_nl_kernel, dim3((unsigned)nfiles), dim3(64), 0, s, st->d_files, (uint32_t)nfiles, st->d_tile_nl);
        hipLaunchKernelGGL(k0_count_kernel, dim3((unsigned)ntiles), dim3(K0_THREADS), 0, s, st->d_raw, st->d_files, (uint32_t)nfiles, st->d_tile_nl,
                           st->d_tile_nbase, st->d_tile_cls, st->d_ctl + 1);
        hipLaunchKernelGGL(k0_carry_base_kernel, dim3((unsigned)nfiles), dim3(64), 0, s, st->d_files, (uint32_t)nfiles, st->d_tile_nbase, st->d_tile_cls,
                           st->d_file_total);
        hipLaunchKernelGGL(k0_file_prefix_kernel, dim3(1), dim3(1024), 0, s, st->d_file_total, (uint32_t)nfiles, st->d_file_base);
        for (int attempt = 0;; ++attempt) {
            hipLaunchKernelGGL(k0_emit_kernel, dim3((unsigned)ntiles), dim3(K0_THREADS), 0, s, st->d_raw, st->d_files, (uint32_t)nfiles, st->d_tile_nl,
                               st->d_tile_nbase, st->d_tile_cls, st->d_file_base, reinterpret_cast<uint32_t *>(sk->d_packed), st->d_run_list,
                               (uint32_t)std::min<size_t>(st->cap_runs, 0xFFFFFFFFu), st->d_ctl, st->d_ctl + 1);
            tm.stop();
            D2G_HIP(ctx, hipGetLastError());
            uint32_t ctl[2] = {0, 0};
            D2G_HIP(ctx, hipMemcpyAsync(ctl, st->d_ctl, 8, hipMemcpyDeviceToHost, s));
            D2G_HIP(ctx, hipMemcpyAsync(fbase.data(), st->d_file_base, (nfiles + 1) * 8, hipMemcpyDeviceToHost, s));
            D2G_HIP(ctx, hipStreamSynchronize(s));
            if (ctl[1] & K0_ST_PLUS) { ctx->last_error = "ingest: a line starts with '+' (FASTQ quality section): host parser"; return D2G_ERR_UNSUPPORTED; }
            if (ctl[1] & K0_ST_RUNLIST) {                                // more run starts than the list holds: grow it and emit again (idempotent)
                if (attempt) { ctx->last_error = "ingest: run list overflow"; return D2G_ERR_INTERNAL; }
                (void)hipFree(st->d_run_list); st->d_run_list = nullptr; st->cap_runs = 0;
                const size_t ncap = (size_t)ctl[0] + 1024;
                D2G_HIP(ctx, hipMalloc((void **)&st->d_run_list, ncap * 8));
                st->cap_runs = ncap;
                D2G_HIP(ctx, hipMemsetAsync(st->d_ctl, 0, 16, s));
                continue;
            }
            starts.resize(ctl[0]);
            if (ctl[0]) D2G_HIP(ctx, hipMemcpy(starts.data(), st->d_run_list, (size_t)ctl[0] * 8, hipMemcpyDeviceToHost));
            break;
        }
    }
    // ---- the run table: lengths from consecutive starts, the >= k filter, the split of very long runs (what
    // d2g_seqpack::close_run_raw does), per-genome offsets and k-mer counts
    std::sort(starts.begin(), starts.end());
    uint32_t max_run = 1u << 30;
    if (const char *e = std::getenv("D2G_MAX_RUN")) { const long v = std::atol(e); if (v >= 64) max_run = (uint32_t)v; }     // (the environment, as d2g_seqpack reads it: ONE source for the host packer and this table -- ADVICE r5)
    size_t si = 0;
    for (size_t g = 0; g < n; ++g) {
        uint64_t nk = 0;
        for (uint64_t f = genome_file_off[g]; f < genome_file_off[g + 1]; ++f) {
            const uint64_t b0 = fbase[f], b1 = fbase[f + 1];
            while (si < starts.size() && starts[si] < b1) {
                const uint64_t s0 = starts[si], s1 = (si + 1 < starts.size() && starts[si + 1] < b1) ? starts[si + 1] : b1;
                ++si;
                if (s0 < b0) { ctx->last_error = "ingest: run start outside its file"; return D2G_ERR_INTERNAL; }
                uint64_t len = s1 - s0, at = s0;
                if (len < (uint64_t)k) continue;
                nk += len - k + 1;
                while (len > max_run) {
                    st->run_start.push_back(at); st->run_len.push_back(max_run);
                    const uint64_t adv = max_run - (uint64_t)(k - 1);
                    at += adv; len -= adv;
                }
                st->run_start.push_back(at); st->run_len.push_back((uint32_t)len);
            }
        }
        st->genome_run_off[g + 1] = st->run_start.size();
        st->genome_nkmers[g] = nk;
    }
    st->nbases = fbase[nfiles];
    st->valid = true;
    return D2G_OK;
}

int d2g_sketcher_ingested_runs(const d2g_sketcher *sk, const uint64_t **run_start, const uint32_t **run_len, size_t *nrun,
                               const uint64_t **genome_run_off, const uint64_t **genome_nkmers, uint64_t *nbases) {
    if (!sk || !sk->k0 || !sk->k0->valid) return D2G_ERR_INVALID;
    const d2g_k0_state *st = sk->k0;
    if (run_start) *run_start = st->run_start.data();
    if (run_len) *run_len = st->run_len.data();
    if (nrun) *nrun = st->run_start.size();
    if (genome_run_off) *genome_run_off = st->genome_run_off.data();
    if (genome_nkmers) *genome_nkmers = st->genome_nkmers.data();
    if (nbases) *nbases = st->nbases;
    return D2G_OK;
}

}  // extern "C"

void d2g_warm_k0() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k0_newline_kernel)); }
