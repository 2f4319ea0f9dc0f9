"""Deterministic synthetic inputs for tests and bench.py.

* genomes: i.i.d. uniform ACGT from splitmix64(0xD2D2 + index), optional mutated families
* sketches: valid One-Permutation register matrices with planted shared registers, so that
  pairwise equality counts span 0..S (clustered collections look like this in practice)
This module only produces INPUTS; it computes nothing that is part of the hot path.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64_stream(seed, n):
    """n uint64 values of the splitmix64 sequence started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


_ASCII4 = None


def random_genome(index, length, seed=0xD2D2):
    """uint8 array of ASCII ACGT, 2 bits per base from splitmix64(seed + index) (base p = bits
    [2(p%32), +2) of word p/32)."""
    global _ASCII4
    if _ASCII4 is None:         # byte -> its four bases as one little-endian u32
        b = np.arange(256, dtype=np.uint32)
        acgt = np.frombuffer(b"ACGT", np.uint8).astype(np.uint32)
        _ASCII4 = acgt[b & 3] | (acgt[(b >> 2) & 3] << 8) | (acgt[(b >> 4) & 3] << 16) | (acgt[(b >> 6) & 3] << 24)
    nwords = (length + 31) // 32
    w = splitmix64_stream(seed + index, nwords)
    return _ASCII4[w.view(np.uint8)].view(np.uint8)[:length]


def mutate(genome, rate, seed):
    """per-base substitution with probability `rate` (always to a different base)."""
    rng = np.random.default_rng(seed)
    g = genome.copy()
    hit = rng.random(g.size) < rate
    lut = np.zeros(256, np.uint8)
    lut[list(b"ACGT")] = [0, 1, 2, 3]
    codes = lut[g[hit]]
    codes = (codes + rng.integers(1, 4, codes.size).astype(np.uint8)) & 3
    g[hit] = np.frombuffer(b"ACGT", np.uint8)[codes]
    return g


def fasta_bytes(name, genome, width=80):
    body = genome.tobytes()
    lines = [body[i:i + width] for i in range(0, len(body), width)]
    return b">" + name.encode() + b"\n" + b"\n".join(lines) + b"\n"


def write_fasta(path, name, genome, width=80):
    with open(path, "wb") as f:
        f.write(fasta_bytes(name, genome, width))


def synthetic_registers(N, S, nclusters=64, seed=1234, expected_kmers_per_bucket=4883, share_lo=0.0, share_hi=0.98):
    """OPH-like u64 register matrix [N][S] with planted equalities.

    register value model: min of ~expected_kmers_per_bucket uniform 64-bit hashes whose
    residue mod S is the bucket index (what K1 produces for a ~5 Mbp genome at S=1024).
    Sketch j belongs to cluster j % nclusters and copies each register from its cluster parent
    with probability f_j (f_j spread over [share_lo, share_hi]), else draws a fresh value.
    """
    rng = np.random.default_rng(seed)
    S = int(S)

    def draw(shape):
        # min of n uniforms ~ Exp(n)/2^-64 ; quantise so that value % S == bucket when S | 2^64
        u = rng.exponential(1.0 / expected_kmers_per_bucket, size=shape)
        v = np.minimum(u, 0.999999) * float(2 ** 64)
        v = v.astype(np.uint64)
        t = np.arange(S, dtype=np.uint64)
        t = np.broadcast_to(t, shape)
        if S & (S - 1) == 0:
            v = (v & ~np.uint64(S - 1)) | t
        else:
            v = v - (v % np.uint64(S)) + t
        return v

    parents = draw((nclusters, S))
    regs = draw((N, S))
    f = rng.uniform(share_lo, share_hi, size=N)
    take = rng.random((N, S)) < f[:, None]
    cl = np.arange(N) % nclusters
    regs = np.where(take, parents[cl], regs)
    return regs.astype(np.uint64)


def fasta_bytes_fast(name, genome, width=80):
    """same bytes as fasta_bytes, built with array operations (5 Mbp in a few ms)"""
    n = genome.size
    full = n // width
    body = np.empty((full, width + 1), np.uint8)
    body[:, :width] = genome[:full * width].reshape(full, width)
    body[:, width] = 10
    tail = genome[full * width:].tobytes()
    return b">" + name.encode() + b"\n" + body.tobytes() + (tail + b"\n" if tail else b"")


def unrelated_registers(N, S, seed=99):
    """every register value distinct within its column (unrelated genomes): no pair shares anything"""
    rng = np.random.default_rng(seed)
    v = (rng.permuted(np.tile(np.arange(N, dtype=np.uint64), (S, 1)), axis=1).T << np.uint64(20))
    return np.ascontiguousarray(v | np.arange(S, dtype=np.uint64)[None, :]) + np.uint64(1 << 40)


def paired_registers(N, S, seed=98):
    """adversarial for the bit-sliced operand: in every column each value occurs exactly twice (random
    pairing per column), so a column holds N/2 shared values and needs ceil(log2(N/2 + 1)) id planes"""
    rng = np.random.default_rng(seed)
    base = np.tile(np.arange(N, dtype=np.uint64) // np.uint64(2), (S, 1))
    v = (rng.permuted(base, axis=1).T << np.uint64(20))
    return np.ascontiguousarray(v | np.arange(S, dtype=np.uint64)[None, :]) + np.uint64(1 << 40)


def skewed_registers(N, S, seed=97, max_shared=64, share=0.7):
    """a collection whose register columns differ widely in how many values they share: column t has K_t shared values,
    K_t log-uniform in [0, max_shared] (K_t + 1 = exp(U(0, ln(max_shared + 1)))); a sketch takes one of them with
    probability `share`, else a value of its own.  The bit-sliced operand needs ceil(log2(K_t + 2)) id planes for such a
    column; a 32-register group needs the maximum over its columns -- which column lands in which group matters."""
    rng = np.random.default_rng(seed)
    K = np.floor(np.exp(rng.uniform(0.0, np.log(max_shared + 1.0), size=S))).astype(np.int64) - 1
    K = np.clip(K, 0, max_shared)
    own = (np.arange(N, dtype=np.uint64)[:, None] + np.uint64(1 << 20)) << np.uint64(24)        # distinct per sketch
    pick = rng.integers(0, np.maximum(K, 1)[None, :], size=(N, S)).astype(np.uint64) << np.uint64(24)
    take = (rng.random((N, S)) < share) & (K[None, :] > 0)
    v = np.where(take, pick, own)
    return np.ascontiguousarray(v | np.arange(S, dtype=np.uint64)[None, :]) + np.uint64(1 << 50)


def add_chance_collisions(regs, c, seed=4242):
    """`c` chance collisions per sketch with random strangers: sketch j takes, in `c` random register columns, the value a random
    OTHER sketch holds there (about half of those are values the stranger copied from its cluster parent, so the collision is with
    every member of that family that kept the register; the rest are values only the stranger had).  What a collection of related
    genomes looks like once a few conserved k-mers are shared across families: every 32 x 256 tile holds a pair with a common
    value, although only the families' own pairs have many.  Returns a new matrix; column residues are preserved (a value moves
    within its column)."""
    rng = np.random.default_rng(seed)
    N, S = regs.shape
    out = regs.copy()
    if c <= 0 or N < 2:
        return out
    cols = np.stack([rng.choice(S, size=min(c, S), replace=False) for _ in range(N)])        # [N][c]
    other = rng.integers(0, N - 1, size=cols.shape)
    other = other + (other >= np.arange(N)[:, None])                                          # != j
    out[np.arange(N)[:, None], cols] = regs[other, cols]
    return out


def family_collection(nfam, per_fam, L, seed=0xFA17, rate_lo=0.0005, rate_hi=0.03, nseg=0, seg_len=1500, segs_per_family=0):
    """A collection of RELATED GENOMES (not planted registers): `nfam` families of `per_fam` genomes of `L` bases.  A family is a random base genome
    (random_genome) and its members are copies with per-base substitutions at a member-specific rate, log-uniform in [rate_lo, rate_hi] (mutate_sparse).
    Conserved segments (nseg > 0): `nseg` random sequences of `seg_len` bases exist in the collection; every family's base genome carries `segs_per_family`
    of them at random positions (drawn without replacement), so that genomes of DIFFERENT families share the k-mers of a segment they both carry -- what
    mobile elements, rRNA operons and other conserved genes do to real collections.  Random 31-mers of different random genomes practically never
    collide, so without segments the sketches of different families share nothing.
    Yields (family index, name, genome as a uint8 ASCII array) in the order family 0 member 0, family 1 member 0, ... (families interleaved, the way
    synthetic_registers deals its clusters)."""
    rng = np.random.default_rng(seed)
    segs = [random_genome(1_000_000 + s, seg_len, seed=seed) for s in range(nseg)]
    bases = []
    for f in range(nfam):
        g = random_genome(f, L, seed=seed + 7).copy()
        if nseg and segs_per_family:
            for s in rng.choice(nseg, size=min(segs_per_family, nseg), replace=False):
                p = int(rng.integers(0, max(1, L - seg_len)))
                g[p:p + seg_len] = segs[s][:max(0, min(seg_len, L - p))]
        bases.append(g)
    rates = np.exp(rng.uniform(np.log(rate_lo), np.log(rate_hi), size=(per_fam, nfam)))
    for m in range(per_fam):
        for f in range(nfam):
            yield f, "f%03d_m%03d" % (f, m), mutate_sparse(bases[f], float(rates[m, f]), seed=int(rng.integers(1 << 62)))


def mutate_sparse(genome, rate, seed):
    """per-base substitution with probability `rate` (always to a different base); draws the mutated POSITIONS (binomial count, then positions with
    replacement: a position hit twice mutates once) instead of one uniform per base -- 200 kbp at 1 % in ~0.1 ms"""
    rng = np.random.default_rng(seed)
    g = genome.copy()
    n = int(rng.binomial(g.size, rate))
    if n == 0:
        return g
    pos = np.unique(rng.integers(0, g.size, size=n))
    lut = np.zeros(256, np.uint8)
    lut[list(b"ACGT")] = [0, 1, 2, 3]
    codes = (lut[g[pos]] + rng.integers(1, 4, pos.size).astype(np.uint8)) & 3
    g[pos] = np.frombuffer(b"ACGT", np.uint8)[codes]
    return g
