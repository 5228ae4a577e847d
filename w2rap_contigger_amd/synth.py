"""Seeded synthetic inputs for Step 2 (SURVEY.md 8d distributions).

Genomes are numpy u8 code arrays (A=0,C=1,G=2,T=3).  Reads are sampled with
torch so the same code fills host memory for fixtures and HBM for the bench:
PE150, insert 400, random strand, R1/R2 interleaved (reference Step 1 layout,
paths/long/large/ExtractReads.cc:474), 0.5 % substitutions with q in U[2,12] at
errors, q in {30,32,35,37,40} elsewhere, 20 % of reads with a Q2 tail of
U[1,20] bases.
"""
from __future__ import annotations

import numpy as np
import torch

READ_LEN = 150
INSERT = 400
GOOD_Q = (30, 32, 35, 37, 40)


def rc_codes(a: np.ndarray) -> np.ndarray:
    return (3 - a[::-1]).astype(np.uint8)


def random_genome(length: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 4, length, dtype=np.uint8)


def fixture_genome(kind: str, seed: int):
    """Adversarial fixture genomes: -> list of linear contigs
    (circular replicons are returned unrolled by INSERT-1 bases so fragments wrap)."""
    rng = np.random.default_rng(seed)
    if kind == "random":
        return [rng.integers(0, 4, 20_000, dtype=np.uint8)]
    if kind == "repeats_snps":
        G = 60_000
        g = rng.integers(0, 4, G, dtype=np.uint8)
        rep = rng.integers(0, 4, 500, dtype=np.uint8)
        for p in (3_000, 15_000, 27_000, 39_000, 51_000):          # 5 x 500 bp exact repeat
            g[p:p + 500] = rep
        inv = g[9_000:9_300].copy()                                # 300 bp inverted repeat
        g[45_000:45_300] = rc_codes(inv)
        hap2 = g.copy()                                            # 60 heterozygous SNPs
        for p in rng.choice(np.arange(1_000, G - 1_000), 60, replace=False):
            hap2[p] = (hap2[p] + 1 + rng.integers(0, 3)) & 3
        return [g, hap2]
    if kind == "palindrome_circle":
        g = rng.integers(0, 4, 20_000, dtype=np.uint8)
        half = rng.integers(0, 4, 30, dtype=np.uint8)
        g[5_000:5_060] = np.concatenate([half, rc_codes(half)])    # a 60-mer palindrome
        half2 = rng.integers(0, 4, 40, dtype=np.uint8)
        g[12_000:12_080] = np.concatenate([half2, rc_codes(half2)])  # an 80 bp palindrome
        plasmid = rng.integers(0, 4, 3_000, dtype=np.uint8)        # circular replicon
        return [g, np.concatenate([plasmid, plasmid[:INSERT - 1]])]
    raise ValueError(kind)


def sample_reads(contigs, n_pairs: int, seed: int, device="cpu", err_rate=0.005, tail_frac=0.2,
                 read_len=READ_LEN, insert=INSERT):
    """-> (codes u8 [2*n_pairs, read_len], quals u8 [2*n_pairs, read_len]) torch tensors on `device`."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    genome = torch.from_numpy(np.concatenate(contigs)).to(dev)
    usable = np.maximum(lens - insert + 1, 0)
    if usable.sum() <= 0:
        raise ValueError("contigs shorter than the insert size")
    cdf = torch.from_numpy(np.cumsum(usable).astype(np.int64)).to(dev)
    u = torch.randint(0, int(usable.sum()), (n_pairs,), generator=gen, device=dev)
    cid = torch.searchsorted(cdf, u, right=True)
    prev = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), cdf[:-1]])
    frag = torch.from_numpy(starts).to(dev)[cid] + (u - prev[cid])          # fragment start in concatenated genome
    strand = torch.randint(0, 2, (n_pairs,), generator=gen, device=dev, dtype=torch.int64)
    ar = torch.arange(read_len, device=dev, dtype=torch.int64)
    left = genome[frag[:, None] + ar[None, :]]                                # fragment left end, forward
    right = 3 - genome[frag[:, None] + (insert - 1) - ar[None, :]]            # rc of fragment right end
    s = strand[:, None].bool()
    r1 = torch.where(s, right, left)
    r2 = torch.where(s, left, right)
    codes = torch.stack([r1, r2], dim=1).reshape(2 * n_pairs, read_len).to(torch.uint8)
    n = 2 * n_pairs
    err = torch.rand((n, read_len), generator=gen, device=dev) < err_rate
    shift = torch.randint(1, 4, (n, read_len), generator=gen, device=dev, dtype=torch.uint8)
    codes = torch.where(err, (codes + shift) & 3, codes)
    gq = torch.tensor(GOOD_Q, dtype=torch.uint8, device=dev)
    quals = gq[torch.randint(0, len(GOOD_Q), (n, read_len), generator=gen, device=dev)]
    eq = torch.randint(2, 13, (n, read_len), generator=gen, device=dev, dtype=torch.uint8)
    quals = torch.where(err, eq, quals)
    has_tail = torch.rand((n,), generator=gen, device=dev) < tail_frac
    tail = torch.randint(1, 21, (n,), generator=gen, device=dev)
    in_tail = has_tail[:, None] & (ar[None, :] >= (read_len - tail)[:, None])
    quals = torch.where(in_tail, torch.full_like(quals, 2), quals)
    return codes.contiguous(), quals.contiguous()


def pack_fixed(codes: torch.Tensor) -> torch.Tensor:
    """[n, L] base codes -> [n, ceil(L/4)] packed bytes (.fastb per-read layout, base i at bits 2*(i%4))"""
    n, L = codes.shape
    pad = (-L) % 4
    if pad:
        codes = torch.cat([codes, torch.zeros((n, pad), dtype=codes.dtype, device=codes.device)], dim=1)
    c = codes.reshape(n, -1, 4).to(torch.uint8)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).contiguous()


def edge_case_reads(rng: np.random.Generator, genome: np.ndarray):
    """Hand-made reads for the quirks of SURVEY.md 8a (Q1, Q2, short reads ...).
    -> list of (codes u8, quals u8), deliberately ragged."""
    out = []
    g = genome
    def q(n, v=35):
        return np.full(n, v, np.uint8)
    # read shorter than K
    out.append((g[100:159].copy(), q(59)))
    out.append((g[100:130].copy(), q(30)))
    # exactly K, all good: contributes nothing (len > K is strict, BuildReadQGraph.cc:1064)
    for _ in range(6):
        out.append((g[200:260].copy(), q(60)))
    # good window ends exactly at base 60 (Q2) / 61
    for _ in range(6):
        qq = q(150); qq[60:] = 2
        out.append((g[300:450].copy(), qq))
        qq = q(150); qq[61:] = 2
        out.append((g[500:650].copy(), qq))
    # Q2 prefix, good window [50,110): k-mers are taken from p=0 anyway (Q1)
    for _ in range(6):
        qq = q(150); qq[:50] = 2; qq[110:] = 2
        out.append((g[700:850].copy(), qq))
    # no 60-long good window at all
    qq = q(150); qq[::50] = 2
    out.append((g[900:1050].copy(), qq))
    # ragged lengths
    for L in (61, 75, 100, 149, 151, 200, 251):
        for _ in range(5):
            out.append((g[1100:1100 + L].copy(), q(L)))
            out.append((rc_codes(g[1100:1100 + L]), q(L)))
    # a k-mer with coverage >= 255 (count saturation, :943-949)
    for _ in range(300):
        out.append((g[2000:2070].copy(), q(70)))
    # an empty read
    out.append((np.zeros(0, np.uint8), np.zeros(0, np.uint8)))
    # qualities below min_qual scattered + quality 63 (max allowed)
    for _ in range(5):
        qq = rng.integers(0, 64, 150).astype(np.uint8)
        out.append((g[3000:3150].copy(), qq))
    return out


def sample_reads_t(genome: torch.Tensor, n_pairs: int, seed: int, device="cuda", err_rate=0.005, tail_frac=0.2,
                   read_len=READ_LEN, insert=INSERT):
    """sample_reads for ONE linear contig already resident on `device` (bench-scale generation)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    G = genome.numel()
    frag = torch.randint(0, G - insert + 1, (n_pairs,), generator=gen, device=dev)
    strand = torch.randint(0, 2, (n_pairs,), generator=gen, device=dev, dtype=torch.int64)
    ar = torch.arange(read_len, device=dev, dtype=torch.int64)
    left = genome[frag[:, None] + ar[None, :]]
    right = 3 - genome[frag[:, None] + (insert - 1) - ar[None, :]]
    s = strand[:, None].bool()
    codes = torch.stack([torch.where(s, right, left), torch.where(s, left, right)], dim=1).reshape(2 * n_pairs, read_len).to(torch.uint8)
    n = 2 * n_pairs
    err = torch.rand((n, read_len), generator=gen, device=dev) < err_rate
    shift = torch.randint(1, 4, (n, read_len), generator=gen, device=dev, dtype=torch.uint8)
    codes = torch.where(err, (codes + shift) & 3, codes)
    gq = torch.tensor(GOOD_Q, dtype=torch.uint8, device=dev)
    quals = gq[torch.randint(0, len(GOOD_Q), (n, read_len), generator=gen, device=dev)]
    eq = torch.randint(2, 13, (n, read_len), generator=gen, device=dev, dtype=torch.uint8)
    quals = torch.where(err, eq, quals)
    has_tail = torch.rand((n,), generator=gen, device=dev) < tail_frac
    tail = torch.randint(1, 21, (n,), generator=gen, device=dev)
    in_tail = has_tail[:, None] & (ar[None, :] >= (read_len - tail)[:, None])
    quals = torch.where(in_tail, torch.full_like(quals, 2), quals)
    return codes.contiguous(), quals.contiguous()


def unpack_fixed(packed: torch.Tensor, read_len: int) -> torch.Tensor:
    """inverse of pack_fixed -> [n, read_len] base codes"""
    n = packed.shape[0]
    c = torch.stack([(packed >> (2 * j)) & 3 for j in range(4)], dim=2).reshape(n, -1)
    return c[:, :read_len].contiguous()


def generate_reads_device(n_reads: int, genome_len: int, seed: int, device="cuda", chunk_pairs=1 << 20, genome=None):
    """Bench-scale synthetic PE150 reads generated directly in HBM.
    -> dict(packed [n,38] u8, quals [n,150] u8, byte_off i64[n+1], qual_off i64[n+1], read_len i32[n], genome)"""
    dev = torch.device(device)
    if genome is None:
        genome = torch.randint(0, 4, (genome_len,), dtype=torch.uint8, device=dev,
                               generator=torch.Generator(device=dev).manual_seed(seed))
    n_pairs = n_reads // 2
    n = 2 * n_pairs
    nb = (READ_LEN + 3) // 4
    packed = torch.empty((n, nb), dtype=torch.uint8, device=dev)
    quals = torch.empty((n, READ_LEN), dtype=torch.uint8, device=dev)
    done = 0
    while done < n_pairs:
        m = min(chunk_pairs, n_pairs - done)
        c, q = sample_reads_t(genome, m, seed * 1000003 + done + 1, device=dev)
        packed[2 * done:2 * (done + m)] = pack_fixed(c)
        quals[2 * done:2 * (done + m)] = q
        done += m
    byte_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * nb
    qual_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * READ_LEN
    read_len = torch.full((n,), READ_LEN, dtype=torch.int32, device=dev)
    return dict(n=n, packed=packed, quals=quals, byte_off=byte_off, qual_off=qual_off, read_len=read_len, genome=genome)
