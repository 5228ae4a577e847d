"""Hand-made Step-3 inputs (small-K edge objects + read paths) for the corners the Step-2 fixtures do not reach:
a smooth circle in the large-K graph, a palindromic K2-mer, places that wrap a circular edge several times, long multi-edge
paths, a place of exactly K2 bases.  Shared by the CPU test (oracle against the reference binary) and the GPU test
(library against the oracle)."""
import numpy as np

from w2rap_contigger_amd import formats as F

K = 60


def rc(a):
    return (3 - np.asarray(a, np.uint8)[::-1]).astype(np.uint8)


def make_hbv(seqs):
    """an HBV holding the given edge objects, every object between two vertices of its own (Step 3 without extend_paths
    reads only the edge objects of the graph; the adjacency just has to load)"""
    n = len(seqs)
    codes = np.concatenate(seqs).astype(np.uint8)
    off = np.zeros(n + 1, np.uint64); np.cumsum([len(s) for s in seqs], out=off[1:])
    packed, boff, lens = F.pack_bases(codes, off)
    nv = 2 * n
    from_off = np.zeros(nv + 1, np.uint64); to_off = np.zeros(nv + 1, np.uint64)
    from_off[1:] = np.repeat(np.arange(1, n + 1), 2)[:nv] if n else 0          # vertex 2o has one out-edge, 2o+1 none
    for o in range(n):
        from_off[2 * o + 1] = o + 1; from_off[2 * o + 2] = o + 1
        to_off[2 * o + 1] = o; to_off[2 * o + 2] = o + 1
    from_v = np.array([2 * o + 1 for o in range(n)], np.int32)
    from_e = np.arange(n, dtype=np.int32); to_e = np.arange(n, dtype=np.int32)
    return F.HBV(K, from_off, from_v, from_e, to_off, to_e, packed, boff, lens)


def make_paths(plist, offsets=None):
    n = len(plist)
    off = np.zeros(n + 1, np.uint64); np.cumsum([len(p) for p in plist], out=off[1:])
    edges = np.array([e for p in plist for e in p], np.int32)
    po = np.zeros(n, np.int32) if offsets is None else np.asarray(offsets, np.int32)
    return po, off, edges


def case(name, seed=5):
    rng = np.random.default_rng(seed)
    R = lambda n: rng.integers(0, 4, n, dtype=np.uint8)
    if name == "circle":
        # a circular small-K edge (its first K-1 bases repeat at the end) read around the junction: the K2-mers close a smooth circle
        circ = R(1000)
        e = np.concatenate([circ, circ[:K - 1]])
        seqs = [e, rc(e)]
        paths = [[0], [0, 0], [1, 1], [0, 0, 0], [1], [], [0, 0], [1]]          # (reads come in pairs: FragDist takes 2i with 2i+1)
        return make_hbv(seqs), make_paths(paths, [3, 900, 950, 990, 7, 0, -20, 40])
    if name == "palindrome":
        x = R(100)
        pal = np.concatenate([x, rc(x)])                       # a 200-base palindrome: one palindromic K2-mer at K2 = 200
        e = np.concatenate([R(300), pal, R(300)])
        short = np.concatenate([R(100), pal[:150]])            # a second edge sharing 150 bases of it (K2-mers differ)
        seqs = [e, rc(e), short, rc(short), pal.copy()]        # the palindrome itself as an edge object: its own reverse complement
        paths = [[0], [1], [2], [3], [4], [4], [0], []]
        return make_hbv(seqs), make_paths(paths, [0, 5, 1, 2, 0, 3, 700, 0])
    if name == "chains":
        # a chain of short edges overlapping by K-1: long multi-edge paths, truncation at both ends, a place of exactly K2 bases
        g = R(4000)
        cuts = [0, 700, 761, 830, 1500, 1561, 1640, 2400, 2470, 3300, 4000 - (K - 1)]
        fw = [g[cuts[i]:cuts[i + 1] + K - 1] for i in range(len(cuts) - 1)]
        seqs = []
        for s in fw:
            seqs += [s, rc(s)]
        nE = len(fw)
        P = lambda *ix: [2 * i for i in ix]
        Pr = lambda *ix: [2 * i + 1 for i in reversed(ix)]
        exact = g[100:100 + 200]                                # an edge of exactly 200 bases
        seqs += [exact, rc(exact)]
        paths = [P(0, 1, 2, 3), Pr(0, 1, 2, 3), P(1, 2), P(2), P(3, 4, 5, 6, 7, 8, 9), Pr(5, 6, 7), P(4, 5), [2 * nE], [2 * nE + 1], P(1), P(0), Pr(9),
                 P(2, 3), [], P(6, 7, 8), Pr(0)]
        return make_hbv(seqs), make_paths(paths, list(range(-7, -7 + len(paths))))
    raise ValueError(name)


CASES = ["circle", "palindrome", "chains"]
