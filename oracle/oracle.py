"""ctypes wrapper around oracle/liboracle_step2.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (w2rap_contigger_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle_step2.so")
REF_BIN = os.path.join(HERE, "_ref", "ref_step2")


def build(ref: bool = True):
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-j8", "-C", HERE, "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build(ref=False)
        L = C.CDLL(LIB)
        L.oracle_run.restype = C.c_void_p
        L.oracle_run.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint,
                                 C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_error.restype = C.c_char_p
        L.oracle_error.argtypes = [C.c_void_p]
        L.oracle_free.argtypes = [C.c_void_p]
        for name, nargs in (("oracle_sizes", 2), ("oracle_good_len", 2), ("oracle_hist", 2), ("oracle_table", 7),
                            ("oracle_edges", 3), ("oracle_objs", 7), ("oracle_paths", 4)):
            getattr(L, name).argtypes = [C.c_void_p] * nargs
            getattr(L, name).restype = None
        L.oracle_adj.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_adj.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@dataclass
class OracleResult:
    n_reads: int
    n_instances: int = 0
    n_distinct: int = 0
    good_len: np.ndarray = None
    hist: np.ndarray = None
    # solid table sorted by k-mer
    k_hi: np.ndarray = None
    k_lo: np.ndarray = None
    k_count: np.ndarray = None
    k_ctx: np.ndarray = None
    k_edge: np.ndarray = None
    k_off: np.ndarray = None
    # unipaths
    edge_codes: np.ndarray = None
    edge_off: np.ndarray = None
    # HBV
    n_vertices: int = 0
    obj_codes: np.ndarray = None
    obj_off: np.ndarray = None
    left: np.ndarray = None
    right: np.ndarray = None
    fwdX: np.ndarray = None
    revX: np.ndarray = None
    from_off: np.ndarray = None
    from_v: np.ndarray = None
    from_e: np.ndarray = None
    to_off: np.ndarray = None
    to_v: np.ndarray = None
    to_e: np.ndarray = None
    # paths
    path_offset: np.ndarray = None
    path_off: np.ndarray = None
    path_edges: np.ndarray = None
    pathed: int = 0
    multipathed: int = 0


def run(codes: np.ndarray, quals: np.ndarray, off: np.ndarray, min_qual=7, min_freq=4,
        hint_codes: np.ndarray = None, hint_off: np.ndarray = None, stop_after=0) -> OracleResult:
    """codes/quals: u8 concatenated (one base code / quality per byte); off: u64[n+1].
    hint_*: canonical edge sequences in the order to replay (None = lexicographic order).
    stop_after: 0 whole Step 2, 1 k-mer table only, 2 graph only."""
    L = lib()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    if hint_codes is not None:
        hint_codes = np.ascontiguousarray(hint_codes, dtype=np.uint8)
        hint_off = np.ascontiguousarray(hint_off, dtype=np.uint64)
        nh, hp, hop = len(hint_off) - 1, _p(hint_codes), _p(hint_off)
    else:
        nh, hp, hop = 0, None, None
    h = L.oracle_run(n, _p(codes), _p(quals), _p(off), min_qual, min_freq, nh, hp, hop, stop_after)
    try:
        e = L.oracle_error(h)
        if e:
            raise RuntimeError(e.decode())
        sz = np.zeros(11, dtype=np.uint64)
        L.oracle_sizes(h, _p(sz))
        sz = [int(x) for x in sz]
        r = OracleResult(n_reads=n, n_instances=sz[0], n_distinct=sz[1])
        r.good_len = np.zeros(n, dtype=np.uint16)
        L.oracle_good_len(h, _p(r.good_len))
        r.hist = np.zeros(101, dtype=np.uint64)
        L.oracle_hist(h, _p(r.hist))
        S = sz[2]
        r.k_hi = np.zeros(S, np.uint64); r.k_lo = np.zeros(S, np.uint64)
        r.k_count = np.zeros(S, np.uint8); r.k_ctx = np.zeros(S, np.uint8)
        r.k_edge = np.zeros(S, np.int32); r.k_off = np.zeros(S, np.uint32)
        L.oracle_table(h, _p(r.k_hi), _p(r.k_lo), _p(r.k_count), _p(r.k_ctx), _p(r.k_edge), _p(r.k_off))
        if stop_after == 1:
            return r
        E, NO, NV = sz[3], sz[4], sz[5]
        r.edge_codes = np.zeros(sz[9], np.uint8); r.edge_off = np.zeros(E + 1, np.uint64)
        L.oracle_edges(h, _p(r.edge_codes), _p(r.edge_off))
        r.n_vertices = NV
        r.obj_codes = np.zeros(sz[10], np.uint8); r.obj_off = np.zeros(NO + 1, np.uint64)
        r.left = np.zeros(NO, np.int32); r.right = np.zeros(NO, np.int32)
        r.fwdX = np.zeros(E, np.int32); r.revX = np.zeros(E, np.int32)
        L.oracle_objs(h, _p(r.obj_codes), _p(r.obj_off), _p(r.left), _p(r.right), _p(r.fwdX), _p(r.revX))
        for which, (o, v) in enumerate((("from_off", "from_v"), ("from_off", "from_e"), ("to_off", "to_v"), ("to_off", "to_e"))):
            offa = np.zeros(NV + 1, np.uint64); vals = np.zeros(NO, np.int32)
            L.oracle_adj(h, which, _p(offa), _p(vals))
            setattr(r, o, offa); setattr(r, v, vals)
        if stop_after == 2:
            return r
        r.path_offset = np.zeros(n, np.int32); r.path_off = np.zeros(n + 1, np.uint64)
        r.path_edges = np.zeros(sz[6], np.int32)
        L.oracle_paths(h, _p(r.path_offset), _p(r.path_off), _p(r.path_edges))
        r.pathed, r.multipathed = sz[7], sz[8]
        return r
    finally:
        L.oracle_free(h)


def to_hbv(r: OracleResult):
    """OracleResult -> w2rap_contigger_amd.formats.HBV (for byte-level comparison with the reference's .hbv)"""
    from w2rap_contigger_amd import formats as F
    packed, boff, lens = F.pack_bases(r.obj_codes, r.obj_off)
    return F.HBV(60, r.from_off, r.from_v, r.from_e, r.to_off, r.to_e, packed, boff, lens)


def edge_hint_from_hbv(hbv):
    """The reference's unipath order = its HBV edge objects whose sequence is not
    REV-canonical, in id order (addEdge canonicalises, BuildReadQGraph.cc:278-281; fwd
    precedes rc, HBVFromEdges.cc:140-149).  -> (codes u8, off u64[n+1])"""
    codes, off = hbv.edge_codes()
    off = off.astype(np.int64)
    keep = []
    for e in range(hbv.n_edges):
        s = codes[off[e]:off[e + 1]]
        if eform(s) != 1:
            keep.append(s)
    hoff = np.zeros(len(keep) + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in keep], out=hoff[1:])
    return (np.concatenate(keep) if keep else np.zeros(0, np.uint8)), hoff


def eform(s: np.ndarray) -> int:
    """bvec::getCanonicalForm (dna/CanonicalForm.h:34-46): 0 FWD, 1 REV, 2 PALINDROME"""
    n = len(s)
    if n & 1:
        return 1 if (int(s[n // 2]) & 2) else 0
    r = 3 - s[::-1]
    d = np.nonzero(s != r)[0]
    if len(d) == 0:
        return 2
    i = d[0]
    return 0 if s[i] < r[i] else 1


def run_reference(workdir: str, prefix="t", threads=1, min_qual=7, min_freq=4) -> float:
    """Run the real reference Step 2 (oracle/_ref/ref_step2) on workdir/frag_reads_orig.{fastb,qualp}.
    -> seconds spent in buildReadQGraph + FixPaths as printed by the driver."""
    if not os.path.exists(REF_BIN):
        raise FileNotFoundError(REF_BIN)
    env = dict(os.environ, OMP_PROC_BIND="spread", MALLOC_PER_THREAD="1", OMP_NUM_THREADS=str(threads))
    out = subprocess.run([REF_BIN, workdir, prefix, str(threads), str(min_qual), str(min_freq)],
                         check=True, capture_output=True, text=True, env=env).stdout
    for line in out.splitlines():
        if line.startswith("REF_TIME"):
            t = line.split()
            return float(t[2]) + float(t[4])
    raise RuntimeError("ref_step2 printed no REF_TIME line:\n" + out[-2000:])
