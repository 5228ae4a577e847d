"""ctypes wrapper around oracle/liboracle_step3.so (Step 3: Involution, FragDist, RepathInMemory) -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package never imports it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle_step3.so")
REF3_BIN = os.path.join(HERE, "_ref", "ref_step3")

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
        L = C.CDLL(LIB)
        L.oracle3_run.restype = C.c_void_p
        L.oracle3_run.argtypes = [C.c_uint, C.c_uint, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
        L.oracle3_error.restype = C.c_char_p
        L.oracle3_error.argtypes = [C.c_void_p]
        L.oracle3_free.argtypes = [C.c_void_p]
        for name, nargs in (("oracle3_sizes", 2), ("oracle3_inv", 2), ("oracle3_frag", 2), ("oracle3_places", 5), ("oracle3_all", 3),
                            ("oracle3_objs", 6), ("oracle3_paths", 4)):
            getattr(L, name).argtypes = [C.c_void_p] * nargs
            getattr(L, name).restype = None
        L.oracle3_adj.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle3_adj.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@dataclass
class Oracle3Result:
    K2: int
    inv: np.ndarray = None
    frag: np.ndarray = None            # FragDist counts, 100 bins
    place_off: np.ndarray = None       # unique places (sorted), CSR
    place_edges: np.ndarray = None
    left_trunc: np.ndarray = None
    right_trunc: np.ndarray = None
    all_codes: np.ndarray = None
    all_off: np.ndarray = None
    n_instances: int = 0
    n_distinct: int = 0
    n_edges: int = 0
    n_vertices: int = 0
    obj_codes: np.ndarray = None
    obj_off: np.ndarray = None
    left: np.ndarray = None
    right: np.ndarray = None
    inv2: np.ndarray = None
    from_off: np.ndarray = None
    from_v: np.ndarray = None
    from_e: np.ndarray = None
    to_off: np.ndarray = None
    to_v: np.ndarray = None
    to_e: np.ndarray = None
    path_offset: np.ndarray = None
    path_off: np.ndarray = None
    path_edges: np.ndarray = None


def run(hbv, paths, K2=200, hint_codes=None, hint_off=None, stop_after=0, extend_paths=False) -> Oracle3Result:
    """hbv: formats.HBV of the small-K graph; paths: (offset i32[n], path_off u64[n+1], edges i32[]) as formats.read_paths gives.
    hint_*: the large-K canonical edges in the order to replay (None = lexicographic).  extend_paths: Repath.cc:72-96 (--extend_paths)."""
    L = lib()
    codes, off = hbv.edge_codes()
    codes = np.ascontiguousarray(codes, np.uint8); off = np.ascontiguousarray(off, np.uint64)
    po = np.ascontiguousarray(paths[0], np.int32); pf = np.ascontiguousarray(paths[1], np.uint64); pe = np.ascontiguousarray(paths[2], np.int32)
    if hint_codes is not None:
        hint_codes = np.ascontiguousarray(hint_codes, np.uint8); hint_off = np.ascontiguousarray(hint_off, np.uint64)
        nh, hp, hop = len(hint_off) - 1, _p(hint_codes), _p(hint_off)
    else:
        nh, hp, hop = 0, None, None
    n = len(po)
    tl = tr = None
    if extend_paths:
        tl, tr = hbv.to_left_right()
        tl = np.ascontiguousarray(tl, np.int32); tr = np.ascontiguousarray(tr, np.int32)
    h = L.oracle3_run(hbv.K, K2, len(off) - 1, _p(codes), _p(off), n, _p(po), _p(pf), _p(pe), nh, hp, hop, stop_after,
                      1 if extend_paths else 0, hbv.n_vertices if extend_paths else 0, _p(tl) if extend_paths else None, _p(tr) if extend_paths else None)
    try:
        e = L.oracle3_error(h)
        if e:
            raise RuntimeError(e.decode())
        sz = np.zeros(10, np.uint64); L.oracle3_sizes(h, _p(sz)); sz = [int(x) for x in sz]
        r = Oracle3Result(K2=K2)
        r.inv = np.zeros(len(off) - 1, np.int32); L.oracle3_inv(h, _p(r.inv))
        r.frag = np.zeros(100, np.float64); L.oracle3_frag(h, _p(r.frag))
        r.place_off = np.zeros(sz[0] + 1, np.uint64); r.place_edges = np.zeros(sz[1], np.int32)
        r.left_trunc = np.zeros(sz[0], np.int32); r.right_trunc = np.zeros(sz[0], np.int32)
        L.oracle3_places(h, _p(r.place_off), _p(r.place_edges), _p(r.left_trunc), _p(r.right_trunc))
        r.all_codes = np.zeros(sz[2], np.uint8); r.all_off = np.zeros(sz[0] + 1, np.uint64)
        L.oracle3_all(h, _p(r.all_codes), _p(r.all_off))
        if stop_after == 1:
            return r
        r.n_instances, r.n_distinct, r.n_edges, NO, r.n_vertices = sz[3], sz[4], sz[5], sz[6], sz[7]
        r.obj_codes = np.zeros(sz[8], np.uint8); r.obj_off = np.zeros(NO + 1, np.uint64)
        r.left = np.zeros(NO, np.int32); r.right = np.zeros(NO, np.int32); r.inv2 = np.zeros(NO, np.int32)
        L.oracle3_objs(h, _p(r.obj_codes), _p(r.obj_off), _p(r.left), _p(r.right), _p(r.inv2))
        for which, (o, v) in enumerate((("from_off", "from_v"), ("from_off", "from_e"), ("to_off", "to_v"), ("to_off", "to_e"))):
            offa = np.zeros(r.n_vertices + 1, np.uint64); vals = np.zeros(NO, np.int32)
            L.oracle3_adj(h, which, _p(offa), _p(vals))
            setattr(r, o, offa); setattr(r, v, vals)
        r.path_offset = np.zeros(n, np.int32); r.path_off = np.zeros(n + 1, np.uint64); r.path_edges = np.zeros(sz[9], np.int32)
        L.oracle3_paths(h, _p(r.path_offset), _p(r.path_off), _p(r.path_edges))
        return r
    finally:
        L.oracle3_free(h)


def to_hbv(r: Oracle3Result):
    from w2rap_contigger_amd import formats as F
    packed, boff, lens = F.pack_bases(r.obj_codes, r.obj_off)
    return F.HBV(r.K2, r.from_off, r.from_v, r.from_e, r.to_off, r.to_e, packed, boff, lens)


def frags_text(count) -> str:
    """the text FragDist writes (GapToyTools3.cc:636-646): iostream default formatting (%g) of count[j] / total"""
    total = float(np.sum(count))
    out = ["# fragment library size distribution", "# bins have diameter 10", "# line format:", "# bin_center mass"]
    for j, c in enumerate(count):
        if total == 0:
            v = "-nan"
        else:
            v = "%g" % (float(c) / total)
        out.append(f"{j * 10 + 5} {v}")
    return "\n".join(out) + "\n"


def run_reference3(workdir: str, prefix="t", K2=200, threads=1, extend_paths=False):
    """the real reference Step 3 (oracle/_ref/ref_step3) on workdir/<prefix>.small_K.{hbv,paths} -> writes <prefix>.large_K.{hbv,paths}"""
    if not os.path.exists(REF3_BIN):
        raise FileNotFoundError(REF3_BIN)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    subprocess.run([REF3_BIN, workdir, prefix, str(K2), str(threads), "1" if extend_paths else "0"], check=True, capture_output=True, text=True, env=env)
