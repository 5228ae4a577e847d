"""ctypes binding of the Step-3 entry points of libw2rap_step2.so (include/w2rap_step3.h) + the host-side mirror of the
reference's Step-3 interface.

`repath_in_memory` mirrors ``hbv.Involution(inv); FragDist(...); RepathInMemory(hbv, edges, inv, paths, hbv.K(), large_K,
hbvr, pathsr, True, True, extend_paths)`` (src/modules/w2rap-contigger.cc:359-371, src/paths/long/large/Repath.cc:23-251);
`run_step3_files` mirrors the reference's ``--from_step 3 --to_step 3`` run on an output directory (w2rap-contigger.cc:352-378):
reads <prefix>.small_K.{hbv,paths}, writes <prefix>.large_K.{hbv,paths} and <prefix>.first.frags.dist.

The HIP library is the only implementation (no CPU fallback)."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import formats as F
from .step2 import EdgeHint, Step2Error, _np_from, _ptr, lib as _lib2, make_hint


class Step3In(C.Structure):
    _fields_ = [("K", C.c_int32), ("n_edge_objs", C.c_uint64), ("edge_packed", C.c_void_p), ("edge_byte_off", C.c_void_p), ("edge_len", C.c_void_p),
                ("n_paths", C.c_uint64), ("path_offset", C.c_void_p), ("path_off", C.c_void_p), ("path_edges", C.c_void_p),
                ("n_vertices", C.c_uint64), ("vleft", C.c_void_p), ("vright", C.c_void_p)]


class Step3Params(C.Structure):
    _fields_ = [("K2", C.c_uint32), ("device", C.c_int32), ("extend_paths", C.c_int32), ("edge_order_hint", C.POINTER(EdgeHint)), ("flags", C.c_uint32),
                ("n_extra_paths", C.c_uint64), ("extra_path_off", C.c_void_p), ("extra_path_edges", C.c_void_p)]


NO_FETCH = 1
PLACES_ONLY = 2
UNIQUE_KMERS = 4        # the graph is a unipath graph whose K-mers occur once (Step 2's output): include/w2rap_step3.h


class Step3Out(C.Structure):
    _fields_ = [("K2", C.c_int32), ("inv", C.c_void_p), ("frag_count", C.c_uint64 * 100),
                ("n_vertices", C.c_uint64), ("n_edge_objs", C.c_uint64),
                ("edge_packed", C.c_void_p), ("edge_byte_off", C.c_void_p), ("edge_len", C.c_void_p), ("vleft", C.c_void_p), ("vright", C.c_void_p),
                ("from_off", C.c_void_p), ("from_v", C.c_void_p), ("from_e", C.c_void_p), ("to_off", C.c_void_p), ("to_v", C.c_void_p), ("to_e", C.c_void_p),
                ("inv2", C.c_void_p),
                ("n_paths", C.c_uint64), ("path_offset", C.c_void_p), ("path_off", C.c_void_p), ("path_edges", C.c_void_p),
                ("n_reads_pathed", C.c_uint64), ("n_reads_multipathed", C.c_uint64), ("n_places", C.c_uint64), ("n_unique_places", C.c_uint64),
                ("n_place_bases", C.c_uint64), ("n_kmer_instances", C.c_uint64), ("n_kmers_distinct", C.c_uint64), ("n_unipaths", C.c_uint64),
                ("ms_places", C.c_float), ("ms_dict", C.c_float), ("ms_graph", C.c_float), ("ms_paths", C.c_float),
                ("n_place_paths", C.c_uint64), ("place_path_off", C.c_void_p), ("place_path_edges", C.c_void_p), ("_owner", C.c_void_p)]


_ready = False


def lib():
    global _ready
    L = _lib2()
    if not _ready:
        L.w2rap_step3_run.argtypes = [C.POINTER(Step3In), C.POINTER(Step3Params), C.POINTER(Step3Out), C.c_char_p, C.c_size_t]
        L.w2rap_step3_free.argtypes = [C.POINTER(Step3Out)]
        L.w2rap_step3_free.restype = None
        L.w2rap_step3_run_after_step2.argtypes = [C.c_void_p, C.POINTER(Step3Params), C.POINTER(Step3Out), C.c_char_p, C.c_size_t]
        L.w2rap_step3_profile.argtypes = [C.c_char_p, C.c_size_t]
        L.w2rap_step3_profile.restype = C.c_size_t
        _ready = True
    return L


@dataclass
class Step3Result:
    hbv: F.HBV                    # the large-K graph (hbvr)
    vleft: np.ndarray
    vright: np.ndarray
    to_v: np.ndarray
    inv: np.ndarray               # Involution of the INPUT graph
    inv2: np.ndarray              # Involution of the large-K graph
    frag_count: np.ndarray        # FragDist counts, 100 bins of 10 bases
    path_offset: np.ndarray       # pathsr
    path_off: np.ndarray
    path_edges: np.ndarray
    n_reads_pathed: int
    n_reads_multipathed: int
    n_places: int
    n_unique_places: int
    n_place_bases: int
    n_kmer_instances: int
    n_kmers_distinct: int
    n_unipaths: int
    ms_places: float
    ms_dict: float
    ms_graph: float
    ms_paths: float
    place_paths: tuple = None     # PLACES_ONLY: (off u64[U+1], edges i32[]) -- a read path per unique place of these reads


def _params(K2, device, extend_paths, hint_p, flags, extra_paths, keep):
    if extra_paths is None:
        return Step3Params(K2, device, 1 if extend_paths else 0, hint_p, flags, 0, None, None)
    xo = np.ascontiguousarray(extra_paths[0], np.uint64); xe = np.ascontiguousarray(extra_paths[1], np.int32)
    keep += [xo, xe]
    return Step3Params(K2, device, 1 if extend_paths else 0, hint_p, flags, len(xo) - 1, _ptr(xo), _ptr(xe) if len(xe) else None)


def repath_in_memory(hbv: F.HBV, paths, K2=200, device=0, edge_order_hint=None, extend_paths=False, extra_paths=None, places_only=False, unique_kmers=False) -> Step3Result:
    """Involution + FragDist + RepathInMemory through the one-shot C entry point (w2rap_step3_run).
    unique_kmers: the caller vouches that `hbv` is the unipath graph of Step 2 (every K-mer once): W2RAP_STEP3_UNIQUE_KMERS.
    paths = (offset i32[n], path_off u64[n+1], edges i32[]); edge_order_hint = (packed, byte_off, len) of the large-K canonical
    edges in the order to replay, or None for the lexicographic order."""
    L = lib()
    keep = [np.ascontiguousarray(hbv.edge_packed, np.uint8), np.ascontiguousarray(hbv.edge_byte_off, np.uint64), np.ascontiguousarray(hbv.edge_len, np.uint32),
            np.ascontiguousarray(paths[0], np.int32), np.ascontiguousarray(paths[1], np.uint64), np.ascontiguousarray(paths[2], np.int32)]
    i = Step3In(hbv.K, len(keep[2]), _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), len(keep[3]), _ptr(keep[3]), _ptr(keep[4]), _ptr(keep[5]), 0, None, None)
    if extend_paths:                          # --extend_paths walks the small-K graph: the vertices every edge object leaves and enters
        tl, tr = hbv.to_left_right()
        keep += [np.ascontiguousarray(tl, np.int32), np.ascontiguousarray(tr, np.int32)]
        i.n_vertices, i.vleft, i.vright = hbv.n_vertices, _ptr(keep[-2]), _ptr(keep[-1])
    hint_p = None
    if edge_order_hint is not None:
        eh, k2 = make_hint(*edge_order_hint)
        keep.append(k2)
        hint_p = C.pointer(eh)
    p = _params(K2, device, extend_paths, hint_p, (PLACES_ONLY if places_only else 0) | (UNIQUE_KMERS if unique_kmers else 0), extra_paths, keep)
    o = Step3Out()
    err = C.create_string_buffer(1024)
    rc = L.w2rap_step3_run(C.byref(i), C.byref(p), C.byref(o), err, 1024)
    if rc:
        raise Step2Error(rc, err.value.decode(errors="replace"))
    return _result3(L, o, len(keep[2]))


def repath_after_step2(ctx, K2=200, edge_order_hint=None, fetch=True, extra_paths=None, places_only=False, extend_paths=False) -> Step3Result:
    """Step 3 straight behind Step 2 on the same GPU context (step2.Step2Context after path_reads): graph and paths stay in HBM
    (w2rap_step3_run_after_step2) -- the reference's default flow of steps 2 and 3 in one process."""
    L = lib()
    keep = []
    hint_p = None
    if edge_order_hint is not None:
        eh, k2 = make_hint(*edge_order_hint)
        keep.append(k2)
        hint_p = C.pointer(eh)
    p = _params(K2, 0, extend_paths, hint_p, (0 if fetch else NO_FETCH) | (PLACES_ONLY if places_only else 0), extra_paths, keep)
    o = Step3Out()
    err = C.create_string_buffer(1024)
    rc = L.w2rap_step3_run_after_step2(ctx.h, C.byref(p), C.byref(o), err, 1024)
    if rc:
        raise Step2Error(rc, err.value.decode(errors="replace"))
    return _result3(L, o, None)


def _result3(L, o, n_in_objs) -> Step3Result:
    try:
        NO, NV, NP = o.n_edge_objs, o.n_vertices, o.n_paths
        boff = _np_from(o.edge_byte_off, np.uint64, NO + 1)
        h2 = F.HBV(o.K2, _np_from(o.from_off, np.uint64, NV + 1), _np_from(o.from_v, np.int32, NO), _np_from(o.from_e, np.int32, NO),
                   _np_from(o.to_off, np.uint64, NV + 1), _np_from(o.to_e, np.int32, NO),
                   _np_from(o.edge_packed, np.uint8, int(boff[-1]) if len(boff) else 0), boff, _np_from(o.edge_len, np.uint32, NO))
        po = _np_from(o.path_off, np.uint64, NP + 1) if NP else np.zeros(1, np.uint64)
        if len(po) == 0:                                     # NO_FETCH: only the counters came back
            po = np.zeros(1, np.uint64)
        pp = None
        if o.n_place_paths or o.place_path_off:
            ppo = _np_from(o.place_path_off, np.uint64, o.n_place_paths + 1)
            pp = (ppo, _np_from(o.place_path_edges, np.int32, int(ppo[-1])))
        return Step3Result(h2, _np_from(o.vleft, np.int32, NO), _np_from(o.vright, np.int32, NO), _np_from(o.to_v, np.int32, NO),
                           _np_from(o.inv, np.int32, n_in_objs) if n_in_objs is not None else None, _np_from(o.inv2, np.int32, NO),
                           np.array(list(o.frag_count), dtype=np.uint64),
                           _np_from(o.path_offset, np.int32, NP), po, _np_from(o.path_edges, np.int32, int(po[-1])),
                           o.n_reads_pathed, o.n_reads_multipathed, o.n_places, o.n_unique_places, o.n_place_bases, o.n_kmer_instances,
                           o.n_kmers_distinct, o.n_unipaths, o.ms_places, o.ms_dict, o.ms_graph, o.ms_paths, pp)
    finally:
        L.w2rap_step3_free(C.byref(o))


def profile():
    """-> {kernel name: (total ms, launches)} of the last repath_in_memory in this process"""
    L = lib()
    n = L.w2rap_step3_profile(None, 0)
    buf = C.create_string_buffer(int(n) + 16)
    L.w2rap_step3_profile(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, ms, k = line.rsplit(" ", 2)
        out[name] = (float(ms), int(k))
    return out


def frags_text(count) -> str:
    """what FragDist writes to <prefix>.first.frags.dist (GapToyTools3.cc:636-646): count[j] / total with iostream's default
    formatting (6 significant digits, %g); "-nan" when no pair qualified (0/0 as the reference prints it)"""
    total = float(np.sum(np.asarray(count, dtype=np.float64)))
    out = ["# fragment library size distribution", "# bins have diameter 10", "# line format:", "# bin_center mass"]
    for j, c in enumerate(count):
        out.append(f"{j * 10 + 5} " + ("-nan" if total == 0 else "%g" % (float(c) / total)))
    return "\n".join(out) + "\n"


def run_step3_files(out_dir, prefix, K2=200, device=0, edge_order_hint=None, extend_paths=False) -> Step3Result:
    """The reference's Step 3 on an output directory (w2rap-contigger.cc:352-378)."""
    hbv = F.read_hbv(os.path.join(out_dir, f"{prefix}.small_K.hbv"))
    paths = F.read_paths(os.path.join(out_dir, f"{prefix}.small_K.paths"))
    res = repath_in_memory(hbv, paths, K2, device, edge_order_hint, extend_paths=extend_paths)
    F.write_hbv(os.path.join(out_dir, f"{prefix}.large_K.hbv"), res.hbv)
    F.write_paths(os.path.join(out_dir, f"{prefix}.large_K.paths"), res.path_offset, res.path_off, res.path_edges)
    with open(os.path.join(out_dir, f"{prefix}.first.frags.dist"), "w") as f:
        f.write(frags_text(res.frag_count))
    return res
