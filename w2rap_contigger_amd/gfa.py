"""ctypes binding of the GFA-dump entry points of libw2rap_step2.so (include/w2rap_gfa.h) + the host-side mirror of the reference's
hbv2gfa tool (src/modules/hbv2gfa.cc, src/GFADump.cc with find_lines = false).

`gfa_dump` mirrors ``hbv.Involution(inv); <graph stats>; GFADump(out_prefix, hbv, inv, paths, 50, 10, false)``;
`run_hbv2gfa` mirrors ``hbv2gfa -i <in_prefix> -o <out_prefix> [-g Kbp] [--stats_only 1]``: reads <in_prefix>.hbv, writes
<out_prefix>_raw.gfa and returns the text the reference prints between "=== Graph stats === " and "Dumping gfa".

The HIP library is the only implementation (no CPU fallback)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import formats as F
from .step2 import Step2Error, _np_from, _ptr, lib as _lib2


class GfaIn(C.Structure):
    _fields_ = [("K", C.c_int32), ("n_vertices", C.c_uint64), ("n_edge_objs", C.c_uint64), ("edge_packed", C.c_void_p), ("edge_byte_off", C.c_void_p),
                ("edge_len", C.c_void_p), ("from_off", C.c_void_p), ("from_e", C.c_void_p), ("to_off", C.c_void_p), ("to_e", C.c_void_p)]


class GfaParams(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32), ("genome_size", C.c_uint64)]


STATS_ONLY = 1
NO_FETCH = 2


class GfaOut(C.Structure):
    _fields_ = [("gfa", C.c_void_p), ("gfa_len", C.c_uint64), ("n_segments", C.c_uint64), ("n_links", C.c_uint64), ("segment_bytes", C.c_uint64), ("inv", C.c_void_p),
                ("canonical_size", C.c_uint64), ("n_canonical", C.c_uint64), ("nxx", C.c_uint64 * 9), ("ngxx", C.c_int64 * 9),
                ("ms_involution", C.c_float), ("ms_dump", C.c_float)]


_ready = False


def lib():
    global _ready
    L = _lib2()
    if not _ready:
        L.w2rap_gfa_dump.argtypes = [C.POINTER(GfaIn), C.POINTER(GfaParams), C.POINTER(GfaOut), C.c_char_p, C.c_size_t]
        L.w2rap_gfa_free.argtypes = [C.POINTER(GfaOut)]
        L.w2rap_gfa_free.restype = None
        L.w2rap_gfa_profile.argtypes = [C.c_char_p, C.c_size_t]
        L.w2rap_gfa_profile.restype = C.c_size_t
        _ready = True
    return L


@dataclass
class GfaResult:
    gfa: bytes                    # the text of <out_prefix>_raw.gfa
    inv: np.ndarray
    n_segments: int
    n_links: int
    canonical_size: int
    n_canonical: int
    nxx: list
    ngxx: list                    # -1 = "n/a"
    genome_size: int
    ms_involution: float
    ms_dump: float
    gfa_len: int = 0
    segment_bytes: int = 0

    def stats_text(self) -> str:
        """what hbv2gfa prints between "=== Graph stats === " and "Dumping gfa" (hbv2gfa.cc:71-92)"""
        out = [f"Canonical graph sequences size: {self.canonical_size}"] + [f"N{10 * (j + 1)}: {v}" for j, v in enumerate(self.nxx)]
        if self.genome_size:
            out += ["", f"User provided size: {self.genome_size}"] + [f"NG{10 * (j + 1)}: " + ("n/a" if v < 0 else str(v)) for j, v in enumerate(self.ngxx)]
        return "\n".join(out) + "\n"


def gfa_dump(hbv: F.HBV, genome_size=0, device=0, flags=0) -> GfaResult:
    L = lib()
    keep = [np.ascontiguousarray(hbv.edge_packed, np.uint8), np.ascontiguousarray(hbv.edge_byte_off, np.uint64), np.ascontiguousarray(hbv.edge_len, np.uint32),
            np.ascontiguousarray(hbv.from_off, np.uint64), np.ascontiguousarray(hbv.from_e, np.int32), np.ascontiguousarray(hbv.to_off, np.uint64),
            np.ascontiguousarray(hbv.to_e, np.int32)]
    i = GfaIn(hbv.K, hbv.n_vertices, hbv.n_edges, *[_ptr(a) for a in keep])
    p = GfaParams(device, flags, genome_size)
    o = GfaOut()
    err = C.create_string_buffer(1024)
    rc = L.w2rap_gfa_dump(C.byref(i), C.byref(p), C.byref(o), err, 1024)
    if rc:
        raise Step2Error(rc, err.value.decode(errors="replace"))
    try:
        text = bytes(_np_from(o.gfa, np.uint8, o.gfa_len)) if o.gfa else b""
        return GfaResult(text, _np_from(o.inv, np.int32, hbv.n_edges), o.n_segments, o.n_links, o.canonical_size, o.n_canonical, list(o.nxx), list(o.ngxx),
                         genome_size, o.ms_involution, o.ms_dump, o.gfa_len, o.segment_bytes)
    finally:
        L.w2rap_gfa_free(C.byref(o))


def run_hbv2gfa(in_prefix: str, out_prefix: str, genome_kb=0, stats_only=False, device=0) -> GfaResult:
    hbv = F.read_hbv(in_prefix + ".hbv")
    res = gfa_dump(hbv, 1000 * genome_kb, device, STATS_ONLY if stats_only else 0)
    if not stats_only:
        with open(out_prefix + "_raw.gfa", "wb") as f:
            f.write(res.gfa)
    return res


def profile():
    """-> {kernel name: (total ms, launches)} of the last gfa_dump in this process"""
    L = lib()
    n = L.w2rap_gfa_profile(None, 0)
    buf = C.create_string_buffer(int(n) + 16)
    L.w2rap_gfa_profile(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, ms, k = line.rsplit(" ", 2)
        out[name] = (float(ms), int(k))
    return out
