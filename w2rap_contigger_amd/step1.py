"""ctypes binding of the Step-1 entry points of libw2rap_step2.so (include/w2rap_step1.h) + the host-side mirror of the reference's
Step-1 interface for a pair of fastq files.

`extract_reads` mirrors ``ExtractReads(read_files, out_dir, ..., &bases, &quals)`` (src/modules/w2rap-contigger.cc:308,
src/paths/long/large/ExtractReads.cc:350-474); `run_step1_files` mirrors the reference's ``--from_step 1 --to_step 1`` run:
reads `r1.fastq,r2.fastq` (plain or .gz), writes <out_dir>/frag_reads_orig.fastb and .qualp (w2rap-contigger.cc:315-316).

The HIP library is the only implementation (no CPU fallback)."""
from __future__ import annotations

import ctypes as C
import gzip
import os
from dataclasses import dataclass

import numpy as np

from . import formats as F
from .step2 import Step2Error, _np_from, lib as _lib2


class Step1In(C.Structure):
    _fields_ = [("fastq1", C.c_void_p), ("len1", C.c_uint64), ("fastq2", C.c_void_p), ("len2", C.c_uint64), ("mem", C.c_int32)]


class Step1Params(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32)]


NO_PQ = 1
NO_FETCH = 2
INTERLEAVED = 4


class Step1Out(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("bases_packed", C.c_void_p), ("base_byte_off", C.c_void_p), ("read_len", C.c_void_p),
                ("quals", C.c_void_p), ("qual_off", C.c_void_p), ("pq", C.c_void_p), ("pq_off", C.c_void_p), ("n_bases", C.c_uint64), ("n_packed_bytes", C.c_uint64), ("n_pq_bytes", C.c_uint64),
                ("ms_upload", C.c_float), ("ms_index", C.c_float), ("ms_encode", C.c_float)]


_ready = False


def lib():
    global _ready
    L = _lib2()
    if not _ready:
        L.w2rap_step1_run.argtypes = [C.POINTER(Step1In), C.POINTER(Step1Params), C.POINTER(Step1Out), C.c_char_p, C.c_size_t]
        L.w2rap_step1_free.argtypes = [C.POINTER(Step1Out)]
        L.w2rap_step1_free.restype = None
        L.w2rap_step1_run_into_step2.argtypes = [C.c_void_p, C.POINTER(Step1In), C.POINTER(Step1Params), C.POINTER(Step1Out), C.c_char_p, C.c_size_t]
        L.w2rap_step1_profile.argtypes = [C.c_char_p, C.c_size_t]
        L.w2rap_step1_profile.restype = C.c_size_t
        _ready = True
    return L


@dataclass
class Step1Result:
    packed: np.ndarray            # frag_reads_orig.fastb payload
    byte_off: np.ndarray
    read_len: np.ndarray
    quals: np.ndarray             # one byte per base
    qual_off: np.ndarray
    pq: np.ndarray | None         # frag_reads_orig.qualp payload (PQVec byte strings)
    pq_off: np.ndarray | None
    n_reads: int
    n_bases: int
    ms_index: float
    ms_encode: float
    ms_upload: float = 0.0
    n_packed_bytes: int = 0
    n_pq_bytes: int = 0


def _text(t):
    """bytes -> (host pointer, length, MEM_HOST, keepalive); a (device pointer, length) pair -> MEM_DEVICE"""
    if isinstance(t, (bytes, bytearray)):
        b = bytes(t)
        return C.cast(C.c_char_p(b), C.c_void_p), len(b), 0, b
    ptr, n = t
    return C.c_void_p(int(ptr)), int(n), 1, None


def extract_reads(fastq1, fastq2, device=0, flags=0, ctx=None) -> Step1Result:
    """the text of the two fastq files -> bases + qualities (w2rap_step1_run).  fastq1/fastq2: bytes, or (device pointer, length) pairs
    for text that already lies in HBM.  With `ctx` (a step2.Step2Context) the reads are also left in HBM as that context's reads
    (w2rap_step1_run_into_step2): count_kmers / build_graph / path_reads follow without another copy."""
    L = lib()
    p1, n1, m1, k1 = _text(fastq1)
    p2, n2, m2, k2 = _text(fastq2)
    if m1 != m2:
        raise Step2Error(1, "both texts must be in host memory or both in device memory")
    i = Step1In(p1, n1, p2, n2, m1)
    p = Step1Params(device, flags)
    o = Step1Out()
    err = C.create_string_buffer(1024)
    if ctx is None:
        rc = L.w2rap_step1_run(C.byref(i), C.byref(p), C.byref(o), err, 1024)
    else:
        rc = L.w2rap_step1_run_into_step2(ctx.h, C.byref(i), C.byref(p), C.byref(o), err, 1024)
    del k1, k2
    if rc:
        raise Step2Error(rc, err.value.decode(errors="replace"))
    try:
        n = o.n_reads
        if flags & NO_FETCH:
            z8, z64 = np.zeros(0, np.uint8), np.zeros(1, np.uint64)
            return Step1Result(z8, z64, np.zeros(0, np.uint32), z8, z64, None, None, n, o.n_bases, o.ms_index, o.ms_encode, o.ms_upload, o.n_packed_bytes, o.n_pq_bytes)
        boff = _np_from(o.base_byte_off, np.uint64, n + 1)
        qoff = _np_from(o.qual_off, np.uint64, n + 1)
        pq = pqo = None
        if not flags & NO_PQ:
            pqo = _np_from(o.pq_off, np.uint64, n + 1)
            pq = _np_from(o.pq, np.uint8, int(pqo[-1]))
        return Step1Result(_np_from(o.bases_packed, np.uint8, int(boff[-1])), boff, _np_from(o.read_len, np.uint32, n),
                           _np_from(o.quals, np.uint8, int(qoff[-1])), qoff, pq, pqo, n, o.n_bases, o.ms_index, o.ms_encode, o.ms_upload, o.n_packed_bytes, o.n_pq_bytes)
    finally:
        L.w2rap_step1_free(C.byref(o))


def profile():
    """-> {kernel name: (total ms, launches)} of the last Step-1 run in this process"""
    L = lib()
    n = L.w2rap_step1_profile(None, 0)
    buf = C.create_string_buffer(int(n) + 16)
    L.w2rap_step1_profile(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, ms, k = line.rsplit(" ", 2)
        out[name] = (float(ms), int(k))
    return out


def _slurp(path) -> bytes:
    """the reference opens .gz through its gzstream wrapper (ExtractReads.cc:372-389); decompression is host work here as there"""
    if path.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


def first_read_name(text: bytes, what="fastq file") -> bytes:
    """ExtractReads.cc:230-243: the first line must start with '@', be longer than one character and not go on with ' ' or '/'; the read
    name is what lies between the '@' and the first ' ' or '/'"""
    line = text.split(b"\n", 1)[0]
    if not line.startswith(b"@") or len(line) == 1 or line[1:2] in (b" ", b"/"):
        raise Step2Error(1, f"Something is wrong with the first line of your {what}")
    p = 0
    while p < len(line) and line[p:p + 1] not in (b" ", b"/"):
        p += 1
    return line[1:p]


def plan_files(texts, names=None):
    """How the reference groups the fastq files of `-r` (ExtractReads.cc:218-258, 370-374, 483): sorted by first read name (order kept among
    equals); two files with one name are a pair (mates interleaved R1, R2), a name shared by more than two files is fatal, every other
    file is read on its own (alternating mates).  -> [(i,) or (i, j)] in output order"""
    rn = [first_read_name(t, f"fastq file {names[i]}" if names else "fastq file") for i, t in enumerate(texts)]
    order = sorted(range(len(texts)), key=lambda i: rn[i])
    plan, j = [], 0
    while j < len(order):
        k = j
        while k < len(order) and rn[order[k]] == rn[order[j]]:
            k += 1
        if k - j > 2:
            raise Step2Error(1, "There are more than two fastq files that start with the read name " + rn[order[j]].decode(errors="replace")
                             + ": it's not clear how to pair the files")
        plan.append(tuple(order[j:k]))
        j = k
    return plan


def extract_read_files(texts, device=0, names=None) -> Step1Result:
    """`-r a.fastq,b.fastq,...` (inflated texts): every group of plan_files through the GPU, the results concatenated in the reference's order"""
    parts = []
    for g in plan_files(texts, names):
        parts.append(extract_reads(texts[g[0]], texts[g[1]], device) if len(g) == 2 else extract_reads(texts[g[0]], b"", device, flags=INTERLEAVED))
    if len(parts) == 1:
        return parts[0]

    def cat_off(off_name, data_name):
        o, base = [np.zeros(1, np.uint64)], 0
        for p in parts:
            o.append(getattr(p, off_name)[1:] + np.uint64(base)); base += len(getattr(p, data_name))
        return np.concatenate(o)
    cat = lambda name: np.concatenate([getattr(p, name) for p in parts])
    return Step1Result(cat("packed"), cat_off("byte_off", "packed"), cat("read_len"), cat("quals"), cat_off("qual_off", "quals"), cat("pq"), cat_off("pq_off", "pq"),
                       sum(p.n_reads for p in parts), sum(p.n_bases for p in parts), sum(p.ms_index for p in parts), sum(p.ms_encode for p in parts),
                       sum(p.ms_upload for p in parts), sum(p.n_packed_bytes for p in parts), sum(p.n_pq_bytes for p in parts))


def run_step1_files(read_files: str, out_dir: str, device=0) -> Step1Result:
    """`-r a.fastq[,b.fastq,...] -o out_dir --from_step 1 --to_step 1` for fastq inputs (plain or .gz)"""
    names = [x for x in read_files.split(",") if x]
    if not names:
        raise Step2Error(1, "no read files")
    res = extract_read_files([_slurp(x) for x in names], device, names)
    F.write_fastb(os.path.join(out_dir, "frag_reads_orig.fastb"), res.packed, res.byte_off, res.read_len)
    F.write_qualp_blobs(os.path.join(out_dir, "frag_reads_orig.qualp"), res.pq, res.pq_off)
    return res
