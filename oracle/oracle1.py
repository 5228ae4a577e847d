"""ctypes wrapper around oracle/liboracle_step1.so (Step 1: paired fastq -> packed bases + PQVec qualities) -- TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle_step1.so")
REF1_BIN = os.path.join(HERE, "_ref", "ref_step1")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
        L = C.CDLL(LIB)
        L.oracle1_run.restype = C.c_void_p
        L.oracle1_run.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
        L.oracle1_run_single.restype = C.c_void_p
        L.oracle1_run_single.argtypes = [C.c_char_p, C.c_uint64]
        L.oracle1_error.restype = C.c_char_p
        L.oracle1_error.argtypes = [C.c_void_p]
        L.oracle1_free.argtypes = [C.c_void_p]
        L.oracle1_sizes.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle1_get.argtypes = [C.c_void_p] * 7
        L.oracle1_pq_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.oracle1_pq_encode.restype = C.c_int64
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def run(fastq1: bytes, fastq2: bytes):
    """a pair of fastq texts -> dict(packed, byte_off, read_len, quals, pq, pq_off) as frag_reads_orig.fastb/.qualp hold them"""
    L = lib()
    return _collect(L, L.oracle1_run(fastq1, len(fastq1), fastq2, len(fastq2)))


def run_single(fastq: bytes):
    """one fastq text with alternating mates (ExtractReads.cc:481-568) -> the same dict"""
    L = lib()
    return _collect(L, L.oracle1_run_single(fastq, len(fastq)))


def first_read_name(text: bytes, what="fastq file") -> bytes:
    """ExtractReads.cc:230-243: the first line must start with '@', be longer than one character and not go on with ' ' or '/';
    the read name is what lies between the '@' and the first ' ' or '/'"""
    line = text.split(b"\n", 1)[0]
    if not line.startswith(b"@") or len(line) == 1 or line[1:2] in (b" ", b"/"):
        raise RuntimeError(f"Something is wrong with the first line of your {what}")
    p = 0
    while p < len(line) and line[p:p + 1] not in (b" ", b"/"):
        p += 1
    return line[1:p]


def plan_files(texts):
    """ExtractReads.cc:218-258,370-374,483: the files sorted by first read name (order kept among equals); two neighbours with one name
    are a pair, a name shared by more than two files is fatal, every other file is read on its own.  -> [(i,) or (i, j)] in output order"""
    names = [first_read_name(t) for t in texts]
    order = sorted(range(len(texts)), key=lambda i: names[i])
    plan, j = [], 0
    while j < len(order):
        k = j
        while k < len(order) and names[order[k]] == names[order[j]]:
            k += 1
        if k - j > 2:
            raise RuntimeError("There are more than two fastq files that start with the read name " + names[order[j]].decode(errors="replace"))
        plan.append(tuple(order[j:k]))
        j = k
    return plan


def run_files(texts):
    """`-r a.fastq,b.fastq,...` (already inflated texts) -> the concatenated result, file groups in the reference's order"""
    parts = [run(texts[g[0]], texts[g[1]]) if len(g) == 2 else run_single(texts[g[0]]) for g in plan_files(texts)]
    out = {k: np.concatenate([p[k] for p in parts]) for k in ("packed", "read_len", "quals", "pq")}
    for off, data in (("byte_off", "packed"), ("pq_off", "pq")):
        o, base = [np.zeros(1, np.uint64)], 0
        for p in parts:
            o.append(p[off][1:] + np.uint64(base)); base += len(p[data])
        out[off] = np.concatenate(o)
    return out


def _collect(L, h):
    try:
        e = L.oracle1_error(h)
        if e:
            raise RuntimeError(e.decode(errors="replace"))
        sz = np.zeros(4, np.uint64); L.oracle1_sizes(h, _p(sz)); n, nb, nq, npq = [int(x) for x in sz]
        r = dict(packed=np.zeros(nb, np.uint8), byte_off=np.zeros(n + 1, np.uint64), read_len=np.zeros(n, np.uint32), quals=np.zeros(nq, np.uint8),
                 pq=np.zeros(npq, np.uint8), pq_off=np.zeros(n + 1, np.uint64))
        L.oracle1_get(h, _p(r["packed"]), _p(r["byte_off"]), _p(r["read_len"]), _p(r["quals"]), _p(r["pq"]), _p(r["pq_off"]))
        return r
    finally:
        L.oracle1_free(h)


def pq_encode(q: np.ndarray) -> bytes:
    """the reference's PQVecEncoder on one quality vector"""
    q = np.ascontiguousarray(q, np.uint8)
    out = np.zeros(3 * len(q) + 8, np.uint8)
    n = lib().oracle1_pq_encode(_p(q), len(q), _p(out))
    if n < 0:
        raise ValueError("quality > 63")
    return out[:n].tobytes()


def run_reference1(workdir: str, reads: str, threads=1) -> float:
    """the real reference Step 1 (oracle/_ref/ref_step1): ExtractReads(reads) + WriteAll -> workdir/frag_reads_orig.{fastb,qualp}; -> seconds in ExtractReads"""
    out = subprocess.run([REF1_BIN, workdir, reads, str(threads)], check=True, capture_output=True, text=True, cwd=workdir).stdout
    for line in out.splitlines():
        if line.startswith("REF_STEP1"):
            return float(line.split()[4])
    raise RuntimeError("ref_step1 printed no REF_STEP1 line:\n" + out[-2000:])
