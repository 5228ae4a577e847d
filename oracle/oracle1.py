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
    """-> dict(packed, byte_off, read_len, quals, pq, pq_off) as frag_reads_orig.fastb/.qualp hold them"""
    L = lib()
    h = L.oracle1_run(fastq1, len(fastq1), fastq2, len(fastq2))
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
