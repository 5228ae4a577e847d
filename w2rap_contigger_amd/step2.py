"""ctypes binding of libw2rap_step2.so + the host-side mirror of the reference's
Step-2 interface.

`build_read_qgraph` mirrors ``buildReadQGraph`` followed by ``FixPaths``
(src/paths/long/BuildReadQGraph.h:24-29, src/modules/w2rap-contigger.cc:338-340);
`run_step2_files` mirrors the reference's ``--from_step 2 --to_step 2`` run on an
output directory (w2rap-contigger.cc:326-346): same input and output file names.

The HIP library is the only implementation: importing this module without
libw2rap_step2.so, or calling it without a gfx950 GPU, raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import formats as F

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("W2RAP_LIB") or os.path.join(HERE, "libw2rap_step2.so")    # (W2RAP_LIB: an alternative build, for A/B measurements)

MEM_HOST, MEM_DEVICE = 0, 1


class Step2Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libw2rap_step2 error {code}: {msg}")
        self.code = code


class Reads(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("bases_packed", C.c_void_p), ("base_byte_off", C.c_void_p),
                ("read_len", C.c_void_p), ("quals", C.c_void_p), ("qual_off", C.c_void_p),
                ("pq", C.c_void_p), ("pq_off", C.c_void_p), ("mem", C.c_int32)]


class EdgeHint(C.Structure):
    _fields_ = [("n_edges", C.c_uint64), ("packed", C.c_void_p), ("byte_off", C.c_void_p), ("len", C.c_void_p)]


F_REPLICATED_GRAPH, F_GRAPH_ONLY = 1, 2      # w2rap_step2_params.flags
X_DONE, X_ALLTOALL, X_ALLGATHER, X_ALLGATHER_HOST, X_ALLREDUCE_U8, X_ALLREDUCE_U32 = range(6)   # w2rap_xchg.op


class Xchg(C.Structure):
    """w2rap_xchg: the exchange the sharded graph phase asks its host layer for (include/w2rap_step2.h)"""
    _fields_ = [("op", C.c_int32), ("elem_bytes", C.c_uint32), ("send", C.c_void_p), ("send_count", C.c_uint64 * 64)]


class Params(C.Structure):
    _fields_ = [("K", C.c_uint32), ("min_qual", C.c_uint32), ("min_freq", C.c_uint32), ("device", C.c_int32),
                ("edge_order_hint", C.POINTER(EdgeHint)), ("freqs_path", C.c_char_p),
                ("n_gpus", C.c_int32), ("n_passes", C.c_uint32), ("devices", C.POINTER(C.c_int32)), ("flags", C.c_uint32)]


class Out(C.Structure):
    _fields_ = [("K", C.c_int32), ("n_vertices", C.c_uint64), ("n_edge_objs", C.c_uint64),
                ("edge_packed", C.c_void_p), ("edge_byte_off", C.c_void_p), ("edge_len", C.c_void_p),
                ("vleft", C.c_void_p), ("vright", C.c_void_p),
                ("from_off", C.c_void_p), ("from_v", C.c_void_p), ("from_e", C.c_void_p),
                ("to_off", C.c_void_p), ("to_v", C.c_void_p), ("to_e", C.c_void_p),
                ("n_unipaths", C.c_uint64), ("fwd_xlat", C.c_void_p), ("rev_xlat", C.c_void_p),
                ("n_paths", C.c_uint64), ("path_offset", C.c_void_p), ("path_off", C.c_void_p), ("path_edges", C.c_void_p),
                ("hist", C.c_uint64 * 101),
                ("n_kmer_instances", C.c_uint64), ("n_kmers_distinct", C.c_uint64), ("n_kmers_solid", C.c_uint64),
                ("n_reads_pathed", C.c_uint64), ("n_reads_multipathed", C.c_uint64),
                ("ms_count", C.c_float), ("ms_graph", C.c_float), ("ms_path", C.c_float)]


_lib = None


def lib():
    """Load libw2rap_step2.so (raises if the HIP extension has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C w2rap_contigger_amd/csrc` "
                              "(there is no CPU fallback)")
        # torch ships its own libamdhip64 / libhsa-runtime64.  A process can use only the HIP runtime that was loaded FIRST (the second
        # one finds no device), so when torch is around its runtime goes first and this library binds to it through the shared SONAME.
        try:
            import torch  # noqa: F401
        except Exception:        # no torch (or a broken one): the library then runs on the system's HIP runtime alone
            pass
        L = C.CDLL(LIB_PATH)
        L.w2rap_step2_abi_version.restype = C.c_int
        L.w2rap_step2_device_count.restype = C.c_int
        L.w2rap_step2_create.restype = C.c_void_p
        L.w2rap_step2_create.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        L.w2rap_step2_destroy.argtypes = [C.c_void_p]
        L.w2rap_step2_destroy.restype = None
        L.w2rap_step2_last_error.restype = C.c_char_p
        L.w2rap_step2_last_error.argtypes = [C.c_void_p]
        L.w2rap_step2_set_reads.argtypes = [C.c_void_p, C.POINTER(Reads)]
        L.w2rap_step2_count_kmers.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Out)]
        L.w2rap_step2_count_kmers_passes.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Out)]
        L.w2rap_step2_copy_bench.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
        L.w2rap_step2_trim_cached.restype = C.c_int
        L.w2rap_step2_last_peer_mode.restype = C.c_int
        L.w2rap_step2_build_graph.argtypes = [C.c_void_p, C.POINTER(EdgeHint)]
        L.w2rap_step2_shard_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64,
                                              C.POINTER(C.c_uint64), C.POINTER(EdgeHint)]
        L.w2rap_step2_shard_next.argtypes = [C.c_void_p, C.POINTER(Xchg)]
        L.w2rap_step2_local_dict_slice.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.w2rap_step2_shard_recv.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_void_p)]
        L.w2rap_step2_shard_host_words.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.w2rap_step2_shard_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.w2rap_step2_device_bytes.argtypes = [C.c_void_p]
        L.w2rap_step2_device_bytes.restype = C.c_uint64
        L.w2rap_step2_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.w2rap_step2_device_peak_bytes.argtypes = [C.c_void_p, C.c_int]
        L.w2rap_step2_device_peak_bytes.restype = C.c_uint64
        L.w2rap_step2_path_reads.argtypes = [C.c_void_p]
        L.w2rap_step2_fetch.argtypes = [C.c_void_p, C.POINTER(Out)]
        L.w2rap_step2_free.argtypes = [C.POINTER(Out)]
        L.w2rap_step2_free.restype = None
        L.w2rap_step2_stream.restype = C.c_void_p
        L.w2rap_step2_stream.argtypes = [C.c_void_p]
        L.w2rap_step2_get_good_len.argtypes = [C.c_void_p, C.c_void_p]
        L.w2rap_step2_get_table.argtypes = [C.c_void_p] * 7
        L.w2rap_step2_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.w2rap_step2_profile.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int]
        L.w2rap_step2_profile.restype = C.c_size_t
        L.w2rap_step2_quality_windows.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
        L.w2rap_step2_default_buckets.argtypes = [C.c_uint64, C.c_uint32]
        L.w2rap_step2_default_buckets.restype = C.c_uint32
        L.w2rap_step2_record_bytes.restype = C.c_uint32
        L.w2rap_step2_partition.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.w2rap_step2_partition_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.w2rap_step2_count_pass.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.w2rap_step2_partition_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.w2rap_step2_count_records.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                                C.POINTER(Out)]
        L.w2rap_step2_solid_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                C.POINTER(C.c_uint64)]
        L.w2rap_step2_set_solid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                            C.POINTER(C.c_uint64)]
        L.w2rap_step2_chunk_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.w2rap_step2_set_solid_chunked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                                    C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_uint64]
        L.w2rap_step2_count_records_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int]
        L.w2rap_step2_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.w2rap_step2_count_records_launch.argtypes = [C.c_void_p, C.c_uint32]
        L.w2rap_step2_count_records_bounds.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.w2rap_step2_count_records_slices.argtypes = [C.c_void_p]
        L.w2rap_step2_count_records_slice.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.w2rap_step2_count_records_end.argtypes = [C.c_void_p, C.POINTER(Out)]
        L.w2rap_step2_dict_begin.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.w2rap_step2_dict_append.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
        L.w2rap_step2_dict_end.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        L.w2rap_step2_dict_abort.argtypes = [C.c_void_p]
        L.w2rap_step2_run.argtypes = [C.POINTER(Reads), C.POINTER(Params), C.POINTER(Out), C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _np_from(ptr, dtype, n):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


@dataclass
class Step2Result:
    hbv: F.HBV
    vleft: np.ndarray
    vright: np.ndarray
    to_v: np.ndarray
    fwd_xlat: np.ndarray
    rev_xlat: np.ndarray
    path_offset: np.ndarray
    path_off: np.ndarray
    path_edges: np.ndarray
    hist: np.ndarray
    n_kmer_instances: int
    n_kmers_distinct: int
    n_kmers_solid: int
    n_reads_pathed: int
    n_reads_multipathed: int
    ms_count: float
    ms_graph: float
    ms_path: float
    peer_access: object = None     # multi-GPU w2rap_step2_run: "peer" or "host-staged" (no peer access between some GPUs: exchanges through pinned host memory)


def _result(o: Out) -> Step2Result:
    NO, NV, E, NP = o.n_edge_objs, o.n_vertices, o.n_unipaths, o.n_paths
    boff = _np_from(o.edge_byte_off, np.uint64, NO + 1)
    hbv = F.HBV(60, _np_from(o.from_off, np.uint64, NV + 1), _np_from(o.from_v, np.int32, NO), _np_from(o.from_e, np.int32, NO),
                _np_from(o.to_off, np.uint64, NV + 1), _np_from(o.to_e, np.int32, NO),
                _np_from(o.edge_packed, np.uint8, int(boff[-1]) if len(boff) else 0), boff, _np_from(o.edge_len, np.uint32, NO))
    po = _np_from(o.path_off, np.uint64, NP + 1) if NP else np.zeros(1, np.uint64)
    return Step2Result(hbv, _np_from(o.vleft, np.int32, NO), _np_from(o.vright, np.int32, NO), _np_from(o.to_v, np.int32, NO),
                       _np_from(o.fwd_xlat, np.int32, E), _np_from(o.rev_xlat, np.int32, E),
                       _np_from(o.path_offset, np.int32, NP), po, _np_from(o.path_edges, np.int32, int(po[-1])),
                       np.array(list(o.hist), dtype=np.uint64), o.n_kmer_instances, o.n_kmers_distinct, o.n_kmers_solid,
                       o.n_reads_pathed, o.n_reads_multipathed, o.ms_count, o.ms_graph, o.ms_path)


def make_hint(hint_packed, hint_byte_off, hint_len):
    """-> (EdgeHint, keepalive tuple)"""
    hp = np.ascontiguousarray(hint_packed, np.uint8)
    ho = np.ascontiguousarray(hint_byte_off, np.uint64)
    hl = np.ascontiguousarray(hint_len, np.uint32)
    return EdgeHint(len(hl), _ptr(hp), _ptr(ho), _ptr(hl)), (hp, ho, hl)


class Step2Context:
    """Staged access to one GPU (w2rap_step2_create .. destroy)."""

    def __init__(self, device=0):
        self.L = lib()
        err = C.create_string_buffer(512)
        self.h = self.L.w2rap_step2_create(device, err, 512)
        if not self.h:
            raise Step2Error(2, err.value.decode(errors="replace"))
        self._keep = None
        self.n_reads = 0

    def close(self):
        if self.h:
            self.L.w2rap_step2_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc:
            raise Step2Error(rc, self.L.w2rap_step2_last_error(self.h).decode())

    @property
    def stream(self):
        return self.L.w2rap_step2_stream(self.h)

    def set_reads_host(self, packed, byte_off, read_len, quals=None, qual_off=None, pq=None, pq_off=None):
        arrs = [np.ascontiguousarray(packed, np.uint8), np.ascontiguousarray(byte_off, np.uint64),
                np.ascontiguousarray(read_len, np.uint32),
                None if quals is None else np.ascontiguousarray(quals, np.uint8),
                None if qual_off is None else np.ascontiguousarray(qual_off, np.uint64),
                None if pq is None else np.ascontiguousarray(pq, np.uint8),
                None if pq_off is None else np.ascontiguousarray(pq_off, np.uint64)]
        r = Reads(len(arrs[2]), *[_ptr(a) for a in arrs], MEM_HOST)
        self._check(self.L.w2rap_step2_set_reads(self.h, C.byref(r)))
        self.n_reads = len(arrs[2])

    def set_reads_device(self, n_reads, d_packed, d_byte_off, d_read_len, d_quals, d_qual_off, keepalive=None):
        """device pointers (ints), used in place; `keepalive` holds the owning tensors"""
        r = Reads(n_reads, d_packed, d_byte_off, d_read_len, d_quals, d_qual_off, None, None, MEM_DEVICE)
        self._keep = keepalive
        self._check(self.L.w2rap_step2_set_reads(self.h, C.byref(r)))
        self.n_reads = n_reads

    def count_kmers(self, min_qual=7, min_freq=4, n_passes=None):
        """n_passes: count in that many hash-range passes over the reads (None: one pass unless the library finds the HBM short)"""
        o = Out()
        if n_passes is None:
            self._check(self.L.w2rap_step2_count_kmers(self.h, min_qual, min_freq, C.byref(o)))
        else:
            self._check(self.L.w2rap_step2_count_kmers_passes(self.h, min_qual, min_freq, n_passes, C.byref(o)))
        return dict(hist=np.array(list(o.hist), dtype=np.uint64), M=o.n_kmer_instances, D=o.n_kmers_distinct,
                    S=o.n_kmers_solid, ms=o.ms_count)

    # ---- multi-GPU building blocks (device pointers are plain ints) ----
    def quality_windows(self, min_qual=7) -> int:
        m = C.c_uint64(0)
        self._check(self.L.w2rap_step2_quality_windows(self.h, min_qual, C.byref(m)))
        return m.value

    def default_buckets(self, total_kmers, multiple_of=1) -> int:
        return self.L.w2rap_step2_default_buckets(total_kmers, multiple_of)

    def partition(self, n_buckets, n_parts, first_bucket=0, end_bucket=None):
        """-> (records ptr, n_records, bucket-counts ptr, records per part); self.kmers_per_part is set too.
        (first_bucket, end_bucket): one hash-range pass -- only the records of that bucket range are kept, the parts divide the range"""
        per = (C.c_uint64 * n_parts)()
        kper = (C.c_uint64 * n_parts)()
        end_bucket = n_buckets if end_bucket is None else end_bucket
        self._check(self.L.w2rap_step2_partition_range(self.h, n_buckets, first_bucket, end_bucket, n_parts, per, kper))
        self.kmers_per_part = [int(x) for x in kper]
        recs, cnts, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self.L.w2rap_step2_partition_buffers(self.h, C.byref(recs), C.byref(cnts), C.byref(n)))
        return recs.value or 0, n.value, cnts.value or 0, [int(x) for x in per]

    def count_pass(self, k, n_passes):
        """the count_records call(s) that follow are hash-range pass k of n_passes on this owner (a later pass appends to the earlier ones)"""
        self._check(self.L.w2rap_step2_count_pass(self.h, k, n_passes))

    def count_records(self, min_freq, n_local_buckets, n_segments, d_records, d_counts, total_kmers):
        o = Out()
        self._check(self.L.w2rap_step2_count_records(self.h, min_freq, n_local_buckets, n_segments, d_records, d_counts,
                                                     total_kmers, C.byref(o)))
        return dict(hist=np.array(list(o.hist), dtype=np.uint64), D=o.n_kmers_distinct, S=o.n_kmers_solid)

    def count_records_begin(self, min_freq, n_local_buckets, n_segments, d_records, d_counts, total_kmers, n_slices, deferred=False) -> int:
        """plans the count in bucket slices [nbl*k//n, nbl*(k+1)//n) and (unless deferred) launches them all; returns at once
        -> number of slices"""
        self._check(self.L.w2rap_step2_count_records_begin(self.h, min_freq, n_local_buckets, n_segments, d_records, d_counts, total_kmers, n_slices,
                                                           1 if deferred else 0))
        return int(self.L.w2rap_step2_count_records_slices(self.h))

    def count_records_bounds(self, k):
        """-> (first bucket, end bucket) of slice k of the planned count"""
        a, b = C.c_uint32(), C.c_uint32()
        self._check(self.L.w2rap_step2_count_records_bounds(self.h, k, C.byref(a), C.byref(b)))
        return a.value, b.value

    def count_records_launch(self, k):
        """deferred mode: launch slice k (in order) -- the records of its buckets are complete in d_records"""
        self._check(self.L.w2rap_step2_count_records_launch(self.h, k))

    def count_records_slice(self, k):
        """waits for slice k -> (solid k-mers, chunks) appended by slices 0..k"""
        s, c = C.c_uint64(), C.c_uint64()
        self._check(self.L.w2rap_step2_count_records_slice(self.h, k, C.byref(s), C.byref(c)))
        return s.value, c.value

    def count_records_end(self):
        o = Out()
        self._check(self.L.w2rap_step2_count_records_end(self.h, C.byref(o)))
        return dict(hist=np.array(list(o.hist), dtype=np.uint64), D=o.n_kmers_distinct, S=o.n_kmers_solid)

    def dict_begin(self, kmer_cap, chunk_cap):
        self._check(self.L.w2rap_step2_dict_begin(self.h, kmer_cap, chunk_cap))

    def dict_append(self, d_hi, d_lo, d_cc, n, d_chunk_start=None, d_chunk_count=None, n_chunks=0):
        self._check(self.L.w2rap_step2_dict_append(self.h, d_hi, d_lo, d_cc, n, d_chunk_start, d_chunk_count, n_chunks))

    def dict_end(self, M, D, hist):
        h = (C.c_uint64 * 101)(*[int(x) for x in hist])
        self._check(self.L.w2rap_step2_dict_end(self.h, M, D, h))

    def dict_abort(self):
        self._check(self.L.w2rap_step2_dict_abort(self.h))

    def solid_buffers(self):
        """-> (hi ptr, lo ptr, cc ptr, n)"""
        hi, lo, cc, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self.L.w2rap_step2_solid_buffers(self.h, C.byref(hi), C.byref(lo), C.byref(cc), C.byref(n)))
        return hi.value or 0, lo.value or 0, cc.value or 0, n.value

    def chunk_buffers(self):
        """-> (chunk-start ptr (u64), chunk-count ptr (u32), n): the contiguous runs of solid k-mers K3 emitted per bucket"""
        st, cn, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self.L.w2rap_step2_chunk_buffers(self.h, C.byref(st), C.byref(cn), C.byref(n)))
        return st.value or 0, cn.value or 0, n.value

    def set_solid(self, d_hi, d_lo, d_cc, n, M, D, hist, d_chunk_start=None, d_chunk_count=None, n_chunks=0):
        h = (C.c_uint64 * 101)(*[int(x) for x in hist])
        self._check(self.L.w2rap_step2_set_solid_chunked(self.h, d_hi, d_lo, d_cc, n, M, D, h, d_chunk_start, d_chunk_count, n_chunks))

    def build_graph(self, hint=None):
        if hint is None:
            self._check(self.L.w2rap_step2_build_graph(self.h, None))
        else:
            eh, keep = make_hint(*hint)
            self._check(self.L.w2rap_step2_build_graph(self.h, C.byref(eh)))

    def path_reads(self):
        self._check(self.L.w2rap_step2_path_reads(self.h))

    # ---- row e-3: dictionary, prune and unipaths sharded by bucket owner (a state machine between exchanges; dist.sharded_graph drives it)
    def shard_begin(self, rank, world, solid_per_rank, n_buckets, n_passes, M, D, hist, hint=None):
        spr = (C.c_uint64 * world)(*[int(x) for x in solid_per_rank])
        h = (C.c_uint64 * 101)(*[int(x) for x in hist])
        self._shard_hint = None
        if hint is None:
            self._check(self.L.w2rap_step2_shard_begin(self.h, rank, world, spr, n_buckets, n_passes, M, D, h, None))
        else:
            eh, keep = make_hint(*hint)
            self._shard_hint = (eh, keep)                       # host memory the library reads until the phase is done
            self._check(self.L.w2rap_step2_shard_begin(self.h, rank, world, spr, n_buckets, n_passes, M, D, h, C.byref(eh)))

    def local_dict_slice(self, n_solid, expected_total):
        """the owner's own dictionary takes the solid k-mers counted so far on the side stream (under the counting of the next bucket slice)"""
        self._check(self.L.w2rap_step2_local_dict_slice(self.h, n_solid, expected_total))

    def shard_next(self) -> Xchg:
        x = Xchg()
        self._check(self.L.w2rap_step2_shard_next(self.h, C.byref(x)))
        return x

    def shard_recv(self, recv_counts, elem_bytes) -> int:
        """room for what an exchange delivers (counts in elements, per source rank) -> device pointer"""
        rc = (C.c_uint64 * len(recv_counts))(*[int(x) for x in recv_counts])
        p = C.c_void_p()
        self._check(self.L.w2rap_step2_shard_recv(self.h, rc, elem_bytes, C.byref(p)))
        return p.value or 0

    def shard_host_words(self, words):
        w = (C.c_uint64 * len(words))(*[int(x) for x in words])
        self._check(self.L.w2rap_step2_shard_host_words(self.h, w))

    def shard_info(self) -> dict:
        o = (C.c_uint64 * 8)()
        self._check(self.L.w2rap_step2_shard_info(self.h, o))
        return dict(zip(("solid_local", "solid_total", "segments_local", "segments_total", "unipaths", "edge_bases", "index_entries", "phase"), [int(x) for x in o]))

    def device_bytes(self) -> int:
        return int(self.L.w2rap_step2_device_bytes(self.h))

    def device_copy(self, dst_ptr: int, src_ptr: int, nbytes: int):
        """device-to-device copy by the library's copy kernel (complete on return)"""
        self._check(self.L.w2rap_step2_device_copy(self.h, dst_ptr, src_ptr, nbytes))

    def device_peak_bytes(self, reset=False) -> int:
        """the maximum of device_bytes() since the context was created / since the last call with reset=True"""
        return int(self.L.w2rap_step2_device_peak_bytes(self.h, 1 if reset else 0))

    def copy_bandwidth(self, nbytes=4 << 30, reps=5) -> float:
        """GB/s (read + written) of a plain 16-B-per-lane device copy on this GPU"""
        g = C.c_double(0)
        self._check(self.L.w2rap_step2_copy_bench(self.h, nbytes, reps, C.byref(g)))
        return g.value

    def counts(self) -> dict:
        """sizes of what the context holds (no transfer of the results themselves)"""
        o = (C.c_uint64 * 8)()
        self._check(self.L.w2rap_step2_counts(self.h, o))
        return dict(zip(("kmer_instances", "kmers_distinct", "kmers_solid", "unipaths", "edge_objects", "vertices", "reads_pathed", "path_elements"),
                        [int(x) for x in o]))

    def fetch(self) -> Step2Result:
        o = Out()
        self._check(self.L.w2rap_step2_fetch(self.h, C.byref(o)))
        try:
            return _result(o)
        finally:
            self.L.w2rap_step2_free(C.byref(o))

    def set_profiling(self, on: bool):
        self.L.w2rap_step2_set_profiling(self.h, 1 if on else 0)

    def profile(self, reset=True):
        """-> {kernel name: (total ms, launches)} from hipEvents on the context's stream"""
        buf = C.create_string_buffer(1 << 16)
        self.L.w2rap_step2_profile(self.h, buf, len(buf), 1 if reset else 0)
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, n = line.rsplit(" ", 2)
            out[name] = (float(ms), int(n))
        return out

    def good_len(self):
        out = np.zeros(self.n_reads, np.uint16)
        self._check(self.L.w2rap_step2_get_good_len(self.h, _ptr(out)))
        return out

    def table(self, S):
        hi = np.zeros(S, np.uint64); lo = np.zeros(S, np.uint64); cnt = np.zeros(S, np.uint8); ctx = np.zeros(S, np.uint8)
        edge = np.zeros(S, np.int32); off = np.zeros(S, np.uint32)
        self._check(self.L.w2rap_step2_get_table(self.h, _ptr(hi), _ptr(lo), _ptr(cnt), _ptr(ctx), _ptr(edge), _ptr(off)))
        return hi, lo, cnt, ctx, edge, off


def build_read_qgraph(packed, byte_off, read_len, quals=None, qual_off=None, pq=None, pq_off=None,
                      min_qual=7, min_freq=4, device=0, edge_order_hint=None, freqs_path=None, n_gpus=1, devices=None,
                      n_passes=0, timing=None, replicated_graph=False, graph_only=False, device_reads=None) -> Step2Result:
    """buildReadQGraph + FixPaths through the one-shot C entry point (w2rap_step2_run).  n_gpus > 1: that many devices from `device`
    on (or the ordinals in `devices`, which may repeat); n_passes: hash-range passes of the counting phase (0 = automatic);
    replicated_graph: with several GPUs gather the dictionary and build the graph on every one (rounds 1-4) instead of keeping
    dictionary, prune and unipaths sharded by bucket owner; graph_only: no read pathing (pPaths == nullptr, BuildReadQGraph.cc:1300-1307)."""
    L = lib()
    if device_reads is not None:
        # device-resident reads (W2RAP_MEM_DEVICE): a dict of n and the raw device pointers of the arrays, all on ONE GPU; with several
        # GPUs every rank takes its shard from there (peer copies, or host-staged ones where the driver grants no peer access)
        g = device_reads.get
        r = Reads(int(device_reads["n"]), *[(int(g(k)) if g(k) else None) for k in ("packed", "byte_off", "read_len", "quals", "qual_off", "pq", "pq_off")], MEM_DEVICE)
    else:
        arrs = [np.ascontiguousarray(packed, np.uint8), np.ascontiguousarray(byte_off, np.uint64),
                np.ascontiguousarray(read_len, np.uint32),
                None if quals is None else np.ascontiguousarray(quals, np.uint8),
                None if qual_off is None else np.ascontiguousarray(qual_off, np.uint64),
                None if pq is None else np.ascontiguousarray(pq, np.uint8),
                None if pq_off is None else np.ascontiguousarray(pq_off, np.uint64)]
        r = Reads(len(arrs[2]), *[_ptr(a) for a in arrs], MEM_HOST)
    keep = None
    hint_p = None
    if edge_order_hint is not None:
        eh, keep = make_hint(*edge_order_hint)
        hint_p = C.pointer(eh)
    dev_arr = None
    if devices is not None:
        n_gpus = len(devices)
        dev_arr = (C.c_int32 * n_gpus)(*[int(d) for d in devices])
    p = Params(60, min_qual, min_freq, device, hint_p, None if freqs_path is None else os.fsencode(freqs_path), n_gpus, n_passes, dev_arr,
               (F_REPLICATED_GRAPH if replicated_graph else 0) | (F_GRAPH_ONLY if graph_only else 0))
    o = Out()
    err = C.create_string_buffer(1024)
    import time
    t0 = time.perf_counter()
    rc = L.w2rap_step2_run(C.byref(r), C.byref(p), C.byref(o), err, 1024)
    if timing is not None:
        timing["run_s"] = time.perf_counter() - t0           # the C call alone (the numpy copies of the result below are this wrapper's)
    if rc:
        raise Step2Error(rc, err.value.decode(errors="replace"))
    try:
        res = _result(o)
        # how the ranks of a multi-GPU call reached each other: "peer" copies, or "host-staged" where the driver granted no peer access
        res.peer_access = {0: None, 1: "peer", 2: "host-staged"}.get(int(L.w2rap_step2_last_peer_mode()), None) if n_gpus > 1 else None
        return res
    finally:
        L.w2rap_step2_free(C.byref(o))


def run_step2_files(out_dir, prefix, min_qual=7, min_freq=4, device=0, edge_order_hint=None) -> Step2Result:
    """The reference's Step 2 on an output directory: reads <out_dir>/frag_reads_orig.{fastb,qualp}
    (w2rap-contigger.cc:326-327), writes <out_dir>/<prefix>.small_K.{hbv,paths} (:345-346) and
    <out_dir>/small_K.freqs (BuildReadQGraph.cc:1108-1112)."""
    packed, byte_off, read_len = F.read_fastb(os.path.join(out_dir, "frag_reads_orig.fastb"))
    pq, pq_off = F.read_qualp(os.path.join(out_dir, "frag_reads_orig.qualp"))
    res = build_read_qgraph(packed, byte_off, read_len, pq=pq, pq_off=pq_off, min_qual=min_qual, min_freq=min_freq,
                            device=device, edge_order_hint=edge_order_hint,
                            freqs_path=os.path.join(out_dir, "small_K.freqs"))
    F.write_hbv(os.path.join(out_dir, f"{prefix}.small_K.hbv"), res.hbv)
    F.write_paths(os.path.join(out_dir, f"{prefix}.small_K.paths"), res.path_offset, res.path_off, res.path_edges)
    return res
