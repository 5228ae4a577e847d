"""Multi-GPU Step 2: one process per GPU, torch.distributed (RCCL over xGMI).

Reads are sharded by rank.  K-mer counting has ONE real exchange step, the k-mer shuffle:
every rank cuts its reads into super-k-mer records, bucketed by canonical minimizer; bucket
b belongs to rank b // (n_buckets / world); records travel to their owner with an
all_to_all_v (the GPU counterpart of MapReduceEngine's swizzle, src/MapReduceEngine.h:320-361),
bucket slice by bucket slice under the counting, and owners count their buckets
(`distributed_count`).  What follows is sharded too (row e-3, the default since round 5,
`sharded_graph`): every owner KEEPS its solid k-mers -- its own dictionary, the adjacency
prune with routed queries for neighbours that live elsewhere, the unipaths by a two-level
list ranking whose exchanges the library's state machine (csrc/step2_shard.hip) asks for
one at a time; only E- and genome-sized results (the ordered edge list, the packed edge
stream, the minimizer-sampled pathing index) end up on every rank, and read pathing of the
rank's own reads runs against that index.  `distributed_count(gather=True)` + build_graph
is the replicated-dictionary path of rounds 1-4 (W2RAP_REPLICATED_GRAPH=1), kept for
comparison.  Step 3 (`distributed_repath`): places exchanged, large-K graph per rank.

The orchestration is written against a small backend interface so that the same code runs on
the HIP library (`GpuBackend`) and, in the CPU tests, on a numpy stand-in over gloo.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist



class _DevArray:
    """exposes a raw device pointer to torch through __cuda_array_interface__ (no copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def dev_bytes(ptr, nbytes, device):
    if nbytes == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(_DevArray(ptr, nbytes), device=device)


class GpuBackend:
    """Backend over libw2rap_step2.so: every array is a torch tensor on this rank's GPU."""

    def __init__(self, ctx, device):
        self.ctx = ctx
        self.device = torch.device(device)

    def copy_on_device(self, out, inp):
        """out <- inp (contiguous tensors of equal byte size on this device) by the library's copy kernel: the part of an exchange that stays on
        the rank.  (hipMemcpy device-to-device runs through SDMA at ~180 GB/s here, an RCCL send to oneself at ~30 GB/s; the kernel: 2.5 TB/s.)"""
        nbytes = inp.numel() * inp.element_size()
        if not nbytes:
            return
        if not (out.is_contiguous() and inp.is_contiguous()) or out.numel() * out.element_size() != nbytes:
            out.copy_(inp)
            return
        torch.cuda.current_stream(self.device).synchronize()        # what produced inp / last read out on torch's stream is complete
        self.ctx.device_copy(out.data_ptr(), inp.data_ptr(), nbytes)

    def quality_windows(self, min_qual):
        return self.ctx.quality_windows(min_qual)

    def default_buckets(self, total_kmers, world):
        return self.ctx.default_buckets(total_kmers, world)

    def partition(self, n_buckets, world, first_bucket=0, end_bucket=None):
        end_bucket = n_buckets if end_bucket is None else end_bucket
        recs, nrec, cnts, per = self.ctx.partition(n_buckets, world, first_bucket, end_bucket)
        rb = int(self.ctx.L.w2rap_step2_record_bytes())
        r = dev_bytes(recs, nrec * rb, self.device).view(nrec, rb)
        c = dev_bytes(cnts, (end_bucket - first_bucket) * 4, self.device).view(torch.int32)
        self.kmers_per_part = self.ctx.kmers_per_part
        return r, c, per

    def count_records(self, min_freq, nbl, nseg, records, counts, total_kmers):
        torch.cuda.synchronize(self.device)
        self._keep = (records, counts)
        return self.ctx.count_records(min_freq, nbl, nseg, records.data_ptr(), counts.data_ptr(), total_kmers)

    # ---- the same in slices: slice k's solid k-mers are exchanged while slice k+1 is being counted
    def count_pass(self, k, n_passes):
        """the count that follows is hash-range pass k of n_passes: its solid k-mers are appended to those of the earlier passes"""
        self._pass = k
        self.ctx.count_pass(k, n_passes)

    def count_begin(self, min_freq, nbl, nseg, records, counts, total_kmers, n_slices):
        """plans the count; slice k is started with count_launch(k) once the records of its buckets have arrived"""
        torch.cuda.current_stream(self.device).synchronize()       # the received per-bucket counts are complete
        self._keep = (records, counts)
        if not getattr(self, "_pass", 0):
            self._prev = (0, 0)                                     # (a later pass goes on behind the solid k-mers of the earlier ones)
            self._appended = []
        return self.ctx.count_records_begin(min_freq, nbl, nseg, records.data_ptr(), counts.data_ptr(), total_kmers, n_slices, deferred=True)

    def count_bounds(self, ns):
        """bucket boundaries of the ns slices"""
        return [self.ctx.count_records_bounds(k)[0] for k in range(ns)] + [self.ctx.count_records_bounds(ns - 1)[1]]

    def count_launch(self, k):
        torch.cuda.current_stream(self.device).synchronize()       # slice k's records are complete
        self.ctx.count_records_launch(k)

    def count_slice(self, k):
        """blocks until slice k is counted -> (hi, lo, cc, chunk start relative to the slice, chunk count) of ITS solid k-mers"""
        s_k, c_k = self.ctx.count_records_slice(k)
        s0, c0 = self._prev
        self._prev = (s_k, c_k)
        hi, lo, cc, _ = self.ctx.solid_buffers()
        st, cn, _ = self.ctx.chunk_buffers()
        d = self.device
        n, nc = s_k - s0, c_k - c0
        cs = dev_bytes(st + 8 * c0 if st else 0, nc * 8, d).view(torch.int64) - s0
        return (dev_bytes(hi + 8 * s0, n * 8, d).view(torch.int64), dev_bytes(lo + 8 * s0, n * 8, d).view(torch.int64),
                dev_bytes(cc + 4 * s0, n * 4, d).view(torch.int32), cs, dev_bytes(cn + 4 * c0 if cn else 0, nc * 4, d).view(torch.int32))

    def count_end(self):
        return self.ctx.count_records_end()

    def dict_begin(self, kmer_cap, chunk_cap):
        self.ctx.dict_begin(kmer_cap, chunk_cap)

    def dict_append(self, hi, lo, cc, cs, cn):
        """one gathered block; its tensors must be complete on the current stream's timeline (we wait for it) and stay alive"""
        torch.cuda.current_stream(self.device).synchronize()
        self._appended.append((hi, lo, cc, cs, cn))
        self.ctx.dict_append(hi.data_ptr(), lo.data_ptr(), cc.data_ptr(), hi.numel(), cs.data_ptr() if cs.numel() else 0,
                             cn.data_ptr() if cn.numel() else 0, cs.numel())

    def dict_end(self, M, D, hist):
        self.ctx.dict_end(M, D, hist)
        self._appended = []
        self._pass = 0

    def dict_abort(self):
        """drops the half-built dictionary (device synchronised by the library) and the gathered blocks it was reading"""
        self.ctx.dict_abort()
        self._appended = []
        self._pass = 0

    def solid(self):
        hi, lo, cc, n = self.ctx.solid_buffers()
        return (dev_bytes(hi, n * 8, self.device).view(torch.int64), dev_bytes(lo, n * 8, self.device).view(torch.int64),
                dev_bytes(cc, n * 4, self.device).view(torch.int32))

    def chunks(self):
        """(first solid k-mer, count) of every bucket's contiguous run in this rank's solid arrays"""
        st, cn, n = self.ctx.chunk_buffers()
        return dev_bytes(st, n * 8, self.device).view(torch.int64), dev_bytes(cn, n * 4, self.device).view(torch.int32)

    def set_solid(self, hi, lo, cc, M, D, hist, chunk_start=None, chunk_count=None):
        torch.cuda.synchronize(self.device)
        self._pass = 0
        if chunk_start is not None and chunk_start.numel():
            self._keep_chunks = (chunk_start, chunk_count)
            self.ctx.set_solid(hi.data_ptr(), lo.data_ptr(), cc.data_ptr(), hi.numel(), M, D, hist,
                               chunk_start.data_ptr(), chunk_count.data_ptr(), chunk_start.numel())
        else:
            self.ctx.set_solid(hi.data_ptr(), lo.data_ptr(), cc.data_ptr(), hi.numel(), M, D, hist)

    def sync(self):
        torch.cuda.synchronize(self.device)

    # ---- the sharded graph phase (row e-3): the library's state machine; arrays are views of its device buffers
    def shard_begin(self, rank, world, solid_per_rank, n_buckets, n_passes, M, D, hist, hint=None):
        self.ctx.shard_begin(rank, world, solid_per_rank, n_buckets, n_passes, M, D, hist, hint)

    def local_dict_slice(self, n_solid, expected_total):
        self.ctx.local_dict_slice(n_solid, expected_total)

    def shard_next(self):
        """-> (op, elem_bytes, send view (bytes) or None, send_count list[64])"""
        x = self.ctx.shard_next()
        cnt = list(x.send_count)
        return x.op, x.elem_bytes, x.send or 0, cnt

    def shard_view(self, ptr, nbytes):
        return dev_bytes(ptr, nbytes, self.device)

    def shard_host_word(self, ptr):
        import ctypes
        return int(ctypes.c_uint64.from_address(ptr).value)

    def shard_recv(self, recv_counts, elem_bytes):
        p = self.ctx.shard_recv(recv_counts, elem_bytes)
        return dev_bytes(p, int(sum(recv_counts)) * elem_bytes, self.device)

    def shard_host_words(self, words):
        self.ctx.shard_host_words(words)


def _host_staged(group) -> bool:
    """gloo has no device all_to_all / all_gather_into_tensor: with that backend (the two-ranks-on-one-GPU test) device
    tensors make the trip through host memory.  RCCL ("nccl") exchanges them in place over xGMI."""
    return dist.get_backend(group) == "gloo"


_scratch = {}


def _buffer(key, nbytes, device):
    """a persistent byte buffer per (purpose, device), grown on demand: the exchange buffers are GBs and would otherwise be
    re-requested from the caching allocator -- at times from the driver, a 200 ms stall -- in every step"""
    b = _scratch.get((key, str(device)))
    if b is None or b.numel() < nbytes:
        _scratch.pop((key, str(device)), None)
        b = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _scratch[(key, str(device))] = b
    return b[:nbytes]


A2A_MAX_BYTES = 512 << 20     # per call and rank; RCCL 2.26 (ROCm 7.0) returns garbage in the second half of an
                               # all_to_all_single of 2 GiB and more (tools/rccl_a2a_check.py), 0.33 GiB is fine
A2A_MAX_PEER_BYTES = 1 << 30   # per call and PEER on the point-to-point path (a send/recv per peer, each far below 2 GiB)


def _a2a_once(out, inp, out_splits, in_splits, group):
    if _host_staged(group) and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None, group=None):
    """all_to_all_v of rows; large exchanges go in several rounds: in round k every rank sends every peer the k-th of R equal
    pieces of that peer's rows (both sides cut the same way, so the pieces line up).  gloo (CPU tests, two ranks on one
    GPU) has only all_to_all_single: there the round's pieces are staged contiguously."""
    if out_splits is None:
        return _a2a_once(out, inp, None, None, group)
    row = inp[0].numel() * inp.element_size() if inp.numel() else (out[0].numel() * out.element_size() if out.numel() else 1)
    in_off = [0]
    for n in in_splits: in_off.append(in_off[-1] + n)
    out_off = [0]
    for n in out_splits: out_off.append(out_off[-1] + n)
    if inp.is_cuda and not _host_staged(group):
        # RCCL: one send/recv pair per peer straight from / into views of the callers' arrays (no staging copies); only a peer's
        # share beyond A2A_MAX_PEER_BYTES is cut into rounds (8 ranks x 50 M reads: 0.8 GB per peer, one round)
        need = torch.tensor([max(list(in_splits) + list(out_splits) + [0]) * row], dtype=torch.int64, device=inp.device)
        dist.all_reduce(need, op=dist.ReduceOp.MAX, group=group)
        rounds = max(1, -(-int(need.item()) // A2A_MAX_PEER_BYTES))
        cut = lambda n, k: n * k // rounds
        for k in range(rounds):
            ins = [inp[in_off[p] + cut(n, k): in_off[p] + cut(n, k + 1)] for p, n in enumerate(in_splits)]
            outs = [out[out_off[p] + cut(n, k): out_off[p] + cut(n, k + 1)] for p, n in enumerate(out_splits)]
            dist.all_to_all(outs, ins, group=group)
        return
    need = torch.tensor([max(sum(in_splits), sum(out_splits)) * row], dtype=torch.int64, device=inp.device)
    _all_reduce(need, group=group, op=dist.ReduceOp.MAX)
    rounds = max(1, -(-int(need.item()) // A2A_MAX_BYTES))
    if rounds == 1:
        return _a2a_once(out, inp, out_splits, in_splits, group)
    cut = lambda n, k: n * k // rounds
    for k in range(rounds):
        isp = [cut(n, k + 1) - cut(n, k) for n in in_splits]
        osp = [cut(n, k + 1) - cut(n, k) for n in out_splits]
        src = _buffer("a2a_src", sum(isp) * row, inp.device).view(inp.dtype).view((sum(isp),) + tuple(inp.shape[1:]))
        o = 0
        for p, n in enumerate(in_splits):
            src[o:o + isp[p]] = inp[in_off[p] + cut(n, k): in_off[p] + cut(n, k + 1)]
            o += isp[p]
        dst = _buffer("a2a_dst", sum(osp) * row, out.device).view(out.dtype).view((sum(osp),) + tuple(out.shape[1:]))
        _a2a_once(dst, src, osp, isp, group)
        o = 0
        for p, n in enumerate(out_splits):
            out[out_off[p] + cut(n, k): out_off[p] + cut(n, k + 1)] = dst[o:o + osp[p]]
            o += osp[p]


def _exchange_views(outs, ins, rounds, group, copier=None):
    """all_to_all_v on lists of row views (ins[p] goes to rank p, outs[p] comes from it).  RCCL: one send/recv per peer and
    round; gloo: the pieces are staged contiguously around all_to_all_single."""
    dev = ins[0].device
    # copier(out, inp): how the part of an exchange that stays on this rank is copied (GpuBackend.copy_on_device: the library's copy kernel)
    local = lambda o, i: copier(o, i) if (copier is not None and i.is_cuda) else o.copy_(i)
    if len(ins) == 1:                                         # world 1 (the forced-distributed runs): nothing travels -- a device copy, not
        local(outs[0], ins[0])                                # a collective with oneself
        return
    direct = ins[0].is_cuda and not _host_staged(group)
    me = dist.get_rank(group)
    if direct and os.environ.get("W2RAP_DIST_SELF_SEND") != "1":      # (W2RAP_DIST_SELF_SEND=1: the rank's own share goes through the collective as before round 6)
        # the rank's own share never enters the collective (RCCL moves a "self send" at ~30 GB/s): the copy kernel takes it, and the
        # collective sees an empty piece in its place
        local(outs[me], ins[me])
        outs = list(outs); ins = list(ins)
        outs[me] = outs[me][:0]; ins[me] = ins[me][:0]
    cut = lambda n, k: n * k // rounds
    for k in range(rounds):                                   # both sides cut every piece the same way, so the parts line up
        oo = [o[cut(o.shape[0], k):cut(o.shape[0], k + 1)] for o in outs]
        ii = [i[cut(i.shape[0], k):cut(i.shape[0], k + 1)] for i in ins]
        if direct:
            dist.all_to_all(oo, ii, group=group)
            continue
        src = torch.cat([i.cpu() for i in ii])
        dst = torch.empty((sum(o.shape[0] for o in oo),) + tuple(src.shape[1:]), dtype=src.dtype)
        dist.all_to_all_single(dst, src.contiguous(), output_split_sizes=[o.shape[0] for o in oo], input_split_sizes=[i.shape[0] for i in ii],
                               group=group)
        o0 = 0
        for o in oo:
            o.copy_(dst[o0:o0 + o.shape[0]].to(dev)); o0 += o.shape[0]


def _all_reduce(t: torch.Tensor, group=None, op=None):
    op = op if op is not None else dist.ReduceOp.SUM
    if _host_staged(group) and t.is_cuda:
        c = t.cpu(); dist.all_reduce(c, op=op, group=group); t.copy_(c)
    else:
        dist.all_reduce(t, op=op, group=group)


def _all_gather_v(t: torch.Tensor, group):
    """all_gather of 1-D tensors of different lengths -> concatenation in rank order"""
    world = dist.get_world_size(group)
    dev = t.device
    if _host_staged(group) and t.is_cuda:
        t = t.cpu()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes + [1])
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    if t.is_cuda:
        out = torch.empty(world * mx, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, pad, group=group)
    else:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out = torch.cat(parts)
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)]).to(dev)


N_SLICES = 4             # bucket slices of the owner-side count (the library uses fewer for tiny inputs)
DICT_HEADROOM = 1.15     # capacity of the gathered dictionary over the first slice's extrapolation
MAX_SOLID = (1 << 32) - (1 << 20) - 1  # solid k-mers per GPU: just below the library's own limit (common.h MAX_SOLID_KMERS);  (the library switches to 64-bit node ids beyond 2^31; its rank words hold 33-bit ids)


def _all_gather_sizes(vals, dev, group):
    """every rank's small list of ints -> [world][len(vals)]"""
    world = dist.get_world_size(group)
    t = torch.tensor(vals, dtype=torch.int64, device=dev)
    if _host_staged(group) and t.is_cuda:
        t = t.cpu()
    if t.is_cuda:
        out = torch.empty(world * t.numel(), dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        return out.view(world, -1).tolist()
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)
    return [x.tolist() for x in parts]


def _all_gather_blocks(hi, lo, cc, cs, cn, sizes, group):
    """ONE all_gather for a slice: every rank packs its (hi | lo | cc | chunk starts | chunk counts) into a byte block padded to the
    largest rank's sizes; yields each rank's (hi, lo, cc, cs, cn) as views of the receive buffer, in rank order."""
    world = dist.get_world_size(group)
    dev = hi.device
    mx = max(max(x[0] for x in sizes), 1); mx += mx & 1                  # even: the 4-byte block keeps 8-byte alignment behind it
    mc = max(max(x[1] for x in sizes), 1); mc += mc & 1
    row = 20 * mx + 12 * mc
    staged = _host_staged(group) and hi.is_cuda
    sdev = torch.device("cpu") if staged else dev
    send = _buffer("dict_send", row, sdev)
    n, nc = hi.numel(), cs.numel()
    send[0:8 * n] = hi.view(torch.uint8).to(sdev)
    send[8 * mx:8 * mx + 8 * n] = lo.view(torch.uint8).to(sdev)
    send[16 * mx:16 * mx + 4 * n] = cc.view(torch.uint8).to(sdev)
    send[20 * mx:20 * mx + 8 * nc] = cs.view(torch.uint8).to(sdev)
    send[20 * mx + 8 * mc:20 * mx + 8 * mc + 4 * nc] = cn.view(torch.uint8).to(sdev)
    # a fresh receive buffer per slice: the library reads it asynchronously (side stream) until dict_end
    if send.is_cuda:
        recv = torch.empty(world * row, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(recv, send, group=group)
    else:
        parts = [torch.empty(row, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
        recv = torch.cat(parts).to(dev)
    for r in range(world):
        b = recv[r * row:(r + 1) * row]
        nr, ncr = sizes[r]
        yield (b[0:8 * nr].view(torch.int64), b[8 * mx:8 * mx + 8 * nr].view(torch.int64), b[16 * mx:16 * mx + 4 * nr].view(torch.int32),
               b[20 * mx:20 * mx + 8 * ncr].view(torch.int64), b[20 * mx + 8 * mc:20 * mx + 8 * mc + 4 * ncr].view(torch.int32))


def distributed_count(backend, min_qual=7, min_freq=4, group=None, n_passes=1, gather=True):
    """The sharded a1-a6: returns job-wide statistics; afterwards every rank's backend holds the
    complete solid-k-mer dictionary (as after count_kmers on one GPU).
    gather=False (row e-3): every owner KEEPS its solid k-mers -- nothing is all-gathered, no job-wide dictionary is built; sharded_graph()
    continues from there.
    n_passes > 1: the counting goes in hash-range passes (SURVEY.md 8e "if HBM is short", MapReduceEngine.h:286-299): pass p cuts all the
    reads again, keeps the records of the p-th part of the bucket range only, the owners divide THAT range; records in flight and on the
    owners shrink by the number of passes, the results are the same."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = backend.device
    import os, time
    trace = os.environ.get("W2RAP_TRACE") and rank == 0
    marks = []

    def mark(what):
        if trace:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            marks.append((what, time.perf_counter()))
    mark("start")
    P = max(1, int(n_passes))
    # a1 on the local reads; agree on the bucket count from the job-wide number of k-mer instances
    m_local = backend.quality_windows(min_qual)
    m = torch.tensor([m_local], dtype=torch.int64, device=dev)
    _all_reduce(m, group=group)
    m_total = int(m.item())
    nb = backend.default_buckets(m_total, world * P)
    nbl = nb // world // P
    mark("quality")
    total = total_c = cap = ccap = 0
    overflow = False
    sent_records = 0
    for pz in range(P):
        # a2: local reads -> super-k-mer records grouped by bucket (hence by owner rank)
        if P > 1:
            recs, counts, send_rows = backend.partition(nb, world, nb // P * pz, nb // P * (pz + 1))
        else:
            recs, counts, send_rows = backend.partition(nb, world)
        sent_records += int(sum(send_rows))
        mark("partition")
        # k-mer instances this rank will own (bounds its solid set: S_local <= owned / min_freq)
        kp = torch.tensor(backend.kmers_per_part, dtype=torch.int64, device=dev)
        kp_recv = torch.empty_like(kp)
        _all_to_all(kp_recv, kp, group=group)
        owned_kmers = int(kp_recv.sum().item())
        # the k-mer shuffle: per-bucket record counts, then the records themselves
        recv_counts = torch.empty(world * nbl, dtype=torch.int32, device=dev)
        _all_to_all(recv_counts, counts, group=group)
        recv_rows = recv_counts.view(world, nbl).sum(dim=1, dtype=torch.int64).tolist()
        recv = _buffer("records", int(sum(recv_rows)) * recs.shape[1], dev).view(int(sum(recv_rows)), recs.shape[1])
        mark("counts")
        # a3-a5 on the owned buckets, in bucket slices [nbl*k//ns, nbl*(k+1)//ns).  A source's records are sorted by bucket, so a
        # slice is one row range per (source, owner): the records of slice k+1 are exchanged WHILE slice k is being counted, and while
        # slice k+1 is counted, slice k's solid k-mers (and their bucket chunks) are all-gathered and every rank inserts them into
        # its copy of the dictionary (on the library's side stream).
        # with several passes the solid arrays are sized in pass 0 for ALL passes: from pass 0's own share (buckets are hash-uniform, every
        # pass brings this owner about as many instances) with a quarter of head room -- not from a blanket "twice the mean": 20 B per entry
        if pz == 0:
            owned_bound = m_total if m_total < (1 << 24) else min(m_total, owned_kmers * P + owned_kmers * P // 4 + (1 << 20))
        backend.count_pass(pz, P)                                  # (every time, P == 1 included: the pass state never carries over from an earlier job)
        ns = backend.count_begin(min_freq, nbl, world, recv, recv_counts, owned_bound if P > 1 else owned_kmers, N_SLICES)
        mark("count_begin")
        bounds = backend.count_bounds(ns)                         # (the first slice is shorter: its exchange is the one nothing hides)

        def slice_offsets(cnt):                                   # [world][ns+1]: rows before each slice boundary, per owner / source
            c64 = cnt.view(world, nbl).to(torch.int64)
            cs = torch.cat([torch.zeros((world, 1), dtype=torch.int64, device=cnt.device), c64.cumsum(dim=1)], dim=1)
            return cs[:, bounds].tolist()
        s_off, r_off = slice_offsets(counts), slice_offsets(recv_counts)
        s_base = [int(sum(send_rows[:p])) for p in range(world)]
        r_base = [int(sum(recv_rows[:p])) for p in range(world)]
        row_bytes = recs.shape[1] * recs.element_size()
        piece = max([o[p][k + 1] - o[p][k] for o in (s_off, r_off) for p in range(world) for k in range(ns)] + [0])
        need = torch.tensor([piece * row_bytes], dtype=torch.int64, device=dev)
        _all_reduce(need, group=group, op=dist.ReduceOp.MAX)
        rounds = max(1, -(-int(need.item()) // A2A_MAX_PEER_BYTES))

        def exchange(k):
            _exchange_views([recv[r_base[p] + r_off[p][k]: r_base[p] + r_off[p][k + 1]] for p in range(world)],
                            [recs[s_base[p] + s_off[p][k]: s_base[p] + s_off[p][k + 1]] for p in range(world)], rounds, group,
                            getattr(backend, "copy_on_device", None))
        mark("offsets")
        exchange(0)
        backend.count_launch(0)
        mark("shuffle[0]")
        for k in range(ns):
            if k + 1 < ns:
                exchange(k + 1)
                backend.count_launch(k + 1)
            hi, lo, cc, cs, cn = backend.count_slice(k)
            if not gather:
                total += hi.numel()                               # (this owner's own k-mers; they stay where they are)
                if hasattr(backend, "local_dict_slice"):
                    if k == 0 and pz == 0:                        # the owner's own dictionary, laid out from the first slice's extrapolation
                        local_expected = int(hi.numel() * (nbl * P / max(bounds[1] - bounds[0], 1)) * DICT_HEADROOM) + 4096
                    backend.local_dict_slice(total, local_expected)
                continue
            if overflow:
                continue
            sizes = _all_gather_sizes([hi.numel(), cs.numel()], dev, group)              # [world][2], identical on every rank
            n_all, c_all = sum(x[0] for x in sizes), sum(x[1] for x in sizes)
            if k == 0 and pz == 0:
                # buckets are hash-uniform: the first slice predicts the whole (with head room); a wrong guess falls back below
                scale = nbl * P / max(bounds[1] - bounds[0], 1)   # the whole from the first slice's share of the buckets
                cap, ccap = int(n_all * scale * DICT_HEADROOM) + 4096, int(c_all * scale * DICT_HEADROOM) + 4096
                cap = min(cap, MAX_SOLID)                             # an ESTIMATE must not trip the limit the real count may respect
                backend.dict_begin(cap, ccap)
            if total + n_all > cap or total_c + c_all > ccap:
                overflow = True
                continue
            for blk in _all_gather_blocks(hi, lo, cc, cs, cn, sizes, group):
                backend.dict_append(*blk)
            total += n_all; total_c += c_all
        st = backend.count_end()
    mark("shuffle+count+gather")
    stats = torch.tensor([int(x) for x in st["hist"]] + [int(st["D"])], dtype=torch.int64, device=dev)
    _all_reduce(stats, group=group)
    hist = stats[:101].tolist()
    d_total = int(stats[101].item())
    if not gather:
        t = torch.tensor([total], dtype=torch.int64, device=dev)
        _all_reduce(t, group=group)
        s_total = int(t.item())
        if hasattr(backend, "_pass"):
            backend._pass = 0
    elif not overflow:
        backend.dict_end(m_total, d_total, hist)
        s_total = total
    else:
        # the classic way: gather every rank's whole solid set, then build the dictionary in one go.  This runs exactly when
        # S is larger than predicted, i.e. at the memory peak: the half-built dictionary and the gathered blocks go first.
        backend.dict_abort()
        hi, lo, cc = backend.solid()
        ghi, glo, gcc = _all_gather_v(hi, group), _all_gather_v(lo, group), _all_gather_v(cc, group)
        s_total = int(ghi.numel())
        if hasattr(backend, "chunks"):
            # the bucket chunks travel with them, renumbered: rank r's solid k-mers start at the sum of the earlier ranks' counts
            n_loc = torch.tensor([hi.numel()], dtype=torch.int64, device=dev)
            n_all = _all_gather_v(n_loc, group)
            my_base = int(n_all[:rank].sum().item())
            cs, cn = backend.chunks()
            gcs, gcn = _all_gather_v(cs + my_base, group), _all_gather_v(cn, group)
            backend.set_solid(ghi, glo, gcc, m_total, d_total, hist, gcs, gcn)
        else:
            backend.set_solid(ghi, glo, gcc, m_total, d_total, hist)
    mark("dictionary")
    if dev.type == "cuda":
        # The gathered blocks are dropped by now, but torch's caching allocator keeps their memory (a replica of BASELINE configs[2] gathers
        # 50 GB of solid k-mers) while the library -- its own pool, plain hipMalloc -- is about to build the graph on S k-mers: ~80 B per
        # solid k-mer at the peak.  Hand the cache back when that would not fit beside it.
        # (sharded graph, gather=False: what this rank OWNS decides, ~90 B per k-mer at the phase's peak; the exchange buffers of the counting
        # -- the received super-k-mer records, GBs -- are dead by now and go first)
        free_b, _ = torch.cuda.mem_get_info(dev)
        need = (90 * total if not gather else 80 * s_total) + (8 << 30)
        if free_b < need:
            for k in [k for k in _scratch if k[0] in ("records", "a2a_src", "a2a_dst", "dict_send") and k[1] == str(dev)]:
                _scratch.pop(k, None)
            recv = recs = None
            if hasattr(backend, "_keep"):
                backend._keep = None                              # (the count is over: nothing reads the received records any more)
            torch.cuda.empty_cache()
    if trace:
        import sys
        print("[w2rap] distributed_count: " + ", ".join(f"{b[0]} {(b[1] - a[1]) * 1e3:.1f} ms" for a, b in zip(marks, marks[1:])), file=sys.stderr)
    return dict(M=m_total, M_local=m_local, D=d_total, S=s_total, S_local=(total if not gather else None), fallback=overflow, hist=np.array(hist, dtype=np.uint64),
                n_buckets=nb, sent_records=sent_records, rank=rank, world=world, n_passes=P)


def distributed_repath(ctx, K2=200, group=None, edge_order_hint=None, fetch=True, extend_paths=False):
    """Step 3 with the reads sharded by rank (SURVEY.md 8e applied to row N1): the small-K graph is replicated (as distributed Step 2 leaves
    it) and every rank holds the paths of ITS reads.  The large-K graph depends on the reads only through the set of unique places, so:
    every rank reduces its paths to one path per unique place (w2rap_step3 PLACES_ONLY), these few paths are all-gathered -- the one
    exchange step, a few MB -- and every rank builds the same large-K graph from the union and translates its own reads.
    extend_paths (Repath.cc:72-96) acts on the union of the places, i.e. in the second call of every rank, never on a rank's own list.
    -> step3.Step3Result for this rank's reads (graph identical on every rank)."""
    from . import step3
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return step3.repath_after_step2(ctx, K2, edge_order_hint, fetch, extend_paths=extend_paths)
    mine = step3.repath_after_step2(ctx, K2, places_only=True).place_paths
    dev = torch.device("cuda", torch.cuda.current_device()) if (torch.cuda.is_available() and not _host_staged(group)) else torch.device("cpu")
    lens = torch.from_numpy(np.diff(mine[0].astype(np.int64))).to(dev)
    edges = torch.from_numpy(mine[1].astype(np.int32)).to(dev)
    counts = _all_gather_v(torch.tensor([lens.numel()], dtype=torch.int64, device=dev), group).tolist()
    all_lens = _all_gather_v(lens, group).cpu().numpy()
    all_edges = _all_gather_v(edges, group).cpu().numpy()
    # the other ranks' places (this rank's own reads are there anyway)
    lo = int(sum(counts[:rank])); hi = lo + counts[rank]
    e_off = np.concatenate([[0], np.cumsum(all_lens)]).astype(np.int64)
    keep_lens = np.concatenate([all_lens[:lo], all_lens[hi:]])
    keep_edges = np.concatenate([all_edges[:e_off[lo]], all_edges[e_off[hi]:]])
    off = np.concatenate([[0], np.cumsum(keep_lens)]).astype(np.uint64)
    res = step3.repath_after_step2(ctx, K2, edge_order_hint, fetch, extra_paths=(off, keep_edges.astype(np.int32)), extend_paths=extend_paths)
    # FragDist (GapToyTools3.cc:616-646) pairs reads 2i and 2i+1 INSIDE the shard and the pathed counters cover this rank's reads only:
    # the job-wide .first.frags.dist and counters are the sums over the ranks.  A shard must therefore hold whole pairs.
    n_local = len(res.path_offset) if res.path_offset is not None else None
    even = torch.tensor([0 if (n_local is None or n_local % 2 == 0 or rank == world - 1) else 1], dtype=torch.int64, device=dev)
    _all_reduce(even, group=group)
    if int(even.item()):
        from .step2 import Step2Error
        raise Step2Error(1, "distributed_repath: every rank but the last must hold an even number of reads (mates 2i, 2i+1 are paired inside a shard)")
    if res.frag_count is not None:
        tot = torch.from_numpy(np.concatenate([res.frag_count.astype(np.int64), [res.n_reads_pathed, res.n_reads_multipathed]])).to(dev)
        _all_reduce(tot, group=group)
        tot = tot.cpu().numpy()
        res.frag_count_local = res.frag_count
        res.frag_count = tot[:-2].astype(res.frag_count.dtype)
        res.n_reads_pathed, res.n_reads_multipathed = int(tot[-2]), int(tot[-1])
    return res


X_DONE, X_ALLTOALL, X_ALLGATHER, X_ALLGATHER_HOST, X_ALLREDUCE_U8, X_ALLREDUCE_U32 = range(6)     # w2rap_xchg.op (include/w2rap_step2.h)


def sharded_graph(backend, solid_local, stats, n_buckets, n_passes=1, group=None, edge_order_hint=None):
    """Row e-3: the dictionary, the adjacency prune and the unipath phase stay with the owners of the k-mers (what distributed_count
    with gather=False left in every rank's backend); this drives the library's state machine and performs the exchanges it asks for --
    three query / response all-to-alls (neighbour membership, neighbour contexts, segment numbers), the segment level (all-gather of one
    8-byte word per chain segment, ~4 % of the k-mers; all-gather of the splitters' and heads' 32-byte records; all-to-all of the walks'
    results), two all-reduces (middle bases, the packed edge stream), all-gathers of the index entries and of the filter words.  Afterwards every rank holds the same graph
    (as after build_graph) and the pathing index; path_reads then paths this rank's reads.  -> job-wide solid k-mers.

    ONE small all-gather per exchange carries everything the ranks must agree on: the operation (a rank whose library call raised says
    X_FAILED, and then EVERY rank raises right behind that collective -- nobody is left waiting in an exchange for a rank that has gone;
    the in-process path does the same with its failing barrier), the element size and every rank's send counts, from which each rank
    reads what it will receive and in how many rounds the pieces travel."""
    import sys, time
    from .step2 import Step2Error
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = backend.device
    spr = [int(x[0]) for x in _all_gather_sizes([int(solid_local)], dev, group)]
    backend.shard_begin(rank, world, spr, n_buckets, n_passes, stats["M"], stats["D"], stats["hist"], edge_order_hint)
    n_x = 0
    trace = os.environ.get("W2RAP_TRACE_SHARD") is not None
    times, log = [], []                                       # wall time of every exchange, ms, and what it moved (bench.py --gpus N: "exchange_ms")
    X_FAILED = -1
    failure = None                                            # this rank's own library error, raised behind the next agreement
    while True:
        op, elem, send, cnt = X_DONE, 0, 0, [0] * world
        if failure is None:
            try:
                op, elem, send, cnt = backend.shard_next()
            except Step2Error as e:
                failure = e
        t_x = time.perf_counter()
        mine = [X_FAILED if failure is not None else op, elem] + [int(c) for c in cnt[:world]]
        head = _all_gather_sizes(mine, dev, group) if world > 1 else [mine]          # (a rank alone has nobody to agree with)
        ops = [h[0] for h in head]
        if failure is not None:
            raise failure
        if X_FAILED in ops:
            raise Step2Error(4, f"sharded graph: rank(s) {[r for r, o in enumerate(ops) if o == X_FAILED]} failed; this rank stops with them")
        if any(o != op for o in ops) or any(h[1] != elem for h in head):
            raise Step2Error(4, f"sharded graph: the ranks disagree about the next exchange ({ops})")
        if op == X_DONE:
            break
        n_x += 1
        if op in (X_ALLTOALL, X_ALLGATHER):
            if op == X_ALLTOALL:
                sc = [int(c) for c in cnt[:world]]
                rc = [int(head[r][2 + rank]) for r in range(world)]
                biggest = max(max(h[2:2 + world]) for h in head)
            else:
                sc = None
                rc = [int(head[r][2]) for r in range(world)]
                biggest = max(rc)
            try:
                out = backend.shard_recv(rc, elem)
            except Step2Error as e:
                # the exchange still has to be performed -- the others are already on their way into it --, into a throw-away buffer;
                # the error is reported at the next agreement
                failure = e
                out = torch.empty(int(sum(rc)) * elem, dtype=torch.uint8, device=dev)
            rounds = max(1, -(-(biggest * elem) // A2A_MAX_PEER_BYTES))
            io = [0]
            for v in rc: io.append(io[-1] + v * elem)
            outs = [out[io[p]:io[p + 1]].view(-1, elem) for p in range(world)]
            if op == X_ALLTOALL:
                inp = backend.shard_view(send, sum(sc) * elem)
                so = [0]
                for v in sc: so.append(so[-1] + v * elem)
                ins = [inp[so[p]:so[p + 1]].view(-1, elem) for p in range(world)]
            else:
                # an all-gather written as an all-to-all in which everybody sends the same block to everyone
                inp = backend.shard_view(send, int(cnt[0]) * elem)
                ins = [inp.view(-1, elem) for _ in range(world)]
            _exchange_views(outs, ins, rounds, group, getattr(backend, "copy_on_device", None))
        elif op == X_ALLGATHER_HOST:
            w = backend.shard_host_word(send)
            backend.shard_host_words([int(x[0]) for x in _all_gather_sizes([w], dev, group)])
        elif op in (X_ALLREDUCE_U8, X_ALLREDUCE_U32):
            n = int(cnt[0])
            buf = backend.shard_view(send, n * elem)
            t = buf.view(torch.uint8 if op == X_ALLREDUCE_U8 else torch.int32)
            # (pieces of at most 1 GiB: the edge stream of a large genome is several GB; sums of disjoint bit groups never carry)
            step = (1 << 30) // elem
            for a in range(0, n, step):
                _all_reduce(t[a:a + step], group=group)
        else:
            raise Step2Error(4, f"sharded graph: unknown exchange {op}")
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        times.append(round((time.perf_counter() - t_x) * 1e3, 3))
        moved = (int(sum(cnt[:world])) if op == X_ALLTOALL else int(cnt[0]) if op != X_ALLGATHER_HOST else 1) * elem
        log.append({"op": ["done", "all_to_all", "all_gather", "all_gather_host", "all_reduce_u8", "all_reduce_u32"][op], "elem_bytes": elem, "bytes_from_this_rank": moved,
                    "ms": times[-1]})
        if trace:
            print(f"[w2rap] exchange {n_x} (op {op}, {elem} B elements): {times[-1]:.2f} ms", file=sys.stderr)
    return dict(solid_total=sum(spr), solid_per_rank=spr, exchanges=n_x, exchange_ms=times, exchange_log=log)
