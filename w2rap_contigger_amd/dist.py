"""Multi-GPU Step 2: one process per GPU, torch.distributed (RCCL over xGMI).

Reads are sharded by rank.  K-mer counting has ONE real exchange step, the k-mer shuffle:
every rank cuts its reads into super-k-mer records, bucketed by canonical minimizer; bucket
b belongs to rank b // (n_buckets / world); records travel to their owner with an
all_to_all_v (the GPU counterpart of MapReduceEngine's swizzle, src/MapReduceEngine.h:320-361),
owners count their buckets, and the solid k-mers (<= M / min_freq of them) are all-gathered so
every rank holds the whole dictionary.  Graph construction is then replicated (it is a small
fraction of Step 2) and read pathing is embarrassingly parallel over the local reads.

The orchestration is written against a small backend interface so that the same code runs on
the HIP library (`GpuBackend`) and, in the CPU tests, on a numpy stand-in over gloo.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

REC_BYTES = 36


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

    def quality_windows(self, min_qual):
        return self.ctx.quality_windows(min_qual)

    def default_buckets(self, total_kmers, world):
        return self.ctx.default_buckets(total_kmers, world)

    def partition(self, n_buckets, world):
        recs, nrec, cnts, per = self.ctx.partition(n_buckets, world)
        r = dev_bytes(recs, nrec * REC_BYTES, self.device).view(nrec, REC_BYTES)
        c = dev_bytes(cnts, n_buckets * 4, self.device).view(torch.int32)
        self.kmers_per_part = self.ctx.kmers_per_part
        return r, c, per

    def count_records(self, min_freq, nbl, nseg, records, counts, total_kmers):
        torch.cuda.synchronize(self.device)
        self._keep = (records, counts)
        return self.ctx.count_records(min_freq, nbl, nseg, records.data_ptr(), counts.data_ptr(), total_kmers)

    def solid(self):
        hi, lo, cc, n = self.ctx.solid_buffers()
        return (dev_bytes(hi, n * 8, self.device).view(torch.int64), dev_bytes(lo, n * 8, self.device).view(torch.int64),
                dev_bytes(cc, n * 4, self.device).view(torch.int32))

    def set_solid(self, hi, lo, cc, M, D, hist):
        torch.cuda.synchronize(self.device)
        self.ctx.set_solid(hi.data_ptr(), lo.data_ptr(), cc.data_ptr(), hi.numel(), M, D, hist)

    def sync(self):
        torch.cuda.synchronize(self.device)


def _host_staged(group) -> bool:
    """gloo has no device all_to_all / all_gather_into_tensor: with that backend (the two-ranks-on-one-GPU test) device
    tensors make the trip through host memory.  RCCL ("nccl") exchanges them in place over xGMI."""
    return dist.get_backend(group) == "gloo"


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None, group=None):
    if _host_staged(group) and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def _all_reduce(t: torch.Tensor, group=None):
    if _host_staged(group) and t.is_cuda:
        c = t.cpu(); dist.all_reduce(c, group=group); t.copy_(c)
    else:
        dist.all_reduce(t, group=group)


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


def distributed_count(backend, min_qual=7, min_freq=4, group=None):
    """The sharded a1-a6: returns job-wide statistics; afterwards every rank's backend holds the
    complete solid-k-mer dictionary (as after count_kmers on one GPU)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = backend.device
    # a1 on the local reads; agree on the bucket count from the job-wide number of k-mer instances
    m_local = backend.quality_windows(min_qual)
    m = torch.tensor([m_local], dtype=torch.int64, device=dev)
    _all_reduce(m, group=group)
    m_total = int(m.item())
    nb = backend.default_buckets(m_total, world)
    nbl = nb // world
    # a2: local reads -> super-k-mer records grouped by bucket (hence by owner rank)
    recs, counts, send_rows = backend.partition(nb, world)
    # k-mer instances this rank will own (bounds its solid set: S_local <= owned / min_freq)
    kp = torch.tensor(backend.kmers_per_part, dtype=torch.int64, device=dev)
    kp_recv = torch.empty_like(kp)
    _all_to_all(kp_recv, kp, group=group)
    owned_kmers = int(kp_recv.sum().item())
    # the k-mer shuffle: per-bucket record counts, then the records themselves
    recv_counts = torch.empty(world * nbl, dtype=torch.int32, device=dev)
    _all_to_all(recv_counts, counts, group=group)
    recv_rows = recv_counts.view(world, nbl).sum(dim=1, dtype=torch.int64).tolist()
    recv = torch.empty((int(sum(recv_rows)), recs.shape[1]), dtype=torch.uint8, device=dev)
    _all_to_all(recv, recs, [int(x) for x in recv_rows], [int(x) for x in send_rows], group=group)
    # a3-a5 on the owned buckets
    st = backend.count_records(min_freq, nbl, world, recv, recv_counts, owned_kmers)
    stats = torch.tensor([int(x) for x in st["hist"]] + [int(st["D"])], dtype=torch.int64, device=dev)
    _all_reduce(stats, group=group)
    hist = stats[:101].tolist()
    d_total = int(stats[101].item())
    # every rank gets the whole solid dictionary
    hi, lo, cc = backend.solid()
    ghi, glo, gcc = _all_gather_v(hi, group), _all_gather_v(lo, group), _all_gather_v(cc, group)
    backend.set_solid(ghi, glo, gcc, m_total, d_total, hist)
    return dict(M=m_total, M_local=m_local, D=d_total, S=int(ghi.numel()), hist=np.array(hist, dtype=np.uint64),
                n_buckets=nb, sent_records=int(sum(send_rows)), rank=rank, world=world)
