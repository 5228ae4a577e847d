"""The N>1 host logic (w2rap_contigger_amd/dist.py) on CPU: world_size 2 over gloo with a numpy
stand-in for the library steps.  Checks bucket ownership, the all_to_all_v splits and the
all_gather_v of the solid dictionary against the oracle's k-mer table of the UNSHARDED reads."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from conftest import load_fixture, run_ranks
from oracle import oracle as O

M60 = (1 << 60) - 1


def kmer_rows(codes, off, good_len):
    """all canonical 60-mer instances of the reads with their context byte -> int64 [n, 3] (hi, lo, ctx)"""
    starts, ctxs = [], []
    off = off.astype(np.int64)
    for r in range(len(off) - 1):
        gl = int(good_len[r])
        if gl <= 60:
            continue
        p = np.arange(0, gl - 59)
        b = codes[off[r]:off[r + 1]].astype(np.int64)
        c = np.zeros(len(p), np.int64)
        c[1:] |= 1 << (4 + b[p[1:] - 1])
        c[:-1] |= 1 << b[p[:-1] + 60]
        starts.append(off[r] + p); ctxs.append(c)
    if not starts:
        return np.zeros((0, 3), np.int64)
    s = np.concatenate(starts); ctx = np.concatenate(ctxs)
    c64 = codes.astype(np.uint64)
    hi = np.zeros(len(s), np.uint64); lo = np.zeros(len(s), np.uint64)
    rhi = np.zeros(len(s), np.uint64); rlo = np.zeros(len(s), np.uint64)
    for j in range(30):
        hi = (hi << np.uint64(2)) | c64[s + j]
        lo = (lo << np.uint64(2)) | c64[s + 30 + j]
        rhi = (rhi << np.uint64(2)) | (np.uint64(3) - c64[s + 59 - j])
        rlo = (rlo << np.uint64(2)) | (np.uint64(3) - c64[s + 29 - j])
    rev = (rhi < hi) | ((rhi == hi) & (rlo < lo))
    brev = np.array([int(f"{x:08b}"[::-1], 2) for x in range(256)], np.int64)
    hi = np.where(rev, rhi, hi); lo = np.where(rev, rlo, lo); ctx = np.where(rev, brev[ctx], ctx)
    return np.stack([hi.astype(np.int64), lo.astype(np.int64), ctx], axis=1)


class NumpyBackend:
    """CPU stand-in with the GpuBackend interface; a 'record' is one k-mer instance (24 B row)."""

    def __init__(self, codes, quals, off):
        self.codes, self.quals, self.off = codes, quals, off
        self.device = torch.device("cpu")

    def quality_windows(self, min_qual):
        r = O.run(self.codes, self.quals, self.off, min_qual=min_qual, stop_after=1)
        self.rows = kmer_rows(self.codes, self.off, r.good_len)
        assert len(self.rows) == r.n_instances
        return len(self.rows)

    def default_buckets(self, total, world):
        nb = total // 2_000 + 1
        return (nb + world - 1) // world * world

    def partition(self, nb, world, first_bucket=0, end_bucket=None):
        end_bucket = nb if end_bucket is None else end_bucket
        h = (self.rows[:, 0].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) ^ self.rows[:, 1].astype(np.uint64)) >> np.uint64(17)
        b = (h % np.uint64(nb)).astype(np.int64)
        mine = (b >= first_bucket) & (b < end_bucket)             # a hash-range pass keeps its own bucket range only
        rows, b = self.rows[mine], b[mine] - first_bucket
        order = np.argsort(b, kind="stable")
        counts = np.bincount(b, minlength=end_bucket - first_bucket).astype(np.int32)
        nbl = (end_bucket - first_bucket) // world
        per = [int(counts[g * nbl:(g + 1) * nbl].sum()) for g in range(world)]
        rec = torch.from_numpy(np.ascontiguousarray(rows[order])).view(torch.uint8).view(len(order), 24)
        self.kmers_per_part = per                       # one k-mer instance per record here
        return rec, torch.from_numpy(counts), per

    def count_records(self, min_freq, nbl, nseg, records, counts, total_kmers):
        assert counts.numel() == nbl * nseg and records.shape[0] == int(counts.sum()) == total_kmers
        rows = records.contiguous().view(torch.int64).view(-1, 3).numpy()
        hist = np.zeros(101, np.uint64)
        if len(rows):
            key = np.ascontiguousarray(rows[:, :2]).view([("h", np.int64), ("l", np.int64)]).reshape(-1)
            uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
            ctx = np.zeros(len(uniq), np.int64)
            np.bitwise_or.at(ctx, inv, rows[:, 2])
            cnt = np.minimum(cnt, 255)
            np.add.at(hist, np.minimum(cnt, 100), 1)
            keep = cnt >= min_freq
            self.s_hi = torch.from_numpy(uniq["h"][keep].copy()); self.s_lo = torch.from_numpy(uniq["l"][keep].copy())
            self.s_cc = torch.from_numpy((cnt[keep] | (ctx[keep] << 8)).astype(np.int32))
            D = len(uniq)
        else:
            self.s_hi = self.s_lo = torch.zeros(0, dtype=torch.int64); self.s_cc = torch.zeros(0, dtype=torch.int32); D = 0
        return dict(hist=hist, D=D, S=int(self.s_hi.numel()))

    # the sliced interface: the owner-side count goes bucket slice by bucket slice -- slice k is counted by count_launch(k), when
    # only the records of ITS buckets have arrived (the later rows of `records` are still unwritten) --, the dictionary is
    # assembled from the gathered pieces (no chunk lists in this stand-in)
    def count_pass(self, k, n_passes):
        self._pass = k

    def count_begin(self, min_freq, nbl, nseg, records, counts, total_kmers, n_slices):
        self._args = (min_freq, nbl, nseg, records, counts.view(nseg, nbl).numpy().astype(np.int64))
        if not getattr(self, "_pass", 0):                         # a later pass goes on behind the earlier ones
            self._hist = np.zeros(101, np.uint64); self._D = 0
            self._done = []
            self._dict = None
        self._slices = []
        self._ns = n_slices
        return n_slices

    def count_bounds(self, ns):
        nbl = self._args[1]
        return [0] + [nbl * (2 * j - 1) // (2 * ns - 1) for j in range(1, ns + 1)]      # a short first slice, like the library

    def count_launch(self, k):
        min_freq, nbl, nseg, records, cnt = self._args
        assert k == len(self._slices)
        lo_b, hi_b = self.count_bounds(self._ns)[k:k + 2]
        seg_base = np.concatenate([[0], np.cumsum(cnt.sum(axis=1))])
        rows = []
        for s in range(nseg):
            a = seg_base[s] + cnt[s, :lo_b].sum(); b = a + cnt[s, lo_b:hi_b].sum()
            rows.append(records[int(a):int(b)].contiguous().view(torch.int64).view(-1, 3).numpy())
        rows = np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)
        if len(rows):
            key = np.ascontiguousarray(rows[:, :2]).view([("h", np.int64), ("l", np.int64)]).reshape(-1)
            uniq, inv, c = np.unique(key, return_inverse=True, return_counts=True)
            ctx = np.zeros(len(uniq), np.int64)
            np.bitwise_or.at(ctx, inv, rows[:, 2])
            c = np.minimum(c, 255)
            np.add.at(self._hist, np.minimum(c, 100), 1)
            keep = c >= min_freq
            self._D += len(uniq)
            self._slices.append((torch.from_numpy(uniq["h"][keep].copy()), torch.from_numpy(uniq["l"][keep].copy()),
                                 torch.from_numpy((c[keep] | (ctx[keep] << 8)).astype(np.int32))))
        else:
            self._slices.append((torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32)))

    def count_slice(self, k):
        e64, e32 = torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int32)
        return self._slices[k] + (e64, e32)

    def count_end(self):
        assert len(self._slices) == self._ns
        self._done += self._slices
        self.s_hi = torch.cat([x[0] for x in self._done]); self.s_lo = torch.cat([x[1] for x in self._done])
        self.s_cc = torch.cat([x[2] for x in self._done])
        return dict(hist=self._hist, D=self._D, S=int(self.s_hi.numel()))

    def dict_begin(self, kmer_cap, chunk_cap):
        self._dict = ([], [], [], kmer_cap)

    def dict_append(self, hi, lo, cc, cs, cn):
        assert sum(x.numel() for x in self._dict[0]) + hi.numel() <= self._dict[3]
        self._dict[0].append(hi.clone()); self._dict[1].append(lo.clone()); self._dict[2].append(cc.clone())

    def dict_end(self, M, D, hist):
        self.set_solid(torch.cat(self._dict[0]), torch.cat(self._dict[1]), torch.cat(self._dict[2]), M, D, hist)
        self.sliced = True

    def dict_abort(self):
        self._dict = None
        self.aborted = True

    def solid(self):
        return self.s_hi, self.s_lo, self.s_cc

    def set_solid(self, hi, lo, cc, M, D, hist):
        self.final = (hi.numpy().copy(), lo.numpy().copy(), cc.numpy().copy(), M, D, list(hist))


def _worker(rank, world, port, name, a2a_max, headroom, q, n_passes=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from w2rap_contigger_amd import dist as wd
        if a2a_max:
            wd.A2A_MAX_BYTES = wd.A2A_MAX_PEER_BYTES = a2a_max      # force the record exchange into many rounds
        if headroom:
            wd.DICT_HEADROOM = headroom                 # < 1: the dictionary's capacity guess fails -> classic gather + set_solid
        fx = load_fixture(name)
        n = len(fx["read_len"])
        cut = (n // world // 2) * 2
        lo_r, hi_r = rank * cut, (n if rank == world - 1 else (rank + 1) * cut)      # whole pairs per rank
        off = fx["off"].astype(np.int64)
        codes = fx["codes"][off[lo_r]:off[hi_r]]
        quals = fx["quals"][off[lo_r]:off[hi_r]]
        be = NumpyBackend(codes, quals, (off[lo_r:hi_r + 1] - off[lo_r]).astype(np.uint64))
        st = wd.distributed_count(be, 7, 4, n_passes=n_passes)
        q.put((rank, st["M"], st["D"], st["S"], st["hist"].tolist(), be.final[0], be.final[1], be.final[2], bool(st["fallback"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,a2a_max,headroom,world", [("random20k", None, None, 2), ("repeats_snps", None, None, 2), ("random20k", 1 << 18, None, 2),
                                                         ("repeats_snps", None, 0.3, 2), ("random20k", None, None, 3), ("repeats_snps", 1 << 18, None, 4)])
def test_two_rank_shuffle_reproduces_the_kmer_table(name, a2a_max, headroom, world):
    """a2a_max: the exchange is cut into rounds (RCCL returns garbage for multi-GiB all_to_all_single calls, see dist.py);
    headroom < 1: the sliced dictionary build runs out of its reserved capacity and the classic gather takes over"""
    outs = run_ranks(_worker, world, (name, a2a_max, headroom), timeout=180)
    fx = load_fixture(name)
    orc = O.run(fx["codes"], fx["quals"], fx["off"], stop_after=1)
    for rank, M, D, S, hist, hi, lo, cc, fallback in outs:
        assert fallback == (headroom is not None)
        assert M == orc.n_instances and D == orc.n_distinct and S == len(orc.k_hi)
        assert hist == [int(x) for x in orc.hist]
        order = np.lexsort((lo.astype(np.uint64), hi.astype(np.uint64)))
        assert np.array_equal(hi.astype(np.uint64)[order], orc.k_hi) and np.array_equal(lo.astype(np.uint64)[order], orc.k_lo)
        assert np.array_equal((cc[order] & 0xFF).astype(np.uint8), orc.k_count)
        assert np.array_equal(((cc[order] >> 8) & 0xFF).astype(np.uint8), orc.k_ctx)
    # all ranks hold the identical dictionary, in the identical order
    for o in outs[1:]:
        assert np.array_equal(outs[0][5], o[5]) and np.array_equal(outs[0][7], o[7])


def _worker_passes(rank, world, port, name, a2a_max, headroom, n_passes, q):
    _worker(rank, world, port, name, a2a_max, headroom, q, n_passes)


@pytest.mark.parametrize("name,world,passes,headroom", [("random20k", 2, 3, None), ("repeats_snps", 3, 2, None), ("palindrome_circle", 2, 5, None), ("repeats_snps", 2, 3, 0.3)])
def test_hash_range_passes_with_several_ranks(name, world, passes, headroom):
    """counting in hash-range passes (MapReduceEngine.h:286-299) TOGETHER with bucket owners: every pass cuts the reads again, keeps one part of
    the bucket range, the owners divide that part; the dictionary is assembled across the passes -- the same table as in one pass"""
    outs = run_ranks(_worker_passes, world, (name, None, headroom, passes), timeout=240)
    fx = load_fixture(name)
    orc = O.run(fx["codes"], fx["quals"], fx["off"], stop_after=1)
    for rank, M, D, S, hist, hi, lo, cc, fallback in outs:
        assert fallback == (headroom is not None)
        assert M == orc.n_instances and D == orc.n_distinct and S == len(orc.k_hi)
        assert hist == [int(x) for x in orc.hist]
        order = np.lexsort((lo.astype(np.uint64), hi.astype(np.uint64)))
        assert np.array_equal(hi.astype(np.uint64)[order], orc.k_hi) and np.array_equal(lo.astype(np.uint64)[order], orc.k_lo)
        assert np.array_equal((cc[order] & 0xFF).astype(np.uint8), orc.k_count)
        assert np.array_equal(((cc[order] >> 8) & 0xFF).astype(np.uint8), orc.k_ctx)
    for o in outs[1:]:
        assert np.array_equal(outs[0][5], o[5]) and np.array_equal(outs[0][7], o[7])


# ---- row e-3: the host layer of the sharded graph phase (dist.sharded_graph) -- the loop that performs whatever exchange the library's
# state machine asks for.  The kernels behind the state machine need the GPU (tests/test_gpu_sharded.py, test_gpu_two_ranks.py); here a
# stand-in asks for one exchange of every kind with recognisable contents and checks what arrives, at world 2, 3 and 4 over gloo.
class ScriptedShard:
    def __init__(self):
        self.device = torch.device("cpu")
        self.step = 0
        self.seen = []

    def shard_begin(self, rank, world, solid_per_rank, n_buckets, n_passes, M, D, hist, hint=None):
        self.rank, self.world, self.spr = rank, world, list(solid_per_rank)

    def shard_view(self, handle, nbytes):
        return handle.view(torch.uint8).view(-1)[:nbytes]

    def shard_host_word(self, handle):
        return int(handle)

    def shard_recv(self, counts, elem):
        self.counts = list(counts)
        self.buf = torch.full((int(sum(counts)) * elem,), 255, dtype=torch.uint8)
        return self.buf

    def shard_host_words(self, words):
        self.words = list(words)

    def shard_next(self):
        r, w = self.rank, self.world
        k = self.step
        self.step += 1
        cnt = [0] * 64
        if k == 0:                                   # all-to-all of 16-byte elements: rank r sends r + p + 1 elements of value 16 r + p to rank p
            for p in range(w): cnt[p] = r + p + 1
            self.send = torch.cat([torch.full(((r + p + 1) * 16,), 16 * r + p, dtype=torch.uint8) for p in range(w)])
            return dist_mod.X_ALLTOALL, 16, self.send, cnt
        if k == 1:
            assert self.counts == [p + r + 1 for p in range(w)]
            o = 0
            for p in range(w):
                n = (p + r + 1) * 16
                assert bool((self.buf[o:o + n] == 16 * p + r).all()); o += n
            cnt[0] = r + 2                             # all-gather of 24-byte elements, r + 2 of them from rank r
            self.send = torch.full(((r + 2) * 24,), 100 + r, dtype=torch.uint8)
            return dist_mod.X_ALLGATHER, 24, self.send, cnt
        if k == 2:
            assert self.counts == [p + 2 for p in range(w)]
            o = 0
            for p in range(w):
                n = (p + 2) * 24
                assert bool((self.buf[o:o + n] == 100 + p).all()); o += n
            cnt[0] = 1
            return dist_mod.X_ALLGATHER_HOST, 8, 1000 + r, cnt
        if k == 3:
            assert self.words == [1000 + p for p in range(w)]
            cnt[0] = 4096
            self.send = torch.ones(4096, dtype=torch.uint8)
            return dist_mod.X_ALLREDUCE_U8, 1, self.send, cnt
        if k == 4:
            assert bool((self.send == w).all())
            cnt[0] = 5000
            self.send = (torch.arange(5000, dtype=torch.int32) * (r + 1))
            return dist_mod.X_ALLREDUCE_U32, 4, self.send, cnt
        if k == 5:
            assert bool((self.send == torch.arange(5000, dtype=torch.int32) * (w * (w + 1) // 2)).all())
        return dist_mod.X_DONE, 0, None, cnt


def _worker_shard(rank, world, port, q):
    global dist_mod
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from w2rap_contigger_amd import dist as wd
        dist_mod = wd
        be = ScriptedShard()
        info = wd.sharded_graph(be, 10 + rank, dict(M=1, D=1, hist=[0] * 101), 64 * world)
        q.put((rank, be.spr, info["solid_total"], info["exchanges"], be.step))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_graph_host_layer_performs_every_kind_of_exchange(world):
    outs = run_ranks(_worker_shard, world, (), timeout=120)
    for rank, spr, total, n_x, steps in outs:
        assert spr == [10 + p for p in range(world)] and total == sum(spr)
        assert n_x == 5 and steps == 6


class FailingShard(ScriptedShard):
    """rank 1's library call raises before the third exchange (shard_next) or inside the second one (shard_recv)"""
    def __init__(self, where):
        super().__init__()
        self.where = where

    def shard_next(self):
        from w2rap_contigger_amd.step2 import Step2Error
        if self.where == "next" and self.rank == 1 and self.step == 2:
            raise Step2Error(5, "injected: list overflow on rank 1")
        return super().shard_next()

    def shard_recv(self, counts, elem):
        from w2rap_contigger_amd.step2 import Step2Error
        if self.where == "recv" and self.rank == 1 and self.step == 2:
            raise Step2Error(3, "injected: pool out of memory on rank 1")
        return super().shard_recv(counts, elem)


def _worker_shard_fail(rank, world, port, where, q):
    global dist_mod
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from w2rap_contigger_amd import dist as wd
        from w2rap_contigger_amd.step2 import Step2Error
        dist_mod = wd
        be = FailingShard(where)
        try:
            wd.sharded_graph(be, 10 + rank, dict(M=1, D=1, hist=[0] * 101), 64 * world)
            q.put((rank, "no error", be.step))
        except Step2Error as e:
            q.put((rank, str(e), be.step))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("where", ["next", "recv"])
def test_sharded_graph_host_layer_every_rank_stops_when_one_fails(where):
    """a rank whose shard_next / shard_recv raises must not leave the others blocked in the next collective: the error
    travels in the per-exchange agreement and every rank raises behind it"""
    outs = dict((r, (m, st)) for r, m, st in run_ranks(_worker_shard_fail, 3, (where,), timeout=120))
    assert "injected" in outs[1][0]
    for r in (0, 2):
        assert "rank(s) [1] failed" in outs[r][0], outs
    assert outs[0][1] == outs[2][1]                            # the survivors stopped at the same point of the state machine


def test_exchange_with_oneself_is_a_copy():
    """world 1 (the forced-distributed runs that feed scale_model.py): nothing travels -- _exchange_views copies, whatever the round count"""
    import torch
    from w2rap_contigger_amd import dist as wd
    inp = torch.arange(48, dtype=torch.uint8).view(-1, 8)
    out = torch.zeros_like(inp)
    wd._exchange_views([out], [inp], 3, None)
    assert torch.equal(out, inp)
