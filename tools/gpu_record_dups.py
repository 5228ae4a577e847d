#!/usr/bin/env python3
"""How many super-k-mer records are byte-identical to another one of their bucket, weighted by the k-mers they carry
(python3 tools/gpu_record_dups.py [reads]): the share of K3's k-mer work a record-level deduplication would save.  Second figure: the same
with a record and its reverse complement (flanks swapped and complemented) counted as one -- reads of both strands over one locus."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from w2rap_contigger_amd import step2, synth
from w2rap_contigger_amd.dist import dev_bytes
dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
g = torch.randint(0, 4, (n * 5,), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(42))
d = synth.generate_reads_device(n, n * 5, 42, device=dev, genome=g); del g; d.pop("genome", None)
torch.cuda.synchronize(); torch.cuda.empty_cache()
with step2.Step2Context(0) as ctx:
    ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(),
                         d["qual_off"].data_ptr(), keepalive=d)
    M = ctx.quality_windows(7)
    nb = ctx.default_buckets(M, 1)
    recs, nrec, cnts, per = ctx.partition(nb, 1)
    rb = int(ctx.L.w2rap_step2_record_bytes()); dw = rb // 4          # 32 B = 8 dwords (36 B / 9 in a -DW2RAP_REC36 build)
    r = dev_bytes(recs, nrec * rb, dev).view(nrec, dw, 4).view(torch.int32).view(nrec, dw).to(torch.int64)
    nk = (r[:, 0] & 63) + 1
    mult = torch.tensor([0x9E3779B97F4A7C15 - (1 << 64), 0xC2B2AE3D27D4EB4F - (1 << 64), 0x165667B19E3779F9, 0x27D4EB2F165667C5, 0x85EBCA77C2B2AE63 - (1 << 64),
                         0x2545F4914F6CDD1D, 0x9FB21C651E98DF25 - (1 << 64), 0xD6E8FEB86659FD93 - (1 << 64), 0x369DEA0F31A53F85][:dw], dtype=torch.int64, device=dev)
    h = ((r & 0xFFFFFFFF) * mult).sum(dim=1)          # wrap-around 64-bit hash of the record bytes (buckets are implied by the content)
    h ^= h >> 29
    del r
    u, inv = torch.unique(h, return_inverse=True)
    first = torch.zeros_like(u)
    first.scatter_reduce_(0, inv, nk, reduce="amax")   # identical records carry identical nk
    print(f"records {nrec:,}  distinct {u.numel():,}  ({u.numel() / nrec:.3f})")
    print(f"k-mer instances {int(nk.sum()):,}  after record dedup {int(first.sum()):,}  ({int(first.sum()) / int(nk.sum()):.3f})")

    # ---- a record and its reverse complement as one: on the records of the first buckets (a sample of whole buckets)
    ns = min(nrec, 6_000_000)
    r = dev_bytes(recs, ns * rb, dev).view(ns, dw, 4).view(torch.int32).view(ns, dw).to(torch.int64) & 0xFFFFFFFF
    nk = (r[:, 0] & 63) + 1; hasL = (r[:, 0] >> 6) & 1; hasR = (r[:, 0] >> 7) & 1
    sh = torch.arange(16, device=dev, dtype=torch.int64) * 2
    bases = ((r[:, 1:9, None] >> sh[None, None, :]) & 3).reshape(ns, 128).to(torch.int8)      # [left flank][nk + 59 bases][right flank]
    del r
    ln = nk + 59
    t = torch.arange(126, device=dev, dtype=torch.int64)
    w = torch.randint(-(1 << 62), 1 << 62, (126,), device=dev, dtype=torch.int64, generator=torch.Generator(device=dev).manual_seed(7))
    live = t[None, :] < ln[:, None]
    body = bases[:, 1:127].to(torch.int64)
    ridx = (ln[:, None] - 1 - t[None, :]).clamp(min=0)
    rbody = 3 - torch.gather(body, 1, ridx)
    L = bases[:, 0].to(torch.int64) * hasL
    R = torch.gather(bases.to(torch.int64), 1, (1 + ln)[:, None]).squeeze(1) * hasR
    def hsh(bd, hl, l, hr, rr):
        h = (torch.where(live, bd + 1, torch.zeros_like(bd)) * w[None, :]).sum(dim=1)
        h = h * 31 + nk; h = h * 31 + hl * 5 + l; h = h * 31 + hr * 5 + rr
        return h ^ (h >> 29)
    hF = hsh(body, hasL, L, hasR, R)
    hRc = hsh(rbody, hasR, (3 - R) * hasR, hasL, (3 - L) * hasL)
    for name, h in (("as stored", hF), ("strand-normalised", torch.minimum(hF, hRc))):
        u, inv = torch.unique(h, return_inverse=True)
        first = torch.zeros_like(u); first.scatter_reduce_(0, inv, nk, reduce="amax")
        print(f"sample of {ns:,} records, {name}: distinct {u.numel() / ns:.3f} of the records, {int(first.sum()) / int(nk.sum()):.3f} of the k-mer instances")
