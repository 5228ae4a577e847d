#!/usr/bin/env python3
"""Scale probe on the GPU box: synthetic PE150 reads generated in HBM, staged Step 2, phase timings."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from w2rap_contigger_amd import formats as F, step2, synth

def gen(n_reads, genome_len, seed, dev="cuda", chunk_pairs=1 << 20):
    g = torch.randint(0, 4, (genome_len,), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    contig = [g]
    n_pairs = n_reads // 2
    packed = torch.empty((n_reads, 38), dtype=torch.uint8, device=dev)
    quals = torch.empty((n_reads, 150), dtype=torch.uint8, device=dev)
    done = 0
    while done < n_pairs:
        m = min(chunk_pairs, n_pairs - done)
        c, q = synth.sample_reads_t(g, m, seed * 1000003 + done, device=dev)
        packed[2 * done:2 * (done + m)] = synth.pack_fixed(c)
        quals[2 * done:2 * (done + m)] = q
        done += m
    return packed, quals

if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
    G = int(float(sys.argv[2])) if len(sys.argv) > 2 else n * 5
    check = len(sys.argv) > 3 and sys.argv[3] == "check"
    t0 = time.time()
    packed, quals = gen(n, G, 42)
    torch.cuda.synchronize()
    print(f"generated {n} reads from {G} bp genome in {time.time()-t0:.1f}s", flush=True)
    boff = torch.arange(n + 1, dtype=torch.int64, device="cuda") * 38
    qoff = torch.arange(n + 1, dtype=torch.int64, device="cuda") * 150
    rlen = torch.full((n,), 150, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_device(n, packed.data_ptr(), boff.data_ptr(), rlen.data_ptr(), quals.data_ptr(), qoff.data_ptr(),
                             keepalive=(packed, boff, rlen, quals, qoff))
        for it in range(2):
            t0 = time.time(); st = ctx.count_kmers(7, 4); t1 = time.time()
            ctx.build_graph(None); t2 = time.time()
            ctx.path_reads(); t3 = time.time()
            print(f"iter {it}: M={st['M']} D={st['D']} S={st['S']} count {t1-t0:.3f}s ({st['M']/(t1-t0)/1e9:.2f} G kmers/s) "
                  f"graph {t2-t1:.3f}s path {t3-t2:.3f}s ({n/(t3-t2)/1e6:.1f} M reads/s) total {t3-t0:.3f}s", flush=True)
            prof = ctx.profile() if hasattr(ctx, "profile") else None
            if prof: print(prof)
        res = ctx.fetch()
        print("E", len(res.fwd_xlat), "NO", res.hbv.n_edges, "NV", res.hbv.n_vertices, "pathed", res.n_reads_pathed, "multi", res.n_reads_multipathed)
        if check:
            from oracle import oracle as O
            codes = synth.unpack_fixed(packed.cpu(), 150).numpy().reshape(-1)
            off = np.arange(n + 1, dtype=np.uint64) * 150
            t0 = time.time()
            orc = O.run(codes, quals.cpu().numpy().reshape(-1), off)
            print(f"oracle {time.time()-t0:.1f}s")
            mine = F.hbv_to_bytes(res.hbv); ref = F.hbv_to_bytes(O.to_hbv(orc))
            print("hist", np.array_equal(res.hist, orc.hist), "hbv", mine == ref,
                  "paths", np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_off, orc.path_off) and np.array_equal(res.path_edges, orc.path_edges))
