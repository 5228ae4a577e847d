#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP path with the oracle on the golden fixtures (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from w2rap_contigger_amd import formats as F, step2
from oracle import oracle as O

G = os.path.join(ROOT, "tests", "golden")
names = sys.argv[1:] or ["random20k", "repeats_snps", "palindrome_circle"]
ok_all = True
for name in names:
    pk, bo, ln = F.read_fastb(f"{G}/{name}.fastb")
    codes, off = F.unpack_bases(pk, bo, ln)
    pq, po = F.read_qualp(f"{G}/{name}.qualp")
    quals, qoff = F.qualp_to_raw(pq, po)
    ref_hbv = F.read_hbv(f"{G}/{name}.ref.hbv")
    hc, ho = O.edge_hint_from_hbv(ref_hbv)
    orc = O.run(codes, quals, off, hint_codes=hc, hint_off=ho)
    hint = F.pack_bases(hc, ho)
    with step2.Step2Context(0) as ctx:
        ctx.set_reads_host(pk, bo, ln, pq=pq, pq_off=po)
        st = ctx.count_kmers(7, 4)
        gl = ctx.good_len()
        print(name, "good_len", np.array_equal(gl, orc.good_len), "M", st["M"], orc.n_instances, "D", st["D"], orc.n_distinct,
              "S", st["S"], len(orc.k_hi), "hist", np.array_equal(st["hist"], orc.hist), f"{st['ms']:.2f} ms")
        S = st["S"]
        hi, lo, cnt, ctx_, edge, off_ = ctx.table(S)
        order = np.lexsort((lo, hi))
        tab_ok = (S == len(orc.k_hi) and np.array_equal(hi[order], orc.k_hi) and np.array_equal(lo[order], orc.k_lo)
                  and np.array_equal(cnt[order], orc.k_count))
        ctx_ok = tab_ok and np.array_equal(ctx_[order], orc.k_ctx)
        print("  table", tab_ok, "pruned ctx", ctx_ok)
        ctx.build_graph(hint)
        hi, lo, cnt, ctx_, edge, off_ = ctx.table(S)
        print("  edge/off", np.array_equal(edge[order], orc.k_edge), np.array_equal(off_[order], orc.k_off))
        ctx.path_reads()
        res = ctx.fetch()
    mine = F.hbv_to_bytes(res.hbv)
    refb = open(f"{G}/{name}.ref.hbv", "rb").read()
    pm = F.paths_to_bytes(res.path_offset, res.path_off, res.path_edges)
    pr = open(f"{G}/{name}.ref.paths", "rb").read()
    print("  hbv bytes", mine == refb, "paths bytes", pm == pr, "NV", res.hbv.n_vertices, orc.n_vertices, "NO", res.hbv.n_edges,
          "pathed", res.n_reads_pathed, orc.pathed, f"graph {res.ms_graph:.2f} ms path {res.ms_path:.2f} ms")
    if pm != pr:
        o2, po2, e2 = F.read_paths(f"{G}/{name}.ref.paths")
        bad = [i for i in range(len(o2)) if o2[i] != res.path_offset[i] or list(e2[int(po2[i]):int(po2[i+1])]) != list(res.path_edges[int(res.path_off[i]):int(res.path_off[i+1])])]
        print("   differing reads", len(bad), bad[:10])
        for i in bad[:5]:
            print("   ", i, "ref", o2[i], e2[int(po2[i]):int(po2[i+1])], "gpu", res.path_offset[i], res.path_edges[int(res.path_off[i]):int(res.path_off[i+1])])
    ok_all &= (mine == refb and pm == pr and tab_ok and ctx_ok)
print("ALL OK" if ok_all else "MISMATCH")
