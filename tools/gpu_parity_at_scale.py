#!/usr/bin/env python3
"""One-off: full parity of the HIP path with the oracle on bench-like synthetic reads (errors, Q2 tails) at a size the unit
tests skip: python3 tools/gpu_parity_at_scale.py [reads] [genome]   (the oracle is single-threaded: ~1 min per million reads)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from w2rap_contigger_amd import formats as F, step2, synth
from oracle import oracle as O
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
g = int(float(sys.argv[2])) if len(sys.argv) > 2 else n * 5
d = synth.generate_reads_device(n, g, 77, device="cuda")
codes = synth.unpack_fixed(d["packed"], synth.READ_LEN).cpu().numpy().reshape(-1)
quals = d["quals"].cpu().numpy().reshape(-1)
off = np.arange(d["n"] + 1, dtype=np.uint64) * synth.READ_LEN
t0 = time.time(); orc = O.run(codes, quals, off); t1 = time.time()
pk, bo, ln = F.pack_bases(codes, off)
res = step2.build_read_qgraph(pk, bo, ln, quals=quals, qual_off=off)
ok = [np.array_equal(res.hist, orc.hist), F.hbv_to_bytes(res.hbv) == F.hbv_to_bytes(O.to_hbv(orc)),
      np.array_equal(res.path_offset, orc.path_offset) and np.array_equal(res.path_off, orc.path_off), np.array_equal(res.path_edges, orc.path_edges)]
print(f"{d['n']} reads, genome {g}: oracle {t1 - t0:.1f} s; histogram {ok[0]}, graph bytes {ok[1]}, path offsets {ok[2]}, path edges {ok[3]}; "
      f"S {res.n_kmers_solid}, edge objects {res.hbv.n_edges}, pathed {res.n_reads_pathed}")
if len(sys.argv) > 3 and sys.argv[3] == "step3":
    # Step 3 (K2 = 200) behind it, GPU result against the Step-3 oracle run on the oracle's Step-2 output
    from w2rap_contigger_amd import step3
    from oracle import oracle3 as O3
    t2 = time.time()
    o3 = O3.run(O.to_hbv(orc), (orc.path_offset, orc.path_off, orc.path_edges), 200)
    t3 = time.time()
    r3 = step3.repath_in_memory(res.hbv, (res.path_offset, res.path_off, res.path_edges), 200)
    ok3 = [F.hbv_to_bytes(r3.hbv) == F.hbv_to_bytes(O3.to_hbv(o3)), np.array_equal(r3.path_offset, o3.path_offset) and np.array_equal(r3.path_off, o3.path_off),
           np.array_equal(r3.path_edges, o3.path_edges), np.array_equal(r3.inv, o3.inv), np.array_equal(r3.frag_count.astype(np.float64), o3.frag)]
    print(f"Step 3 at K2=200: oracle {t3 - t2:.1f} s; graph bytes {ok3[0]}, path offsets {ok3[1]}, path edges {ok3[2]}, involution {ok3[3]}, fragment counts {ok3[4]}; "
          f"{r3.n_unique_places} unique places, {r3.n_kmer_instances} K2-mer occurrences, {r3.hbv.n_edges} large-K edge objects")
    ok += ok3
sys.exit(0 if all(ok) else 1)
