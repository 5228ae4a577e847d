#!/usr/bin/env python3
"""One-off: the SHARDED graph path (row e-3) behind the one in-process call -- `world` ranks sharing the one GPU of the box, with the two
test hooks that give few ranks the query and segment counts of a many-rank job -- against the one-GPU path on the same reads, at sizes the
unit tests skip; below `oracle_max` reads also against the oracle (single-threaded: ~1 min per million reads).
    python3 tools/gpu_sharded_at_scale.py reads genome world [cut] [virtual] [oracle_max]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from w2rap_contigger_amd import formats as F, step2, synth
n = int(float(sys.argv[1])); g = int(float(sys.argv[2])); world = int(sys.argv[3])
cut = int(sys.argv[4]) if len(sys.argv) > 4 else 0
virt = int(sys.argv[5]) if len(sys.argv) > 5 else 0
oracle_max = int(float(sys.argv[6])) if len(sys.argv) > 6 else 0
d = synth.generate_reads_device(n, g, 77, device="cuda")
codes = synth.unpack_fixed(d["packed"], synth.READ_LEN).cpu().numpy().reshape(-1)
quals = d["quals"].cpu().numpy().reshape(-1)
off = np.arange(d["n"] + 1, dtype=np.uint64) * synth.READ_LEN
del d; torch.cuda.empty_cache()
pk, bo, ln = F.pack_bases(codes, off)
t0 = time.time(); one = step2.build_read_qgraph(pk, bo, ln, quals=quals, qual_off=off); t1 = time.time()
if cut: os.environ["W2RAP_TEST_SHARD_CUT"] = str(cut)
if virt: os.environ["W2RAP_TEST_SHARD_VIRTUAL"] = str(virt)
sh = step2.build_read_qgraph(pk, bo, ln, quals=quals, qual_off=off, devices=[0] * world); t2 = time.time()
ok = [np.array_equal(sh.hist, one.hist), (sh.n_kmer_instances, sh.n_kmers_distinct, sh.n_kmers_solid) == (one.n_kmer_instances, one.n_kmers_distinct, one.n_kmers_solid),
      F.hbv_to_bytes(sh.hbv) == F.hbv_to_bytes(one.hbv), np.array_equal(sh.path_offset, one.path_offset) and np.array_equal(sh.path_off, one.path_off),
      np.array_equal(sh.path_edges, one.path_edges)]
print(f"{len(ln)} reads, genome {g}: one GPU {t1 - t0:.1f} s, {world} ranks sharded (cut {cut}, virtual {virt}) {t2 - t1:.1f} s: histogram {ok[0]}, counts {ok[1]}, "
      f"graph bytes {ok[2]}, path offsets {ok[3]}, path edges {ok[4]}; S {sh.n_kmers_solid}, edge objects {sh.hbv.n_edges}, pathed {sh.n_reads_pathed}")
if len(ln) <= oracle_max:
    from oracle import oracle as O
    t3 = time.time(); orc = O.run(codes, quals, off); t4 = time.time()
    oko = [np.array_equal(sh.hist, orc.hist), F.hbv_to_bytes(sh.hbv) == F.hbv_to_bytes(O.to_hbv(orc)),
           np.array_equal(sh.path_offset, orc.path_offset) and np.array_equal(sh.path_off, orc.path_off), np.array_equal(sh.path_edges, orc.path_edges)]
    print(f"  against the oracle ({t4 - t3:.1f} s): histogram {oko[0]}, graph bytes {oko[1]}, path offsets {oko[2]}, path edges {oko[3]}")
    ok += oko
sys.exit(0 if all(ok) else 1)
