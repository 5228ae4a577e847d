# K1 diagnosis: the partition step alone on the bench reads, kernel times from the library's own event profile
import os, sys, json, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from w2rap_contigger_amd import step2, synth
dev = torch.device("cuda", 0)
n_reads, genome_len = 50_000_000, 250_000_000
gen = torch.Generator(device=dev).manual_seed(42)
genome = torch.randint(0, 4, (genome_len,), dtype=torch.uint8, device=dev, generator=gen)
d = synth.generate_reads_device(n_reads, genome_len, 42, device=dev, genome=genome)
del genome; d.pop("genome", None); torch.cuda.synchronize(); torch.cuda.empty_cache()
ctx = step2.Step2Context(0)
ctx.set_reads_device(d["n"], d["packed"].data_ptr(), d["byte_off"].data_ptr(), d["read_len"].data_ptr(), d["quals"].data_ptr(), d["qual_off"].data_ptr(), keepalive=d)
M = ctx.quality_windows(7)
nb = ctx.default_buckets(M)
ctx.set_profiling(True)
ctx.partition(nb, 1)
ctx.profile(reset=True)
for _ in range(3):
    recs, n, cnts, per = ctx.partition(nb, 1)
torch.cuda.synchronize()
prof = ctx.profile(reset=True)
print(json.dumps({"lib": os.environ.get("W2RAP_LIB", "default"), "batches": os.environ.get("W2RAP_BATCHES", ""), "n_records": n, "prof": prof}, default=str))
