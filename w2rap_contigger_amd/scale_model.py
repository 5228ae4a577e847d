"""Per-phase cost model of `bench.py --gpus N` (BASELINE configs[2] scaled to N GPUs: fixed reads per GPU, ONE genome of N x 312.5 Mbp).

No multi-GPU node was available to the builder; this model says what the first hardware run should show, phase by phase, from numbers that
WERE measured: the per-phase times of the distributed code path at world 1 (W2RAP_FORCE_DIST=1 W2RAP_TRACE=1 python bench.py, or the defaults
below taken from profiles/r04_*), the sizes of what travels, and the link rates of MI355X_MICROARCH.md.  `tests/test_scale_model.py` pins
its arithmetic on CPU and, on the GPU box, checks its world-1 prediction against a forced-distributed run of bench.py.

What scales how:
  * per GPU, constant in N (weak scaling): quality windows, partition (K1/K2), owner-side counting (K3), read pathing;
  * the k-mer shuffle: every rank sends (N-1)/N of its records, point to point over xGMI, slice by slice under the counting -- only the
    first slice is exposed;
  * REPLICATED, proportional to the JOB's solid k-mers S_total = N x S_1: the gathered dictionary (all-gather + k_table_insert), the
    adjacency prune, the whole graph phase.  This is the term that breaks weak scaling; row e-3 (sharded dictionary and graph) removes it.
"""
from dataclasses import dataclass, asdict

XGMI_LINK_GBS = 153.0       # per link and direction, 7 links per GPU (MI355X_MICROARCH.md)
LINK_EFF = 0.7              # fraction of the link rate a large point-to-point copy reaches (assumption until measured)


@dataclass
class World1:
    """per-phase milliseconds of the DISTRIBUTED code path at world 1 on the per-GPU share of the workload (62.5 M reads, 312.5 Mbp)"""
    # measured: W2RAP_FORCE_DIST=1 W2RAP_TRACE=1 python bench.py --reads 62.5e6 --genome 312.5e6 (profiles/r04_dist_world1.json / .txt)
    quality: float = 2.3
    partition: float = 26.4          # K1 + K2 over the rank's reads (round 4: the lane-per-read K1; was 40.0)
    count: float = 68.5              # exchange + owner-side counting + gathers of the four slices (the kernels alone: ~49; was 92.4)
    first_slice_exposed: float = 9.1  # counts, offsets, exchange + launch of slice 0, which nothing hides
    insert: float = 12.0             # k_table_insert of S_1 = 312 M solid k-mers (hidden under the counting at world 1)
    prune: float = 11.5              # adjacency prune over S_1 ("dictionary" mark of the trace)
    graph: float = 29.5              # unipaths, vertices, k-mer records, 31-mer filter over S_1
    path: float = 18.4
    record_bytes_per_gpu: float = 9.1e9      # super-k-mer records a rank produces (284 M of 32 B each)
    solid_bytes_per_gpu: float = 312e6 * 20  # S_1 x (16 B key + 4 B count | context) this rank contributes to the gathered dictionary


def predict(n_gpus: int, w: World1 = World1()) -> dict:
    """-> per-phase ms at n_gpus and the step time; weak-scaling efficiency = t(1) / t(N)"""
    n = max(1, int(n_gpus))
    links = min(n - 1, 7)
    # the shuffle: (n-1)/n of the records leave the rank, over `links` links at once; it runs under the counting except for slice 0
    shuffle = 0.0 if n == 1 else w.record_bytes_per_gpu * (n - 1) / n / (links * XGMI_LINK_GBS * 1e9 * LINK_EFF) * 1e3
    exposed_shuffle = max(0.0, shuffle - w.count) + w.first_slice_exposed
    # the gathered dictionary: every rank receives the other ranks' solid k-mers and inserts ALL n x S_1 of them; the counting hides what it can
    gather = 0.0 if n == 1 else w.solid_bytes_per_gpu * (n - 1) / (links * XGMI_LINK_GBS * 1e9 * LINK_EFF) * 1e3
    insert_all = w.insert * n
    exposed_dict = max(0.0, insert_all + gather - w.count)
    phases = {"quality": w.quality, "partition": w.partition, "count": w.count, "shuffle_exposed": exposed_shuffle, "dictionary_exposed": exposed_dict,
              "prune": w.prune * n, "graph": w.graph * n, "path": w.path}
    total = sum(phases.values())
    return {"n_gpus": n, "phase_ms": phases, "ms_per_step": total, "replicated_ms": exposed_dict + w.prune * n + w.graph * n}


def table(w: World1 = World1()):
    t1 = predict(1, w)["ms_per_step"]
    rows = []
    for n in (1, 2, 4, 8):
        p = predict(n, w)
        rows.append({"n_gpus": n, "ms_per_step": round(p["ms_per_step"], 1), "weak_scaling_efficiency": round(t1 / p["ms_per_step"], 3),
                     "replicated_share": round(p["replicated_ms"] / p["ms_per_step"], 3)})
    return rows


if __name__ == "__main__":
    import json
    print(json.dumps({"assumptions": asdict(World1()), "link_GBs": XGMI_LINK_GBS, "link_eff": LINK_EFF, "prediction": table()}, indent=1))
