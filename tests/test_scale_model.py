"""The per-phase cost model of `bench.py --gpus N` (w2rap_contigger_amd/scale_model.py): its arithmetic, pinned on CPU."""
from w2rap_contigger_amd import scale_model as SM


def test_world_1_is_the_sum_of_its_measured_phases():
    w = SM.World1()
    p = SM.predict(1, w)
    assert abs(p["ms_per_step"] - (w.quality + w.partition + w.count + w.first_slice_exposed + w.prune + w.graph + w.path)) < 1e-9
    assert p["phase_ms"]["dictionary_exposed"] == 0.0           # the insert hides under the counting at world 1


def test_replicated_phases_grow_with_the_job_and_sharded_ones_do_not():
    w = SM.World1()
    p1, p8 = SM.predict(1, w), SM.predict(8, w)
    for k in ("quality", "partition", "count", "path"):
        assert p8["phase_ms"][k] == p1["phase_ms"][k]
    for k in ("prune", "graph"):
        assert abs(p8["phase_ms"][k] - 8 * p1["phase_ms"][k]) < 1e-9
    rows = SM.table(w)
    eff = [r["weak_scaling_efficiency"] for r in rows]
    assert eff[0] == 1.0 and all(a > b for a, b in zip(eff, eff[1:]))
    assert rows[-1]["replicated_share"] > 0.5                  # what row e-3 (sharded dictionary + graph) has to remove


def test_without_the_replicated_part_the_model_scales():
    w = SM.World1(insert=0.0, prune=0.0, graph=0.0, solid_bytes_per_gpu=0.0)
    rows = SM.table(w)
    assert rows[-1]["weak_scaling_efficiency"] > 0.9           # the shuffle hides under the counting: 8.3 GB x 7/8 over 7 links
