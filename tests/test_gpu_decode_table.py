"""GPU: the table form of the run-length decode (rocco_hip_decode_runs_table: every chromosome's merged runs as rows
(unit, begin, end) of ONE table, on the device and in pinned host memory) against the per-chromosome decode and the
oracle's records (rocco/rocco.py:139-191)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solutions(rng, sizes):
    out = []
    for n in sizes:
        z = np.zeros(n, dtype=np.uint8)
        if n > 4:
            for p in rng.integers(0, n, size=max(1, n // 40)):
                z[p:p + int(rng.integers(1, 30))] = 1
        if n > 2 and rng.random() < 0.5:
            z[-3:] = 1  # a run into the last locus (never emitted: rocco/rocco.py:180)
        if n > 0 and rng.random() < 0.5:
            z[0] = 1
        out.append(z)
    return out


@pytest.mark.parametrize("sizes", [[1], [2, 0, 5], [4096, 4097, 8191, 3], [70000, 1, 33000, 12, 200000], list(range(0, 48))])
def test_table_equals_per_chromosome_decode_and_oracle(gpu, oracle, sizes):
    import torch
    from rocco_amd import rocco as impl

    rng = np.random.default_rng(len(sizes))
    zs = _solutions(rng, sizes)
    sols = [torch.from_numpy(z).to(gpu) for z in zs]
    units = [100 + 3 * i for i in range(len(zs))]
    table_t, offsets, host = impl.decode_runs_table_device(sols, units=units)
    assert len(offsets) == len(zs) + 1 and offsets[0] == 0 and offsets[-1] == table_t.shape[0] == host.shape[0]
    assert np.array_equal(table_t.cpu().numpy(), host)
    for i, z in enumerate(zs):
        rows = host[offsets[i]:offsets[i + 1]]
        n = z.shape[0]
        want = oracle.chrom_solution_records("c", np.arange(n, dtype=np.int64), z) if n > 1 else []
        assert [(int(b), int(e)) for _u, b, e in rows] == [(s, e) for _c, s, e in want], i
        assert np.all(rows[:, 0] == units[i])
    # too little room: the call is repeated with what the first one asked for; nothing eager: a second copy
    small_t, small_off, small_host = impl.decode_runs_table_device(sols, units=units, capacity_rows=1, eager_rows=0)
    assert small_off == offsets and np.array_equal(small_host, host)
    dev_only = impl.decode_runs_table_device(sols, units=units, to_host=False)
    assert dev_only[2] is None and np.array_equal(dev_only[0].cpu().numpy(), host)


def test_pipeline_results_carry_the_table(gpu):
    from rocco_amd import pipeline, synth

    works = [pipeline.ChromWork(f"c{i}", synth.hash_matrix_device(8, n, 5 + i), 0.03, 1.0) for i, n in enumerate([90000, 30000, 5000])]
    for groups in (1, 2):
        res = pipeline.solve_rank(works, groups=groups, units=[7, 9, 11])
        rows_h = pipeline.interval_rows(res, host=True)
        rows_d = pipeline.interval_rows(res, host=False).cpu().numpy()
        assert sorted(map(tuple, rows_h.tolist())) == sorted(map(tuple, rows_d.tolist()))
        for unit, r in zip([7, 9, 11], res):
            mine = rows_h[rows_h[:, 0] == unit]
            assert np.array_equal(mine[:, 1], r["begin"].cpu().numpy()) and np.array_equal(mine[:, 2], r["end"].cpu().numpy())
            lo, hi = r["row_range"]
            assert np.array_equal(r["rows_host"][lo:hi], mine)
