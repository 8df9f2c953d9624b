"""Row f2 (rocco/readtracks.py:94-186, 575-633): the CPU oracle's matrix assembly and bigWig dense fill against fixtures the
REFERENCE's own functions wrote (tests/golden/make_golden_assemble.py: `generate_chrom_matrix` with its readers replaced in
the module, `get_bigwig_chrom_scores` over a stand-in pyBigWig).  Arrays bit for bit (dtype included), errors word for word."""
import numpy as np
import pytest

import assemble_golden as ag


@pytest.fixture(scope="module")
def gold():
    return ag.load()


def test_assembly_scenarios(oracle, gold):
    arrays, meta = gold
    assert len(meta["matrix"]) >= 12
    for record in meta["matrix"]:
        name = record["name"]
        tracks = [t for t in ag.matrix_tracks(arrays, record) if t is not None]  # (files without data are excluded: 597-604)
        types = {f.rsplit(".", 1)[1].lower() for f in record["files"]}
        if "error" in record and "same type" in record["error"]:
            assert len({("bam" if t == "bam" else "bigwig") for t in types}) == 2
            continue
        track_type = "bam" if types == {"bam"} else "bigwig"
        if record.get("none"):
            assert not tracks
            continue
        kwargs = dict(track_type=track_type, low_memory=bool(record["kwargs"].get("low_memory", False)), chromosome="chrT")
        if "error" in record:
            with pytest.raises(ValueError) as info:
                oracle.assemble_chrom_matrix([t[0] for t in tracks], [t[1] for t in tracks], **kwargs)
            assert str(info.value) == record["error"], name
            continue
        got_i, got_m = oracle.assemble_chrom_matrix([t[0] for t in tracks], [t[1] for t in tracks], **kwargs)
        want_i, want_m = arrays[f"m_{name}_intervals"], arrays[f"m_{name}_matrix"]
        assert got_i.dtype == want_i.dtype and np.array_equal(got_i, want_i), name
        assert got_m.dtype == want_m.dtype and got_m.shape == want_m.shape and got_m.tobytes() == want_m.tobytes(), name


def test_bigwig_dense_fill_scenarios(oracle, gold):
    arrays, meta = gold
    checked = 0
    for record in meta["bigwig"]:
        name = record["name"]
        intervals = ag.bigwig_intervals(arrays, record)
        if record.get("none") or not intervals or "sizes file" in record.get("error", ""):
            continue  # (decided before the intervals are looked at: covered by the product-level test)
        starts, ends, vals = (np.asarray([e[k] for e in intervals]) for k in range(3))
        kwargs = dict(const_scale=record["kwargs"].get("const_scale", 1.0), round_digits=record["kwargs"].get("round_digits", 5),
                      bigwig_file="{file}", chromosome=record["chromosome"])
        if "error" in record:
            with pytest.raises(ValueError) as info:
                oracle.bigwig_dense_fill(starts, ends, vals, **kwargs)
            assert str(info.value) == record["error"], name
        else:
            got_i, got_v = oracle.bigwig_dense_fill(starts, ends, vals, **kwargs)
            want_i, want_v = arrays[f"b_{name}_intervals"], arrays[f"b_{name}_out"]
            assert got_i.dtype == want_i.dtype and np.array_equal(got_i, want_i), name
            assert got_v.dtype == want_v.dtype and got_v.tobytes() == want_v.tobytes(), name
        checked += 1
    assert checked >= 14
