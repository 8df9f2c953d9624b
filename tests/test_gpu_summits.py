"""GPU: narrowPeak summit offsets (rocco/rocco.py:809-872) through the C ABI
against the reference's own known answer (tests/test_rocco.py:301-325) and the NumPy restatement: exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_reference_known_answer(gpu, tmp_path):
    from rocco_amd.rocco import _write_narrowpeak_summit_offsets

    peak_file = tmp_path / "peaks.bed"
    peak_file.write_text("chr1\t100\t250\nchr1\t150\t250\nchr2\t0\t50\n", encoding="utf-8")
    summit_track_file = tmp_path / "chr1_summit_track.npz"
    np.savez(summit_track_file, starts=np.array([100, 150, 200], dtype=np.int64),
             centers=np.array([125, 175, 225], dtype=np.int64), mean=np.array([1.0, 5.0, 2.0], dtype=np.float32))
    chrom_cache = {"chr1": {"summit_track_file": str(summit_track_file)}, "chr2": {"summit_track_file": None}}
    output_file = tmp_path / "pointsource.tsv"
    _write_narrowpeak_summit_offsets(str(peak_file), chrom_cache, str(output_file))
    assert output_file.read_text(encoding="utf-8").splitlines() == ["chr1_100_250\t75", "chr1_150_250\t25", "chr2_0_50\t-1"]


def random_case(rng, n, step, n_peaks):
    intervals = 1000 + step * np.arange(n + 1, dtype=np.int64)
    mean = rng.normal(size=n)
    mean = np.round(mean, 1)  # ties: the first maximum must win
    mean[rng.random(n) < 0.05] = np.nan
    mean[rng.random(n) < 0.01] = np.inf
    mean[rng.random(n) < 0.01] = -np.inf
    a = rng.integers(0, n, size=n_peaks)
    w = rng.integers(0, 400, size=n_peaks)
    starts = intervals[a] + rng.integers(-step, step, size=n_peaks) * (rng.random(n_peaks) < 0.3)
    ends = starts + w * step // 3 - (rng.random(n_peaks) < 0.1) * 7
    return intervals, mean, starts.astype(np.int64), ends.astype(np.int64)


@pytest.mark.parametrize("n,step,n_peaks", [(1, 50, 5), (2, 50, 20), (300, 50, 200), (5000, 10, 3000), (200000, 50, 40000)])
def test_offsets_match_the_restatement(gpu, oracle, tmp_path, n, step, n_peaks):
    from rocco_amd.rocco import _cpy_narrowpeak_summit_track, _write_narrowpeak_summit_offsets

    rng = np.random.default_rng(n + n_peaks)
    intervals, mean, starts, ends = random_case(rng, n, step, n_peaks)
    mean[:3] = np.nan  # a peak of NaNs only
    records = [("chrA", int(s), int(e)) for s, e in zip(starts, ends)] + [("chrB", 0, 100)]
    track = oracle.narrowpeak_summit_track(intervals, mean)
    want = oracle.narrowpeak_summit_offsets(records, {"chrA": track, "chrB": None})
    peak_file = tmp_path / "peaks.bed"
    peak_file.write_text("".join(f"{c}\t{s}\t{e}\n" for c, s, e in records), encoding="utf-8")
    path = _cpy_narrowpeak_summit_track("chrA", intervals, mean)
    try:
        with np.load(path) as stored:
            assert np.array_equal(stored["starts"], track[0]) and np.array_equal(stored["centers"], track[1])
            assert stored["mean"].tobytes() == track[2].tobytes()
        out = tmp_path / "offsets.tsv"
        _write_narrowpeak_summit_offsets(str(peak_file), {"chrA": {"summit_track_file": path}, "chrB": {}}, str(out))
        got = [tuple(line.split("\t")) for line in out.read_text(encoding="utf-8").splitlines()]
    finally:
        import os
        os.remove(path)
    assert got == [(name, str(off)) for name, off in want]


def test_device_entry_without_stored_centers(gpu, oracle):
    import torch

    from rocco_amd.rocco import narrowpeak_summit_offsets_device

    rng = np.random.default_rng(77)
    intervals, mean, starts, ends = random_case(rng, 100000, 25, 20000)
    want = [off for _, off in oracle.narrowpeak_summit_offsets(
        [("c", int(s), int(e)) for s, e in zip(starts, ends)], {"c": oracle.narrowpeak_summit_track(intervals, mean)})]
    got = narrowpeak_summit_offsets_device(torch.from_numpy(intervals).cuda(), torch.from_numpy(mean).cuda(),
                                           torch.from_numpy(starts).cuda(), torch.from_numpy(ends).cuda())
    assert got.cpu().numpy().tolist() == want
    assert _cpy_none()


def _cpy_none():
    from rocco_amd.rocco import _cpy_narrowpeak_summit_track

    return _cpy_narrowpeak_summit_track("c", np.array([5], dtype=np.int64), np.array([1.0])) is None
