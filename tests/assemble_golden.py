"""(not a test module) Shared by the CPU (oracle) and GPU (product) tests of row f2: the scenarios tests/golden/make_golden_assemble.py ran through
the REFERENCE's `generate_chrom_matrix` (readers replaced in the module) and `get_bigwig_chrom_scores` (stand-in pyBigWig)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    arrays = np.load(os.path.join(HERE, "golden", "assemble_vectors.npz"))
    with open(os.path.join(HERE, "golden", "assemble_vectors.json"), "r", encoding="utf-8") as handle:
        meta = json.load(handle)
    return arrays, meta


def matrix_tracks(arrays, record):
    """Per input file: (starts, values) or None, as the stand-in readers returned them."""
    name = record["name"]
    return [(arrays[f"m_{name}_t{k}_starts"], arrays[f"m_{name}_t{k}_values"]) if has else None for k, has in enumerate(record["has"])]


def bigwig_intervals(arrays, record):
    """The list of (start, end, value) tuples the stand-in pyBigWig handed to the reference, or None."""
    if not record["has_intervals"]:
        return None
    name = record["name"]
    return [(int(s), int(e), float(v)) for s, e, v in zip(arrays[f"b_{name}_starts"], arrays[f"b_{name}_ends"], arrays[f"b_{name}_values"])]


class FakeBigWig:
    """Stands in for a pyBigWig file object: chroms() / intervals(chromosome) / close()."""

    def __init__(self, chroms, intervals):
        self._chroms, self._intervals, self.closed = chroms, intervals, False

    def chroms(self):
        return self._chroms

    def intervals(self, chromosome):
        return self._intervals.get(chromosome)

    def close(self):
        self.closed = True
