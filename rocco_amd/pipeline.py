"""Device-resident per-rank pipeline: K x n matrices -> scores -> budgeted solve -> merged runs.

This is the GPU form of the reference's `_build_chrom_cache` scoring step for bigWig inputs
(rocco/rocco.py:983-991), `_solve_cached_chromosomes` (rocco/rocco.py:1146-1196) and the decode in
`chrom_solution_to_bed` (rocco/rocco.py:139-191), for the chromosomes owned by one rank: every
array stays in HBM, every chromosome of the rank shares each device pass of the solve, and only the
interval lists (a few thousand index pairs per chromosome) come back to the host.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import dp as _dp
from . import rocco as _rocco


class ChromWork:
    """One chromosome owned by this rank."""

    def __init__(self, name: str, matrix_t, budget: float, gamma: float, step: int = 50, start: int = 0):
        self.name = name
        self.matrix_t = matrix_t  # [K, n] float64 / float32 CUDA tensor
        self.budget = float(budget)
        self.gamma = float(gamma)
        self.step = int(step)
        self.start = int(start)
        self.n = int(matrix_t.shape[1])


def solve_rank(chroms: Sequence[ChromWork], scores_out: Optional[list] = None):
    """Score, solve and decode every chromosome of this rank.

    Returns a list of dicts: name, n, selected_count, selection_penalty, penalized_objective, path,
    begin / end (int64 CUDA tensors: half-open locus index pairs of the merged runs) and the
    solution tensor.
    """
    import torch

    scores = []
    for c in chroms:
        s_t = _rocco.score_central_tendency_chrom_device(c.matrix_t)
        scores.append(s_t)
    if scores_out is not None:
        scores_out.extend(scores)
    targets = [int(np.floor(c.n * c.budget)) for c in chroms]  # rocco/dp.py:197
    solved = _dp.calibrate_batch_device(scores, [c.gamma for c in chroms], targets)
    out = []
    for c, s_t, (penalty, sol_t, value, count, info) in zip(chroms, scores, solved):
        begin_t, end_t = _rocco.decode_runs_device(sol_t, capacity=max(1024, c.n // 64))
        out.append({
            "name": c.name, "n": c.n, "selected_count": count, "selection_penalty": penalty,
            "penalized_objective": value, "path": info["path"], "info": info,
            "begin": begin_t, "end": end_t, "solution": sol_t, "step": c.step, "start": c.start,
        })
    return out


def runs_to_records(result: dict, min_length_bp: Optional[int] = None) -> List[Tuple[str, int, int]]:
    """BED3 records of one chromosome result (fixed-step loci: start + index * step)."""
    b = result["begin"].cpu().numpy()
    e = result["end"].cpu().numpy()
    step, start = result["step"], result["start"]
    recs = []
    for bi, ei in zip(b.tolist(), e.tolist()):
        s_bp, e_bp = start + bi * step, start + ei * step
        if min_length_bp is None or (e_bp - s_bp) >= int(min_length_bp):
            recs.append((result["name"], int(s_bp), int(e_bp)))
    return recs
