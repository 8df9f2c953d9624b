"""Test helper: the CPU oracle on several chromosomes at once, one spawned process per job (no torch, no GPU in
the children; fork after HIP initialisation is unsafe).  TEST INFRASTRUCTURE ONLY."""
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _budgeted(job):
    path, budget, gamma = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po

    s = np.load(path, mmap_mode="r")
    sol, obj, det = po.solve_chrom_exact(np.asarray(s), budget=budget, gamma=gamma, return_details=True)
    np.save(path + ".sol.npy", sol)
    return det["selection_penalty"], det["selected_count"], det["penalized_objective"]


def _fixed(job):
    path, gamma, penalty = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po

    s = np.asarray(np.load(path, mmap_mode="r"))
    sol, value, count = po.solve_penalized_chain(s, po.build_switch_costs(s, gamma), penalty)
    np.save(path + ".sol.npy", sol)
    return value, count


def run(jobs, kind, processes=None):
    """jobs: list of (scores array, ...) tuples; returns [(result tuple, solution array)] in order."""
    procs = processes or max(1, min(16, len(jobs), len(os.sched_getaffinity(0))))
    tmp_root = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(prefix="rocco_oracle_", dir=tmp_root) as tmp:
        packed = []
        for i, job in enumerate(jobs):
            path = os.path.join(tmp, f"{i}.npy")
            np.save(path, np.ascontiguousarray(job[0], dtype=np.float64))
            packed.append((path,) + tuple(job[1:]))
        fn = _budgeted if kind == "budgeted" else _fixed
        if procs == 1:
            res = [fn(j) for j in packed]
        else:
            with mp.get_context("spawn").Pool(procs) as pool:
                res = pool.map(fn, packed, chunksize=1)
        return [(r, np.load(p[0] + ".sol.npy")) for r, p in zip(res, packed)]
