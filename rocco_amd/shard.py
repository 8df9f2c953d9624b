"""Chromosome sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference's only parallelism over chromosomes is a fork pool of at most 4 workers
(rocco/rocco.py:792-806, 1176-1180) whose results come back as pickled tuples and temporary BED
files.  Here chromosomes are independent units assigned to ranks by longest-processing-time-first
on their locus counts; there is no collective on the data path.  The only exchange is the gather
of the final interval lists to rank 0 (a few hundred KB for a whole genome): one all_gather of the
per-rank interval counts and one all_gather of a padded int64 tensor -- RCCL over xGMI when the
backend is "nccl", Gloo in the CPU tests.  Latency-bound, not link-bound: no ring is engineered.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def lpt_partition(sizes: Sequence[int], n_ranks: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of units to ranks.  Returns, per rank, the indices
    of its units in descending size order.  Deterministic (ties by index)."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    loads = [0] * n_ranks
    owned: List[List[int]] = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (loads[k], k))
        owned[r].append(i)
        loads[r] += int(sizes[i])
    return owned


def makespan(sizes: Sequence[int], owned: List[List[int]]) -> int:
    return max((sum(int(sizes[i]) for i in part) for part in owned), default=0)


def gather_budget_counts(local: Dict[int, Tuple[float, float]], n_units: int, device=None, group=None) -> Dict[int, Tuple[float, float]]:
    """The one cross-chromosome exchange BEFORE the solve (rocco/rocco.py:1117-1127 read every chromosome's
    `budget_count_hat` and `total_count` to pool the budgets): `local` maps the units this rank owns to that pair;
    returns the pairs of every unit, identical on every rank, so that each rank runs the same host-side
    empirical-Bayes fit (rocco_amd/budget.py) and keeps the budgets of its own units.  One all_gather of a
    [n_units, 3] float64 tensor (flag, count, total) per rank: 24 x 24 bytes on a genome."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(local)
    world = dist.get_world_size(group)
    dev = device if device is not None else "cpu"
    mine = torch.zeros((n_units, 3), dtype=torch.float64, device=dev)
    for u, (count, total) in local.items():
        mine[u, 0], mine[u, 1], mine[u, 2] = 1.0, float(count), float(total)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    merged: Dict[int, Tuple[float, float]] = {}
    for part in parts:
        arr = part.cpu().numpy()
        for u in np.flatnonzero(arr[:, 0] > 0.5):
            if int(u) in merged:
                raise ValueError(f"unit {int(u)} is owned by more than one rank")
            merged[int(u)] = (float(arr[u, 1]), float(arr[u, 2]))
    return merged


def gather_interval_rows(rows_t, group=None) -> Dict[int, np.ndarray]:
    """The same gather from rows that are still on the device: `rows_t` is an int64 tensor [m, 3] of (unit, start, end),
    on the GPU under RCCL (backend "nccl") or on the CPU under Gloo.  Two collectives (row counts, padded rows) the first
    time, ONE afterwards (a header row + the room the ranks remember from the last exchange; two again when a table has
    outgrown it), and ONE transfer to the host at the end -- the intervals are not brought to the host first and sent
    back for the exchange.
    Returns unit -> [m_u, 2] arrays, identical on every rank."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        arr = rows_t.cpu().numpy()
    else:
        world = dist.get_world_size(group)
        m = int(rows_t.shape[0])
        key = (id(group), str(rows_t.device), world)
        arr = None
        room = _gather_room.get(key)
        if room is not None:
            # ONE collective: a header row (my row count) in front of `room` rows, the room every rank remembers from the
            # last exchange (twice the longest table then); every rank reads the same headers, so every rank takes the
            # same decision when some table has outgrown the room
            mine = torch.zeros((room + 1, 3), dtype=torch.int64, device=rows_t.device)
            mine[0, 0] = m
            if m:
                mine[1:1 + min(m, room)] = rows_t[:room]
            parts = torch.empty((world * (room + 1), 3), dtype=torch.int64, device=rows_t.device)
            dist.all_gather_into_tensor(parts, mine, group=group)
            host = parts.cpu().numpy().reshape(world, room + 1, 3)
            sizes = [int(v) for v in host[:, 0, 0]]
            if max(sizes) <= room:
                arr = np.concatenate([host[r, 1:1 + sizes[r]] for r in range(world)], axis=0) if sum(sizes) else np.zeros((0, 3), dtype=np.int64)
        if arr is None:
            count = torch.tensor([m], dtype=torch.int64, device=rows_t.device)
            counts = torch.zeros(world, dtype=torch.int64, device=rows_t.device)
            dist.all_gather_into_tensor(counts, count, group=group)
            sizes = [int(c) for c in counts.cpu().tolist()]
            width = max(max(sizes), 1)
            mine = torch.zeros((width, 3), dtype=torch.int64, device=rows_t.device)
            if m:
                mine[:m] = rows_t
            parts = torch.empty((world * width, 3), dtype=torch.int64, device=rows_t.device)
            dist.all_gather_into_tensor(parts, mine, group=group)
            host = parts.cpu().numpy().reshape(world, width, 3)
            arr = np.concatenate([host[r, : sizes[r]] for r in range(world)], axis=0) if sum(sizes) else np.zeros((0, 3), dtype=np.int64)
        _gather_room[key] = max(1024, 2 * max(sizes))
    return _split_rows_by_unit(arr)


_gather_room: Dict[tuple, int] = {}  # per (group, device, world size): rows of room in the one-collective form of the gather


def _split_rows_by_unit(arr: np.ndarray) -> Dict[int, np.ndarray]:
    """Rows (unit, start, end) -> {unit: [m_u, 2]}.  Every rank hands its units over one after the other, so a unit's rows
    are one run of the gathered table: the runs' ends are where the unit column changes (one pass; a stable sort by unit --
    2 ms for a genome's 60 000 rows, as long as a rank's whole step at N = 8 -- only if some unit comes in two runs)."""
    merged: Dict[int, np.ndarray] = {}
    if not arr.shape[0]:
        return merged
    units = arr[:, 0]
    cuts = np.flatnonzero(units[1:] != units[:-1]) + 1
    starts = np.concatenate(([0], cuts))
    heads = units[starts]
    if np.unique(heads).size != heads.size:  # a unit in two runs: order by unit first (rows of a unit keep their order)
        arr = arr[np.argsort(units, kind="stable")]
        units = arr[:, 0]
        cuts = np.flatnonzero(units[1:] != units[:-1]) + 1
        starts = np.concatenate(([0], cuts))
        heads = units[starts]
    ends = np.concatenate((cuts, [arr.shape[0]]))
    for u, a, b in zip(heads.tolist(), starts.tolist(), ends.tolist()):
        merged[int(u)] = arr[a:b, 1:]
    return merged


def gather_intervals(local: Dict[int, np.ndarray], device=None, group=None) -> Dict[int, np.ndarray]:
    """Gather per-unit interval arrays to every rank.

    `local` maps unit index -> int64 array of shape [m, 2] (start, end).  Returns the union over all
    ranks.  Uses two collectives on a flat int64 tensor of rows (unit, start, end)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(local)
    world = dist.get_world_size(group)
    rows = [np.concatenate([np.full((a.shape[0], 1), u, dtype=np.int64), a.astype(np.int64)], axis=1)
            for u, a in sorted(local.items()) if a.shape[0] > 0]
    flat = np.concatenate(rows, axis=0) if rows else np.zeros((0, 3), dtype=np.int64)
    dev = device if device is not None else "cpu"
    count = torch.tensor([flat.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    sizes = [int(c.item()) for c in counts]
    width = max(max(sizes), 1)
    mine = torch.zeros((width, 3), dtype=torch.int64, device=dev)
    if flat.shape[0]:
        mine[: flat.shape[0]] = torch.from_numpy(flat).to(dev)
    parts = [torch.zeros((width, 3), dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    merged: Dict[int, List[np.ndarray]] = {}
    for r, part in enumerate(parts):
        arr = part[: sizes[r]].cpu().numpy()
        for u in np.unique(arr[:, 0]) if arr.shape[0] else []:
            merged.setdefault(int(u), []).append(arr[arr[:, 0] == u][:, 1:])
    return {u: np.concatenate(v, axis=0) for u, v in merged.items()}
