"""N > 1 path on the CPU: two ranks over Gloo run the chromosome partition and the interval gather
that bench.py / the driver use over RCCL."""
import os

import numpy as np
import pytest


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rocco_amd import shard

    sizes = [50, 40, 30, 20, 10, 5]
    owned = shard.lpt_partition(sizes, world)
    local = {}
    for u in owned[rank]:
        m = sizes[u] // 10
        local[u] = np.stack([np.arange(m) * 100 + u, np.arange(m) * 100 + u + 50], axis=1).astype(np.int64)
    merged = shard.gather_intervals(local)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{str(k): v for k, v in merged.items()})
    # the same exchange from rows that never left the "device" (here: CPU tensors under Gloo)
    import torch

    rows = [np.concatenate([np.full((a.shape[0], 1), u, dtype=np.int64), a], axis=1) for u, a in sorted(local.items()) if a.shape[0]]
    rows_t = torch.from_numpy(np.concatenate(rows, axis=0) if rows else np.zeros((0, 3), dtype=np.int64))
    again = shard.gather_interval_rows(rows_t)
    assert sorted(again) == sorted(merged) and all(np.array_equal(again[u], merged[u]) for u in merged)
    # the second exchange takes ONE collective (a header row + the room remembered from the first) ...
    assert len(shard._gather_room) == 1
    once_more = shard.gather_interval_rows(rows_t)
    assert sorted(once_more) == sorted(merged) and all(np.array_equal(once_more[u], merged[u]) for u in merged)
    # ... a table that has outgrown the room falls back to two (every rank reads the same headers), and the next fits again
    big = torch.stack([torch.full((3000 + 7 * rank,), 100 + rank, dtype=torch.int64), torch.arange(3000 + 7 * rank),
                       torch.arange(3000 + 7 * rank) + 1], dim=1)
    for _ in range(2):
        grown = shard.gather_interval_rows(big)
        assert sorted(grown) == [100 + r for r in range(world)]
        for r in range(world):
            assert grown[100 + r].shape == (3000 + 7 * r, 2) and np.array_equal(grown[100 + r][:, 0], np.arange(3000 + 7 * r))
    none = shard.gather_interval_rows(torch.zeros((0, 3), dtype=torch.int64))
    assert none == {}
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_intervals_two_ranks(tmp_path):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sizes = [50, 40, 30, 20, 10, 5]
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    for g in got:
        for u, sz in enumerate(sizes):
            m = sz // 10
            if m == 0:
                assert str(u) not in g.files
                continue
            want = np.stack([np.arange(m) * 100 + u, np.arange(m) * 100 + u + 50], axis=1)
            assert np.array_equal(g[str(u)], want)


def _budget_worker(rank, world, port, out_dir):
    import json

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rocco_amd import budget, shard

    sizes = [50, 40, 30, 20, 10, 5]
    counts = [3.0, 40.0, 11.0, 26.5, 0.0, 88.0]
    totals = [1000.0, 1500.0, 800.0, 1200.0, 600.0, 2000.0]
    owned = shard.lpt_partition(sizes, world)
    local = {u: (counts[u], totals[u]) for u in owned[rank]}
    merged = shard.gather_budget_counts(local, len(sizes))
    # every rank pools the same pairs in the same (global) order and keeps its own chromosomes' budgets
    order = sorted(merged)
    budgets, meta = budget.estimate_empirical_bayes_budgets({f"u{u}": merged[u][0] for u in order},
                                                            {f"u{u}": merged[u][1] for u in order})
    with open(os.path.join(out_dir, f"budget{rank}.json"), "w") as fh:
        json.dump({"merged": {str(u): merged[u] for u in order}, "budgets": budgets, "alpha": meta["alpha"],
                   "owned": owned[rank]}, fh)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_budget_counts_exchange_two_ranks(tmp_path):
    """SURVEY.md section 8(e), exchange 1: all-gather of (budget_count_hat, total_count) per chromosome, then the same
    host-side empirical-Bayes fit on every rank."""
    import json

    import torch.multiprocessing as mp

    from rocco_amd import budget

    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_budget_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [json.load(open(tmp_path / f"budget{r}.json")) for r in range(2)]
    counts = [3.0, 40.0, 11.0, 26.5, 0.0, 88.0]
    totals = [1000.0, 1500.0, 800.0, 1200.0, 600.0, 2000.0]
    want, meta = budget.estimate_empirical_bayes_budgets({f"u{u}": counts[u] for u in range(6)}, {f"u{u}": totals[u] for u in range(6)})
    assert sorted(got[0]["owned"] + got[1]["owned"]) == list(range(6))
    for g in got:
        assert g["merged"] == {str(u): [counts[u], totals[u]] for u in range(6)}
        assert g["budgets"] == want and g["alpha"] == meta["alpha"]  # identical on every rank, equal to the one-process fit
