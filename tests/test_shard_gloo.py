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
