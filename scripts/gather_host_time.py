"""Ad-hoc (round 5): the interval gather (rocco_amd.shard.gather_interval_rows) under Gloo on the CPU with N ranks holding a
genome's worth of rows between them (~2 500 intervals per chromosome as on the benchmark): host time per exchange.
    python scripts/gather_host_time.py 8"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
from rocco_amd import shard, synth


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    genome = synth.chrom_loci(50, None)
    owned = shard.lpt_partition([n for _, n in genome], world)[rank]
    rng = np.random.default_rng(rank)
    rows = []
    for unit in owned:
        m = genome[unit][1] // 2000
        start = np.sort(rng.integers(0, genome[unit][1] * 50, size=m))
        rows.append(np.stack([np.full(m, unit), start, start + 500], axis=1))
    rows_t = torch.from_numpy(np.concatenate(rows).astype(np.int64))
    times = []
    for _rep in range(12):
        dist.barrier()
        t0 = time.perf_counter()
        out = shard.gather_interval_rows(rows_t)
        times.append(1e3 * (time.perf_counter() - t0))
    if rank == 0:
        print(f"N={world}: {sum(len(v) for v in out.values())} rows of {len(out)} chromosomes gathered on every rank: "
              f"{np.median(times[2:]):.3f} ms per exchange (Gloo over loopback, CPU tensors; first two calls {times[0]:.2f}, {times[1]:.2f})", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    for world in [int(x) for x in (sys.argv[1:] or ["2", "4", "8"])]:
        mp.spawn(worker, args=(world, 29611 + world), nprocs=world, join=True)
