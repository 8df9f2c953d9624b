"""`python bench.py --gpus 2` without a launcher starts two ranks by itself (torch.distributed.run, fresh
processes) and relays rank 0's line.  Run here with --rehearse (no device work: launch, LPT partition, interval
gather over Gloo, timing reduction and reporting only) -- the N > 1 host path of the driver's scaling run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=280)


@pytest.mark.timeout(300)
def test_gpus_2_starts_two_ranks_and_reports_them():
    proc = _run(["--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "1"])
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [t for t in proc.stdout.splitlines() if t.startswith("{")]
    assert len(lines) == 1  # one JSON line, rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["rehearsal"] is True
    assert line["parity_ranks_ok"] == 2  # every rank found its shortest chromosome's rows in the gathered table
    assert line["chromosomes_gathered"] == 24  # every rank's chromosomes arrived
    assert len(line["shard_loci"]) == 2 and sum(line["shard_loci"]) == 61765409
    assert max(line["shard_loci"]) <= 1.01 * 61765409 / 2  # LPT balance


def test_rank_count_mismatch_is_an_error():
    proc = _run(["--gpus", "2", "--rehearse"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert proc.returncode != 0 and "WORLD_SIZE=3" in (proc.stderr + proc.stdout)


def test_single_rank_needs_no_launcher():
    proc = _run(["--gpus", "1", "--rehearse", "--steps", "1", "--warmup", "0"])
    assert proc.returncode == 0
    line = json.loads([t for t in proc.stdout.splitlines() if t.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["parity_ranks_ok"] == 1
