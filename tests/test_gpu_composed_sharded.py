"""GPU: the composed driver over TWO ranks (rocco_amd.rocco.run_chromosomes inside a torch.distributed job: chromosomes
dealt to the ranks, exchange 1 = every chromosome's budget counts before the pooled fit, exchange 2 = the merged intervals
to rank 0) must write the bytes the REFERENCE's single-process composition wrote for the same
matrices (tests/golden/composed_vectors.npz).  Two worker processes share the box's GPU and talk over Gloo."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "composed_vectors.npz")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("fixture", ["bigwig", "counts_exact_log"])
def test_two_ranks_write_the_references_combined_bed(gpu, fixture, tmp_path):
    gold = np.load(GOLD)
    port = _free_port()
    output = str(tmp_path / "peaks.bed")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "tools", "sharded_driver_worker.py"), fixture, output,
                                       str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    reports = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err[-3000:]
        reports.append(json.loads(out.strip().splitlines()[-1]))
    assert sorted(r["rank"] for r in reports) == [0, 1]
    assert all(r["final"] == output for r in reports)
    assert all(r["left_in_workdir"] == [] for r in reports)  # per-chromosome and summit files removed on every rank
    assert open(output).read() == str(gold[f"{fixture}_combined_bed"][0])
