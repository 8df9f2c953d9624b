"""One rank of the sharded composed driver (tests/test_gpu_composed_sharded.py): RANK / WORLD_SIZE / MASTER_* from the
environment, Gloo for the two small exchanges (every rank of the test shares the box's one GPU, which RCCL does not take),
the composed fixture `sys.argv[1]` of tests/golden/composed_vectors.npz, output `sys.argv[2]`."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist

    from rocco_amd import rocco as impl

    fixture, output, workdir = sys.argv[1], sys.argv[2], sys.argv[3]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "composed_vectors.npz"))
    chroms = [str(c) for c in gold[f"{fixture}_chroms"]]
    args = json.loads(str(gold[f"{fixture}_args"][0]))
    inputs = {c: (gold[f"{fixture}_{c}_intervals"], gold[f"{fixture}_{c}_matrix"]) for c in chroms}
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    os.makedirs(os.path.join(workdir, f"rank{rank}"), exist_ok=True)
    os.chdir(os.path.join(workdir, f"rank{rank}"))  # per-chromosome and summit files go to the working directory
    args["output"] = output
    args["narrowPeak"] = fixture.startswith("counts")
    final = impl.run_chromosomes(chroms + ["chrMissing"], inputs, args, run_id="31")
    left = sorted(os.listdir("."))
    print(json.dumps({"rank": rank, "final": final, "left_in_workdir": left}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
