"""Derive profiles/rNN_pmc_median.json (the `traffic` figure of bench.py's roofline block) from the two PMC passes
of scripts/pmc_median.py:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 scripts/pmc_median.py
and the same with --pmc WRITE_SIZE (separate passes: the two counters do not fit one).  Units and the gfx950 correction
follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are KiB; FETCH_SIZE reports half of the bytes
of a coalesced streaming read on gfx950 -- checked here on the calibration launch, whose bytes are known.

    python scripts/pmc_derive.py <fetch counter_collection.csv> <write counter_collection.csv> <round tag, e.g. r02>
"""
import csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_csv, write_csv, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def rows(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if r["Counter_Name"] == counter and ("rocco::" in n) and ("median" in n or "copy_row" in n):
            out.append(r)
    return out


def keep(path, dst, counter):
    rs = rows(path, counter)
    with open(dst, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rs[0].keys()))
        w.writeheader()
        w.writerows(rs)
    return rs


f_rows = keep(fetch_csv, os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_size.csv"), "FETCH_SIZE")
w_rows = keep(write_csv, os.path.join(ROOT, "profiles", f"{tag}_pmc_write_size.csv"), "WRITE_SIZE")
cal_f = [r for r in f_rows if "copy_row" in r["Kernel_Name"]][0]
cal_w = [r for r in w_rows if "copy_row" in r["Kernel_Name"]][0]
n_cal = int(cal_f["Grid_Size"])
known = n_cal * 8
ratio_f = known / (float(cal_f["Counter_Value"]) * 1024.0)
ratio_w = known / (float(cal_w["Counter_Value"]) * 1024.0)
correction = 2.0  # the guide's gfx950 figure; the calibration ratio below must agree with it
assert abs(ratio_f - correction) < 0.01 and abs(ratio_w - 1.0) < 0.01, (ratio_f, ratio_w)
med_f = [r for r in f_rows if "median" in r["Kernel_Name"]]
med_w = [r for r in w_rows if "median" in r["Kernel_Name"]]
K = 100
launch_threads = sorted({int(r["Grid_Size"]) for r in med_f}, reverse=True)
per_step = len(launch_threads)
fetch_b = sum(float(r["Counter_Value"]) for r in med_f) * 1024.0 * correction / len(med_f)
write_b = sum(float(r["Counter_Value"]) for r in med_w) * 1024.0 / len(med_w)
sys.path.insert(0, ROOT)
from rocco_amd import synth  # noqa: E402

loci = sum(n for _, n in synth.chrom_loci(50, None))
alg = (8 * K + 8) * loci / per_step
out = {
    "kernel": "median_batch_kernel<double,100,true>",
    "workload": "the median launches of one benchmark step at N = 1 (whole genome, one launch per group of chromosomes)",
    "K": K, "loci_per_step": loci, "launches_per_step": per_step, "launch_grid_threads": launch_threads,
    "median_hip_sha256": hashlib.sha256(open(os.path.join(ROOT, "rocco_amd", "csrc", "median.hip"), "rb").read()).hexdigest(),
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 scripts/pmc_median.py  (and a second pass with --pmc WRITE_SIZE)",
    "fetch_correction": correction,
    "calibration": {"kernel": "copy_row_kernel<double> over 100e6 loci (8 B per lane, same access pattern)",
                    "known_read_bytes": known, "fetch_size_kib": float(cal_f["Counter_Value"]), "ratio_known_over_counter": ratio_f,
                    "known_written_bytes": known, "write_size_kib": float(cal_w["Counter_Value"]), "ratio_written": ratio_w},
    "hbm_read_bytes_per_launch": fetch_b, "hbm_written_bytes_per_launch": write_b,
    "traffic_bytes_per_launch": fetch_b + write_b, "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": (fetch_b + write_b) / alg,
}
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_median.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
