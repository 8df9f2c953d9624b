"""profiles/r05_pmc_count_path.json from the two PMC passes of scripts/pmc_count_path.sh: per kernel of the count path (rocco:: kernels
but the probe's own generator), HBM-side bytes per call -- FETCH_SIZE (KiB, x 2: gfx950 tallies the 128-byte requests of coalesced
streaming reads at 64, profiles/r03_pmc_median.json) and WRITE_SIZE (KiB) -- summed over the dispatches of the file and divided by its
number of calls (the whole-genome rolling launch runs once per call).
    python scripts/pmc_count_path_derive.py <fetch counter_collection.csv> <write counter_collection.csv>"""
import csv, json, re, sys

VALUES = 100 * 61765409  # K x loci of the probe's genome


def totals(path, counter):
    out, calls = {}, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if "rocco::" not in name or "synth_kernel" in name:
            continue
        short = re.sub(r"\(.*", "", name.replace("rocco::(anonymous namespace)::", "").replace("void ", ""))
        out[short] = out.get(short, 0.0) + float(r["Counter_Value"])
        if "wls_rolling_rows_kernel<8>" in short:
            calls += 1
    return out, calls


fetch, calls_f = totals(sys.argv[1], "FETCH_SIZE")
write, calls_w = totals(sys.argv[2], "WRITE_SIZE")
calls = max(1, calls_f)
assert calls_f == calls_w, (calls_f, calls_w)
kernels = {}
for name in sorted(set(fetch) | set(write), key=lambda k: -(2.0 * fetch.get(k, 0.0) + write.get(k, 0.0))):
    rd, wr = 2.0 * 1024.0 * fetch.get(name, 0.0) / calls, 1024.0 * write.get(name, 0.0) / calls
    kernels[name] = {"read_GB": round(rd / 1e9, 2), "written_GB": round(wr / 1e9, 2), "bytes_per_value": round((rd + wr) / VALUES, 1)}
total = sum(k["read_GB"] + k["written_GB"] for k in kernels.values())
print(json.dumps({"workload": "score_loci_wls_batch_device, 24 chromosomes, K = 100, 61765409 loci, one pipeline; per call", "calls_in_the_files": calls,
                  "fetch_correction": 2.0, "total_GB_per_call": round(total, 1), "bytes_per_value": round(total * 1e9 / VALUES, 1),
                  "bytes_per_value_by_the_passes_count": 192, "kernels": kernels}, indent=1))
