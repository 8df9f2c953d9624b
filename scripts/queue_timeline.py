"""Ad-hoc: per-queue summary of the last step in a rocprofv3 kernel trace (the part after the last synth kernel,
last repetition): for every queue, first start / last end, busy time, kernels; and the median kernels' span."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'(\w+_kernel|__amd_\w+)', n)
    return m.group(1) if m else n[:30]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r['Queue_Id'], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) for r in rows]
# steps are delimited by the first median kernel after a gap: take the last 24 median kernels
med = [i for i, e in enumerate(ev) if e[2] == 'median_kernel']
last = med[-24:]
t0 = ev[last[0]][0]
tail = [e for e in ev if e[0] >= t0 - 1000]
print(f"median span: {(ev[last[-1]][1] - t0) / 1e3:.1f} us; sum of median kernel time {sum(ev[i][1] - ev[i][0] for i in last) / 1e3:.1f} us")
byq = collections.defaultdict(list)
for e in tail:
    byq[e[3]].append(e)
for q, es in byq.items():
    busy = sum(e[1] - e[0] for e in es)
    print(f"queue {q}: {len(es)} kernels, first start {(es[0][0] - t0) / 1e3:8.1f}, last end {(es[-1][1] - t0) / 1e3:8.1f}, busy {busy / 1e3:8.1f} us")
if len(sys.argv) > 2:
    for e in tail:
        print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f} q{e[3]} {e[2]:30s} wgs={e[4]}")
