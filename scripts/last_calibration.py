"""Ad-hoc: kernels of the LAST calibration in a rocprofv3 kernel trace (from its last stats kernel on): start offset,
duration, gap to the previous kernel's end, name, workgroups; and the totals."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    return n.split('::')[-1][:40]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) for r in rows]
stats = [i for i, e in enumerate(ev) if 'stats' in e[2]]
start = stats[-2] if len(stats) >= 2 and ev[stats[-1]][0] - ev[stats[-2]][0] < 200000 else stats[-1]
# the stats pass is two kernels (partials, final): begin at the first of the last pair
tail = ev[start:]
t0 = tail[0][0]
prev_end = t0
busy = 0
for e in tail:
    print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:8.1f}  gap {(e[0] - prev_end) / 1e3:7.1f}  {e[2]:42s} wgs={e[3]}")
    busy += e[1] - e[0]
    prev_end = max(prev_end, e[1])
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels {len(tail)}, busy {busy / 1e3:.1f} us")
