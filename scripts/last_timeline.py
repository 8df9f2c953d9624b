"""Ad-hoc: print the LAST run of kernels that starts with `first` in a rocprofv3 kernel-trace CSV (start offset, duration,
gap to the previous kernel's end, name, workgroups)."""
import csv, sys, glob
path = sys.argv[1]
if not path.endswith('.csv'):
    path = sorted(glob.glob(path + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = sys.argv[2] if len(sys.argv) > 2 else 'stats_partial'
def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    return n.split('::')[-1][:40]
out = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) for r in rows]
idx = [i for i, o in enumerate(out) if first in o[2]]
start = idx[-1]
base = out[start][0]
prev_end = base
busy = 0
for o in out[start:]:
    print(f"{(o[0]-base)/1e3:9.1f} {(o[1]-o[0])/1e3:8.1f}  gap {(o[0]-prev_end)/1e3:7.1f}  {o[2]:42s} wgs={o[3]}")
    prev_end = o[1]
    busy += o[1] - o[0]
print(f"span {(prev_end-base)/1e3:.1f} us, kernels {len(out)-start}, busy {busy/1e3:.1f} us")
