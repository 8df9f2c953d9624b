"""Ad-hoc: print the kernels of a rocprofv3 --kernel-trace CSV in launch order (start offset, duration, name, grid)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = sys.argv[2] if len(sys.argv) > 2 else 'lean_eval'
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
def short(n):
    n = n.replace('void ', '')
    n = n.split('(')[0]
    return n.split('::')[-1][:34]
out = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r['Grid_Size_X'], r['Workgroup_Size_X']) for r in rows]
idx = [i for i, o in enumerate(out) if first in o[2]]
start = max(0, idx[0] - 2) if idx else 0
base = out[start][0]
for o in out[start:start + count]:
    print(f"{(o[0]-base)/1e3:9.1f} {(o[1]-o[0])/1e3:8.1f} {o[2]:36s} wgs={int(o[3])//max(1,int(o[4]))}")
