"""Ad-hoc: timeline (kernel, duration, gap to the previous kernel's end) of the last solve in a kernel trace csv."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
med = [i for i, r in enumerate(rows) if 'median_kernel' in r['Kernel_Name']]
# the last burst of medians starts the last solve
start = med[-1]
while start - 1 in med: start -= 1
prev_end, t0 = None, int(rows[start]['Start_Timestamp'])
busy = 0
for r in rows[start:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = re.sub(r'^void |rocco::|\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:34]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    busy += e - s
    print(f"t={(s-t0)/1e3:9.1f} us  {nm:34s} {(e-s)/1e3:8.1f} us  gap {gap:7.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size'))}")
    prev_end = e
print(f"total {(prev_end-t0)/1e3:.1f} us, kernels busy {busy/1e3:.1f} us")
