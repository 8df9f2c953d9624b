"""Ad-hoc: per-step kernel totals and the round timeline from a rocprofv3 kernel trace csv."""
import csv, sys
path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tot = {}
for r in rows:
    import re; nm = re.sub(r'^void |rocco::|\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:40]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = tot.setdefault(nm, [0, 0]); a[0] += d; a[1] += 1
for nm, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{nm:42s} {c:6d} calls {d/1e6/steps:9.2f} ms/step  avg {d/c/1e3:9.1f} us")
if len(sys.argv) > 3:
    # timeline of the last step: find the last 'median' burst start
    med = [i for i, r in enumerate(rows) if 'median_kernel' in r['Kernel_Name']]
    # bursts of 24 medians per step; walk back from the parity/baseline sections: print all apply launches after the 4th-from-last burst
    ap = [r for r in rows if 'fast_apply' in r['Kernel_Name'] or 'spine_kernel' in r['Kernel_Name']]
    t0 = None
    prev_end = None
    for r in ap[-int(sys.argv[3]):]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if t0 is None: t0 = s
        nm = 'spine' if 'spine' in r['Kernel_Name'] else 'apply'
        gap = (s - prev_end) / 1e3 if prev_end else 0
        print(f"t={(s-t0)/1e6:8.3f} {nm} {(e-s)/1e3:8.1f} us grid {int(r['Grid_Size_X'])//256 if 'Grid_Size_X' in r else r.get('Grid_Size')} gap {gap:8.1f}")
        prev_end = e
