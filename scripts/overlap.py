"""Ad-hoc: for the last solve in a kernel trace: wall span, summed kernel time, time with >= 1 / >= 2 kernels running."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
med = [i for i, r in enumerate(rows) if 'median_kernel' in r['Kernel_Name']]
start = med[-1]
while start - 1 in med: start -= 1
ev = []
tot = 0
for r in rows[start:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ev.append((s, 1)); ev.append((e, -1)); tot += e - s
ev.sort()
t_prev, depth, busy1, busy2 = ev[0][0], 0, 0, 0
for t, d in ev:
    if depth >= 1: busy1 += t - t_prev
    if depth >= 2: busy2 += t - t_prev
    depth += d; t_prev = t
span = ev[-1][0] - ev[0][0]
print(f"kernels {len(rows)-start}, span {span/1e6:.2f} ms, summed kernel time {tot/1e6:.2f} ms, >=1 running {busy1/1e6:.2f} ms, >=2 running {busy2/1e6:.2f} ms, idle {(span-busy1)/1e6:.2f} ms")
streams = {}
for r in rows[start:]:
    streams.setdefault(r.get('Queue_Id', r.get('Stream_Id', '?')), 0)
    streams[r.get('Queue_Id', r.get('Stream_Id', '?'))] += 1
print("kernels per queue:", streams)
