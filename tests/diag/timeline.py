"""Timeline of one steady-state bench step from a rocprofv3 --kernel-trace CSV: per kernel start / end relative to the step,
grouped by queue.  usage: timeline.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv, re, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'\bk_\w+', r['Kernel_Name'])
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(0) if m else r['Kernel_Name'][:24], r.get('Queue_Id', '?')))
rows.sort()
# a step starts with k_seed of a prefetch-less round 0 ... simpler: cut at k_init_state launches (one per batch swap)
cuts = [i for i, r in enumerate(rows) if r[2] == 'k_init_state']
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = cuts[-back - 1], cuts[-back]
t0 = rows[a][0]
print('step of %d kernels, %.2f ms' % (b - a, (rows[b][0] - t0) / 1e6))
big = [r for r in rows[a:b] if r[1] - r[0] > 150000]
for s, e, n, q in big:
    print('%7.2f - %7.2f  (%5.2f ms)  q%-3s %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
