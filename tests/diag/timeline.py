"""Print the kernel timeline of a few work items from a rocprofv3 kernel trace (st_kernel_trace.csv).
usage: python tests/diag/timeline.py trace.csv [n_from_end] [n_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 70
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:24], r['Stream_Id']) for r in rows)
idx = [i for i, e in enumerate(ev) if e[2] in ('k_pair',)]
i0 = idx[-back]
t0 = ev[i0][0]
for e in ev[max(i0 - 25, 0):i0 + n]:
    if e[1] - e[0] < 20000 and not e[2].startswith('k_'): continue
    print(f"{(e[0]-t0)/1e6:8.3f} {(e[1]-t0)/1e6:8.3f} {(e[1]-e[0])/1e6:7.3f} {e[2]:24s} s{e[3]}")
