"""How many kernels are in flight over time, from a rocprofv3 --kernel-trace CSV of a multi-stream run (the last ``--tail``
fraction of the trace): the union of the kernel intervals against the wall time, and the share of time with 0 / 1 / 2 / ...
kernels active.    python scratch/trace_concurrency.py <kernel_trace.csv> [--tail 0.5]"""
import csv, sys
rows = []
for row in csv.DictReader(open(sys.argv[1])):
    rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp'])))
rows.sort()
tail = float(sys.argv[sys.argv.index('--tail') + 1]) if '--tail' in sys.argv else 0.5
rows = rows[int(len(rows) * (1 - tail)):]
events = sorted([(s, 1) for s, e in rows] + [(e, -1) for s, e in rows])
first, last = events[0][0], events[-1][0]
active, previous, share = 0, first, {}
for time, delta in events:
    share[active] = share.get(active, 0) + time - previous
    previous, active = time, active + delta
span = last - first
print(f'{len(rows)} kernels over {span / 1e6:.2f} ms; sum of kernel durations {sum(e - s for s, e in rows) / 1e6:.2f} ms '
      f'(average {sum(e - s for s, e in rows) / span:.2f} in flight)')
for count in sorted(share):
    print(f'  {count} kernels in flight: {share[count] / 1e6:8.2f} ms ({share[count] / span:.1%})')
