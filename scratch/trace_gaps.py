"""GPU busy time and inter-kernel gaps from a rocprofv3 --kernel-trace CSV (the last ``--tail`` fraction of the trace, i.e.
steady-state steps).    python scratch/trace_gaps.py <kernel_trace.csv> [--tail 0.5]"""
import csv, sys
rows = []
for row in csv.DictReader(open(sys.argv[1])):
    rows.append((int(row['Start_Timestamp']), int(row['End_Timestamp']), row['Kernel_Name']))
rows.sort()
tail = float(sys.argv[sys.argv.index('--tail') + 1]) if '--tail' in sys.argv else 0.5
rows = rows[int(len(rows) * (1 - tail)):]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = [max(rows[i + 1][0] - rows[i][1], 0) for i in range(len(rows) - 1)]
print(f'{len(rows)} kernels over {span / 1e6:.2f} ms: kernel time {busy / 1e6:.2f} ms ({busy / span:.1%}), '
      f'gaps {sum(gaps) / 1e6:.2f} ms, mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us, median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us')
durations = sorted(e - s for s, e, _ in rows)
for q in (0.1, 0.25, 0.5, 0.75, 0.9):
    print(f'  duration q{int(q * 100)}: {durations[int(q * len(durations))] / 1e3:.1f} us')
short = [(e - s) for s, e, _ in rows if e - s < 10000]
print(f'  {len(short)} kernels shorter than 10 us: {sum(short) / 1e6:.2f} ms')
by = {}
for s, e, name in rows:
    key = name.split('(')[0][:70]
    c = by.setdefault(key, [0, 0])
    c[0] += 1; c[1] += e - s
for key, (count, total) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'  {total / 1e6:8.2f} ms {count:6d} x {total / count / 1e3:7.1f} us  {key}')
