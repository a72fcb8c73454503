"""Average of every collected counter per kernel name (rocprofv3 counter_collection.csv)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0].replace('void srgan::', '')[:60]
    a = acc[name][r['Counter_Name']]
    a[0] += float(r['Counter_Value']); a[1] += 1
for name, counters in acc.items():
    if not any(k in name for k in sys.argv[2:]):
        continue
    print(name)
    for c, (v, n) in sorted(counters.items()):
        print(f'   {c:32s} {v / n:16.1f}  (n={n})')
