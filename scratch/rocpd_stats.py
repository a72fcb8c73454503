"""Per-kernel statistics (the rocprofv3 --stats table) from a rocprofv3 rocpd SQLite file, as CSV.
usage: rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import statistics
import sys

connection = sqlite3.connect(sys.argv[1])
durations = {}
for name, duration in connection.execute('select name, duration from kernels'):
    durations.setdefault(name, []).append(int(duration))
total = sum(sum(v) for v in durations.values())
rows = sorted(durations.items(), key=lambda item: -sum(item[1]))
with open(sys.argv[2], 'w', newline='') as handle:
    writer = csv.writer(handle, quoting=csv.QUOTE_NONNUMERIC)
    writer.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
    for name, values in rows:
        writer.writerow([name, len(values), sum(values), round(sum(values) / len(values), 3),
                         round(100.0 * sum(values) / total, 4), min(values), max(values),
                         round(statistics.pstdev(values), 3)])
print(f'{len(rows)} kernels, {sum(len(v) for v in durations.values())} dispatches, {total / 1e6:.1f} ms')
