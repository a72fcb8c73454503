"""Per-STEP kernel statistics from two rocprofv3 --kernel-trace --stats runs of bench.py (--steps 3 and --steps 1, both
--warmup 0): (3 steps - 1 step) / 2, so that the set-up launches cancel.

    python scratch/per_step_stats.py gpurun_out/r02t_prof_s1/t_kernel_stats.csv gpurun_out/r02t_prof_s3/t_kernel_stats.csv > profiles/r02t_kernel_stats_per_step.md
"""
import csv, sys


def read(path):
    rows = {}
    for row in csv.DictReader(open(path)):
        rows[row['Name']] = (int(row['Calls']), float(row['TotalDurationNs']))
    return rows


one, three = read(sys.argv[1]), read(sys.argv[2])
table = []
for name, (calls, total) in three.items():
    c1, t1 = one.get(name, (0, 0.0))
    launches, ms = (calls - c1) / 2, (total - t1) / 2 / 1e6
    if launches > 0:
        table.append((ms, launches, name))
table.sort(reverse=True)
print('Per-STEP kernel statistics: rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 0` minus the same with '
      '`--steps 1`, halved\n(set-up launches -- parameter copies into the arenas, batch-norm constants -- cancel).\n')
print(f'Total: {sum(t[1] for t in table):.0f} launches, {sum(t[0] for t in table):.1f} ms of kernel time per step.\n')
print('| kernel | launches / step | ms / step | average us |\n|---|---|---|---|')
for ms, launches, name in table:
    print(f'| {name[:120]} | {launches:.1f} | {ms:.3f} | {1e3 * ms / launches:.1f} |')
