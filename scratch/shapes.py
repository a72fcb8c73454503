"""Ranks the per-shape contraction table written by ``bench.py --shape-report`` and prints the per-kernel table
(ms, launches, executed GF, TF/s, fraction of the fp32 MFMA peak, algorithmic GB and GB/s) that profiles/*.md keep.

    python scratch/shapes.py gpurun_out/shapes.txt [top-n]
"""
import collections
import sys

KINDS = {0: 'gg_direct_kernel', 1: 'gg_mfma_kernel', 2: 'conv3x3_lds_kernel', 3: 'pointwise_kernel', 4: 'conv3x3_wgrad_kernel',
         5: 'gg_rows_kernel', 6: 'pointwise_wgrad_kernel', 8: 'pointwise_ksplit_kernel', 9: 'gg_dot_kernel',
         10: 'stem7x7_fwd_kernel', 11: 'stem7x7_wgrad_kernel', 12: 'stem7x7_bwd_data_kernel', 13: 'pointwise_ring_kernel',
         14: 'hconv3x3_kernel', 15: 'hwgrad3x3_kernel', 16: 'hgemm_kernel', 17: 'hlinear_wgrad_kernel', 18: 'hconv4x4s2_kernel',
         19: 'hwgrad4x4s2_kernel'}
PEAK = 157.3

rows = []
for line in open(sys.argv[1]):
    if line.startswith('#') or not line.strip():
        continue
    fields = line.split()
    M, N, K, kind, bm, bn, split, akf, bkf, count = map(int, fields[:10])
    ms = float(fields[10])
    nbytes = float(fields[11]) if len(fields) > 11 else 0.0
    rows.append((ms, M, N, K, kind, bm, bn, split, akf, bkf, count, 2.0 * M * N * K * count, nbytes))
total_ms, total_flops = sum(r[0] for r in rows), sum(r[11] for r in rows)
print(f'all contraction launches: {total_ms:.1f} ms, {total_flops / 1e9:.0f} GF, {total_flops / total_ms / 1e9:.1f} TF/s '
      f'= {total_flops / total_ms / 1e9 / PEAK:.3f} of {PEAK} TF/s, {sum(r[10] for r in rows)} launches\n')
print('| kernel | ms | launches | GF | TF/s | frac of fp32 MFMA peak | algorithmic GB | GB/s |')
print('|---|---|---|---|---|---|---|---|')
families = collections.defaultdict(lambda: [0.0, 0, 0.0, 0.0])
for r in rows:
    entry = families[r[4]]
    entry[0] += r[0]; entry[1] += r[10]; entry[2] += r[11]; entry[3] += r[12]
for kind, (ms, count, flops, nbytes) in sorted(families.items(), key=lambda kv: -kv[1][0]):
    print(f'| {KINDS.get(kind, kind)} | {ms:.2f} | {count} | {flops / 1e9:.0f} | {flops / ms / 1e9:.1f} | '
          f'{flops / ms / 1e9 / PEAK:.3f} | {nbytes / 1e9:.1f} | {nbytes / ms / 1e6:.0f} |')
rows.sort(reverse=True)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print(f'\ntop {top} (shape, kernel) pairs:\n')
print('   ms    %   cum%  count      M        N       K kernel                    bm   bn split akf bkf   TF/s  us/launch')
cumulative = 0.0
for ms, M, N, K, kind, bm, bn, split, akf, bkf, count, flops, nbytes in rows[:top]:
    cumulative += ms
    print(f'{ms:7.2f} {100 * ms / total_ms:5.1f} {100 * cumulative / total_ms:5.1f} {count:6d} {M:6d} {N:8d} {K:7d} '
          f'{str(KINDS.get(kind, kind)):24s} {bm:4d} {bn:4d} {split:4d} {akf:3d} {bkf:3d} {flops / ms / 1e9:6.1f} {1e3 * ms / count:9.1f}')
