"""Summarises the rocprofv3 PMC passes of one bench step into profiles/ (VERDICT r1 item 2c/2d).

    python scratch/pmc_summarise.py <tag> <image_size> <batch> <dir with one sub-directory per pass> [workload, default crowd]

Passes (each `rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir>/<pass> -- python3 bench.py --steps 1
--warmup 0 --no-cpu-baseline --no-roofline [--image-size S]`; counters never combined with other trace domains):
  fetch : FETCH_SIZE                      (KiB; x2 on gfx950 for wide coalesced reads, MI355X_MICROARCH.md "HBM")
  write : WRITE_SIZE                      (KiB)
  sq    : SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
Writes profiles/<tag>_pmc_per_kernel.md and adds / replaces the entry of (image size, batch, kernel source id) in
profiles/pmc_traffic.json, which bench.py reads for roofline.traffic."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import srgan_amd  # noqa: E402,F401
from srgan_amd import _build  # noqa: E402

CONTRACTION = ('gg_mfma_kernel', 'gg_direct_kernel', 'gg_rows_kernel', 'gg_dot_kernel', 'conv3x3_lds_kernel', 'conv3x3_wgrad_kernel',
               'pointwise_ksplit_kernel', 'pointwise_kernel', 'pointwise_ring_kernel', 'pointwise_wgrad_kernel', 'pointwise_wgrad_grouped_kernel',
               'conv3x3_wgrad_grouped_kernel', 'conv3x3_mixed_kernel', 'stem7x7_fwd_kernel', 'stem7x7_wgrad_kernel',
               'stem7x7_bwd_data_kernel', 'pointwise_wgrad_lds_kernel', 'pointwise_wgrad_lds_grouped_kernel',
               'hconv3x3_kernel', 'hconv3x3_dma_kernel', 'hwgrad3x3_kernel', 'hgemm_kernel', 'hlinear_wgrad_kernel', 'hconv2x2_kernel',
               'hwgrad4x4s2_kernel', 'hwgrad4x4s2_f32_kernel', 'conv3x3_mixed_small_kernel',
               # the finish / reduce launches a contraction call's bracket covers (their partial-tile traffic belongs to the call)
               'hwgrad3x3_finish_kernel', 'hwgrad4x4s2_finish_kernel', 'pointwise_wgrad_grouped_finish_kernel',
               'conv3x3_wgrad_grouped_finish_kernel', 'stem7x7_wgrad_finish_kernel', 'gg_reduce_partials_kernel',
               'gg_reduce_partials_wide_kernel')
HELPERS = ('_finish_kernel', 'gg_reduce_partials')        # counted in the bytes, not in the launches


def family(name):
    name = name.split('(')[0].replace('void ', '').replace('srgan::', '')
    return name.split('<')[0]


def read(directory):
    """{kernel family: {counter: [sum, launches]}}; the pseudo counter 'duration_ns' sums End - Start once per dispatch
    (durations under counter collection are longer than un-profiled ones: use them only against counters of the same pass)."""
    counters = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    seen = set()
    for path in glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(path)):
            kernel = family(row['Kernel_Name'])
            entry = counters[kernel][row['Counter_Name']]
            entry[0] += float(row['Counter_Value'])
            entry[1] += 1
            if row['Dispatch_Id'] not in seen:
                seen.add(row['Dispatch_Id'])
                duration = counters[kernel]['duration_ns']
                duration[0] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
                duration[1] += 1
    return counters


def main():
    tag, image_size, batch, base = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    workload = sys.argv[5] if len(sys.argv) > 5 else 'crowd'
    fetch, write, sq = (read(os.path.join(base, name)) for name in ('fetch', 'write', 'sq'))
    kernels = sorted(set(fetch) | set(write) | set(sq))
    lines = [f'PMC summary of ONE training step (workload {workload}, {image_size}x{image_size}, batch {batch}; setup kernels of the process included in the '
             'non-contraction rows), kernel sources ' + _build.source_id() + '.', '',
             'HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE reports half of a wide coalesced read). MFMA busy = '
             'SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe cycles summed over the 1024 SIMDs; = 64 x the v_mfma_f32_32x32x2_f32 count) / '
             '(1024 x the kernel\'s summed duration in the same pass x 2.4 GHz): a LOWER bound, the clock under load is 1.9-2.1 GHz. '
             'issue stall / parked / issuing = SQ_WAIT_INST_ANY / SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES.', '',
             '| kernel | launches | HBM GB (corrected) | FETCH KiB raw | WRITE KiB raw | MFMA busy | issue stall | parked | issuing |',
             '|---|---|---|---|---|---|---|---|---|']
    totals = {'fetch': 0.0, 'write': 0.0, 'launches': 0, 'other_fetch': 0.0, 'other_write': 0.0, 'other_launches': 0}
    rows = []
    for kernel in kernels:
        f = fetch[kernel].get('FETCH_SIZE', [0.0, 0])
        w = write[kernel].get('WRITE_SIZE', [0.0, 0])
        s = {name: value[0] for name, value in sq[kernel].items()}
        launches = max(f[1], w[1])
        hbm = (2 * f[0] + w[0]) * 1024
        wave = s.get('SQ_WAVE_CYCLES', 0.0)
        busy = 1024 * s.get('duration_ns', 0.0) * 2.4 / 4.0        # (the row below divides by 4 x busy)
        ratio = lambda a, b: f'{a / b:.3f}' if b else '-'
        rows.append((hbm, f'| {kernel} | {launches} | {hbm / 1e9:.2f} | {f[0]:.0f} | {w[0]:.0f} | '
                          f'{ratio(s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), 4 * busy)} | {ratio(s.get("SQ_WAIT_INST_ANY", 0.0), wave)} | '
                          f'{ratio(s.get("SQ_WAIT_ANY", 0.0), wave)} | {ratio(s.get("SQ_ACTIVE_INST_ANY", 0.0), wave)} |'))
        if any(kernel == c for c in CONTRACTION):
            totals['fetch'] += f[0]; totals['write'] += w[0]
            if not any(h in kernel for h in HELPERS):
                totals['launches'] += launches
        else:
            totals['other_fetch'] += f[0]; totals['other_write'] += w[0]; totals['other_launches'] += launches
    lines += [row for _, row in sorted(rows, reverse=True)]
    per_step = (2 * totals['fetch'] + totals['write']) * 1024
    lines += ['', f'Contraction kernels: {totals["launches"]} launches, {per_step / 1e9:.1f} GB per step = '
                  f'{per_step / max(totals["launches"], 1) / 1e6:.1f} MB per launch; other kernels '
                  f'{(2 * totals["other_fetch"] + totals["other_write"]) * 1024 / 1e9:.1f} GB over {totals["other_launches"]} launches.']
    with open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_per_kernel.md'), 'w') as handle:
        handle.write('\n'.join(lines) + '\n')
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        document = json.load(open(path))
    except (OSError, ValueError):
        document = {'note': 'HBM traffic of the contraction kernels of one training step from separate rocprofv3 --pmc FETCH_SIZE / '
                            '--pmc WRITE_SIZE passes (scratch/pmc_summarise.py); hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB. '
                            'bench.py uses an entry only for the same image size, batch and kernel sources.', 'entries': []}
    entry = {'workload': workload, 'image_size': image_size, 'batch_per_gpu': batch, 'kernel_source_id': _build.source_id(), 'source': f'profiles/{tag}_pmc_per_kernel.md',
             'fetch_size_kb_raw': totals['fetch'], 'write_size_kb_raw': totals['write'], 'launches': totals['launches'],
             'hbm_bytes_per_step': per_step, 'hbm_bytes_per_launch': per_step / max(totals['launches'], 1)}
    document['entries'] = [e for e in document['entries']
                           if (e.get('workload', 'crowd'), e['image_size'], e['batch_per_gpu'], e['kernel_source_id']) !=
                           (workload, image_size, batch, entry['kernel_source_id'])] + [entry]
    json.dump(document, open(path, 'w'), indent=1)
    print('\n'.join(lines[-3:]))


if __name__ == '__main__':
    main()
