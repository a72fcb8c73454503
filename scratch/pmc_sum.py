"""Sum a rocprofv3 --pmc counter over the contraction kernels of a counter_collection.csv."""
import csv, sys
path, counter = sys.argv[1], sys.argv[2]
names = ('gg_mfma_kernel', 'gg_direct_kernel', 'gg_rows_kernel', 'gg_dot_kernel', 'conv3x3_lds_kernel', 'conv3x3_wgrad_kernel',
         'pointwise_ksplit_kernel',
         'pointwise_kernel', 'pointwise_ring_kernel', 'pointwise_wgrad_kernel', 'pointwise_wgrad_grouped_kernel', 'pointwise_wgrad_lds_',
         'conv3x3_wgrad_grouped_kernel', 'conv3x3_mixed_kernel', 'stem7x7_fwd_kernel', 'stem7x7_wgrad_kernel', 'stem7x7_bwd_data_kernel')
total, launches, other, other_launches = 0.0, 0, 0.0, 0
for r in csv.DictReader(open(path)):
    if r['Counter_Name'] != counter:
        continue
    v = float(r['Counter_Value'])
    if any(n in r['Kernel_Name'] for n in names):
        total += v; launches += 1
    else:
        other += v; other_launches += 1
print(f'{counter} contraction_sum {total} launches {launches} other_sum {other} other_launches {other_launches}')
