import csv, sys, collections
groups = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0].replace('void srgan::', '').replace('srgan::', '')
    key = (name, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    g = groups[key]
    g[0] += 1
    g[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
steps = int(sys.argv[2])
want = sys.argv[3:] 
rows = sorted(groups.items(), key=lambda kv: -kv[1][1])
for (name, gx, gy, gz), (count, us) in rows:
    if want and not any(w in name for w in want):
        continue
    print(f'{us/steps/1e3:8.3f} ms/step {count/steps:7.1f} calls/step {us/count:9.1f} us  grid {gx},{gy},{gz}  {name[:70]}')
