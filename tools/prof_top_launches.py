"""Heaviest (kernel, grid) combinations of ONE training step from a rocprofv3 --kernel-trace CSV of bench.py."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'step_increment' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
agg, tot = {}, 0.0
for r in rows[a + 1:b + 1]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    key = (n[:95], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    agg.setdefault(key, [0, 0.0])
    agg[key][0] += 1
    agg[key][1] += d
print('kernel time of the step: %.1f us' % tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print('%4d x %8.1f us  %5.1f%%  grid %s,%s,%s  %s' % (v[0], v[1] / v[0], 100 * v[1] / tot, k[1], k[2], k[3], k[0]))
