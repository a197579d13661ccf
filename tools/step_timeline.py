"""One replayed step of a rocprofv3 --kernel-trace CSV of bench.py as a timeline: start, end, duration, queue, grid, kernel.

    python tools/step_timeline.py <kernel_trace.csv> [step-index-from-the-end]"""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'step_increment_kernel' in r['Kernel_Name']]
    a, b = idx[-back - 1], idx[-back]
    t0 = int(rows[a]['Start_Timestamp'])
    for r in rows[a:b]:
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        n = re.sub(r'^void ', '', n)
        n = re.sub(r'at::native::', '', n)[:48]
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        g = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
        print(f"{s:8.1f} {e:8.1f} {e - s:7.1f} q{r['Queue_Id']} g{g}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} {n}")
    print('step span %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))


if __name__ == '__main__':
    main()
