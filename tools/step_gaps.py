"""Idle time inside one replayed step of a rocprofv3 --kernel-trace CSV of bench.py: the step's span, the time during which at least one
kernel runs (union over all queues), the idle remainder, and the kernels that follow the largest / most frequent idle gaps.

    python tools/step_gaps.py <kernel_trace.csv> [step-index-from-the-end]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'at::native::', '', n).split('(')[0][:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'step_increment_kernel' in r['Kernel_Name']]
    a, b = idx[-back - 1], idx[-back]
    seg = rows[a + 1:b + 1]
    t0, t1 = int(rows[a]['End_Timestamp']), int(rows[b]['End_Timestamp'])
    busy, cur_end, gaps = 0, t0, defaultdict(lambda: [0, 0.0])
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > cur_end:
            g = gaps[short(r['Kernel_Name'])]
            g[0] += 1
            g[1] += (s - cur_end) / 1e3
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    span = (t1 - t0) / 1e3
    ksum = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3
    print('step span %.1f us, %d kernels, sum of kernel durations %.1f us, some kernel running %.1f us, idle %.1f us (%.1f %%)'
          % (span, len(seg), ksum, busy / 1e3, span - busy / 1e3, 100 * (span - busy / 1e3) / span))
    print('idle gaps by the kernel that ends them:')
    for k, (n, us) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print('%9.1f us %5d gaps %6.2f us avg  %s' % (us, n, us / n, k))


if __name__ == '__main__':
    main()
