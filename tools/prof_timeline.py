"""Kernel timeline of ONE replayed training step from a rocprofv3 --kernel-trace CSV (bench.py run).

    python tools/prof_timeline.py <kernel_trace.csv> [step index]

The step is delimited by two consecutive launches of the optimizer's step-count kernel; overlapping kernels show a negative gap."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'step_increment' in r['Kernel_Name']]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2          # default: a step from the middle of the run
    a, b = idx[which], idx[which + 1]
    t0 = int(rows[a]['End_Timestamp'])
    prev_end, tot, busy_until, idle = t0, 0, t0, 0
    for r in rows[a + 1:b + 1]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        print(f'{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name[:110]}')
        tot += e - s
        if s > busy_until:
            idle += s - busy_until
        busy_until = max(busy_until, e)
        prev_end = e
    print(f'kernel time {tot / 1e3:.1f} us, idle {idle / 1e3:.1f} us, span {(busy_until - t0) / 1e3:.1f} us')


if __name__ == '__main__':
    main()
