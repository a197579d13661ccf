"""Per-step kernel table from a rocprofv3 `--kernel-trace --stats` run of bench.py.

    python tools/prof_summary.py <kernel_stats.csv> [divisor-kernel-substring]

The number of executed steps is taken from the call count of a kernel that runs exactly once per step (default: the
optimizer's step-count kernel), so warm-up, graph capture and the instrumented eager steps are all accounted for."""
import csv
import sys


def main():
    path = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else 'step_increment'
    rows = list(csv.DictReader(open(path)))
    name_k = 'Name' if 'Name' in rows[0] else 'KernelName'
    def calls(r): return int(r.get('Calls') or r.get('Count'))
    def total_ns(r): return float(r.get('TotalDurationNs') or r.get('TotalDuration(ns)') or r['TotalDuration'])
    fw = [r for r in rows if key in r[name_k]]
    steps = calls(fw[0]) if fw else 1
    tot = sum(total_ns(r) for r in rows)
    print(f'steps executed: {steps}; total kernel time per step: {tot / steps / 1e3:.1f} us\n')
    print('| calls/step | us/step | avg us | % | kernel |\n|---|---|---|---|---|')
    for r in sorted(rows, key=lambda r: -total_ns(r))[:28]:
        t = total_ns(r)
        print(f'| {calls(r) / steps:.1f} | {t / steps / 1e3:.1f} | {t / calls(r) / 1e3:.1f} | {100 * t / tot:.2f} | `{r[name_k][:110]}` |')


if __name__ == '__main__':
    main()
