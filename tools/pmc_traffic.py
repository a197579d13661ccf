"""HBM-side traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/f -o p -- python3 bench.py --no_graph --steps 6 --warmup 2 --no_cpu_baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir>/w -o p -- python3 bench.py --no_graph --steps 6 --warmup 2 --no_cpu_baseline
    python tools/pmc_traffic.py <dir>/f/p_counter_collection.csv <dir>/w/p_counter_collection.csv out.json out.md

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the counters are kilobytes at the
L2's memory side (Infinity-Cache hits included: an upper bound on HBM bytes); on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes, so it is doubled.  bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / launches."""
import csv
import json
import re
import sys
from collections import defaultdict

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from spatiotemporal_variable_separation_amd.profiling import group_of_kernel, source_sha  # noqa: E402


def family(name):
    return group_of_kernel(name)


def read(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        tot[r['Kernel_Name']] += float(r['Counter_Value'])
        n[r['Kernel_Name']] += 1
    return tot, n


def main():
    fetch, nf = read(sys.argv[1], 'FETCH_SIZE')
    write, _ = read(sys.argv[2], 'WRITE_SIZE')
    fam = defaultdict(lambda: [0.0, 0.0, 0])
    rows = []
    for k in fetch:
        f, w, n = fetch[k], write.get(k, 0.0), nf[k]
        rows.append((2 * f + w, k, n, f, w))
        name = family(k)
        if name:
            fam[name][0] += 2 * f * 1024
            fam[name][1] += w * 1024
            fam[name][2] += n
    # steps executed in the profiled run = launches of the once-per-step counter kernel
    steps = max([n for k, n in nf.items() if 'step_increment' in k] or [1])
    groups = {name: {'bytes_per_launch': (v[0] + v[1]) / v[2], 'fetch_bytes_per_launch_x2': v[0] / v[2], 'write_bytes_per_launch': v[1] / v[2],
                     'launches': v[2], 'bytes_per_step': (v[0] + v[1]) / steps} for name, v in fam.items()}
    meta = {'_source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --no_graph --steps 6 --warmup 2`; FETCH_SIZE x2 '
                       '(gfx950); kernel groups: spatiotemporal_variable_separation_amd/profiling.py', '_source_sha': source_sha(), '_steps': steps, 'groups': groups}
    json.dump(meta, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
    out = groups
    with open(sys.argv[4], 'w') as md:
        md.write('# HBM-side traffic (rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE), bf16: %s\n\n' % (sys.argv[5] if len(sys.argv) > 5 else 'WaveEq'))
        md.write('FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported.\n'
                 'Infinity-Cache hits are counted by these fabric-side counters, so this is traffic beyond L2, an upper bound on HBM bytes.\n\n')
        md.write('| kernel group | launches | fetch MB/launch (x2) | write MB/launch |\n|---|---|---|---|\n')
        for name, v in sorted(out.items(), key=lambda kv: -kv[1]['bytes_per_launch'] * kv[1]['launches']):
            md.write(f"| `{name}` | {v['launches']} | {v['fetch_bytes_per_launch_x2'] / 1e6:.1f} | {v['write_bytes_per_launch'] / 1e6:.1f} |\n")
        md.write('\n| kernel | launches | fetch MB/launch (x2) | write MB/launch |\n|---|---|---|---|\n')
        for tot, k, n, f, w in sorted(rows, reverse=True)[:16]:
            md.write(f'| `{k[:110]}` | {n} | {2 * f * 1024 / n / 1e6:.1f} | {w * 1024 / n / 1e6:.1f} |\n')


if __name__ == '__main__':
    main()
