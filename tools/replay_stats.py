"""rocprofv3 `--kernel-trace --stats` table of a bench.py run -> per-step time of every kernel group of the REPLAYED step.

    cd /tmp && export TMPDIR=/tmp
    VARSEP_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o p -- python3 bench.py --config <w> \
        --no_cpu_baseline --extra_configs none
    python tools/replay_stats.py <dir>/.../p_kernel_stats.csv profiles/<round>_<w>_<dtype>_replay.json [label]

With VARSEP_BENCH_NO_EVENTS=1 bench.py runs no instrumented eager steps, so all but the recording's three warm-up steps of the launches in
the table are hipGraph replays.  Groups: spatiotemporal_variable_separation_amd/profiling.py.  bench.py attaches the result to its line
(`roofline.*`: duration from here, algorithmic FLOPs / bytes from its live accounting)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporal_variable_separation_amd.profiling import replay_table, source_sha  # noqa: E402


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps, table, rest = replay_table(rows)
    out = {'_source': 'rocprofv3 --kernel-trace --stats of `VARSEP_BENCH_NO_EVENTS=1 python3 bench.py ...` (%s): per group, sum of '
                      'TotalDurationNs over its kernel symbols / %d steps executed' % (sys.argv[3] if len(sys.argv) > 3 else os.path.basename(sys.argv[1]), steps),
           '_source_sha': source_sha(), '_steps': steps, '_unassigned_us_per_step': round(rest, 2), 'groups': {g: {k: round(v, 3) for k, v in e.items()} for g, e in table.items()}}
    json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
    tot = sum(e['us_per_step'] for e in table.values()) + rest
    print('steps %d, kernel time per step %.1f us (unassigned %.1f us)' % (steps, tot, rest))
    for g, e in sorted(table.items(), key=lambda kv: -kv[1]['us_per_step']):
        print('%-48s %9.1f us/step %7.1f launches/step %8.1f us avg  %5.1f %%' % (g, e['us_per_step'], e['launches_per_step'], e['avg_launch_us'],
                                                                              100 * e['us_per_step'] / tot))


if __name__ == '__main__':
    main()
