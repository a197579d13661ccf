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

GEMM = re.compile(r'gemm_kernel<1, (?:\(anonymous namespace\)::)?Dense<1, (\d)>, (?:\(anonymous namespace\)::)?Dense<1, (\d)>,')
GEMM_BIG = re.compile(r'gemm_big_kernel<1, (\d), (\d),')          # 256x256 LDS-DMA ring tile
GEMM_GLDS = re.compile(r'gemm_glds_kernel<(\d), (\d),')          # 128x128 LDS-DMA tile
GEMM_MID = re.compile(r'gemm_mid_kernel<1, (\d), (\d), false, \d+, false>')      # 128x128 ring tile
GEMM_ADAM = re.compile(r'gemm_mid_kernel<1, (\d), (\d), false, \d+, true>')      # the same with the Adam epilogue (vs_gemm_adam)
FAMILIES = [
    (r'rollout_ws_kernel<\d+, true', 'vs_mlp_rollout_fwd<bf16>'),
    (r'rollout_ws_kernel<\d+, false', 'vs_mlp_rollout_bwd<bf16>'),
    (r'colsum_multi_kernel', 'vs_colsum_multi'),
    (r'adam_multi_kernel', 'vs_adam_multi'),
    (r'train_losses_fwd_kernel', 'vs_train_losses_fwd'),
    (r'train_losses_bwd_kernel', 'vs_train_losses_bwd'),
    (r'splitk_reduce_kernel', 'splitk_reduce'),
    (r'mix_codes_fwd_kernel', 'vs_mix_codes_fwd'),
    (r'mix_codes_bwd_kernel', 'vs_mix_codes_bwd'),
    (r'wgrad3_band_kernel', 'vs_conv3_wgrad_band<bf16> (row-band 3x3 weight gradient)'),
    (r'conv3_band_kernel', 'vs_conv3_band<bf16> (row-band 3x3)'),
    (r'conv3_img16_kernel', 'vs_conv3_img16<bf16> (few-maps 3x3)'),
    (r'bn_fwd_small_kernel', 'vs_bn_train_fwd_small(_slabs)'),
    (r'bn_bwd_small', 'vs_bn_act_bwd (one-launch forms)'),
    (r'slab_sum_kernel', 'vs_slab_sum'),
    (r'convt_k4s2_tap_kernel', 'vs_convT_fwd<bf16> (tap kernel)'),
    (r'conv_k3s1_tap_kernel', 'vs_conv_fwd<bf16> (3x3 tap kernel)'),
    (r'im2col_', 'im2col (column-matrix gathers)'),
    (r'bn_act_fwd_kernel', 'vs_bn_act_fwd'),
    (r'bn_bwd_', 'vs_bn_act_bwd'),
    (r'bn_stats_kernel', 'vs_bn_stats'),
    (r'gemm_kernel<1, .*ChanRows', 'vs_conv_wgrad<bf16> (GEMM part)'),
]


def family(name):
    m = GEMM_ADAM.search(name)
    if m:
        return 'vs_gemm_adam<bf16,%s%s>' % ('RS'[int(m.group(1))], 'RS'[int(m.group(2))])
    for pat in (GEMM, GEMM_BIG, GEMM_GLDS, GEMM_MID):
        m = pat.search(name)
        if m:
            return 'vs_gemm<bf16,%s%s>' % ('RS'[int(m.group(1))], 'RS'[int(m.group(2))])
    for pat, fam in FAMILIES:
        if re.search(pat, name):
            return fam
    return None


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
    out = {name: {'bytes_per_launch': (v[0] + v[1]) / v[2], 'fetch_bytes_per_launch_x2': v[0] / v[2], 'write_bytes_per_launch': v[1] / v[2],
                  'launches': v[2]} for name, v in fam.items()}
    out['_source'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `bench.py --no_graph --steps 6 --warmup 2`; FETCH_SIZE x2 (gfx950)'
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    with open(sys.argv[4], 'w') as md:
        md.write('# HBM-side traffic (rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE), bf16: %s\n\n' % (sys.argv[5] if len(sys.argv) > 5 else 'WaveEq'))
        md.write('FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE as reported.\n'
                 'Infinity-Cache hits are counted by these fabric-side counters, so this is traffic beyond L2, an upper bound on HBM bytes.\n\n')
        md.write('| kernel family | launches | fetch MB/launch (x2) | write MB/launch |\n|---|---|---|---|\n')
        for name, v in sorted(out.items(), key=lambda kv: -kv[1]['bytes_per_launch'] * kv[1]['launches'] if kv[0] != '_source' else 0):
            if name == '_source':
                continue
            md.write(f"| `{name}` | {v['launches']} | {v['fetch_bytes_per_launch_x2'] / 1e6:.1f} | {v['write_bytes_per_launch'] / 1e6:.1f} |\n")
        md.write('\n| kernel | launches | fetch MB/launch (x2) | write MB/launch |\n|---|---|---|---|\n')
        for tot, k, n, f, w in sorted(rows, reverse=True)[:16]:
            md.write(f'| `{k[:110]}` | {n} | {2 * f * 1024 / n / 1e6:.1f} | {w * 1024 / n / 1e6:.1f} |\n')


if __name__ == '__main__':
    main()
