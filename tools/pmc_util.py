"""Per-kernel MFMA / LDS / vector-memory counters from rocprofv3 --pmc passes of eager steps (`bench.py --no_graph`).

    python tools/pmc_util.py out.md out.json <label> pass1_counter_collection.csv [pass2.csv ...]

Every pass is one `rocprofv3 --pmc <counters> --output-format csv` run of the same command (tools/collect_mfma_util.sh); counters are
per dispatch, summed here per kernel symbol (and per (symbol, grid) for the convolution kernels, so that the layer shapes stay apart).
Derived columns (MI355X_MICROARCH.md, sections "rocprofv3 PMC slots" and "Per-instruction cycle constants"):

  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)      share of the chip's 1024 matrix pipes' cycles in which an MFMA
                executes (the counter is in cycles, summed over SIMDs: 32 per v_mfma_f32_32x32x16; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
  mfma_busy_cu = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)           the same over the cycles in which the CU holds a wave at all
  wait / stall / issue = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycle units, disjoint: parked at a
                wait or barrier / stalled at issue / issuing)
  lds_conf    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                     extra LDS-array cycles from bank conflicts
  lds_busy    = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 * 256)              share of the LDS arrays' cycles in use
  l1_hit      = 1 - TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum      vector-L1 hit rate (requests that did not go to L2)
  ta_busy     = TA_TA_BUSY_sum / (GRBM_GUI_ACTIVE / 8 * 256)                 share of the texture-address units' cycles busy
  l2_hit      = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporal_variable_separation_amd.profiling import group_of_kernel, source_sha  # noqa: E402

SPLIT_BY_GRID = re.compile(r'conv3_band_kernel|wgrad3_band_kernel|conv3_img16_kernel|convt_k4s2_tap_kernel|gemm_big_kernel|gemm_p8_kernel|gemm_mid_kernel|gemm_kernel<')


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0][:72]


def read(paths):
    tot = defaultdict(lambda: defaultdict(float))      # key -> counter -> sum
    n = defaultdict(lambda: defaultdict(int))          # key -> counter -> dispatches
    meta = {}
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = r['Kernel_Name']
            key = (k, r.get('Grid_Size', '')) if SPLIT_BY_GRID.search(k) else (k, '')
            c = r['Counter_Name']
            tot[key][c] += float(r['Counter_Value'])
            n[key][c] += 1
            if key not in meta:
                meta[key] = {'lds': r.get('LDS_Block_Size', ''), 'vgpr': r.get('VGPR_Count', ''), 'agpr': r.get('Accum_VGPR_Count', ''),
                             'wg': r.get('Workgroup_Size', '')}
    return tot, n, meta


def ratio(a, b):
    return a / b if b else float('nan')


def derive(c):
    g = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0            # cycles of the dispatch (per XCD)
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    out = {
        'mfma_busy': ratio(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), g * 1024),
        'mfma_busy_cu': ratio(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), 4 * c.get('SQ_BUSY_CU_CYCLES', 0.0)),
        'wait': ratio(c.get('SQ_WAIT_ANY', 0.0), wc),
        'stall': ratio(c.get('SQ_WAIT_INST_ANY', 0.0), wc),
        'issue': ratio(c.get('SQ_ACTIVE_INST_ANY', 0.0), wc),
        'lds_conf': ratio(c.get('SQ_LDS_BANK_CONFLICT', 0.0), c.get('SQ_LDS_IDX_ACTIVE', 0.0)),
        'lds_busy': ratio(c.get('SQ_LDS_IDX_ACTIVE', 0.0), g * 256),
        'l1_hit': 1.0 - ratio(c.get('TCP_TCC_READ_REQ_sum', 0.0), c.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0.0)),
        'ta_busy': ratio(c.get('TA_TA_BUSY_sum', 0.0), g * 256),
        'l2_hit': ratio(c.get('TCC_HIT_sum', 0.0), c.get('TCC_HIT_sum', 0.0) + c.get('TCC_MISS_sum', 0.0)),
        'mfma_per_wave': ratio(c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0) + c.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0.0), c.get('SQ_WAVES', 0.0)),
        'cycles': g,
    }
    return out


def main():
    out_md, out_json, label = sys.argv[1], sys.argv[2], sys.argv[3]
    tot, n, meta = read(sys.argv[4:])
    rows = []
    for key, c in tot.items():
        # per-dispatch means (every counter of a pass has the same dispatch count; passes may differ by warm-up launches)
        mean = {name: v / n[key][name] for name, v in c.items()}
        d = derive(mean)
        d['launches'] = max(n[key].values())
        d['kernel'], d['grid'] = key
        d['group'] = group_of_kernel(key[0]) or ''
        d.update(meta.get(key, {}))
        d['counters'] = mean
        rows.append(d)
    rows.sort(key=lambda r: -(r['cycles'] if r['cycles'] == r['cycles'] else 0) * r['launches'])
    json.dump({'_source': 'rocprofv3 --pmc passes of `bench.py --no_graph` (tools/collect_mfma_util.sh); per-dispatch means', '_source_sha': source_sha(),
               'label': label, 'rows': rows}, open(out_json, 'w'), indent=1)

    def f(x, pct=True):
        if x != x:
            return '-'
        return ('%.0f%%' % (100 * x)) if pct else ('%.2f' % x)
    with open(out_md, 'w') as md:
        md.write('# MFMA / LDS / vector-memory counters per kernel (rocprofv3 --pmc, eager steps): %s\n\n' % label)
        md.write(__doc__.split('Derived columns')[1].split('"""')[0].replace('\n  ', '\n* ').strip() + '\n\n')
        md.write('Kernels ordered by (cycles per launch x launches); convolution / GEMM symbols are listed per grid size (= per layer shape).\n\n')
        md.write('| kernel | grid | launches | kcycles/launch | mfma_busy | mfma_busy_cu | wait | stall | issue | lds_busy | lds_conf | ta_busy | l1_hit | l2_hit | LDS B | VGPR+AGPR |\n')
        md.write('|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n')
        for r in rows[:60]:
            md.write('| `%s` | %s | %d | %.1f | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s+%s |\n' % (
                short(r['kernel']), r['grid'], r['launches'], r['cycles'] / 1e3, f(r['mfma_busy']), f(r['mfma_busy_cu']), f(r['wait']), f(r['stall']),
                f(r['issue']), f(r['lds_busy']), f(r['lds_conf']), f(r['ta_busy']), f(r['l1_hit']), f(r['l2_hit']), r.get('lds', ''), r.get('vgpr', ''),
                r.get('agpr', '')))
        # per kernel group
        md.write('\n## per kernel group (cycle-weighted)\n\n| group | launches | Mcycles | mfma_busy | wait | stall | lds_busy | ta_busy | l1_hit | l2_hit |\n|---|---|---|---|---|---|---|---|---|---|\n')
        grp = defaultdict(lambda: defaultdict(float))
        for r in rows:
            g = grp[r['group'] or '(none)']
            for name, v in r['counters'].items():
                g[name] += v * r['launches']
            g['_launches'] += r['launches']
        for name, c in sorted(grp.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0.0)):
            d = derive(c)
            md.write('| `%s` | %d | %.1f | %s | %s | %s | %s | %s | %s | %s |\n' % (name, c['_launches'], d['cycles'] / 1e6, f(d['mfma_busy']), f(d['wait']),
                                                                                f(d['stall']), f(d['lds_busy']), f(d['ta_busy']), f(d['l1_hit']), f(d['l2_hit'])))


if __name__ == '__main__':
    main()
