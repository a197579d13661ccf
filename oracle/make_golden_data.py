"""Generate tests/golden/wave_loader.npz by running the REFERENCE's own WaveEq datasets (imported read-only from
/root/reference) behind a seeded `DataLoader(shuffle=True)`, the way main.py:113 builds it.

TEST INFRASTRUCTURE ONLY; runs in the build container:   python -m oracle.make_golden_data

The simulation files are synthetic and deterministic (oracle.wave_data_ref.write_fixture_set), so the tests rebuild them
instead of committing them; the fixture holds what the reference made of them: dataset lengths, single items and the
first batches of a seeded epoch for `WaveEq` and `WaveEqPartial`, train and test split (with `os.listdir` sorted, see
oracle.wave_data_ref.sorted_listdir).  The CPU restatement
(oracle.wave_data_ref) is required to reproduce every recorded tensor bit for bit before the file is written.
"""
import os
import shutil
import sys

import numpy as np
import torch
from torch.utils.data import DataLoader

sys.dont_write_bytecode = True
REF = os.environ.get('VARSEP_REFERENCE', '/root/reference')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.wave_data_ref import FIXTURE, WaveEqRef, WaveEqPartialRef, write_fixture_set, fixture_dir, sorted_listdir   # noqa: E402


def record(out, tag, ds, seed, batch_size, n_batches=3):
    out[tag + ':len'] = np.int64(len(ds))
    for i in (0, 1, len(ds) - 1, len(ds) // 2):
        c, t = ds[i]
        out['%s:item%d:cond' % (tag, i)] = c.numpy().copy()
        out['%s:item%d:target' % (tag, i)] = t.numpy().copy()
    torch.manual_seed(seed)
    loader = DataLoader(ds, batch_size=batch_size, shuffle=True)
    out[tag + ':n_batches'] = np.int64(len(loader))
    last = None
    for b, (c, t) in enumerate(loader):
        if b < n_batches:
            out['%s:batch%d:cond' % (tag, b)] = c.numpy().copy()
            out['%s:batch%d:target' % (tag, b)] = t.numpy().copy()
        last = (c, t)
    out[tag + ':last:cond'] = last[0].numpy().copy()                 # the ragged final batch (drop_last=False)
    out[tag + ':last:target'] = last[1].numpy().copy()


def main():
    sys.path.insert(0, REF)
    from var_sep.data.wave_eq import WaveEq, WaveEqPartial
    d = fixture_dir()
    shutil.rmtree(d, ignore_errors=True)
    write_fixture_set(d)
    f = FIXTURE
    ref, mine = {}, {}
    for train in (True, False):
      with sorted_listdir():                                          # file order independent of the file system
        tag = 'train' if train else 'test'
        record(ref, 'full:' + tag, WaveEq(d, f['nt_cond'], f['seq_len'], train, f['downsample']), f['seed'], f['batch_size'])
        record(mine, 'full:' + tag, WaveEqRef(d, f['nt_cond'], f['seq_len'], train, f['downsample']), f['seed'], f['batch_size'])
        record(ref, 'partial:' + tag, WaveEqPartial(d, f['nt_cond'], f['seq_len'], train, f['downsample'], f['n_pixels']), f['seed'],
               f['batch_size'])
        record(mine, 'partial:' + tag, WaveEqPartialRef(d, f['nt_cond'], f['seq_len'], train, f['downsample'], f['n_pixels']), f['seed'],
               f['batch_size'])
    assert ref.keys() == mine.keys()
    for k in ref:
        assert np.array_equal(ref[k], mine[k]), 'oracle.wave_data_ref differs from the reference at ' + k
    path = os.path.join(ROOT, 'tests', 'golden', 'wave_loader.npz')
    np.savez_compressed(path, **ref)
    print('wrote %s: %d arrays, %.1f KB; oracle restatement identical' % (path, len(ref), os.path.getsize(path) / 1e3))
    shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    main()
