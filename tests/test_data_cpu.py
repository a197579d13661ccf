"""The CPU restatement of the reference's WaveEq datasets (oracle/wave_data_ref.py) against the golden batches recorded from the
reference itself (tests/golden/wave_loader.npz, made by oracle/make_golden_data.py)."""
import os
import shutil

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'wave_loader.npz')


@pytest.fixture(scope='module')
def wave_dir():
    from oracle.wave_data_ref import write_fixture_set, fixture_dir
    d = fixture_dir()
    shutil.rmtree(d, ignore_errors=True)
    write_fixture_set(d)
    from oracle.wave_data_ref import sorted_listdir
    with sorted_listdir():                                           # the golden batches were recorded under a sorted listing
        yield d
    shutil.rmtree(d, ignore_errors=True)


def replay(ds, seed, batch_size, make_loader):
    """The quantities oracle/make_golden_data.record() stores, for any dataset / loader pair."""
    out = {'len': np.int64(len(ds))}
    for i in (0, 1, len(ds) - 1, len(ds) // 2):
        c, t = ds[i]
        out['item%d:cond' % i], out['item%d:target' % i] = c.cpu().numpy(), t.cpu().numpy()
    torch.manual_seed(seed)
    loader = make_loader(ds, batch_size)
    out['n_batches'] = np.int64(len(loader))
    last = None
    for b, (c, t) in enumerate(loader):
        if b < 3:
            out['batch%d:cond' % b], out['batch%d:target' % b] = c.cpu().numpy(), t.cpu().numpy()
        last = (c, t)
    out['last:cond'], out['last:target'] = last[0].cpu().numpy(), last[1].cpu().numpy()
    return out


def check_against_golden(tag, got):
    gold = np.load(GOLDEN)
    keys = [k for k in gold.files if k.startswith(tag + ':')]
    assert len(keys) == len(got) and keys
    for k in keys:
        g = got[k[len(tag) + 1:]]
        assert g.shape == gold[k].shape, k
        assert np.array_equal(g, gold[k]), k                        # slicing and min-max in fp32: bit-exact


@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('kind', ['full', 'partial'])
def test_oracle_wave_datasets_match_reference_batches(wave_dir, kind, train):
    from oracle.wave_data_ref import FIXTURE as f, WaveEqRef, WaveEqPartialRef
    if kind == 'full':
        ds = WaveEqRef(wave_dir, f['nt_cond'], f['seq_len'], train, f['downsample'])
    else:
        ds = WaveEqPartialRef(wave_dir, f['nt_cond'], f['seq_len'], train, f['downsample'], f['n_pixels'])
    got = replay(ds, f['seed'], f['batch_size'], lambda d, bs: DataLoader(d, batch_size=bs, shuffle=True))
    check_against_golden('%s:%s' % (kind, 'train' if train else 'test'), got)


def test_product_wave_dataset_refuses_cpu(wave_dir):
    from oracle.wave_data_ref import FIXTURE as f
    from spatiotemporal_variable_separation_amd.data.wave_eq import WaveEq
    from spatiotemporal_variable_separation_amd._lib import VarsepHipError
    with pytest.raises(VarsepHipError):
        WaveEq(wave_dir, f['nt_cond'], f['seq_len'], True, f['downsample'], device='cpu')


# ---- Moving MNIST: the oracle restatement against the frames the reference's own generator produced ------------------------------
def _mmnist_cases():
    import os
    from golden_util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'moving_mnist.npz'))
    tags = sorted({k.split(':')[0] for k in z.files if ':' in k})
    return z, tags


def test_moving_mnist_restatement_reproduces_reference_frames():
    from oracle import mmnist_ref
    z, tags = _mmnist_cases()
    digits = z['digits']
    assert np.array_equal(digits, mmnist_ref.blobs())
    for tag in tags:
        frame, nt_cond, seq_len, max_speed, nd, batch, seed = [int(v) for v in z[tag + ':params']]
        np.random.seed(seed)
        init = mmnist_ref.draw(len(digits), digits.shape[1:], frame, max_speed, nd, batch)
        assert np.array_equal(init, z[tag + ':init']), tag + ': draws differ from the reference\'s use of the NumPy stream'
        frames = mmnist_ref.render(digits, init, seq_len, frame)
        assert np.array_equal(frames, z[tag + ':frames_u8'].astype(np.float32) / 255), tag
