"""HBM-resident WaveEq datasets + device batch loader (data/wave_eq.py, vs_gather_windows) against the reference's own batches
(tests/golden/wave_loader.npz) and the CPU restatement: bit-exact -- the path only moves fp32 values."""
import os
import shutil

import numpy as np
import pytest
import torch

from test_data_cpu import replay, check_against_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def wave_dir():
    from oracle.wave_data_ref import write_fixture_set, fixture_dir
    d = fixture_dir() + 'gpu'
    shutil.rmtree(d, ignore_errors=True)
    write_fixture_set(d)
    from oracle.wave_data_ref import sorted_listdir
    with sorted_listdir():                                           # the golden batches were recorded under a sorted listing
        yield d
    shutil.rmtree(d, ignore_errors=True)


def _dataset(wave_dir, kind, train):
    from oracle.wave_data_ref import FIXTURE as f
    from spatiotemporal_variable_separation_amd.data.wave_eq import WaveEq, WaveEqPartial
    if kind == 'full':
        return WaveEq(wave_dir, f['nt_cond'], f['seq_len'], train, f['downsample'], device='cuda')
    return WaveEqPartial(wave_dir, f['nt_cond'], f['seq_len'], train, f['downsample'], f['n_pixels'], device='cuda')


@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('kind', ['full', 'partial'])
def test_device_loader_reproduces_reference_batches(wave_dir, kind, train):
    from oracle.wave_data_ref import FIXTURE as f
    from spatiotemporal_variable_separation_amd.data.wave_eq import DeviceBatchLoader
    ds = _dataset(wave_dir, kind, train)
    assert ds.all_data.is_cuda
    got = replay(ds, f['seed'], f['batch_size'], lambda d, bs: DeviceBatchLoader(d, bs, shuffle=True))
    check_against_golden('%s:%s' % (kind, 'train' if train else 'test'), got)


def test_gather_windows_every_item_and_bf16(wave_dir):
    """Every window of the set (also the ones `__len__` never reaches, wave_eq.py:62-65) against plain slicing of the CPU
    restatement's tensors; bf16 output = the rounded fp32 output; bad items are rejected on the host."""
    from oracle.wave_data_ref import FIXTURE as f, WaveEqRef
    ds = _dataset(wave_dir, 'full', True)
    ref = WaveEqRef(wave_dir, f['nt_cond'], f['seq_len'], True, f['downsample'])
    n_items = ds.size * ds.windows_per_seq
    cond, target = ds.batch(list(range(n_items)))
    for i in range(n_items):
        c, t = ref[i]
        assert torch.equal(cond[i].cpu(), c) and torch.equal(target[i].cpu(), t)
    c16, t16 = ds.batch(list(range(n_items)), out_dtype=torch.bfloat16)
    assert torch.equal(c16, cond.bfloat16()) and torch.equal(t16, target.bfloat16())
    with pytest.raises(IndexError):
        ds.batch([n_items])
    with pytest.raises(ValueError):
        from spatiotemporal_variable_separation_amd.data.wave_eq import WaveEq
        WaveEq(wave_dir, 3, 10 ** 6, True, f['downsample'], device='cuda')


def test_main_trains_on_a_resident_wave_set(wave_dir, tmp_path):
    """`python -m ...main --data wave --data_dir <dir>` end to end: dataset in HBM, device loader, MLP step."""
    from oracle.wave_data_ref import FIXTURE as f
    from spatiotemporal_variable_separation_amd import main as vmain
    xp = str(tmp_path / 'xp')
    vmain.main(['--xp_dir', xp, '--data_dir', wave_dir, '--data', 'wave', '--architecture', 'mlp', '--device', '0', '--nt_cond',
                str(f['nt_cond']), '--nt_pred', str(f['seq_len'] - f['nt_cond']), '--downsample', str(f['downsample']),
                '--batch_size', '4', '--epochs', '1', '--offset', '0', '--enc_hidden_size', '32', '--dec_hidden_size', '32',
                '--res_hidden_size', '64', '--code_size_t', '8', '--code_size_s', '8', '--mixing', 'mul', '--n_blocks', '1',
                '--seed', '5', '--num_workers', '1', '--log_interval', '1', '--chkpt_interval', '1'])
    for stem in ('ov_Et', 'ov_Es', 'decoder', 't_resnet'):
        assert os.path.exists(os.path.join(xp, stem + '.pt'))


def test_device_loader_sequential_drop_last_and_external_sampler(wave_dir):
    """Loader plumbing around the gather: sequential order, drop_last, and a caller-supplied sampler (what main() passes under
    torchrun: a DistributedSampler over the item indices)."""
    from torch.utils.data.distributed import DistributedSampler
    from spatiotemporal_variable_separation_amd.data.wave_eq import DeviceBatchLoader
    ds = _dataset(wave_dir, 'full', True)
    n = len(ds)
    seq = list(DeviceBatchLoader(ds, 4, shuffle=False))
    assert len(seq) == (n + 3) // 4 and sum(c.shape[0] for c, _ in seq) == n
    want_c, want_t = ds.batch(list(range(n)))
    assert torch.equal(torch.cat([c for c, _ in seq]), want_c) and torch.equal(torch.cat([t for _, t in seq]), want_t)
    dropped = DeviceBatchLoader(ds, 4, shuffle=False, drop_last=True)
    assert len(dropped) == n // 4 and all(c.shape[0] == 4 for c, _ in dropped)
    parts = []
    for rank in range(2):
        sampler = DistributedSampler(ds, num_replicas=2, rank=rank, shuffle=True, seed=11)
        items = list(sampler)
        got = torch.cat([c for c, _ in DeviceBatchLoader(ds, 3, sampler=sampler)])
        assert torch.equal(got, ds.batch(items)[0])
        parts.append(set(items))
    assert parts[0] | parts[1] == set(range(n))              # the two ranks cover the set between them
