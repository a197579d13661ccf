"""GPU: Moving MNIST generated on the device (`vs_moving_mnist_batch`) -- bit-exact against the frames the REFERENCE's generator
produced (tests/golden/moving_mnist.npz, oracle/make_golden_mmnist.py) and against the CPU restatement on fresh random draws."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import mmnist_ref
from golden_util import GOLDEN_DIR

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dataset(digits, frame, nt_cond, seq_len, max_speed, nd):
    from spatiotemporal_variable_separation_amd.data.moving_mnist import MovingMNIST
    return MovingMNIST(digits, frame, nt_cond, seq_len, max_speed, True, nd, True, device='cuda')


def test_device_generator_reproduces_reference_videos_bit_for_bit():
    z = np.load(os.path.join(GOLDEN_DIR, 'moving_mnist.npz'))
    digits = z['digits']
    for tag in sorted({k.split(':')[0] for k in z.files if ':' in k}):
        frame, nt_cond, seq_len, max_speed, nd, batch, seed = [int(v) for v in z[tag + ':params']]
        ds = _dataset(digits, frame, nt_cond, seq_len, max_speed, nd)
        np.random.seed(seed)                                       # the product draws from the global NumPy stream like the reference
        cond, target = ds.batch(batch)
        got = torch.cat([cond, target], dim=1).cpu().numpy()
        want = z[tag + ':frames_u8'].astype(np.float32) / 255
        assert got.shape == want.shape and got.dtype == np.float32
        assert np.array_equal(got, want), f'{tag}: {int((got != want).sum())} pixels differ from the reference generator'
        assert cond.shape[1] == nt_cond and target.shape[1] == seq_len - nt_cond


@pytest.mark.parametrize('frame,max_speed,nd,seq_len', [(64, 4, 2, 15), (64, 13, 4, 40), (40, 6, 3, 30)])
def test_device_generator_matches_cpu_restatement_on_random_draws(frame, max_speed, nd, seq_len):
    digits = mmnist_ref.blobs(n=20, seed=5)
    ds = _dataset(digits, frame, 5, seq_len, max_speed, nd)
    rng = np.random.RandomState(123)
    init = mmnist_ref.draw(len(digits), digits.shape[1:], frame, max_speed, nd, 64, rnd=rng.randint)
    got = ds.render(init).cpu().numpy()
    want = mmnist_ref.render(digits, init, seq_len, frame)
    assert np.array_equal(got, want), f'{int((got != want).sum())} pixels differ'
    lowp = ds.render(init, out_dtype=torch.bfloat16).float().cpu().numpy()
    assert np.array_equal(lowp, torch.from_numpy(want).to(torch.bfloat16).float().numpy())


def test_main_trains_on_device_generated_moving_mnist(tmp_path):
    cmd = [sys.executable, '-m', 'spatiotemporal_variable_separation_amd.main', '--xp_dir', str(tmp_path), '--data_dir', 'synthetic_digits',
           '--data', 'mnist', '--device', '0', '--epochs', '1', '--batch_size', '8', '--num_workers', '0', '--seed', '3', '--log_interval', '1',
           '--nt_cond', '2', '--nt_pred', '3', '--offset', '2', '--enc_hidden_size', '8', '--dec_hidden_size', '8', '--res_hidden_size', '16',
           '--code_size_s', '12', '--code_size_t', '6', '--precision', 'bf16', '--hip_graph']
    env = dict(os.environ, VARSEP_MMNIST_EPOCH_LEN='32')
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'frames/s' in r.stdout and (tmp_path / 'decoder.pt').exists()
