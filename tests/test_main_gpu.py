"""GPU: the training entry point end to end on synthetic data (CLI -> networks -> train loop -> checkpoints)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('extra', [
    ['--data', 'wave', '--architecture', 'mlp', '--nt_cond', '3', '--nt_pred', '4', '--offset', '3', '--code_size_t', '8',
     '--code_size_s', '8', '--mixing', 'mul', '--enc_hidden_size', '64', '--dec_hidden_size', '64', '--res_hidden_size', '32',
     '--n_blocks', '2', '--precision', 'bf16', '--hip_graph'],
    ['--data', 'mnist', '--nt_cond', '2', '--nt_pred', '3', '--offset', '2', '--enc_hidden_size', '8', '--dec_hidden_size', '8',
     '--res_hidden_size', '16', '--code_size_s', '12', '--code_size_t', '6', '--torch_amp', '--hip_graph'],
    ['--data', 'taxibj', '--architecture', 'vgg', '--nt_cond', '2', '--nt_pred', '2', '--offset', '2', '--enc_hidden_size', '8',
     '--dec_hidden_size', '8', '--res_hidden_size', '16', '--code_size_s', '12', '--code_size_t', '6', '--skipco'],
    ['--data', 'chairs', '--architecture', 'resnet', '--decoder_architecture', 'dcgan', '--nt_cond', '2', '--nt_pred', '2', '--offset', '2',
     '--dec_hidden_size', '8', '--res_hidden_size', '16', '--code_size_s', '12', '--code_size_t', '6', '--lamb_ae', '1', '--lamb_s', '1',
     '--precision', 'bf16'],
])
def test_main_trains_and_checkpoints(tmp_path, extra):
    cmd = [sys.executable, '-m', 'spatiotemporal_variable_separation_amd.main', '--xp_dir', str(tmp_path), '--data_dir',
           'synthetic', '--device', '0', '--epochs', '1', '--batch_size', '8', '--synthetic_len', '24', '--num_workers', '0',
           '--seed', '3', '--log_interval', '1', '--chkpt_interval', '1'] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'frames/s' in r.stdout
    for stem in ('ov_Et', 'ov_Es', 'decoder', 't_resnet'):
        assert (tmp_path / f'{stem}.pt').exists() and (tmp_path / f'{stem}_1.pt').exists()
    assert (tmp_path / 'params.json').exists()
