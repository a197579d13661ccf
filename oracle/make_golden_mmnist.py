"""Generate tests/golden/moving_mnist.npz by running the REFERENCE's own Moving-MNIST generator (`var_sep.data.moving_mnist.
MovingMNIST.__getitem__`, imported read-only from /root/reference) on seeded digit stand-ins.

TEST INFRASTRUCTURE ONLY; runs in the build container:   python -m oracle.make_golden_mmnist

The reference module does `from torchvision import datasets` at import time; torchvision is not installed offline and is only
used by `make_dataset(train=True)` to READ the MNIST files, which this script never calls (the dataset object is constructed
directly from the blobs, as `make_dataset` does after reading).  An empty placeholder module satisfies that import line; every
line that is exercised -- the draws, `_compute_trajectory`, `_process_collision`, the intersections and the compositing -- is the
reference's own code.  The restatement `oracle.mmnist_ref` must reproduce every frame bit for bit before the file is written.
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get('VARSEP_REFERENCE', '/root/reference')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import mmnist_ref      # noqa: E402

CASES = [
    # (tag, frame, nt_cond, seq_len, max_speed, num_digits, batch, seed)
    ('default', 64, 5, 15, 4, 2, 6, 4321),           # main.py:81-82: 64x64, max speed 4, 2 digits, nt_cond 5 + nt_pred 10
    ('fast3', 64, 3, 24, 9, 3, 4, 99),               # many bounces per step (speed 9 > remaining room), 3 digits, long horizon
    ('tight', 32, 2, 12, 4, 1, 4, 7),                # 28x28 digit in a 32x32 frame: a bounce almost every step
]


def main():
    if 'torchvision' not in sys.modules:             # see the module docstring
        tv = types.ModuleType('torchvision')
        tv.datasets = types.ModuleType('torchvision.datasets')
        sys.modules['torchvision'], sys.modules['torchvision.datasets'] = tv, tv.datasets
    sys.path.insert(0, REF)
    from var_sep.data.moving_mnist import MovingMNIST
    digits = mmnist_ref.blobs()
    out = {'digits': digits}
    for tag, frame, nt_cond, seq_len, max_speed, nd, batch, seed in CASES:
        ds = MovingMNIST([d for d in digits], frame, nt_cond, seq_len, max_speed, True, nd, True)
        np.random.seed(seed)
        vids = []
        for i in range(batch):
            c, t = ds[i]
            vids.append(np.concatenate([c.numpy(), t.numpy()], axis=0))
        ref = np.stack(vids)
        np.random.seed(seed)
        init = mmnist_ref.draw(len(digits), digits.shape[1:], frame, max_speed, nd, batch)
        mine = mmnist_ref.render(digits, init, seq_len, frame)
        assert mine.dtype == ref.dtype == np.float32 and np.array_equal(mine, ref), 'oracle.mmnist_ref differs from the reference: ' + tag
        out[tag + ':params'] = np.array([frame, nt_cond, seq_len, max_speed, nd, batch, seed], dtype=np.int64)
        out[tag + ':init'] = init
        out[tag + ':frames_u8'] = np.round(ref * 255).astype(np.uint8)       # exact: frames are k / 255 with integer k <= 255
        assert np.array_equal((out[tag + ':frames_u8'].astype(np.float32)) / 255, ref)
        nz = int((ref > 0).sum())
        print('%-8s %d videos of %d frames, %d non-zero pixels; oracle restatement identical' % (tag, batch, seq_len, nz))
    path = os.path.join(ROOT, 'tests', 'golden', 'moving_mnist.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1e3))


if __name__ == '__main__':
    main()
