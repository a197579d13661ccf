"""Moving MNIST generated on the device (reference: data/moving_mnist.py:39-341).

Same class name, constructor / `make_dataset` arguments and sampling semantics as the reference: every training item is a fresh
video of `num_digits` digits bouncing elastically inside a `nx` x `nx` frame, built from five draws of the GLOBAL NumPy stream
per digit (digit index, start row, start column, row speed, column speed -- `moving_mnist.py:121-123, 156-160`).  What differs is
where the work happens: the host only draws those integers (in the reference's order, so a seeded single-process run yields the
reference's videos bit for bit) and ONE kernel launch (`vs_moving_mnist_batch`) computes the trajectories and renders the whole
batch into HBM.  A 4-worker host generator tops out far below the > 100 k frames/s the MI355X training step consumes.

Digits come from the raw MNIST idx files under `data_dir` (the files torchvision downloads: `MNIST/raw/train-images-idx3-ubyte[.gz]`;
torchvision itself is not needed), or -- `data_dir='synthetic_digits'` -- from seeded 28x28 blobs for benchmarks and tests.
Only the deterministic variant (the one main.py constructs) is generated on the device.
"""
import gzip
import os

import numpy as np
import torch

from .. import _lib
from .._lib import check, dtype_code, require_cuda, stream_ptr


def synthetic_digits(n=256, size=28, seed=1234):
    """Seeded digit-like blobs (uint8 [n, size, size]): a few soft strokes on a dark background, MNIST's value range."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    out = np.zeros((n, size, size), dtype=np.float32)
    for i in range(n):
        img = np.zeros((size, size), dtype=np.float32)
        for _ in range(rng.randint(2, 5)):
            cy, cx = rng.uniform(6, size - 6, 2)
            ang, length, width = rng.uniform(0, np.pi), rng.uniform(4, 10), rng.uniform(1.0, 2.2)
            dy, dx = np.sin(ang), np.cos(ang)
            along = (yy - cy) * dy + (xx - cx) * dx
            across = -(yy - cy) * dx + (xx - cx) * dy
            img += np.exp(-(across / width) ** 2) * (np.abs(along) <= length)
        out[i] = np.clip(img, 0, 1) * 255.0
    return out.astype(np.uint8)


def read_mnist_images(data_dir, train=True):
    """uint8 [N, 28, 28] from the idx3-ubyte file torchvision's MNIST keeps under <data_dir>/MNIST/raw (gzip or plain)."""
    stem = 'train-images-idx3-ubyte' if train else 't10k-images-idx3-ubyte'
    for base in (os.path.join(data_dir, 'MNIST', 'raw'), data_dir):
        for name, opener in ((stem, open), (stem + '.gz', gzip.open)):
            path = os.path.join(base, name)
            if os.path.exists(path):
                with opener(path, 'rb') as f:
                    raw = f.read()
                magic, n, h, w = np.frombuffer(raw[:16], dtype='>i4')
                if magic != 2051:
                    raise ValueError('%s is not an idx3-ubyte image file' % path)
                return np.frombuffer(raw, dtype=np.uint8, offset=16).reshape(int(n), int(h), int(w)).copy()
    raise FileNotFoundError('no %s[.gz] under %s (expected the raw MNIST files; pass data_dir=synthetic_digits for seeded blobs)'
                            % (stem, data_dir))


class MovingMNIST:
    """Device-resident Moving MNIST.  `train=True`: videos generated per batch (`batch(B)` / DeviceMovingLoader); `train=False`: the
    precomputed test videos of the reference (`[s]mmnist_test_<n>digits_<nx>.npz`) kept in HBM."""
    eps = 1e-8
    device_resident = True
    generated_on_device = True
    rng = None                   # None: the global NumPy stream (the reference's); a np.random.RandomState: a private stream

    def __init__(self, data, nx, nt_cond, seq_len, max_speed, deterministic, num_digits, train, device='cuda'):
        self.frame_size, self.nt_cond, self.seq_len = nx, nt_cond, seq_len
        self.max_speed, self.deterministic, self.num_digits, self.train = max_speed, deterministic, num_digits, train
        self.device = torch.device(device)
        if train:
            if not deterministic:
                raise NotImplementedError('stochastic Moving MNIST redraws speeds inside the bounce loop (data-dependent RNG use); only the '
                                          'deterministic variant -- the one main.py trains on -- is generated on the device')
            arr = np.ascontiguousarray(np.stack([np.asarray(d, dtype=np.uint8) for d in data]) if not isinstance(data, np.ndarray) else data)
            assert arr.dtype == np.uint8 and arr.ndim == 3, 'digits: uint8 [N, h, w]'
            self.digit_shape = (int(arr.shape[1]), int(arr.shape[2]))
            self.n_source = int(arr.shape[0])
            self.data = torch.from_numpy(arr).to(self.device)
        else:
            self.data = torch.stack([torch.as_tensor(np.asarray(v), dtype=torch.float32) for v in data]).to(self.device)

    def __len__(self):
        return 200000 if self.train else int(self.data.shape[0])       # moving_mnist.py:104-111: arbitrary epoch length when training

    def draw(self, batch):
        """The reference's draws for `batch` consecutive items from the global NumPy stream (moving_mnist.py:121-123, 156-160)."""
        h, w = self.digit_shape
        rnd = (self.rng or np.random).randint
        init = np.empty((batch, self.num_digits, 5), dtype=np.int32)
        for b in range(batch):
            for n in range(self.num_digits):
                init[b, n, 0] = rnd(self.n_source)
                init[b, n, 1] = rnd(0, self.frame_size - h + 1)
                init[b, n, 2] = rnd(0, self.frame_size - w + 1)
                init[b, n, 3] = rnd(-self.max_speed, self.max_speed + 1)
                init[b, n, 4] = rnd(-self.max_speed, self.max_speed + 1)
        return init

    def render(self, init, out_dtype=torch.float32):
        """init int32 [B, num_digits, 5] (host array or device tensor) -> videos [B, seq_len, 1, nx, nx] on the device."""
        if not isinstance(init, torch.Tensor):
            init = torch.from_numpy(np.ascontiguousarray(init, dtype=np.int32)).to(self.device, non_blocking=True)
        require_cuda(init, self.data)
        B = init.shape[0]
        out = torch.empty((B, self.seq_len, 1, self.frame_size, self.frame_size), dtype=out_dtype, device=self.device)
        h, w = self.digit_shape
        check(_lib.load_library().vs_moving_mnist_batch(self.data.data_ptr(), self.n_source, h, w, init.data_ptr(), B, self.num_digits,
                                                        self.seq_len, self.frame_size, out.data_ptr(), dtype_code(out), stream_ptr()),
              'vs_moving_mnist_batch')
        return out

    def batch(self, batch, out_dtype=torch.float32):
        """(cond, target) of `batch` fresh videos (training) -- `__getitem__` of the reference for a whole batch."""
        x = self.render(self.draw(batch), out_dtype)
        return x[:, :self.nt_cond], x[:, self.nt_cond:]

    def __getitem__(self, index):
        if not self.train:
            v = self.data[index]
            return v[:self.nt_cond] / 255, v[self.nt_cond:self.seq_len] / 255
        cond, target = self.batch(1)
        return cond[0], target[0]

    @classmethod
    def make_dataset(cls, data_dir, nx, nt_cond, seq_len, max_speed, deterministic, num_digits, train, device='cuda', seed=None):
        if train:
            data = synthetic_digits(seed=seed or 1234) if data_dir == 'synthetic_digits' else read_mnist_images(data_dir, train=True)
        else:
            prefix = '' if deterministic else 's'
            dataset = np.load(os.path.join(data_dir, f'{prefix}mmnist_test_{num_digits}digits_{nx}.npz'), allow_pickle=True)
            sequences = dataset['sequences']
            data = [sequences[:, i].astype(np.single) for i in range(sequences.shape[1])]
        return cls(data, nx, nt_cond, seq_len, max_speed, deterministic, num_digits, train, device=device)


class DeviceMovingLoader:
    """`DataLoader(MovingMNIST, batch_size, shuffle=True)` of main.py:113 for the device generator: len(dataset) // batch_size
    (+1 ragged) batches per epoch, each rendered by one launch.  Item indices are irrelevant to a generator, so no sampler runs."""

    def __init__(self, dataset, batch_size, drop_last=False, out_dtype=torch.float32, world=1, epoch_len=None):
        self.dataset, self.batch_size, self.drop_last, self.out_dtype = dataset, batch_size, drop_last, out_dtype
        self.sampler = None
        self.n_items = (epoch_len if epoch_len is not None else len(dataset)) // max(world, 1)

    def __len__(self):
        n = self.n_items
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.n_items
        for start in range(0, n, self.batch_size):
            size = min(self.batch_size, n - start)
            if size < self.batch_size and self.drop_last:
                return
            yield self.dataset.batch(size, self.out_dtype)
