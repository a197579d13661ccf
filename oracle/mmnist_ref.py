"""CPU restatement of the reference's Moving-MNIST training generator.  TEST INFRASTRUCTURE ONLY (see cpu_ref.py): nothing under
`spatiotemporal_variable_separation_amd/` imports this file; tests and fixtures use it as the checker of `vs_moving_mnist_batch`.

Pinned: `oracle/make_golden_mmnist.py` runs the reference's own `MovingMNIST.__getitem__` (imported from /root/reference) on the
seeded blobs below and requires this restatement to reproduce every frame bit for bit before writing
`tests/golden/moving_mnist.npz`; `tests/test_data_cpu.py` replays the fixture against this file.

Follows data/moving_mnist.py (deterministic variant):
  draws           :121-123 (digit), :156-160 (start position, speed)     -> `draw`
  trajectory      :162-175                                               -> `trajectory`
  border bounces  :177-255, intersections :257-299                       -> `process_collision`
  compositing     :119-130 (add digit at its position, clip 255, /255)   -> `render`
"""
import numpy as np

EPS = 1e-8


def blobs(n=12, size=28, seed=77):
    """Deterministic digit stand-ins (uint8 [n, size, size]); hashed, no RNG stream involved."""
    from oracle.detdata import det_uniform
    u = det_uniform([n, size, size], seed).numpy()
    yy, xx = np.mgrid[0:size, 0:size]
    out = np.zeros((n, size, size), dtype=np.uint8)
    for i in range(n):
        cy, cx, r = 8 + (i * 5) % 12, 8 + (i * 7) % 12, 4 + i % 5
        mask = ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r) | ((np.abs(yy - 14) <= 1 + i % 3) & (xx > 3 + i % 4) & (xx < 24))
        out[i] = np.where(mask, 40 + np.floor(u[i] * 215), 0).astype(np.uint8)
    return out


def draw(n_source, digit_shape, frame, max_speed, num_digits, batch, rnd=None):
    """int32 [batch, num_digits, 5] = (digit, start row, start column, row speed, column speed): the reference's calls of the
    global np.random.randint, in its order, for `batch` consecutive items."""
    rnd = rnd or np.random.randint
    h, w = digit_shape
    init = np.empty((batch, num_digits, 5), dtype=np.int32)
    for b in range(batch):
        for n in range(num_digits):
            init[b, n] = (rnd(n_source), rnd(0, frame - h + 1), rnd(0, frame - w + 1), rnd(-max_speed, max_speed + 1),
                          rnd(-max_speed, max_speed + 1))
    return init


def _inter_x(a, b, x_lim, lo, hi):
    y = a * x_lim + b
    return (y >= lo - EPS) and (y <= hi + EPS), (x_lim, y)


def _inter_y(a, b, y_lim, lo, hi):
    x = (y_lim - b) / a
    return (x >= lo - EPS) and (x <= hi + EPS), (x, y_lim)


def process_collision(sx, sy, dx, dy, x_max, y_max):
    x_min = y_min = 0
    left, upper, right, bottom = sx < x_min - EPS, sy < y_min - EPS, sx > x_max + EPS, sy > y_max + EPS
    while left or right or upper or bottom:
        if dx == 0:
            cx, cy = (sx, y_min) if upper else (sx, y_max)
        elif dy == 0:
            cx, cy = (x_min, sy) if left else (x_max, sy)
        else:
            a = dy / dx
            b = sy - a * sx
            if left:
                left, n = _inter_x(a, b, x_min, y_min, y_max)
                if left:
                    cx, cy = n
            if right:
                right, n = _inter_x(a, b, x_max, y_min, y_max)
                if right:
                    cx, cy = n
            if upper:
                upper, n = _inter_y(a, b, y_min, x_min, x_max)
                if upper:
                    cx, cy = n
            if bottom:
                bottom, n = _inter_y(a, b, y_max, x_min, x_max)
                if bottom:
                    cx, cy = n
        p = ((sx - cx) / dx) if dx != 0 else ((sy - cy) / dy)
        if left:
            dx = abs(dx)
        if right:
            dx = -abs(dx)
        if upper:
            dy = abs(dy)
        if bottom:
            dy = -abs(dy)
        sx, sy = cx + dx * p, cy + dy * p
        left, upper, right, bottom = sx < x_min - EPS, sy < y_min - EPS, sx > x_max + EPS, sy > y_max + EPS
    return sx, sy, dx, dy


def trajectory(sx, sy, dx, dy, seq_len, x_max, y_max):
    sx, sy, dx, dy = int(sx), int(sy), int(dx), int(dy)
    out = []
    for _ in range(seq_len):
        sx, sy, dx, dy = process_collision(sx, sy, dx, dy, x_max, y_max)
        out.append((int(round(sx)), int(round(sy))))
        sy += dy
        sx += dx
    return out


def render(digits, init, seq_len, frame):
    """float32 [B, seq_len, 1, frame, frame] from uint8 digits [N, h, w] and the draws `init`."""
    B, nd = init.shape[0], init.shape[1]
    h, w = digits.shape[1], digits.shape[2]
    x = np.zeros((B, seq_len, 1, frame, frame), dtype=np.float32)
    for b in range(B):
        for n in range(nd):
            img = digits[init[b, n, 0]]
            for t, (px, py) in enumerate(trajectory(init[b, n, 1], init[b, n, 2], init[b, n, 3], init[b, n, 4], seq_len, frame - h, frame - w)):
                x[b, t, 0, px:px + h, py:py + w] += img
    x[x > 255] = 255
    return x / 255
