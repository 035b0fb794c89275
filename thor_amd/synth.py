#!/usr/bin/env python3
"""Seeded synthetic YUV 4:2:0 clip generator: the input of bench.py and of the tests.

Content model: 3-octave value-noise texture (cells 64/16/4 px, amplitudes
60/35/18), global sub-pel pan (1.37, 0.61) px/frame plus 0.2 %/frame zoom
(bilinear resample), three inverted-texture squares moving at non-integer
velocities, additive Gaussian noise sigma (luma) / sigma/2 (chroma) per frame.
Sub-pel motion + temporal noise keep P frames from being trivially early-skipped.

  python -m thor_amd.synth out.yuv W H FRAMES SEED [--sigma 2] [--bits 8]
"""
import argparse
import numpy as np


def _value_noise(rng, h, w, cell, amp):
    gh, gw = h // cell + 3, w // cell + 3
    g = rng.uniform(-1.0, 1.0, size=(gh, gw))
    y = np.arange(h) / cell
    x = np.arange(w) / cell
    y0 = y.astype(int); x0 = x.astype(int)
    fy = (y - y0)[:, None]; fx = (x - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)


def _bilinear(img, ys, xs):
    h, w = img.shape
    ys = np.clip(ys, 0, h - 1.001); xs = np.clip(xs, 0, w - 1.001)
    y0 = ys.astype(int); x0 = xs.astype(int)
    fy = ys - y0; fx = xs - x0
    return (img[y0, x0] * (1 - fx) + img[y0, x0 + 1] * fx) * (1 - fy) + \
           (img[y0 + 1, x0] * (1 - fx) + img[y0 + 1, x0 + 1] * fx) * fy


def make_clip(w, h, frames, seed, sigma=2.0, bits=8):
    rng = np.random.default_rng(seed)
    m = 96  # margin so that pan/zoom never leaves the canvas
    H, W = h + 2 * m + frames * 2, w + 2 * m + frames * 3
    tex = 128.0 + sum(_value_noise(rng, H, W, c, a) for c, a in ((64, 60), (16, 35), (4, 18)))
    cu = 128.0 + _value_noise(rng, H, W, 48, 40) + _value_noise(rng, H, W, 12, 12)
    cv = 128.0 + _value_noise(rng, H, W, 40, 40) + _value_noise(rng, H, W, 10, 12)
    sq = [(rng.uniform(0.1, 0.7) * h, rng.uniform(0.1, 0.7) * w,
           rng.uniform(-2.3, 2.3), rng.uniform(-3.1, 3.1), int(rng.integers(24, 96))) for _ in range(3)]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    scale = (1 << bits) / 256.0
    maxv = (1 << bits) - 1
    out = []
    for f in range(frames):
        z = 1.0 + 0.002 * f
        ys = m + 0.61 * f + (yy - h / 2) / z + h / 2
        xs = m + 1.37 * f + (xx - w / 2) / z + w / 2
        Y = _bilinear(tex, ys, xs)
        U = _bilinear(cu, ys, xs)
        V = _bilinear(cv, ys, xs)
        for (y0, x0, vy, vx, s) in sq:
            py = int(round(y0 + vy * f)) % max(1, h - s); px = int(round(x0 + vx * f)) % max(1, w - s)
            fy = (y0 + vy * f) - np.floor(y0 + vy * f)
            Y[py:py + s, px:px + s] = 255.0 - Y[py:py + s, px:px + s] * (1.0 - 0.1 * fy)
            U[py:py + s, px:px + s] = 255.0 - U[py:py + s, px:px + s]
        Y = Y + rng.normal(0.0, sigma, size=Y.shape)
        U = U[::2, ::2] + rng.normal(0.0, sigma / 2, size=(h // 2, w // 2))
        V = V[::2, ::2] + rng.normal(0.0, sigma / 2, size=(h // 2, w // 2))
        dt = np.uint8 if bits == 8 else np.dtype('<u2')
        out.append(tuple(np.clip(np.rint(p * scale), 0, maxv).astype(dt) for p in (Y, U, V)))
    return out


def make_stream_frames(base, sid, nframes):
    """Frames of stream `sid` of a multi-stream run (bench.py, tests): a window of the seeded base clip, flipped /
    offset per stream so that streams do not do identical work.  Returns nframes flat planar 4:2:0 uint8 frames."""
    off = sid % max(1, len(base) - nframes + 1)
    mode = (sid // 3) % 4
    out = []
    for f in range(nframes):
        Y, U, V = base[off + f]
        if mode & 1:
            Y, U, V = Y[:, ::-1], U[:, ::-1], V[:, ::-1]
        if mode & 2:
            Y, U, V = Y[::-1], U[::-1], V[::-1]
        d = sid % 5
        Y = np.clip(Y.astype(np.int16) + d, 0, 255).astype(np.uint8)
        out.append(np.concatenate([np.ascontiguousarray(Y).ravel(), np.ascontiguousarray(U).ravel(), np.ascontiguousarray(V).ravel()]))
    return out


def write_clip(path, clip):
    with open(path, 'wb') as f:
        for (Y, U, V) in clip:
            f.write(Y.tobytes()); f.write(U.tobytes()); f.write(V.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out'); ap.add_argument('w', type=int); ap.add_argument('h', type=int)
    ap.add_argument('frames', type=int); ap.add_argument('seed', type=int)
    ap.add_argument('--sigma', type=float, default=2.0); ap.add_argument('--bits', type=int, default=8)
    a = ap.parse_args()
    write_clip(a.out, make_clip(a.w, a.h, a.frames, a.seed, a.sigma, a.bits))


if __name__ == '__main__':
    main()
