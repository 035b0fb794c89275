#!/usr/bin/env python3
"""bench.py - encoder throughput of the MI355X hot path on BASELINE config 2:
1920x1080 8-bit, LDB_high_efficiency operating point (configs/ldb_high_efficiency.cfg), qp 32.

Workload ("step"): one lock-step frame of S independent closed streams (sequence chunks - the only
partition of this path that is bit-exact, SURVEY.md 8e: stream s is exactly what the reference
produces with -skip/-n for its chunk).  Warm-up steps include each stream's I frame; the timed K
steps are the following P frames (the low-delay GOP has one I frame per chunk).  Defaults: warm-up 1 (the I
frame) + 4 timed P frames, so that the last timed frame searches all 4 reference frames of the operating
point ("I+4P so all 4 refs are exercised").  Inputs are staged in HBM before the
timed region.  value = luma pixels coded by all ranks / max-over-ranks wall time.
S defaults to 1024 per GPU: a frame's superblock dependency chain is ~32 superblocks long and the
kernel keeps 3072 wavefronts resident, so fewer than ~730 streams cannot fill the chip.

  python bench.py --gpus N --steps K --warmup W [--streams S] [--width 1920 --height 1080]
For N > 1 launch with torch.distributed.run (one rank per GPU); streams are sharded across ranks
with no data-path collective ("weak" scaling: S streams per GPU).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
CFG = os.path.join(ROOT, 'configs', 'ldb_high_efficiency.cfg')
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


# ---- helpers that the CPU (gloo) test exercises --------------------------------------------------
def shard_streams(total, world):
    base, rem = divmod(total, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def stream_ids(total, world, rank):
    parts = shard_streams(total, world)
    start = sum(parts[:rank])
    return list(range(start, start + parts[rank]))


def _reduce_device(dist):
    """Collectives run on the GPU with RCCL (backend "nccl") and on the host with gloo (CPU tests)."""
    import torch
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')


def reduce_max_time(t, dist):
    import torch
    x = torch.tensor([t], dtype=torch.float64, device=_reduce_device(dist))
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return float(x.item())


def reduce_sum(v, dist):
    import torch
    x = torch.tensor([v], dtype=torch.float64, device=_reduce_device(dist))
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    return float(x.item())


# ---- synthetic input -------------------------------------------------------------------------------
def make_stream_frames(base, sid, nframes):
    """Frames of stream `sid`: a window of the seeded base clip, flipped / offset per stream so that
    streams do not do identical work."""
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


def cpu_baseline(frames, w, h, warmup, steps):
    """Reference encoder (oracle/_ref/Thorenc, SIMD path) on ONE host core: stream 0's first
    warmup+steps frames; the timed region is isolated as t(warmup+steps) - t(warmup)."""
    ref = os.path.join(ROOT, 'oracle', '_ref', 'Thorenc')
    if not os.path.exists(ref):
        return None
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 'in.yuv'), 'wb').write(b''.join(f.tobytes() for f in frames[:warmup + steps]))

        def run(n):
            t0 = time.perf_counter()
            subprocess.run([ref, '-cf', CFG, '-if', os.path.join(d, 'in.yuv'), '-width', str(w), '-height', str(h), '-qp', '32',
                            '-n', str(n), '-f', '30', '-of', os.path.join(d, 'o.bit')], check=True, stdout=subprocess.DEVNULL)
            return time.perf_counter() - t0
        t_all = run(warmup + steps)
        t_w = run(warmup) if warmup > 0 else 0.0
    dt = max(t_all - t_w, 1e-9)
    return {'value': round(w * h * steps / dt / 1e6, 4), 'unit': 'Mpixels/s', 'cores': 1, 'kind': 'reference',
            'sample': f'stream 0, frames {warmup}..{warmup + steps - 1} of the same {w}x{h} clip (t[{warmup + steps} frames] - t[{warmup} frames] = {dt:.2f} s), Thorenc SIMD build, 1 thread'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--streams', type=int, default=int(os.environ.get('THOR_BENCH_STREAMS', '1024')), help='streams PER GPU')
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--height', type=int, default=1080)
    ap.add_argument('--qp', type=int, default=32)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl')

    import thor_amd
    from thor_amd import synth
    w, h, S = a.width, a.height, a.streams
    nframes = a.warmup + a.steps
    base = synth.make_clip(w, h, nframes + 3, 2, 2.0)  # BASELINE cfg 2 content model, seed 2
    p = thor_amd.load_config(CFG, width=w, height=h, qp=a.qp, f=30)
    enc = thor_amd.Encoder(p, S, device=local_rank)
    first = None
    for s in range(S):
        fr = make_stream_frames(base, rank * S + s, nframes)
        if s == 0:
            first = fr
        for f in range(nframes):
            enc.stage(s, f, fr[f])
    for f in range(a.warmup):
        enc.encode_staged([f] * S)
    enc.kernel_time_reset()
    if dist is not None:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(a.warmup, nframes):
        enc.encode_staged([f] * S)   # blocks until all streams' bits are on the host
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sb_ms, launches, filt_ms = enc.kernel_time()
    if dist is not None:
        dist.barrier()
        dt = reduce_max_time(dt, dist)
    total_px = float(w) * h * a.steps * S * world
    value = total_px / dt / 1e6

    if rank == 0:
        # roofline of the dominant kernel (k_superblocks): algorithmic HBM bytes per luma pixel of a
        # P frame with R references = 1.5 * (1 orig + R refs + 1 rec) (block-path terms).
        # frame f of a chunk (f >= 1) references min(max_num_ref, f) earlier frames: average over the timed frames
        R = sum(min(int(p.max_num_ref), f) for f in range(a.warmup, nframes)) / max(a.steps, 1)
        bytes_per_px = 1.5 * (2 + R)
        alg_bytes_per_launch = (w * h * a.steps * S * bytes_per_px) / max(launches, 1)
        avg_launch_s = (sb_ms / 1e3) / max(launches, 1)
        achieved = alg_bytes_per_launch / max(avg_launch_s, 1e-12) / 1e9
        out = {
            'metric': 'encoder Mpixels/s at fixed qp, bit-exact recon vs ref', 'value': round(value, 3), 'unit': 'Mpixels/s',
            'fps': round(value * 1e6 / (w * h), 3),
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt * 1e3 / max(a.steps, 1), 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'config': {'workload': f'{w}x{h} 8-bit 4:2:0, LDB_high_efficiency (configs/ldb_high_efficiency.cfg), qp {a.qp}, '
                                   f'{S} independent closed streams per GPU in lock step, timed frames = P frames {a.warmup}..{nframes - 1} of each stream',
                       'streams_per_gpu': S, 'frames_timed_per_stream': a.steps, 'parallelism': f'stream-sharded x{world}'},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 4), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 8), 'traffic': None,
                         'kernel': 'k_superblocks', 'launches': launches, 'avg_launch_ms': round(avg_launch_s * 1e3, 3),
                         'alg_bytes_per_px': round(bytes_per_px, 3),
                         'note': 'one persistent dependency-driven launch per frame; path is latency/VALU-bound, not HBM-bound (SURVEY.md 0.7); filters+ref kernels took %.1f ms' % filt_ms},
        }
        if not a.no_cpu_baseline and world == 1:
            cb = cpu_baseline(first, w, h, a.warmup, a.steps)
            if cb:
                out['cpu_baseline'] = cb
        print(json.dumps(out), flush=True)
    enc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
