#!/usr/bin/env python3
"""bench.py - encoder throughput of the MI355X hot path on the configuration the BASELINE target is quoted on:
3840x2160 8-bit, LDB_high_efficiency operating point (configs/ldb_high_efficiency.cfg), qp 32.

Workload ("step"): one lock-step frame of S independent closed streams (sequence chunks - the only partition of this
path that is bit-exact, SURVEY.md 8e: stream s is exactly what the reference produces with -skip/-n for its chunk).
Warm-up steps include each stream's I frame; the timed K steps are the following P frames.  Defaults: warm-up 1 (the I
frame) + 4 timed P frames, the last of which searches all 4 reference frames of the operating point; the driver's
`--steps 20 --warmup 5` times P frames 5..24, all with 4 references + bi-prediction.  Inputs are resident in HBM before
the timed region: rank 0 generates the seeded clip, broadcasts it over RCCL (world > 1) and every rank cuts its chunks'
frames out of it ON THE GPU (torch ops) and hands them to the encoder device-to-device.
value = luma pixels coded by all ranks / max-over-ranks wall time.

The line is self-verifying: after the timed region the bitstreams of the first and the last stream of rank 0 (first
`--verify-frames` frames: header + I + P) and the reconstruction of the last of those frames are compared with a live run
of the reference encoder (oracle/_ref/Thorenc, the checker - never the thing measured); "bit_exact": false zeroes the metric.

  python bench.py --gpus N --steps K --warmup W [--streams S] [--width 3840 --height 2160]
For N > 1 launch with torch.distributed.run (one rank per GPU); streams are sharded across ranks with no data-path
collective in the timed region ("weak" scaling: S streams per GPU); RCCL carries the clip broadcast before it and the
ordered gather of the per-chunk bitstreams + the all-reduce of the bit/frame counts after it.
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
REF_ENC = os.path.join(ROOT, 'oracle', '_ref', 'Thorenc')
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


# ---- helpers that the CPU (gloo) tests exercise --------------------------------------------------
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


def broadcast_clip(clip_u8, dist, src=0):
    """Rank `src` owns the clip (1-D uint8 numpy array of all frames); every rank returns a torch uint8 tensor with the
    same bytes on its collective device.  This is the "scatter of the raw input" of SURVEY.md 8e: the chunks of all
    ranks are windows of this one clip."""
    import torch
    dev = _reduce_device(dist)
    n = torch.tensor([0 if clip_u8 is None else clip_u8.size], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if dist.get_rank() == src:
        t = torch.from_numpy(clip_u8).to(dev)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return t


def gather_bitstreams(local_streams, dist, dst=0):
    """Ordered gather of variable-length byte strings: rank r holds the bitstreams of its chunks (in chunk order); rank
    `dst` returns the list of ALL chunks' bitstreams in global chunk order (rank-major), the other ranks return None.
    Sizes travel in one all_gather, payloads padded to the largest rank total in a second one."""
    import torch
    dev = _reduce_device(dist)
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = torch.tensor([len(local_streams)], dtype=torch.int64, device=dev)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    maxn = max(int(c.item()) for c in all_counts)
    sizes = torch.zeros(maxn, dtype=torch.int64, device=dev)
    if local_streams:
        sizes[:len(local_streams)] = torch.tensor([len(b) for b in local_streams], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    maxb = max(1, max(int(s.sum().item()) for s in all_sizes))
    buf = torch.zeros(maxb, dtype=torch.uint8, device=dev)
    blob = b''.join(local_streams)
    if blob:
        buf[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    all_buf = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(all_buf, buf)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        raw = all_buf[r].cpu().numpy().tobytes()
        pos = 0
        for i in range(int(all_counts[r].item())):
            n = int(all_sizes[r][i].item())
            out.append(raw[pos:pos + n])
            pos += n
    return out


# ---- bitstream helpers ---------------------------------------------------------------------------
def stream_prefix(bits, nframes):
    """First `nframes` framed frames of a Thor stream file (each frame: 4-byte big-endian byte count + payload; the
    sequence header is part of frame 0's payload - enc/putbits.c:45-83)."""
    pos = 0
    for _ in range(nframes):
        if pos + 4 > len(bits):
            return None
        pos += 4 + int.from_bytes(bits[pos:pos + 4], 'big')
    return bits[:pos]


# ---- CPU legs (checker + reported baseline; run after the timed region, in parallel on host cores) --------------
def _ref_cmd(d, tag, w, h, qp, n, with_rec):
    cmd = [REF_ENC, '-cf', CFG, '-if', os.path.join(d, tag + '.yuv'), '-width', str(w), '-height', str(h), '-qp', str(qp),
           '-n', str(n), '-f', '30', '-of', os.path.join(d, f'{tag}_{n}.bit')]
    if with_rec:
        cmd += ['-rf', os.path.join(d, f'{tag}_{n}.yuv')]
    return cmd


def cpu_legs(verify, base_frames, w, h, qp, nv, want_baseline):
    """verify: {stream id: (frames, gpu_bits, gpu_rec_last)}.  Runs the reference on the first nv frames of each of
    those streams (bit-exactness check) and, for the reported CPU baseline, on a bounded sample: the top-left
    1920x1080 crop of stream-0's frames, n = k and n = k-1 frames (the difference isolates one P frame with
    min(4, k-1) references).  All runs are separate processes started together, one host core each."""
    res = {'bit_exact': None, 'checked': [], 'cpu_baseline': None}
    if not os.path.exists(REF_ENC):
        res['note'] = 'oracle/_ref/Thorenc not in the snapshot: bit-exactness not checked in this run'
        return res
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for sid, (frames, _, _) in verify.items():
            open(os.path.join(d, f'v{sid}.yuv'), 'wb').write(b''.join(f.tobytes() for f in frames[:nv]))
            procs.append(('v', sid, time.perf_counter(), subprocess.Popen(_ref_cmd(d, f'v{sid}', w, h, qp, nv, True), stdout=subprocess.DEVNULL)))
        kb = 0
        if want_baseline:
            cw, ch = min(w, 1920), min(h, 1080)
            kb = min(len(base_frames), 6)
            crop = []
            for f in base_frames[:kb]:
                Y = f[:w * h].reshape(h, w)[:ch, :cw]
                U = f[w * h:w * h + (w // 2) * (h // 2)].reshape(h // 2, w // 2)[:ch // 2, :cw // 2]
                V = f[w * h + (w // 2) * (h // 2):].reshape(h // 2, w // 2)[:ch // 2, :cw // 2]
                crop.append(np.concatenate([np.ascontiguousarray(Y).ravel(), np.ascontiguousarray(U).ravel(), np.ascontiguousarray(V).ravel()]))
            open(os.path.join(d, 'b.yuv'), 'wb').write(b''.join(c.tobytes() for c in crop))
            if kb >= 2:
                for n in (kb, kb - 1):
                    procs.append(('b', n, time.perf_counter(), subprocess.Popen(_ref_cmd(d, 'b', cw, ch, qp, n, False), stdout=subprocess.DEVNULL)))
        tb = {}
        pending = list(procs)
        while pending:  # poll so that every process gets its own wall time
            for it in list(pending):
                kind, key, t0, pr = it
                if pr.poll() is not None:
                    if pr.returncode != 0:
                        raise RuntimeError(f'reference encoder failed ({kind} {key})')
                    if kind == 'b':
                        tb[key] = time.perf_counter() - t0
                    pending.remove(it)
            time.sleep(0.05)
        ok = True
        for sid, (frames, gbits, grec) in verify.items():
            rbits = open(os.path.join(d, f'v{sid}_{nv}.bit'), 'rb').read()
            rrec = open(os.path.join(d, f'v{sid}_{nv}.yuv'), 'rb').read()
            fsz = w * h * 3 // 2
            same = stream_prefix(gbits, nv) == rbits
            if grec is not None:
                same = same and rrec[(nv - 1) * fsz:nv * fsz] == grec.tobytes()
            res['checked'].append({'stream': sid, 'frames': nv, 'bitstream_bytes': len(rbits), 'recon_checked': grec is not None, 'ok': bool(same)})
            ok = ok and same
        res['bit_exact'] = bool(ok)
        if want_baseline and kb >= 2:
            cw, ch = min(w, 1920), min(h, 1080)
            dt = max(tb[kb] - tb[kb - 1], 1e-9)
            res['cpu_baseline'] = {
                'value': round(cw * ch / dt / 1e6, 4), 'unit': 'Mpixels/s', 'cores': 1, 'kind': 'reference',
                'sample': f'top-left {cw}x{ch} crop of stream 0 of the same clip, P frame {kb - 1} ({min(4, kb - 1)} references): '
                          f't[{kb} frames] - t[{kb - 1} frames] = {tb[kb]:.1f} s - {tb[kb - 1]:.1f} s, Thorenc SIMD build, 1 thread per run'}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--streams', type=int, default=int(os.environ.get('THOR_BENCH_STREAMS', '96')), help='streams PER GPU')
    ap.add_argument('--width', type=int, default=3840)
    ap.add_argument('--height', type=int, default=2160)
    ap.add_argument('--qp', type=int, default=32)
    ap.add_argument('--verify-frames', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-verify', action='store_true')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if a.gpus != world:
        if rank == 0:
            print(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run --nnodes=1 '
                  f'--nproc-per-node {a.gpus} --master-addr 127.0.0.1 bench.py --gpus {a.gpus} ... (one rank per GPU)', file=sys.stderr)
        sys.exit(2)
    dist = None
    import torch
    have_gpu = torch.cuda.is_available()  # without one thor_amd.Encoder() below aborts: the hot path has no CPU fallback
    dev = torch.device('cuda', local_rank) if have_gpu else torch.device('cpu')
    if have_gpu:
        torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group('nccl')

    import thor_amd
    from thor_amd import synth
    w, h, S = a.width, a.height, a.streams
    nframes = a.warmup + a.steps
    fsz = w * h * 3 // 2
    extra = 3
    # ---- input: rank 0 generates the clip; RCCL broadcast; chunks are cut on the GPU -------------------------------
    t_in = time.perf_counter()
    base = None
    if rank == 0:
        base = synth.make_clip(w, h, nframes + extra, 4 if w >= 3840 else 2, 2.0)  # BASELINE cfg 4 / cfg 2 content model and seed
        flat = np.concatenate([np.concatenate([p.ravel() for p in fr]) for fr in base])
    if dist is not None:
        clip_t = broadcast_clip(flat if rank == 0 else None, dist)
    else:
        clip_t = torch.from_numpy(flat).to(dev)
    dev_frames = clip_t.view(nframes + extra, fsz)
    p = thor_amd.load_config(CFG, width=w, height=h, qp=a.qp, f=30)
    enc = thor_amd.Encoder(p, S, device=local_rank)
    c2 = (w // 2) * (h // 2)

    def dev_stream_frame(sid, f):  # the torch restatement of synth.make_stream_frames (checked against it below)
        off = sid % max(1, (nframes + extra) - nframes + 1)
        mode = (sid // 3) % 4
        fr = dev_frames[off + f]
        Y, U, V = fr[:w * h].view(h, w), fr[w * h:w * h + c2].view(h // 2, w // 2), fr[w * h + c2:].view(h // 2, w // 2)
        dims = ([1] if mode & 1 else []) + ([0] if mode & 2 else [])
        if dims:
            Y, U, V = torch.flip(Y, dims), torch.flip(U, dims), torch.flip(V, dims)
        Y = torch.clamp(Y.to(torch.int16) + (sid % 5), 0, 255).to(torch.uint8)
        return torch.cat([Y.reshape(-1), U.reshape(-1), V.reshape(-1)])

    my_ids = [rank * S + s for s in range(S)]
    for s in range(S):
        for f in range(nframes):
            t = dev_stream_frame(my_ids[s], f)
            if have_gpu:
                torch.cuda.synchronize()
            enc.stage_device(s, f, t.data_ptr())
    verify = {}
    if rank == 0:
        for s in sorted({0, S - 1}):
            fr = synth.make_stream_frames(base, my_ids[s], nframes)
            assert np.array_equal(dev_stream_frame(my_ids[s], 0).cpu().numpy(), fr[0]), 'GPU chunk cutter differs from synth.make_stream_frames'
            verify[s] = [fr, None, None]
    t_in = time.perf_counter() - t_in
    nv = max(1, min(a.verify_frames, a.warmup if a.warmup > 0 else 1, nframes))

    def grab_recon(f):
        if rank == 0 and f == nv - 1 and not a.no_verify:
            for s in verify:
                verify[s][2] = enc.recon(s)

    for f in range(a.warmup):
        enc.encode_staged([f] * S)
        grab_recon(f)
    enc.kernel_time_reset()
    if dist is not None:
        dist.barrier()
    if have_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(a.warmup, nframes):
        enc.encode_staged([f] * S)   # blocks until all streams' bits are on the host
    if have_gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sb_ms, launches, filt_ms = enc.kernel_time()
    if dist is not None:
        dist.barrier()
        dt = reduce_max_time(dt, dist)
    total_px = float(w) * h * a.steps * S * world
    value = total_px / dt / 1e6

    # ---- output side: ordered gather of the chunk bitstreams to rank 0, totals by all-reduce ------------------------
    t_g = time.perf_counter()
    local_bits = [enc.bitstream(s) for s in range(S)]
    total_bytes = float(sum(len(b) for b in local_bits))
    total_frames = float(S * nframes)
    if dist is not None:
        allbits = gather_bitstreams(local_bits, dist)
        total_bytes = reduce_sum(total_bytes, dist)
        total_frames = reduce_sum(total_frames, dist)
        if rank == 0:
            assert len(allbits) == S * world and allbits[:S] == local_bits, 'ordered gather lost the chunk order'
    t_g = time.perf_counter() - t_g

    rc = 0
    if rank == 0:
        for s in verify:
            verify[s][1] = local_bits[s]
        legs = {'bit_exact': None, 'checked': [], 'cpu_baseline': None, 'note': 'verification disabled (--no-verify)'}
        if not a.no_verify or not a.no_cpu_baseline:
            legs = cpu_legs({} if a.no_verify else {my_ids[s]: tuple(v) for s, v in verify.items()}, verify[0][0], w, h, a.qp, nv,
                            not a.no_cpu_baseline and world == 1)
            if a.no_verify:
                legs['bit_exact'] = None
        if legs['bit_exact'] is False:
            value, rc = 0.0, 1
        # roofline of the dominant kernel (k_superblocks): algorithmic HBM bytes per luma pixel of a P frame with R
        # references = 1.5 * (1 orig + R refs + 1 rec) (block-path terms).  Frame f of a chunk (f >= 1)
        # references min(max_num_ref, f) earlier frames: average over the timed frames.
        R = sum(min(int(p.max_num_ref), f) for f in range(a.warmup, nframes)) / max(a.steps, 1)
        bytes_per_px = 1.5 * (2 + R)
        alg_bytes_per_launch = (w * h * a.steps * S * bytes_per_px) / max(launches, 1)
        avg_launch_s = (sb_ms / 1e3) / max(launches, 1)
        achieved = alg_bytes_per_launch / max(avg_launch_s, 1e-12) / 1e9
        # HBM traffic per launch from the committed PMC passes of this kernel (profiles/r02_pmc.json: FETCH_SIZE + WRITE_SIZE
        # with the guide's gfx950 corrections, collected in their own runs on a saturated workload of the same operating
        # point) scaled from bytes per luma pixel to the pixels one launch of THIS run covers; null if the file is missing.
        traffic, traffic_src = None, None
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', 'r02_pmc.json')))
            traffic = round((pm['fetch_bytes_per_px'] + pm['write_bytes_per_px']) * w * h * S * a.steps / max(launches, 1))
            traffic_src = 'profiles/r02_pmc.md (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE: %.0f + %.0f B per luma pixel on: %s)' % (
                pm['fetch_bytes_per_px'], pm['write_bytes_per_px'], pm['workload'])
        except (OSError, KeyError, ValueError):
            pass
        out = {
            'metric': 'encoder Mpixels/s at fixed qp, bit-exact recon vs ref', 'value': round(value, 3), 'unit': 'Mpixels/s',
            'fps': round(value * 1e6 / (w * h), 3),
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt * 1e3 / max(a.steps, 1), 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
            'bit_exact': legs['bit_exact'], 'bit_exact_checked': legs['checked'],
            'config': {'workload': f'{w}x{h} 8-bit 4:2:0, LDB_high_efficiency (configs/ldb_high_efficiency.cfg), qp {a.qp}, '
                                   f'{S} independent closed streams per GPU in lock step, timed frames = P frames {a.warmup}..{nframes - 1} of each stream '
                                   f'({R:.2f} references on average)',
                       'streams_per_gpu': S, 'frames_timed_per_stream': a.steps, 'parallelism': f'stream-sharded x{world}',
                       'per_stream_fps': round(a.steps / dt, 4), 'per_stream_mpx_s': round(w * h * a.steps / dt / 1e6, 4)},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 4), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 8), 'traffic': traffic, 'traffic_unit': 'bytes per launch', 'traffic_source': traffic_src,
                         'alg_bytes_per_launch': round(alg_bytes_per_launch),
                         'kernel': 'k_superblocks', 'launches': launches, 'avg_launch_ms': round(avg_launch_s * 1e3, 3),
                         'alg_bytes_per_px': round(bytes_per_px, 3),
                         'note': 'one persistent dependency-driven launch per frame; the path is latency/VALU-bound, not HBM-bound (SURVEY.md 0.7); '
                                 'filters+ref kernels took %.1f ms in the timed region' % filt_ms},
            'io': {'input_setup_s': round(t_in, 2), 'gather_s': round(t_g, 3), 'stream_bytes_total': int(total_bytes), 'frames_total': int(total_frames)},
        }
        if legs.get('note'):
            out['bit_exact_note'] = legs['note']
        if legs['cpu_baseline']:
            out['cpu_baseline'] = legs['cpu_baseline']
        print(json.dumps(out), flush=True)
    enc.close()
    if dist is not None:
        dist.barrier()   # the other ranks wait here while rank 0 runs the CPU legs (reference runs) above
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == '__main__':
    main()
