#!/usr/bin/env python3
"""bench.py - encoder throughput of the MI355X hot path on the configuration the BASELINE target is quoted on:
3840x2160 8-bit, LDB_high_efficiency operating point (configs/ldb_high_efficiency.cfg), qp 32.

Workload ("step"): one lock-step frame of S independent closed streams (sequence chunks - the only partition of this
path that is bit-exact, SURVEY.md 8e: stream s is exactly what the reference produces with -skip/-n for its chunk).
160 streams per GPU: a stream's superblocks form a 62-step dependency chain at 3840x2160, so a frame takes at least 62
superblock times whatever the number of streams; the streams are coded in two groups half a frame apart so that the ramp-up of
one group's dependency wavefront fills the slots the other's ramp-down leaves (93-95 % of the 768 resident workgroups busy).
Measured in the driver's regime (round 6, call 12, same library): 128 streams ~124.7, 144 -> 126.9, 160 -> 128.2 Mpixels/s
(round 5 had 192 streams 3 % below 128): 160 is the default.
Warm-up steps include each stream's I frame; the timed K steps are the following frames in coding order.  Defaults:
warm-up 1 (the I frame) + 4 timed P frames, the last of which searches all 4 reference frames of the operating point; the
driver's `--steps 20 --warmup 5` times P frames 5..24, all with 4 references + bi-prediction.  Inputs are resident in HBM
before the timed region: every rank generates the seeded clip, cuts its chunks' frames out of it ON THE GPU (torch ops)
and hands them to the encoder device-to-device.
value = luma pixels coded by all ranks / max-over-ranks wall time.

The line is self-verifying.  BEFORE the timed region rank 0 starts live runs of the reference encoder
(oracle/_ref/Thorenc: the checker and the reported CPU baseline - never the thing measured) on the first, the middle and the
last of its streams; they run on host cores concurrently with the GPU work and cover `verify_frames` coded frames: at least
max(warmup + 2, 7), more when the reference can be expected to finish inside the GPU run anyway - i.e. they reach INTO the
timed region (with the driver's flags: I, P1..P4 and the first timed 4-reference frames).  The GPU bitstream prefix of those
frames and the reconstruction of EVERY one of them must equal the reference's; `bit_exact_scope` says which timed frames that
covers; "bit_exact": false zeroes the metric and the exit code is 1.  On top of the live runs EVERY RANK compares every coded frame -
the whole bitstream and the reconstruction of each frame, so all the timed frames - of those of its streams for which a reference run
was recorded in the build container (scripts/record_bench_refs.py -> tests/golden/bench_refs.json: the first, middle and last stream of rank 0 and ten more of its streams of the headline
workload with the driver's flags, and streams of every further rank up to 8 GPUs; the reference needs ~11 minutes per 25-frame 3840x2160
stream on one core, which no default run can wait for).  `--verify recorded` uses the records only (the RA / HDB16 operating points at
3840x2160), `--verify live` the live runs only.
cpu_baseline: CPU seconds (wait4 rusage) of reference runs taken when the processes exit; an implausible value is an error, not a number.
`value` is sampled on EXACTLY the timed frames: two runs of the top-left 1920x1080 crop of stream 0 with warmup + steps and with warmup frames
(LDB; a bounded sample - the full-size pair would take the GPU run's time several times over); the one-frame figure at the benched geometry
(the two live runs that differ by one coded frame) and the build container's full-size figure over the timed frames are reported beside it.

  python bench.py --gpus N --steps K --warmup W [--streams S] [--width 3840 --height 2160] [--config ldb|ra|hdb16]
                  [--bitdepth 8|10] [--sigma 2.0] [--verify auto|live|recorded] [--cpu-sample WxH] [--clip-frames F]
For N > 1 launch with torch.distributed.run (one rank per GPU); streams are sharded across ranks with no data-path
collective in the timed region ("weak" scaling: S streams per GPU); RCCL carries a consistency broadcast of the input
before it and the ordered gather of the per-chunk bitstreams + the all-reduce of the bit/frame counts after it.  A launch
through torch.distributed.run with ONE process runs the same collectives over RCCL on one GPU.  Rank 0 runs the reference
legs (verification + cpu_baseline) whatever N is.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
CONFIGS = {  # operating points of BASELINE.json: config file, default qp
    'ldb': ('ldb_high_efficiency.cfg', 32),
    'ra': ('ra_high_efficiency.cfg', 27),
    'hdb16': ('hdb16_high_efficiency.cfg', 32),
}
REF_ENC = os.path.join(ROOT, 'oracle', '_ref', 'Thorenc')
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
PMC_JSON = 'r06_pmc_bench.json'
# v_sad_u8: 4 sample differences per lane and instruction, 64 lanes, one wave64 VALU instruction per 4 clocks and SIMD,
# 1024 SIMDs at 2.4 GHz (MI355X_MICROARCH.md) -> pixel-differences per second the chip could accumulate
SAD_PEAK_PXOPS = 1024 * 2.4e9 / 4 * 64 * 4


# ---- helpers that the CPU (gloo) tests exercise --------------------------------------------------
def shard_streams(total, world):
    base, rem = divmod(total, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def stream_ids(total, world, rank):
    """Global ids of the streams (sequence chunks) rank `rank` encodes: contiguous, rank-major."""
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
    """Rank `src` owns the bytes (1-D uint8 numpy array); every rank returns a torch uint8 tensor with the same bytes on
    its collective device (the "scatter of the raw input" of SURVEY.md 8e)."""
    import torch
    dev = _reduce_device(dist)
    n = torch.tensor([0 if clip_u8 is None else clip_u8.size], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(clip_u8)).to(dev)
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


# ---- the streams of the workload (shared with scripts/record_bench_refs.py) ------------------------------------------------------
CONTENT_SEED = lambda config, w: {'ldb': 4 if w >= 3840 else 2, 'ra': 3, 'hdb16': 5}[config]   # cfg 4 / 2 / 3 / 5 content seeds
EXTRA_FRAMES = 3   # the base clip is this much longer than a chunk: streams start at different offsets


def host_stream_frames(base, sid, nframes, bitdepth):
    """Frames (flat planar 4:2:0, uint8 or little-endian uint16) of stream `sid`: a window of the seeded base clip, flipped / offset per
    stream so that the streams do different work (8-bit: thor_amd.synth.make_stream_frames; 10-bit: the same rule on uint16 samples)."""
    from thor_amd import synth
    if bitdepth == 8:
        return synth.make_stream_frames(base, sid, nframes)
    maxv = (1 << bitdepth) - 1
    off = sid % max(1, len(base) - nframes + 1)
    mode = (sid // 3) % 4
    out = []
    for f in range(nframes):
        Y, U, V = base[off + f]
        if mode & 1:
            Y, U, V = Y[:, ::-1], U[:, ::-1], V[:, ::-1]
        if mode & 2:
            Y, U, V = Y[::-1], U[::-1], V[::-1]
        Y = np.clip(Y.astype(np.int32) + (sid % 5), 0, maxv).astype('<u2')
        out.append(np.concatenate([np.ascontiguousarray(Y).ravel(), np.ascontiguousarray(U).ravel(), np.ascontiguousarray(V).ravel()]))
    return out


def ref_key(config, w, h, bitdepth, qp, nframes, sigma, sid):
    """Key of a recorded reference run (tests/golden/bench_refs.json, scripts/record_bench_refs.py)."""
    return f'{config}_{w}x{h}_{bitdepth}b_q{qp}_n{nframes}_sigma{sigma:g}_s{sid}'


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


# ---- CPU legs: the checker and the reported baseline (host processes, started BEFORE the timed region) ------------
def csrc_digest():
    """sha1 over the engine sources (thor_amd/csrc/*, include/*): identifies the library a profile was taken from (the GPU box
    has no .git; scripts/pmc_summary.py stamps the same digest into the PMC summaries bench.py reads)."""
    hsh = hashlib.sha1()
    for d in ('thor_amd/csrc', 'include'):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            hsh.update(f.encode())
            hsh.update(open(os.path.join(ROOT, d, f), 'rb').read())
    return hsh.hexdigest()[:16]


class CpuLegs:
    """Live runs of the reference encoder on host cores, one process per leg:
       'v<sid>'  stream sid, first nv coded frames, stream + reconstruction   (bit-exactness of the GPU path)
       'b'       stream of the first verify leg, nv - 1 frames                (1-core baseline: t[v] - t[b] = one frame at the
                                                                               benched geometry with every reference in use)
       'm<k>'    further streams, min(nv, 3) frames                           (N-process figure: N cores busy at once)
    The legs run while the GPU encodes.  Every process has a reaper thread that blocks in wait4(): the leg's wall time ends
    when the process EXITS (not when somebody polls it) and its CPU time (user + system, from the kernel's rusage) does not
    depend on scheduling at all; collect() joins the reapers after the timed region."""

    def __init__(self, cfg_path, w, h, qp, extra, total_frames, reordered):
        self.cfg, self.w, self.h, self.qp, self.extra = cfg_path, w, h, qp, list(extra)
        self.total, self.reordered = total_frames, reordered
        self.dir = tempfile.TemporaryDirectory()
        self.procs = []  # dicts: tag, n, t0, popen, thread, wall, cpu, rc
        self.ncores = os.cpu_count() or 1

    def available(self):
        return os.path.exists(REF_ENC)

    @staticmethod
    def _reap(it):
        _, status, ru = os.wait4(it['popen'].pid, 0)
        it['wall'] = time.perf_counter() - it['t0']
        it['cpu'] = ru.ru_utime + ru.ru_stime
        it['rc'] = os.waitstatus_to_exitcode(status)
        it['popen'].returncode = it['rc']   # the child has been reaped here: keep subprocess from waiting for it again

    def start(self, tag, frames_bytes, n, with_rec, geom=None):
        import threading
        d = self.dir.name
        path = os.path.join(d, tag + '.yuv')
        if not os.path.exists(path):
            with open(path, 'wb') as f:
                for fr in frames_bytes:
                    f.write(fr)
        gw, gh = geom or (self.w, self.h)
        cmd = [REF_ENC, '-cf', self.cfg, '-if', path, '-width', str(gw), '-height', str(gh), '-qp', str(self.qp),
               '-n', str(n), '-f', '30', '-of', os.path.join(d, f'{tag}_{n}.bit')] + self.extra
        if with_rec:
            cmd += ['-rf', os.path.join(d, f'{tag}_{n}.yuv')]
        it = {'tag': tag, 'n': n, 't0': time.perf_counter(), 'popen': subprocess.Popen(cmd, stdout=subprocess.DEVNULL),
              'wall': None, 'cpu': None, 'rc': None}
        it['thread'] = threading.Thread(target=self._reap, args=(it,), daemon=True)
        it['thread'].start()
        self.procs.append(it)

    def collect(self):
        """-> {(tag, n): (wall seconds until exit, CPU seconds)}"""
        for it in self.procs:
            it['thread'].join()
            if it['rc'] != 0:
                raise RuntimeError(f"reference encoder failed ({it['tag']}, -n {it['n']}): exit code {it['rc']}")
        return {(it['tag'], it['n']): (it['wall'], it['cpu']) for it in self.procs}

    def read(self, tag, n, what):
        return open(os.path.join(self.dir.name, f'{tag}_{n}.{what}'), 'rb').read()


def est_run_s(nframes, w, h, S):
    """Rough wall time of the GPU run (input staging + frames), for sizing the CPU legs that should end with it."""
    return nframes * w * h * S / 90e6 + 30.0


def file_md5(path):
    return hashlib.md5(open(path, 'rb').read()).hexdigest()


def load_bench_refs(cfg_path):
    """tests/golden/bench_refs.json: recorded reference runs {ref_key: record}.  Records carry the md5 of the configuration file and of the
    reference binary they were made with: a record made with another configuration file is dropped (it describes another workload); another
    reference binary (a rebuilt oracle/_ref) is reported in '_note'."""
    path = os.path.join(ROOT, 'tests', 'golden', 'bench_refs.json')
    if not os.path.exists(path):
        return {}
    refs = json.load(open(path))
    cfg_md5 = file_md5(cfg_path)
    enc_md5 = file_md5(REF_ENC) if os.path.exists(REF_ENC) else None
    out, other_bin = {}, 0
    for k, r in refs.items():
        if r.get('cfg_md5') not in (None, cfg_md5) and k.split('_')[0] == os.path.basename(cfg_path).split('_')[0]:
            continue
        if enc_md5 and r.get('ref_enc_md5') not in (None, enc_md5):
            other_bin += 1
        out[k] = r
    if other_bin:
        out['_note'] = f'{other_bin} records were made with another build of oracle/_ref/Thorenc than the one in this snapshot'
    return out


def one_frame_baseline(t_hi, t_lo, frame_px, what):
    """1-core figure from two legs that differ by one coded frame: (wall, cpu) of the longer and the shorter run.  CPU seconds
    (rusage) are the measure; the result must be a plausible single-core rate of this encoder - a difference that is not
    (a leg timed when it was polled instead of when it ended, a leg that lost its core) is an ERROR, never a number."""
    d_cpu, d_wall = t_hi[1] - t_lo[1], t_hi[0] - t_lo[0]
    per_frame_lo = t_lo[1] / max(what['frames_lo'], 1)
    if d_cpu <= 0 or d_cpu < 0.1 * per_frame_lo:
        raise RuntimeError(f'cpu_baseline: the two reference legs differ by {d_cpu:.3f} CPU-seconds for one more frame '
                           f'({t_hi[1]:.1f} s vs {t_lo[1]:.1f} s; the shorter run spends {per_frame_lo:.1f} s per frame): timing is broken')
    v = frame_px / d_cpu / 1e6
    if not 0.01 < v < 8.0:
        raise RuntimeError(f'cpu_baseline: {v:.3f} Mpixels/s is not a single-core rate of the reference encoder '
                           f'(t = {t_hi[1]:.2f} s - {t_lo[1]:.2f} s of CPU time)')
    return v, d_cpu, d_wall


def base_n_lo(reordered, warmup, nv):
    """Length of the shorter of the two full-geometry baseline legs.  Low delay: the warm-up frames, so that the difference to the nv-frame leg is the
    TIMED coded frames warmup..nv-1 at the benched geometry (round 6; it was nv - 1 = one frame); when the live leg does not reach into the timed
    region, one frame less.  Frame reordering: the I frame alone (the difference is every other frame of the chunk)."""
    if reordered:
        return 1
    return warmup if 1 <= warmup < nv else nv - 1


def cpu_baseline_object(times, base_tag, tf_legs, nv, n_ref, reordered, a, p, w, h, cw, ch, legs, world, my_ids, refs, qp, clipn):
    """The cpu_baseline object of the line from the (wall, CPU seconds) of the reference legs.  `value` is measured on the GPU box's host at the
    geometry of `geometry` - the benched one unless --cpu-sample asked for a crop -: the CPU seconds of the timed frames the live leg reaches."""
    nframes = a.warmup + a.steps
    n_lo = base_n_lo(reordered, a.warmup, nv)
    nproc = len(legs.procs)
    if not (n_lo >= 1 and n_lo < n_ref):
        raise RuntimeError(f'cpu_baseline needs two reference runs of different length (have {n_ref} frames)')
    t_hi, t_lo = times[(base_tag, n_ref)], times[(base_tag, n_lo)]
    v1, d_cpu, d_wall = one_frame_baseline(t_hi, t_lo, float(cw) * ch * (n_ref - n_lo), {'frames_lo': n_lo})
    # aggregate of all concurrently running reference processes of the benched geometry: pixels coded / wall of the slowest
    same_geom = {k: t for k, t in times.items() if k[0] != 't'}
    agg = sum(n for (_, n) in same_geom) * float(cw) * ch / max(t[0] for t in same_geom.values()) / 1e6
    what = ((f'the TIMED coded frames {n_lo}..{nv - 1}' if n_lo == a.warmup and nv - n_lo > 1 else f'coded frame {nv - 1}') +
            f' ({min(int(p.max_num_ref), nv - 1)} references)' if not reordered
            else f'coded frames 1..{n_ref - 1} of the chunk (everything but the I frame)')
    geom = (f'at the benched geometry {w}x{h}' if (cw, ch) == (w, h) else
            f'top-left {cw}x{ch} crop of the {w}x{h} frames (a bounded sample: the full-size reference run takes many minutes per stream)')
    tail = (f'Thorenc SIMD build, 1 thread per process, {nproc} reference processes running at the same time beside the GPU job on rank 0 of {world} '
            f'(the other ranks run no reference process)')
    one = {'value': round(v1, 4), 'unit': 'Mpixels/s',
           'sample': f'stream {my_ids[0]} of the same workload, {geom}: {what} = CPU time (user + system, wait4 rusage) of the {n_ref}-frame run - of the '
                     f'{n_lo}-frame run = {t_hi[1]:.1f} s - {t_lo[1]:.1f} s = {d_cpu:.1f} s (wall until exit: {t_hi[0]:.1f} s - {t_lo[0]:.1f} s = {d_wall:.1f} s)'}
    out = {'value': one['value'], 'unit': 'Mpixels/s', 'cores': 1, 'kind': 'reference', 'geometry': f'{cw}x{ch}', 'host': 'the GPU box (live, beside the GPU run)',
           'frames': list(range(n_lo, n_ref)), 'sample': one['sample'] + '; ' + tail}
    if tf_legs is not None:   # every timed frame on a crop: a SECONDARY figure (round 6; it was the value, which then had another geometry than the GPU figure)
        tw, th = tf_legs
        t_hi, t_lo = times[('t', nframes)], times[('t', a.warmup)]
        vt, dt_cpu, dt_wall = one_frame_baseline(t_hi, t_lo, float(tw) * th * a.steps, {'frames_lo': a.warmup})
        out['all_timed_frames_on_a_crop'] = {
            'value': round(vt, 4), 'unit': 'Mpixels/s', 'geometry': f'{tw}x{th}',
            'sample': (f'stream {my_ids[0]}, ALL timed coded frames {a.warmup}..{nframes - 1} on the top-left {tw}x{th} crop of the {w}x{h} frames (the full-size pair of runs would take '
                       f'{nframes * w * h / 0.3e6 / 60:.0f} minutes): CPU time of the {nframes}-frame run - of the {a.warmup}-frame run = {t_hi[1]:.1f} s - {t_lo[1]:.1f} s = {dt_cpu:.1f} s '
                       f'(wall until exit: {t_hi[0]:.1f} s - {t_lo[0]:.1f} s)')}
    # the same difference at the full geometry, recorded in the build container (another host: reported, never the value)
    k_hi = refs.get(ref_key(a.config, w, h, a.bitdepth, qp, nframes, a.sigma, my_ids[0]))
    k_lo = refs.get(ref_key(a.config, w, h, a.bitdepth, qp, a.warmup, a.sigma, my_ids[0]) + f'_cpu_of_n{nframes}')
    if not reordered and clipn == nframes and k_hi and k_lo and 'cpu_s' in k_hi and k_hi['cpu_s'] > k_lo['cpu_s']:
        out['recorded_full_geometry_timed_frames'] = {
            'value': round(float(w) * h * a.steps / (k_hi['cpu_s'] - k_lo['cpu_s']) / 1e6, 4), 'unit': 'Mpixels/s',
            'sample': f'coded frames {a.warmup}..{nframes - 1} of stream {my_ids[0]} at {w}x{h}: CPU seconds of the recorded {nframes}-frame reference run - of the {a.warmup}-frame '
                      f'run = {k_hi["cpu_s"]:.1f} s - {k_lo["cpu_s"]:.1f} s, measured in the build container ({k_hi.get("cpu_host")}; {k_hi.get("cpu_concurrent_jobs")} reference '
                      f'processes at once), not on the GPU box'}
    out['n_process'] = {'procs': nproc, 'host_cores': legs.ncores, 'value': round(agg, 4), 'unit': 'Mpixels/s',
                        'note': f'N = {len(same_geom)} concurrently running reference processes of the benched geometry (verify + baseline + fill-up chunks): pixels coded / wall of the slowest'}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--streams', type=int, default=int(os.environ.get('THOR_BENCH_STREAMS', '160')), help='streams PER GPU')
    ap.add_argument('--width', type=int, default=3840)
    ap.add_argument('--height', type=int, default=2160)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='ldb')
    ap.add_argument('--qp', type=int, default=None)
    ap.add_argument('--bitdepth', type=int, default=8, choices=(8, 10))
    ap.add_argument('--sigma', type=float, default=2.0, help='temporal noise of the synthetic clip (SURVEY 8d: 2 = default, 6 = hard variant)')
    ap.add_argument('--verify-frames', type=int, default=None, help='coded frames compared with the live reference (default: min(frames, max(warmup + 2, 7)))')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-verify', action='store_true')
    ap.add_argument('--verify', choices=('auto', 'live', 'recorded'), default='auto',
                    help='auto (default): live reference processes beside the GPU run on rank 0 (first / middle / last stream, the frames they can finish) AND, on every rank, '
                         'every frame of the streams that have a reference run recorded with scripts/record_bench_refs.py (tests/golden/bench_refs.json); live / recorded: only that')
    ap.add_argument('--lockstep', action='store_true', help='one lock-step frame of all streams per step (thor_hip_encode_staged, the rounds 1-4 path) instead of '
                                                            'thor_hip_encode_staged_run (two stream groups half a frame apart)')
    ap.add_argument('--clip-frames', type=int, default=None,
                    help='generate the seeded clip as a run of this many coded frames would (the clip depends on its length) and encode its first warmup + steps frames: '
                         'a shorter low-delay run is then a prefix of the recorded 25-frame reference runs and is verified against them')
    ap.add_argument('--cpu-sample', default=None, metavar='WxH',
                    help='with --verify recorded: geometry of the live CPU-baseline sample (top-left crop of stream 0, all frames); default: the benched geometry')
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
    # a torch.distributed.run launch (even with one process) runs the collectives: RCCL on the GPU box, gloo on CPU (tests)
    if world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ):
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(os.environ.get('THOR_BENCH_BACKEND', 'nccl' if have_gpu else 'gloo'))

    import thor_amd
    from thor_amd import synth
    w, h, S = a.width, a.height, a.streams
    cfg_name, qp_default = CONFIGS[a.config]
    cfg_path = os.path.join(ROOT, 'configs', cfg_name)
    qp = a.qp if a.qp is not None else qp_default
    hbd = a.bitdepth > 8
    bps = 2 if hbd else 1
    ref_extra = ['-bitdepth', str(a.bitdepth), '-input_bitdepth', str(a.bitdepth)] if hbd else []
    nframes = a.warmup + a.steps           # coded frames = display frames of every chunk
    clipn = a.clip_frames if a.clip_frames is not None else nframes   # the length the seeded clip is generated for
    assert clipn >= nframes and (clipn == nframes or a.config == 'ldb'), '--clip-frames: low-delay configuration, at least warmup + steps'
    fpx = w * h * 3 // 2                   # samples per frame
    extra = EXTRA_FRAMES
    reordered = a.config != 'ldb'
    # ---- input: every rank generates the seeded clip; chunks are cut on the GPU --------------------------------------
    t_in = time.perf_counter()
    seed = CONTENT_SEED(a.config, w)
    base = synth.make_clip(w, h, clipn + extra, seed, a.sigma, a.bitdepth)
    flat = np.concatenate([np.concatenate([p.ravel() for p in fr]) for fr in base])
    my_ids = stream_ids(S * world, world, rank)
    if dist is not None:  # consistency of the inputs across ranks: rank 0's first frame travels over RCCL and is compared
        f0 = broadcast_clip(flat[:fpx].view(np.uint8) if rank == 0 else None, dist)
        assert f0.cpu().numpy().tobytes() == flat[:fpx].tobytes(), 'ranks generated different clips'
    clip_t = torch.from_numpy(flat.view(np.int16) if hbd else flat).to(dev)
    dev_frames = clip_t.view(clipn + extra, fpx)
    over = {'bitdepth': a.bitdepth, 'input_bitdepth': a.bitdepth} if hbd else {}
    p = thor_amd.load_config(cfg_path, width=w, height=h, qp=qp, f=30, **over)
    enc = thor_amd.Encoder(p, S, device=local_rank)
    c2 = (w // 2) * (h // 2)
    maxv = (1 << a.bitdepth) - 1

    def stream_frames(sid):
        return host_stream_frames(base, sid, clipn, a.bitdepth)[:nframes]

    def dev_stream_frame(sid, f):  # the torch restatement of host_stream_frames (checked against it below)
        off = sid % (extra + 1)
        mode = (sid // 3) % 4
        fr = dev_frames[off + f]
        Y, U, V = fr[:w * h].view(h, w), fr[w * h:w * h + c2].view(h // 2, w // 2), fr[w * h + c2:].view(h // 2, w // 2)
        dims = ([1] if mode & 1 else []) + ([0] if mode & 2 else [])
        if dims:
            Y, U, V = torch.flip(Y, dims), torch.flip(U, dims), torch.flip(V, dims)
        Y = torch.clamp(Y.to(torch.int32) + (sid % 5), 0, maxv).to(Y.dtype)
        return torch.cat([Y.reshape(-1), U.reshape(-1), V.reshape(-1)])

    # ---- checker / CPU baseline: reference processes start NOW and run beside the GPU ------------------------------
    # Live-verified frames: at least I + the frames that fill the reference list + two timed frames (7 with the driver's flags);
    # more when the reference legs can be expected to finish inside the GPU run anyway (one host core codes a frame of this
    # geometry in ~w*h/0.33e6 s; the GPU run takes ~frames * w*h*S / 90e6 s): they cost no wall time.
    nv = a.verify_frames
    if nv is None:
        nv = max(a.warmup + 2, 7)
        t_cpu_frame = w * h / 0.33e6 * (1.3 if hbd else 1.0)
        est_gpu_s = nframes * w * h * S / (110e6 if a.config == "ldb" and not hbd else 60e6)
        nv = max(nv, int(est_gpu_s / t_cpu_frame))
    nv = max(1, min(nv, nframes))
    hq = int(p.HQperiod) if not reordered else 0
    if hq > 1 and nv - 1 > 0 and (nv - 1) % hq == 0:   # the baseline frame (coded frame nv - 1) must not be a high-quality frame
        nv = nv + 1 if nv + 1 <= nframes else nv - 1
    cw, ch = w, h      # geometry of the live CPU legs
    if a.cpu_sample:
        cw, ch = (int(x) for x in a.cpu_sample.lower().split('x'))
        assert cw <= w and ch <= h and cw % 8 == 0 and ch % 8 == 0
    legs = CpuLegs(cfg_path, cw, ch, qp, ref_extra, nframes, reordered)
    # Recorded reference runs (every rank, its own streams): {local stream index: record}.  A record of the same stream with more frames
    # serves a shorter low-delay run through its per-frame prefix hashes (the clip must have been generated for that length: --clip-frames).
    recorded, refs_note = {}, None
    if a.verify in ('auto', 'recorded') and not a.no_verify:
        refs = load_bench_refs(cfg_path)
        refs_note = refs.pop('_note', None)
        for s_ in range(S):
            r_ = refs.get(ref_key(a.config, w, h, a.bitdepth, qp, clipn, a.sigma, my_ids[s_]))
            if r_ is not None and (clipn == nframes or len(r_.get('prefix_md5', [])) >= nframes):
                recorded[s_] = r_
        if a.verify == 'recorded' and rank == 0:
            missing = [my_ids[s_] for s_ in sorted({0, S // 2, S - 1}) if s_ not in recorded]
            if missing:
                raise KeyError(f'--verify recorded: no recorded reference run for streams {missing} of this workload (scripts/record_bench_refs.py)')
    live = a.verify in ('auto', 'live') and not a.no_verify and rank == 0 and legs.available() and (cw, ch) == (w, h)
    # N > 1 (round 6): rank 0 starts NO reference process beside the timed region of a weak-scaling figure when recorded reference runs verify the
    # run anyway (every rank checks its own recorded streams); live legs remain the fall-back for workloads nobody recorded.  cpu_baseline: N = 1 only.
    if world > 1 and live and recorded and a.verify == 'auto':
        live = False
    if a.verify == 'recorded':
        nv = nframes
    verify = {}        # local stream index -> [host frames or None, {coded frame: (display index, md5 of the GPU reconstruction)}]
    do_verify = not a.no_verify and (live or bool(recorded))
    do_base = rank == 0 and world == 1 and not a.no_cpu_baseline and legs.available()
    n_ref = nframes if reordered else nv   # with frame reordering the coding order depends on the chunk length: run it all
    vs_live = sorted({0, S // 2, S - 1}) if live else []
    for s_ in recorded:
        verify[s_] = [None, {}]

    def crop(fr, cw_, ch_):   # top-left crop of a flat 4:2:0 frame (CPU-baseline samples)
        if (cw_, ch_) == (w, h):
            return fr
        Y, U, V = fr[:w * h].reshape(h, w), fr[w * h:w * h + c2].reshape(h // 2, w // 2), fr[w * h + c2:].reshape(h // 2, w // 2)
        return np.concatenate([np.ascontiguousarray(Y[:ch_, :cw_]).ravel(), np.ascontiguousarray(U[:ch_ // 2, :cw_ // 2]).ravel(), np.ascontiguousarray(V[:ch_ // 2, :cw_ // 2]).ravel()])
    base_tag, tf_legs = None, None
    if rank == 0 and (live or do_base):
        for s_ in (vs_live or [0]):
            fr = stream_frames(my_ids[s_])
            verify.setdefault(s_, [None, {}])[0] = fr
            if live or s_ == 0:
                legs.start(f'v{my_ids[s_]}', [crop(x, cw, ch).tobytes() for x in fr[:n_ref]], n_ref, live and s_ in vs_live)
        if not live and 0 in verify and 0 not in recorded:
            del verify[0]     # stream 0 only feeds the CPU-baseline legs
        if do_base:
            base_tag = f'v{my_ids[0]}'
            # second leg of the one-frame figure: one frame less (LDB: the difference is coded frame nv - 1 with every reference in
            # use); with frame reordering the I frame alone (the difference is every other frame of the chunk)
            n_lo = base_n_lo(reordered, a.warmup, nv)
            if n_lo >= 1 and n_lo < n_ref:
                legs.start(base_tag, None, n_lo, False)
            # LDB: the CPU figure over EXACTLY the timed frames - a top-left crop of stream 0 (at most 1920x1080) coded with warmup + steps
            # and with warmup frames; the difference of the CPU seconds is coded frames warmup .. warmup + steps - 1
            if not reordered and a.warmup >= 1 and not a.cpu_sample:
                tw, th = min(w, 1920), min(h, 1080)
                tw, th = tw - tw % 8, th - th % 8
                if nframes * tw * th / 0.3e6 < max(60.0, 0.8 * est_run_s(nframes, w, h, S)):   # bounded: it must end with the GPU run
                    fr0 = stream_frames(my_ids[0])
                    legs.start('t', [crop(x, tw, th).tobytes() for x in fr0], nframes, False, geom=(tw, th))
                    legs.start('t', None, a.warmup, False, geom=(tw, th))
                    tf_legs = (tw, th)
            # N-process figure: fill the remaining host cores (one is left to this process) with further chunks
            nm = max(0, min(legs.ncores - 1, 8) - len(legs.procs))
            km = min(n_ref, 3)
            for k in range(nm):
                sid = my_ids[(1 + k) % S]
                legs.start(f'm{k}', [crop(x, cw, ch).tobytes() for x in stream_frames(sid)[:km]], km, False)
    if verify:
        s_chk = sorted(verify)[-1]
        fr_chk = verify[s_chk][0][0] if verify[s_chk][0] is not None else stream_frames(my_ids[s_chk])[0]
        assert np.array_equal(dev_stream_frame(my_ids[s_chk], 0).cpu().numpy().view(np.uint8), fr_chk.view(np.uint8)), \
            'GPU chunk cutter differs from the host stream generator'

    for s in range(S):
        for f in range(nframes):
            t = dev_stream_frame(my_ids[s], f)
            if have_gpu:
                torch.cuda.synchronize()
            enc.stage_device(s, f, t.data_ptr())
        if reordered:
            enc.begin_sequence(s, 0, nframes, nframes)
    t_in = time.perf_counter() - t_in

    coded = [0]
    from concurrent.futures import ThreadPoolExecutor
    hasher = ThreadPoolExecutor(2)
    done_frames = [0] * S     # coded frames finished per stream

    # Host buffers of the verified reconstructions: a small pool of PINNED frames that is reused (a fresh pageable array per frame costs page faults and
    # a staged copy: ~8 ms per 3840x2160 frame; 14 verified streams per step).  A buffer goes back to the pool when its hash has been taken.
    import queue
    fbytes_host = enc.frame_bytes
    pool = queue.SimpleQueue()
    n_pool = min(2 * max(len(verify), 1), 32) if do_verify else 0
    pool_bufs = [torch.empty(fbytes_host, dtype=torch.uint8, pin_memory=have_gpu) for _ in range(n_pool)]
    for b_ in pool_bufs:
        pool.put(b_)
    verify_host_s = [0.0]

    def hash_and_release(buf):
        try:
            return hashlib.md5(memoryview(buf.numpy())).hexdigest()
        finally:
            pool.put(buf)

    def take_recon(s, di):
        """Reconstruction of the frame stream s has just finished: (display index, md5).  The frame is fetched here (D2H into a pinned buffer of the
        pool); the hash is taken by a worker thread while the GPU codes on (hashlib releases the GIL)."""
        cf = done_frames[s]
        if s in verify and (s in recorded or cf < nv):
            t_ = time.perf_counter()
            buf = pool.get()     # blocks only if every buffer is still being hashed
            enc.recon_into(s, buf.data_ptr())
            verify[s][1][cf] = (di, hasher.submit(hash_and_release, buf))
            verify_host_s[0] += time.perf_counter() - t_
        done_frames[s] = cf + 1

    def step():
        """One lock-step frame of every stream, in coding order (--lockstep)."""
        if reordered:
            idx = [enc.next_frame(s) for s in range(S)]
            assert idx[0] is not None
        else:
            idx = [coded[0]] * S
        enc.encode_staged(idx)      # blocks until all streams' bits are on the host
        coded[0] += 1
        for s in range(S):
            take_recon(s, idx[s]) if do_verify else None

    def run(k):
        """k steps = k frames of every stream.  Default: ONE call of thor_hip_encode_staged_run - the streams in two groups half a frame apart
        inside the call, which starts and ends on a frame boundary of every stream; --lockstep: k lock-step frames (the rounds 1-4 path)."""
        if a.lockstep:
            for _ in range(k):
                step()
            return

        def on_done(first, count):
            if do_verify:
                for s in range(first, first + count):
                    take_recon(s, enc.last_display_index(s))
        enc.encode_run(k, on_done)

    run(a.warmup)
    enc.kernel_time_reset()
    verify_host_s[0] = 0.0
    if dist is not None:
        dist.barrier()
    if have_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    if have_gpu:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sb_ms, launches, filt_ms = enc.kernel_time()
    if dist is not None:
        dist.barrier()
        dt = reduce_max_time(dt, dist)
    total_px = float(w) * h * a.steps * S * world
    value = total_px / dt / 1e6

    # ---- what the hand-over of HOST frames would add (measured, never part of `value`: the contract quotes value with the inputs resident in HBM) ----
    # One step's worth of input frames (S frames, in pieces of at most 32 frames of pinned host memory) copied host-to-device after the timed region,
    # timed with events: `io.h2d.value_including_h2d` adds that time to every step as if nothing overlapped it (a double-buffered upload would hide it
    # behind the seconds a step takes).
    h2d = None
    if have_gpu and rank == 0:
        try:
            nb = min(S, 32)
            hbuf = torch.empty(nb * fpx * bps, dtype=torch.uint8).pin_memory()
            dbuf = torch.empty(nb * fpx * bps, dtype=torch.uint8, device=dev)
            dbuf.copy_(hbuf, non_blocking=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range((S + nb - 1) // nb):
                dbuf.copy_(hbuf, non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) * S / (((S + nb - 1) // nb) * nb)
            step_ms = dt * 1e3 / max(a.steps, 1)
            h2d = {'ms_per_step': round(ms, 2), 'GB_per_s': round(S * fpx * bps / (ms / 1e3) / 1e9, 1), 'bytes_per_step': S * fpx * bps,
                   'value_including_h2d': round(value * step_ms / (step_ms + ms), 3),
                   'how': f'{S} frames of {fpx * bps / 1e6:.1f} MB from pinned host memory, copied after the timed region and timed with HIP events; added to every step '
                          'without any overlap (an upper bound of the cost: the upload of the next frame can run while a step codes)'}
            del hbuf, dbuf
        except Exception as e:   # pinned memory is a host resource: a failure here must not take the line down
            h2d = {'error': str(e)}

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
    stats = enc.stats() if hasattr(enc, 'stats') else None
    kinfo = None
    if have_gpu:   # registers / LDS / workgroups per CU of the superblock kernel as the runtime reports them (an occupancy regression shows in the line)
        try:
            import ctypes as C_
            r_, l_, p_, w_ = C_.c_int(), C_.c_int(), C_.c_int(), C_.c_int()
            # the build of the 8-bit kernel this run launched (the library picks by stream count and geometry): its resources, not the throughput build's
            in_use = thor_amd.lib().thor_hip_superblock_kernel_in_use() if bps == 1 else 0
            which = {0: bps, 1: 0, 2: 3}[in_use]
            if thor_amd.lib().thor_hip_superblock_kernel_info(which, C_.byref(r_), C_.byref(l_), C_.byref(p_), C_.byref(w_)) == 0:
                kinfo = {'build': ('throughput (thor_hip.cpp, four wavefronts per workgroup)', 'latency (thor_hip_lat.cpp, four wavefronts per workgroup)',
                                   'wide (thor_hip_wide.cpp, eight wavefronts per workgroup)')[in_use],
                         'vgprs': r_.value, 'lds_bytes': l_.value, 'private_bytes_per_lane': p_.value, 'workgroups_per_cu': w_.value}
        except (AttributeError, OSError):
            pass

    rc = 0
    if rank == 0:
        res = {'bit_exact': None, 'checked': [], 'cpu_baseline': None}
        if not legs.available():
            res['note'] = 'oracle/_ref/Thorenc not in the snapshot: bit-exactness not checked in this run'
        elif a.no_verify:
            res['note'] = 'verification disabled (--no-verify)'
        t_w = time.perf_counter()
        times = legs.collect() if legs.procs else {}
        t_w = time.perf_counter() - t_w
        if do_base:
            try:
                res['cpu_baseline'] = cpu_baseline_object(times, base_tag, tf_legs, nv, n_ref, reordered, a, p, w, h, cw, ch, legs, world, my_ids,
                                                          load_bench_refs(cfg_path) if not a.no_verify else {}, qp, clipn)
            except (RuntimeError, KeyError) as e:
                print(f'bench.py: {e}', file=sys.stderr)
                res['cpu_baseline'] = {'value': None, 'unit': 'Mpixels/s', 'cores': 1, 'kind': 'reference', 'sample': None, 'error': str(e)}
    # ---- verification: every rank checks its own streams, rank 0 collects ------------------------------------------------
    my_checked, my_ok = [], True
    if do_verify:
        fbytes = fpx * bps
        for s in sorted(verify):
            fr, recs = verify[s]
            recs = {cf: (di, fut.result()) for cf, (di, fut) in recs.items()}
            entry = {'rank': rank, 'stream': my_ids[s], 'ok': True}
            if s in vs_live:   # bitstream prefix + reconstruction of the first nv coded frames against the live reference run
                tag = f'v{my_ids[s]}'
                rbits, rrec = legs.read(tag, n_ref, 'bit'), legs.read(tag, n_ref, 'yuv')
                gpre, rpre = stream_prefix(local_bits[s], nv), stream_prefix(rbits, nv)
                same = gpre is not None and rpre is not None and gpre == rpre   # a stream with fewer than nv frames is a failure
                cmp_frames = []
                for cf in sorted(recs):   # the reference writes its reconstruction in DISPLAY order
                    if cf >= nv:
                        continue
                    di, md5 = recs[cf]
                    same = same and len(rrec) >= (di + 1) * fbytes and hashlib.md5(rrec[di * fbytes:(di + 1) * fbytes]).hexdigest() == md5
                    cmp_frames.append(cf)
                entry['live'] = {'frames': nv, 'timed_coded_frames_compared': list(range(a.warmup, nv)), 'bitstream_bytes': len(rpre or b''),
                                 'recon_coded_frames_compared': cmp_frames, 'ok': bool(same)}
                entry['ok'] = entry['ok'] and bool(same)
            if s in recorded:  # whole stream (or its prefix) + every reconstructed frame against the recorded reference run
                rr = recorded[s]
                if clipn == nframes:
                    same = len(local_bits[s]) == rr['bit_bytes'] and hashlib.md5(local_bits[s]).hexdigest() == rr['bit_md5']
                else:
                    same = hashlib.md5(local_bits[s]).hexdigest() == rr['prefix_md5'][nframes - 1]
                same = same and len(recs) == nframes
                for cf in sorted(recs):
                    di, md5 = recs[cf]
                    same = same and rr['rec_md5'][di] == md5
                entry['recorded'] = {'frames': nframes, 'timed_coded_frames_compared': list(range(a.warmup, nframes)), 'bitstream_bytes': len(local_bits[s]),
                                     'recon_coded_frames_compared': sorted(recs), 'ok': bool(same)}
                entry['ok'] = entry['ok'] and bool(same)
            entry['frames'] = nframes if s in recorded else nv
            entry['timed_frames_covered'] = max(0, entry['frames'] - a.warmup)
            my_checked.append(entry)
            my_ok = my_ok and entry['ok']
    all_checked = [my_checked]
    if dist is not None:
        all_checked = [None] * world
        dist.all_gather_object(all_checked, my_checked)
    if rank == 0:
        checked = [e for part in all_checked for e in part]
        if checked:
            n_live = sum(1 for e in checked if 'live' in e)
            n_rec = sum(1 for e in checked if 'recorded' in e)
            res['checked'] = checked
            res['bit_exact'] = all(e['ok'] for e in checked)
            src, scope = [], []
            if n_live:
                src.append('live runs of oracle/_ref/Thorenc beside the GPU run')
                scope.append(f'{n_live} of {S * world} streams (first, middle, last of rank 0) against the live reference run: coded frames 0..{nv - 1} of {nframes} - '
                             f'bitstream prefix + reconstruction of every one of those frames; timed frames {a.warmup}..{nv - 1} of {a.warmup}..{nframes - 1}')
            if n_rec:
                src.append('reference runs recorded in the build container (scripts/record_bench_refs.py -> tests/golden/bench_refs.json)')
                ranks_rec = sorted({e['rank'] for e in checked if 'recorded' in e})
                scope.append(f'{n_rec} of {S * world} streams (ranks {ranks_rec}) against recorded reference runs: ALL coded frames 0..{nframes - 1} - the '
                             f'{"whole bitstream" if clipn == nframes else "bitstream (a prefix of the recorded " + str(clipn) + "-frame run)"} + reconstruction of every frame; '
                             f'every timed frame {a.warmup}..{nframes - 1} is covered')
            res['bit_exact_source'] = ' + '.join(src)
            res['bit_exact_scope'] = '; '.join(scope)
            if refs_note:
                res['bit_exact_scope'] += f' ({refs_note})'
        if res['bit_exact'] is False:
            value, rc = 0.0, 1
        # roofline of the dominant kernel (k_superblocks): algorithmic HBM bytes per luma pixel of a P frame with R
        # references = 1.5 * bps * (1 orig + R refs + 1 rec) (block-path terms).  Coded frame f of an LDB chunk
        # (f >= 1) references min(max_num_ref, f) earlier frames: average over the timed frames.
        mref = int(p.max_num_ref)
        R = sum(min(mref, f) for f in range(a.warmup, nframes)) / max(a.steps, 1)
        bytes_per_px = 1.5 * bps * (2 + R)
        alg_bytes_per_launch = (w * h * a.steps * S * bytes_per_px) / max(launches, 1)
        avg_launch_s = (sb_ms / 1e3) / max(launches, 1)
        achieved = alg_bytes_per_launch / max(avg_launch_s, 1e-12) / 1e9
        # HBM traffic and SQ figures: only from a committed PMC pass of THIS workload geometry AND of this very library
        # (profiles/r06_pmc_bench.json, written by scripts/pmc_summary.py from rocprofv3 --pmc passes of the same bench geometry; it
        # carries csrc_digest() of the sources it profiled); otherwise null - a counter of other code is not attached to this line.
        traffic, traffic_src, valu_util = None, None, None
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', PMC_JSON)))
            # `same_device_code`: digests of later source states whose DEVICE code of the profiled translation unit is identical to the profiled one (host-side or
            # comment-only changes), each with the check that established it (scripts/device_code_same.sh: hipcc --cuda-device-only -S of both states, diffed)
            same = {e.get('csrc_digest'): e.get('check', '') for e in pm.get('same_device_code', [])}
            if pm.get('csrc_digest') != csrc_digest() and csrc_digest() not in same:
                traffic_src = f'profiles/{PMC_JSON} describes other engine sources (digest {pm.get("csrc_digest")} != {csrc_digest()}): not attached'
            elif (pm.get('width') == w and pm.get('height') == h and pm.get('streams') == S and pm.get('config', 'ldb') == a.config and
                  pm.get('sigma', 2.0) == a.sigma and not hbd):   # same geometry, operating point and content as the profiled workload
                traffic = round((pm['fetch_bytes_per_px'] + pm['write_bytes_per_px']) * w * h * S * a.steps / max(launches, 1))
                traffic_src = f'profiles/{PMC_JSON[:-5]}.md: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this geometry and of these sources (%s)' % pm['workload']
                if pm.get('csrc_digest') != csrc_digest():
                    traffic_src += f'; profiled at source digest {pm.get("csrc_digest")}, these sources ({csrc_digest()}) have the same device code: ' + same[csrc_digest()]
                valu_util = pm.get('valu_util_chip')
        except (OSError, KeyError, ValueError):
            pass
        # op model: ~1.1e4 SAD pixel-differences per pixel of a 4-reference P frame at speed 0 (+30 % for the
        # bi-prediction search), scaled by the references in use; against the v_sad_u8 rate of the chip
        sad_ops_px = 1.1e4 * (R / 4.0) * (1.3 if R > 1 else 1.0)
        sad_frac = value * 1e6 * sad_ops_px / SAD_PEAK_PXOPS
        out = {
            'metric': 'encoder Mpixels/s at fixed qp, bit-exact recon vs ref', 'value': round(value, 3), 'unit': 'Mpixels/s',
            'fps': round(value * 1e6 / (w * h), 3),
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt * 1e3 / max(a.steps, 1), 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u16' if hbd else 'u8', 'data': 'synthetic',
            'bit_exact': res['bit_exact'], 'bit_exact_source': res.get('bit_exact_source'), 'bit_exact_scope': res.get('bit_exact_scope'), 'bit_exact_checked': res['checked'],
            'config': {'workload': f'{w}x{h} {a.bitdepth}-bit 4:2:0, {cfg_name[:-4]} (configs/{cfg_name}), qp {qp}, '
                                   f'{S} independent closed streams per GPU ' + ('in lock step' if a.lockstep else 'in two groups half a frame apart (thor_hip_encode_staged_run; the timed '
                                   'run starts and ends on a frame boundary of every stream)') + f', timed frames = coded frames {a.warmup}..{nframes - 1} of each stream '
                                   f'({R:.2f} references on average), synthetic content sigma {a.sigma:g}',
                       'streams_per_gpu': S, 'frames_timed_per_stream': a.steps, 'parallelism': f'stream-sharded x{world}',
                       'per_stream_fps': round(a.steps / dt, 4), 'per_stream_mpx_s': round(w * h * a.steps / dt / 1e6, 4),
                       'superblock_queue': 'fifo', 'schedule': 'lockstep' if a.lockstep else 'two groups half a frame apart', 'csrc_digest': csrc_digest(),
                       'superblock_kernel': kinfo},   # thor_amd/csrc/tk_sched.h
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 4), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 8), 'traffic': traffic, 'traffic_unit': 'bytes per launch', 'traffic_source': traffic_src,
                         'alg_bytes_per_launch': round(alg_bytes_per_launch),
                         'kernel': 'k_superblocks', 'launches': launches, 'avg_launch_ms': round(avg_launch_s * 1e3, 3),
                         'alg_bytes_per_px': round(bytes_per_px, 3),
                         'ops': {'model': 'SAD pixel-differences per luma pixel (SURVEY 8d): %.3g' % sad_ops_px,
                                 'achieved_pxops_per_s': round(value * 1e6 * sad_ops_px, 0), 'peak_pxops_per_s': SAD_PEAK_PXOPS,
                                 'unit': 'v_sad_u8 pixel-differences/s', 'frac': round(sad_frac, 6), 'valu_util_chip': valu_util},
                         'traffic_frac_of_peak': (round(traffic / max(avg_launch_s, 1e-12) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
                         'note': ('one persistent dependency-driven launch per frame' if a.lockstep else
                                  'persistent dependency-driven launches over ranges of anti-diagonals of the superblock grid: two per frame + one per run (two stream groups half a frame apart)') +
                                 '; by algorithmic bytes the path is latency/instruction-bound, not HBM-bound (SURVEY.md 0.7) - `traffic` (L2 misses of the resident working set, '
                                 'DESIGN.md 8a) is what loads the memory system; filters+ref kernels took %.1f ms in the timed region' % filt_ms},
            'io': {'input_setup_s': round(t_in, 2), 'gather_s': round(t_g, 3), 'cpu_legs_wait_s': round(t_w, 1),
                   'stream_bytes_total': int(total_bytes), 'frames_total': int(total_frames),
                   'inputs': 'resident in HBM before the timed region (staged device-to-device), as the measurement contract quotes `value`; the host-to-device transfer of the '
                             f'input frames is EXCLUDED from value and MEASURED beside it (io.h2d: one {fpx * bps / 1e6:.1f} MB frame per stream and step); the device-to-host '
                             'transfer of every stream\'s bits and of the verified reconstructions is inside the timed region',
                   'h2d': h2d,
                   'timed_region_ms_per_step': {'k_superblocks': round(sb_ms / max(a.steps, 1), 1), 'filters_reference_creation_bit_gather_kernels': round(filt_ms / max(a.steps, 1), 1),
                                                'verified_reconstructions_d2h': round(verify_host_s[0] * 1e3 / max(a.steps, 1), 1),
                                                'other_host_work_and_launch_gaps': round((dt * 1e3 - sb_ms - filt_ms - verify_host_s[0] * 1e3) / max(a.steps, 1), 1),
                                                'note': 'rank-local figures of this rank; the kernels of a step do not overlap (a launch, then the filters of the stream group whose frame it completed)'},
                   'rank0_host_load': (f'{len(legs.procs)} reference processes (checker + cpu_baseline legs) ran on rank 0\'s host beside its GPU work; the other ranks ran none' +
                                       ('' if world == 1 or legs.procs else ' (N > 1: recorded reference runs verify every rank, no reference process runs beside a timed region)'))},
        }
        if stats:
            out['content'] = stats
        if res.get('note'):
            out['bit_exact_note'] = res['note']
        if res['cpu_baseline']:
            out['cpu_baseline'] = res['cpu_baseline']
        print(json.dumps(out), flush=True)
    enc.close()
    if dist is not None:
        dist.barrier()   # the other ranks wait here while rank 0 collects the CPU legs (reference runs) above
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == '__main__':
    main()
