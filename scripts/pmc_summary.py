#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of tools/thorenc_hip (one directory per pass: gpurun_out/<prefix>_<tag>/) for
k_superblocks: writes a markdown table and a small JSON with the per-pixel figures bench.py reports as roofline.traffic.
  python scripts/pmc_summary.py gpurun_out/r4pmc 3840 2160 128 6 gpurun_out/r04_pmc_bench "<workload description>" [SKIP]   (run on the GPU box, in the tree that was profiled)
SKIP (round 5): number of leading k_superblocks launches to leave out - the warm-up frames of a bench.py run - so that the counters are those of the TIMED
frames only; `frames` is then the number of timed frames (steps). """
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import csrc_digest   # the summary is stamped with the digest of the engine sources it profiled; bench.py attaches it only to the same sources

prefix, w, h, S, n, out, desc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], sys.argv[7]
skip = int(sys.argv[8]) if len(sys.argv) > 8 else 0
px = float(w) * h * S * n
agg = collections.OrderedDict()
times = {}
for tag in ('sq1', 'sq2', 'fetch', 'write'):   # one rocprofv3 --pmc pass each (a missing pass is skipped)
    fs = glob.glob(f'{prefix}_{tag}/*/*_counter_collection.csv')
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(fs[0])) if 'k_superblocks' in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    keep = set(ids[skip:])   # the launches of the timed frames
    for r in rows:
        if int(r['Dispatch_Id']) in keep:
            agg[r['Counter_Name']] = agg.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    kt = glob.glob(f'{prefix}_{tag}/*/*_kernel_trace.csv')[0]
    times[tag] = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e9 for r in sorted((r for r in csv.DictReader(open(kt)) if 'k_superblocks' in r['Kernel_Name']),
                                                                                                   key=lambda r: int(r['Start_Timestamp']))][skip:]
g = lambda k: agg.get(k, float('nan'))
# gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE tallies 128-B requests at
# 64 B -> doubled; WRITE_SIZE is uncalibrated on gfx950 and is reported as is.
fetch_b = g('FETCH_SIZE') * 1024 * 2
write_b = g('WRITE_SIZE') * 1024
tsum = sum(times.get('sq1', [0]))
res = {'csrc_digest': csrc_digest(), 'workload': desc, 'width': w, 'height': h, 'streams': S, 'frames': n, 'config': 'ldb', 'luma_pixels': px, 'kernel_seconds_total': tsum, 'launches': len(times.get('sq1', [])),
       'fetch_bytes_per_px': fetch_b / px, 'write_bytes_per_px': write_b / px,
       'valu_insts_per_px': g('SQ_INSTS_VALU') / px, 'salu_insts_per_px': g('SQ_INSTS_SALU') / px, 'lds_insts_per_px': g('SQ_INSTS_LDS') / px,
       'vmem_rd_insts_per_px': g('SQ_INSTS_VMEM_RD') / px, 'vmem_wr_insts_per_px': g('SQ_INSTS_VMEM_WR') / px, 'flat_insts_per_px': g('SQ_INSTS_FLAT') / px,
       # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (guide, per-instruction constants table)
       'wave_cycles_busy_frac_valu': g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'), 'wave_cycles_wait_any_frac': g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'),
       'wave_cycles_active_any_frac': g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'), 'wave_cycles_wait_inst_frac': g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'),
       # VALU issue utilisation of the chip: one wave64 VALU instruction occupies a SIMD for one quad-cycle (4 clocks); 1024 SIMDs at 2.4 GHz
       'valu_util_chip': g('SQ_ACTIVE_INST_VALU') * 4 / (1024 * 2.4e9 * tsum) if tsum else None,
       'raw': agg}
json.dump(res, open(out + '.json', 'w'), indent=1)
L = ['# rocprofv3 PMC passes on `k_superblocks`', '', f'Workload: {desc}.', 'Each counter group was collected in its own run (`rocprofv3 --pmc ... --kernel-trace`, no other tracing), '
     'values summed over the launches of `k_superblocks`.', '', '| counter | sum | per luma pixel |', '|---|---|---|']
for k, v in agg.items():
    L.append(f'| {k} | {v:.4g} | {v / px:.4g} |')
L += ['', f'Kernel time of the SQ pass: {tsum:.2f} s in {len(times.get("sq1", []))} launches ({px / 1e6 / tsum if tsum else 0:.1f} Mpx/s inside the kernel).', '',
      '| derived | value |', '|---|---|',
      f'| HBM-side read traffic (FETCH_SIZE KiB x 1024 x 2, gfx950 correction) | {fetch_b / 1e9:.2f} GB = {fetch_b / px:.1f} B per luma pixel |',
      f'| HBM-side write traffic (WRITE_SIZE KiB x 1024, uncalibrated on gfx950) | {write_b / 1e9:.2f} GB = {write_b / px:.1f} B per luma pixel |',
      f'| wave-instructions per luma pixel: VALU / SALU / LDS / VMEM rd / VMEM wr | {res["valu_insts_per_px"]:.0f} / {res["salu_insts_per_px"]:.0f} / {res["lds_insts_per_px"]:.0f} / {res["vmem_rd_insts_per_px"]:.0f} / {res["vmem_wr_insts_per_px"]:.0f} |',
      f'| share of resident wave-cycles: issuing VALU / any instruction / stalled at issue / parked (waitcnt, barrier, sleep) | {100 * res["wave_cycles_busy_frac_valu"]:.1f} % / {100 * res["wave_cycles_active_any_frac"]:.1f} % / {100 * res["wave_cycles_wait_inst_frac"]:.1f} % / {100 * res["wave_cycles_wait_any_frac"]:.1f} % |',
      f'| VALU issue utilisation of the chip (1024 SIMDs x 2.4 GHz / 4 clocks per wave64 instruction) | {100 * (res["valu_util_chip"] or 0):.1f} % |']
open(out + '.md', 'w').write('\n'.join(L) + '\n')
print('\n'.join(L[-8:]))
