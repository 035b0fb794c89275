#!/usr/bin/env python3
"""kernel_stats_md.py <rocprofv3 output dir> <title> <command> -> markdown table of the per-kernel statistics (from *_kernel_stats.csv, or
aggregated from *_kernel_trace.csv when the stats file is missing)."""
import collections, csv, glob, sys
d, title, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
rows = []
fs = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
if fs:
    for r in csv.DictReader(open(fs[0])):
        rows.append((r['Name'], int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6, float(r['MinNs']) / 1e6, float(r['MaxNs']) / 1e6, float(r['Percentage'])))
else:
    agg = collections.OrderedDict()
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            agg.setdefault(r['Kernel_Name'], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    tot = sum(sum(v) for v in agg.values()) or 1.0
    for k, v in agg.items():
        rows.append((k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))
rows.sort(key=lambda r: -r[2])
print(f'# {title}\n\nCommand: `{cmd}`\n\n| kernel | calls | total ms | average ms | min ms | max ms | % |\n|---|---|---|---|---|---|---|')
for n, c, t, a, mn, mx, p in rows[:14]:
    print(f'| `{n[:72]}` | {c} | {t:.1f} | {a:.3f} | {mn:.3f} | {mx:.3f} | {p:.2f} |')
