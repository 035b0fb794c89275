#!/usr/bin/env python3
"""pc_hist.py DIR [TOP] - histogram of rocprofv3 PC-sampling CSVs (beta): samples per source line (Instruction_Comment, needs a
-gline-tables-only build), per instruction and per stall reason, for the dominant kernel.  Prints a summary small enough to travel back
from the GPU box (the raw CSVs are hundreds of MB).  Development aid."""
import csv, collections, glob, os, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 150
files = [f for f in glob.glob(os.path.join(d, '**', '*.csv'), recursive=True) if 'pc_sampling' in os.path.basename(f)]
print('files:', files)
for f in files:
    with open(f, newline='') as fh:
        rd = csv.reader(fh)
        hdr = next(rd)
        print('==', f, 'columns:', hdr)
        ix = {h: i for i, h in enumerate(hdr)}
        ci = ix.get('Instruction')
        cc = ix.get('Instruction_Comment')
        extra = [i for h, i in ix.items() if any(k in h.lower() for k in ('stall', 'reason', 'issued', 'wave_in_group', 'type'))]
        by_line, by_ins, by_extra = collections.Counter(), collections.Counter(), {i: collections.Counter() for i in extra}
        by_line_extra = collections.defaultdict(collections.Counter)
        n = 0
        first = []
        for row in rd:
            if n < 3:
                first.append(row)
            n += 1
            line = row[cc] if cc is not None else ''
            ins = row[ci] if ci is not None else ''
            by_line[line] += 1
            by_ins[(ins.split(' ')[0], line)] += 1
            for i in extra:
                by_extra[i][row[i]] += 1
                by_line_extra[line][hdr[i] + '=' + row[i]] += 1
        print('samples:', n)
        for r in first:
            print('  row:', r)
        for i in extra:
            print('--', hdr[i], by_extra[i].most_common(20))
        print('-- top source lines')
        for line, c in by_line.most_common(top):
            ex = ' '.join(f'{k}:{v}' for k, v in by_line_extra[line].most_common(4))
            print(f'{100.0 * c / max(n, 1):6.2f}% {c:8d}  {line}   [{ex}]')
        print('-- top (opcode, line)')
        for (op, line), c in by_ins.most_common(top):
            print(f'{100.0 * c / max(n, 1):6.2f}% {c:8d}  {op:28s} {line}')
        # per opcode class
        cls = collections.Counter()
        for (op, line), c in by_ins.items():
            k = op.split('_')[0] + '_' + (op.split('_')[1] if '_' in op else '')
            cls[k] += c
        print('-- by opcode family', cls.most_common(40))
