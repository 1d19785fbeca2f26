#!/usr/bin/env python3
"""Compact timeline (ms) of a rocprofv3 --kernel-trace --memory-copy-trace run: trace_timeline.py DIR"""
import csv
import glob
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e - s > 100000:
            ev.append((s, e, 'H2D' if 'HOST_TO_DEVICE' in r['Direction'] else 'D2H'))
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if 'copyBuffer' in r['Kernel_Name']:
            ev.append((s, e, 'BLIT'))
        elif 'row_mul' in r['Kernel_Name']:
            ev.append((s, e, 'row'))
ev.sort()
t0 = ev[0][0]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, len(ev))
print(len(ev), 'events')
print(' '.join('%s@%.1f-%.1f' % (t, (s - t0) / 1e6, (e - t0) / 1e6) for s, e, t in ev[lo:hi]))
