"""Bitwise repeatability of the NLL+grad evaluation per option setting: determinism.py legs=T64,shard8,nll4096 [reps=40] opt=a,opt2=b ...
Prints, per leg and setting, how many of the repetitions differ from the first one (value, gradient) and the largest deviation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv_saved = sys.argv[:]
args = sys.argv[1:]
legs = 'T64,shard8'
reps = 40
rest = []
for a in args:
    if a.startswith('legs='): legs = a[5:]
    elif a.startswith('reps='): reps = int(a[5:])
    else: rest.append(a)
sys.argv = [sys.argv[0], 'legs=none']
import numpy as np
import ab_suite as ab          # (runs no leg: 'none' is skipped below)
for leg in legs.split(','):
    f = ab.make(leg)
    for s in (rest or ['-']):
        ab.apply(s)
        v0, g0 = f(); g0 = ab.flat(g0)
        nv = ng = 0; dv = dg = 0.0
        for _ in range(reps):
            v, g = f(); g = ab.flat(g)
            if v != v0: nv += 1; dv = max(dv, abs(v - v0) / abs(v0))
            if not np.array_equal(g, g0): ng += 1; dg = max(dg, float(np.max(np.abs(g - g0)) / np.max(np.abs(g0))))
        print('%-8s %-40s value differs %2d/%d (max %.1e)   gradient differs %2d/%d (max %.1e)' % (leg, s, nv, reps, dv, ng, reps, dg), flush=True)
ab.apply('-')
