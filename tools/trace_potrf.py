"""Timeline of the Cholesky panel chain from a rocprofv3 kernel trace (csv) of tools/prof_nll.py level 0.
usage: python tools/trace_potrf.py <kernel_trace.csv>"""
import csv, sys, collections
ALL = len(sys.argv) > 2 and sys.argv[2] == 'all'
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r.get('Queue_Id', 0) or 0),
                     int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)))
rows.sort()
# last evaluation: find the last gram_kernel launch
starts = [i for i, r in enumerate(rows) if 'gram_kernel' in r[2]]
i0 = starts[-1]
ev = rows[i0:]
t0 = ev[0][0]
def short(n):
    for k in ('potf2', 'trsm_kernel', 'gemm_kernel', 'syrk3', 'split3_panel', 'split3_block', 'gram_kernel', 'nll_reduce', 'trtri', 'wtz', 'grad_contract', 'grad_finalize', 'aug_rows', 'dmu'):
        if k in n:
            return k + ('<64>' if 'Li64E' in n else ('<128>' if 'Li128E' in n else ''))
    return n[:30]
potf2 = [r for r in ev if 'potf2' in r[2]]
print('kernels in eval:', len(ev), 'potf2 launches:', len(potf2), 'eval span ms:', (ev[-1][1] - t0) / 1e6)
# per panel: interval between consecutive potf2 starts, and what ran on the same queue in between
pq = potf2[0][3]
chain = [r for r in ev if r[3] == pq and r[0] >= potf2[0][0] and r[0] <= potf2[-1][1] + 1]
print('chain queue', pq, 'kernels on it:', len(chain))
tot_dur = collections.Counter(); tot_cnt = collections.Counter(); gap = 0.0
for a, b in zip(chain, chain[1:]):
    gap += max(0, b[0] - a[1])
for r in chain:
    tot_dur[short(r[2])] += r[1] - r[0]; tot_cnt[short(r[2])] += 1
print('chain span ms %.3f, sum kernel ms %.3f, sum gaps ms %.3f' % ((chain[-1][1] - chain[0][0]) / 1e6, sum(tot_dur.values()) / 1e6, gap / 1e6))
for k in tot_dur:
    print('   %-24s %8.3f ms %4d launches %7.1f us avg' % (k, tot_dur[k] / 1e6, tot_cnt[k], tot_dur[k] / 1e3 / tot_cnt[k]))
# panel-by-panel
print('panel  dt_us  potf2_us  (kernels between)')
for j, (a, b) in enumerate(zip(potf2, potf2[1:])):
    if ALL or j % 4 == 0 or j > len(potf2) - 6:
        between = [r for r in chain if a[0] <= r[0] < b[0]]
        print('%4d %7.1f %7.1f   %s' % (j, (b[0] - a[0]) / 1e3, (a[1] - a[0]) / 1e3,
              ' '.join('%s:%.0f' % (short(r[2])[:9], (r[1] - r[0]) / 1e3) for r in between)))
# other queues: busy intervals summary
byq = collections.defaultdict(list)
for r in ev:
    byq[r[3]].append(r)
for qid, lst in byq.items():
    print('queue', qid, 'kernels', len(lst), 'busy ms %.3f' % (sum(r[1] - r[0] for r in lst) / 1e6), 'span %.3f..%.3f ms' % ((lst[0][0] - t0) / 1e6, (lst[-1][1] - t0) / 1e6))
