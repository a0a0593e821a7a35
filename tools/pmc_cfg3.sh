#!/bin/bash
# Matrix-core counters of cfg 3's fp32-on-fp16/bf16 kernels (post2h_kernel, syrk3_kernel<true/false>, post3_kernel):
#   pass 1: SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES        -> MFMA busy = busy / (4 SIMDs x CU-busy cycles)
#   pass 2: GRBM_GUI_ACTIVE (+ the kernel trace's durations)    -> effective shader clock = GUI_ACTIVE / duration
# One counter group per pass, --kernel-trace only beside --pmc.  usage (through gpurun, from the repo root):
#   bash tools/pmc_cfg3.sh [name=value ...]  > gpurun_out/pmc_cfg3.txt      (options go to tools/prof_cfg3.py, e.g. post_f16x2=0 chol_f16x2=0)
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc3
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc3 -- python $ROOT/tools/prof_cfg3.py "$@" > /tmp/pmc3.log 2>&1
  f=$(find /tmp/pmc3 -name "*counter_collection.csv" | head -1)
  k=$(find /tmp/pmc3 -name "*kernel_trace.csv" | head -1)
  echo "== $grp   ($*)"
  [ -n "$f" ] && python3 - "$f" "$k" <<'PY'
import csv, sys, collections, re
clean = lambda n: re.sub(r'\(.*$', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))
per = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    n = clean(r['Kernel_Name'])
    per[n][r['Counter_Name']] += float(r['Counter_Value']); disp[n].add(r['Dispatch_Id'])
dur = collections.defaultdict(float)
if len(sys.argv) > 2 and sys.argv[2]:
    for r in csv.DictReader(open(sys.argv[2])):
        dur[clean(r['Kernel_Name'])] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-9
for n, d in sorted(per.items(), key=lambda kv: -dur.get(kv[0], 0))[:8]:
    line = '%-40s launches %4d  %8.2f ms' % (n[:40], len(disp[n]), dur.get(n, 0) * 1e3)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
        line += '  mfma_busy %.4e  cu_busy %.4e  busy/(4 x cu_busy) %.3f' % (d['SQ_VALU_MFMA_BUSY_CYCLES'], d['SQ_BUSY_CU_CYCLES'], d['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * d['SQ_BUSY_CU_CYCLES']) if d['SQ_BUSY_CU_CYCLES'] else float('nan'))
    if 'GRBM_GUI_ACTIVE' in d and dur.get(n):
        # GRBM_GUI_ACTIVE is reported per XCD instance and summed over the dispatch's instances: divide by the 8 XCDs
        line += '  gui_active %.4e  clock %.3f GHz (gui_active / 8 XCDs / duration)' % (d['GRBM_GUI_ACTIVE'], d['GRBM_GUI_ACTIVE'] / 8 / dur[n] / 1e9)
    print(line)
PY
done
