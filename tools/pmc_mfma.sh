#!/bin/bash
# MFMA utilisation of the hot kernels from rocprofv3 PMC passes (one counter group per pass; --kernel-trace only beside --pmc):
#   busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES, per kernel over the launches of 3 evaluations of the cfg-2 bench leg
# usage (through gpurun, from the repo root): bash tools/pmc_mfma.sh > gpurun_out/pmc_mfma.txt
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  rm -rf /tmp/pmcm
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcm -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-multitask --no-extra > /tmp/pmcm.log 2>&1
  f=$(find /tmp/pmcm -name "*counter_collection.csv" | head -1)
  echo "== $grp"
  [ -n "$f" ] && python3 - $f <<'PY'
import csv,sys,collections,re
per=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ',''); n=re.sub(r'\(.*$','',n)
    per[n][r['Counter_Name']]+=float(r['Counter_Value']); disp[n].add(r['Dispatch_Id'])
names=sorted({c for d in per.values() for c in d})
for n,d in sorted(per.items(), key=lambda kv:-sum(kv[1].values()))[:9]:
    vals='  '.join('%s %.4e'%(c,d.get(c,0)) for c in names)
    ratio=(d.get(names[0],0)/d.get(names[1],1)) if len(names)>1 and d.get(names[1],0) else float('nan')
    print('%-44s launches %4d  %s  ratio %.3f'%(n[:44],len(disp[n]),vals,ratio))
PY
done
