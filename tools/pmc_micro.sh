#!/bin/bash
# rocprofv3 --pmc pass over one conv_microbench invocation; prints per-kernel counter sums.  usage: pmc_micro.sh <outdir> <counters...> -- <microbench args>
O=$1; shift
ctr=()
while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $R/$O -o p -- python $R/tools/conv_microbench.py "$@" > $R/$O/log.txt 2>&1)
python - "$O" <<'PY'
import csv, glob, sys, collections
d=sys.argv[1]
f=glob.glob(d+'/**/*counter_collection.csv', recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'][:70]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='GRBM_GUI_ACTIVE' or len(agg[k])==1: cnt[k]+=1
for k,v in agg.items():
    if 'axpby' in k or 'at::' in k: continue
    print(k, dict((a, round(b)) for a,b in v.items()))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and v.get('GRBM_GUI_ACTIVE'): print('   mfma_busy_frac', round(v['SQ_VALU_MFMA_BUSY_CYCLES']/(128.0*v['GRBM_GUI_ACTIVE']),4))
    if v.get('SQ_WAVE_CYCLES'):
        for c in ('SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_WAIT_INST_LDS','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_VMEM','SQ_ACTIVE_INST_MISC','SQ_ACTIVE_INST_SCA'):
            if c in v: print('   %s/WAVE_CYCLES %.4f'%(c, v[c]/v['SQ_WAVE_CYCLES']))
PY
find $O -name "*.csv" -size +4M -delete
