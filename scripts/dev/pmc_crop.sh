set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/t9
mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/a -- python3 $R/scripts/bench_train.py --only crop --steps 3 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 $R/scripts/bench_train.py --only crop --steps 3 > $O/b.log 2>&1
cd $R
python3 - <<'PY'
import csv,glob,collections,os
for tag in 'ab':
    f=glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/t9',tag,'*','*counter_collection.csv'))[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        cnt[(k,r['Counter_Name'])]+=1
    for k in agg:
        if any(s in k for s in ('conv9h','xtd9b','xtdb','gate_bwd','xw64_kernel')):
            print(tag,k,{c:round(v/cnt[(k,c)]) for c,v in agg[k].items()})
PY
rm -rf $O/a $O/b
