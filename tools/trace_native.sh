cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tn; mkdir -p $OUT
python $R/tools/trace_native.py 20 2>/dev/null | tail -2
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python $R/tools/trace_native.py 6 > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find last occurrences: split into native part (first) and python part (second) by locating reverb inverse kernels
names=[r['Kernel_Name'] for r in rows]
inv=[i for i,n in enumerate(names) if 'part_inv_kernel' in n]
def dump(lo,hi,tag):
    t0=int(rows[lo]['Start_Timestamp'])
    print('==',tag)
    for r in rows[lo:hi+1]:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} q{r.get('Queue_Id','?')} {r['Kernel_Name'][:70]}")
# native: steps 3 warm + 6 = 9 invs, python next 9
k=inv[7]; prev=inv[6]
dump(prev+1,k,'native step')
k=inv[16]; prev=inv[15]
dump(prev+1,k,'python step')
PY
rm -rf $OUT/kt
