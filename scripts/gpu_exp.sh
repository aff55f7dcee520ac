export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/clk -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --frames 1 --no-cpu-baseline --conv-iters 1 > /dev/null 2>&1 )
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob('/tmp/clk/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('/tmp/clk/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
agg = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    if 'conv_halo' not in r['Kernel_Name'] and 'render_kernel' not in r['Kernel_Name']: continue
    k = (r['Kernel_Name'][34:58], r.get('Grid_Size', ''))
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value']); a[2] += dur[r['Dispatch_Id']]
for k, a in agg.items():
    print(k, 'n', a[0], 'avg_us %.1f' % (a[2]/a[0]/1e3), 'clock GHz (GRBM/8/time) %.3f' % (a[1]/8/a[2]))
PY
