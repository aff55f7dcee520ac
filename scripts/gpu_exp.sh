export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --compute-dtype f32_bf16x3 --steps 1 --warmup 1 --frames 1 --no-cpu-baseline --conv-iters 1"
pass () { n=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/px_$n -o p -- $CMD > /dev/null 2>&1 ); }
pass a SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
pass b GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD TCP_PENDING_STALL_CYCLES_sum SQ_LDS_ADDR_CONFLICT
python3 - <<'PY'
import csv, glob, collections
for n in 'ab':
    f = glob.glob(f'/tmp/px_{n}/**/*counter_collection.csv', recursive=True)
    if not f: print('no data', n); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        kn = r['Kernel_Name']
        if 'conv_halo_split_kernel<2>' not in kn: continue
        if int(r.get('Grid_Size', r.get('Grid_Size_X', 0))) != 2048*512: continue
        k = ('split<2> 64^3', r['Counter_Name'])
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, a in agg.items(): print(k[0], k[1], 'n', a[0], 'avg %.4g' % (a[1]/a[0]))
PY
