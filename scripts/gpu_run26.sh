mkdir -p gpurun_out gpurun_out/prof3; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_IFETCH"
for var in "8:0,1,8,0" "4:0,1,44,0"; do
  IFS=: read tag tun <<< "$var"
  i=0
  for P in "$P1" "$P2" "$P3"; do i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $PWD/gpurun_out/prof3/mi${tag}_p$i -o r -- python bench.py --workload a16w4_4096_m256 --tuning $tun --steps 2 --warmup 1 --no-cpu-baseline --no-graph --kernel-samples 4 > gpurun_out/prof3/mi${tag}_p$i.log 2>&1
  done
done
python - <<'PY'
import csv,collections,glob
for d in sorted(glob.glob('gpurun_out/prof3/mi*_p*')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'tiled' in r['Kernel_Name'] or 'pipe' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(d.split('/')[-1],k,round(sum(v)/len(v)),len(v))
PY
