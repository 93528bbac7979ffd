mkdir -p gpurun_out/official3; export TMPDIR=/tmp; O=gpurun_out/official3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for w in a16w4_4096_m16; do
  timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline >> $O/bench_others.jsonl 2>> $O/bench_others.err
done
tail -3 $O/smoke.log; tail -6 $O/pytest.log; cat $O/bench_default.json
python - <<'PY'
import json
for l in open('gpurun_out/official3/bench_others.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:34], '|', r['kernel'], 'kern_us',r['kernel_us'],'achieved',r['achieved'],r['unit'],'frac',r['frac'])
PY
