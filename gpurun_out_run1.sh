set -x
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/probe.txt 2>&1
import os, torch
print("ref exists:", os.path.isdir('/root/reference/gemlite'))
p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, getattr(p,'gcnArchName','?'), p.total_memory>>30, "GiB")
print("cpus", os.cpu_count()); os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)' ; free -g | head -2")
PY
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/bench_m1.json 2> gpurun_out/bench_m1.err
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $PWD/gpurun_out/prof_m1 -o m1 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_m1.log 2>&1
tail -5 gpurun_out/smoke.log; tail -30 gpurun_out/pytest.log; cat gpurun_out/bench_m1.json
