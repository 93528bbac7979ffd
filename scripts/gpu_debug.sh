set -x
mkdir -p gpurun_out
timeout 900 python scripts/debug_dump.py > gpurun_out/debug.log 2>&1; echo "rc=$?" >> gpurun_out/debug.log
tail -40 gpurun_out/debug.log
