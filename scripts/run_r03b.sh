export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r03b; mkdir -p $O
timeout 700 python oracle/run_ref_gpu.py --which ref --budget-s 400 > $O/ref.log 2>&1; echo "ref rc=$?" >> $O/ref.log
timeout 400 python oracle/run_ref_gpu.py --which hip --budget-s 300 > $O/hip.log 2>&1; echo "hip rc=$?" >> $O/hip.log
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -n 6 -p no:cacheprovider > $O/pytest_parity.log 2>&1 ); tail -5 $O/pytest_parity.log
cp gemlite_amd/csrc/libgemlite_hip.so /tmp/lib_keep.so
for v in nt1 nt0; do
  cp scripts/ab/lib_$v.so gemlite_amd/csrc/libgemlite_hip.so
  echo "== $v" >> $O/probe_gemv3.log
  timeout 600 python scripts/probe_gemv3.py 2>&1 | grep '^{' >> $O/probe_gemv3.log
done
cp /tmp/lib_keep.so gemlite_amd/csrc/libgemlite_hip.so
cat $O/probe_gemv3.log
