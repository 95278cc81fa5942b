set -x
cd /root/repo
timeout 600 python -m pytest tests/test_inflate.py -m gpu -x -q -s 2>&1 | tail -15 > gpurun_out/wave1_pytest.log
for k in lane wave; do
  DEEPBINNER_INFLATE_KERNEL=$k timeout 300 python tools/inflate_rate.py > gpurun_out/wave1_rate_$k.json 2>&1
done
cd /tmp && export TMPDIR=/tmp
DEEPBINNER_INFLATE_KERNEL=wave timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/wave1_prof -o w -- python /root/repo/tools/inflate_rate.py > /root/repo/gpurun_out/wave1_prof.log 2>&1
cd /root/repo
find gpurun_out/wave1_prof -name "*kernel_stats.csv" | head -1 | xargs cat > gpurun_out/wave1_kernel_stats.csv
