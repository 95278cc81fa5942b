cd $GRAFT_REPO_ROOT
D=/tmp/splitdir
python tools/gpu_inflate_split.py $D --write 16 > gpurun_out/split_write.log 2>&1
python tools/gpu_inflate_split.py $D --share 0 --loops 4 --repeat 3 2>&1 | tail -2 | tee -a gpurun_out/split_cpu2.txt
