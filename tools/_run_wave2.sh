cd $GRAFT_REPO_ROOT
for i in 1 2; do
for name in base s21 s29; do
  lib=deepbinner_amd/csrc/_variants/$name.so
  [ "$name" = base ] && lib=deepbinner_amd/libdeepbinner_hip.so
  echo "$name: $(DEEPBINNER_HIP_LIB=$PWD/$lib timeout 200 python tools/inflate_rate.py 4000 27000 uniform | cut -c1-200)"
  echo "$name lognormal: $(DEEPBINNER_HIP_LIB=$PWD/$lib timeout 200 python tools/inflate_rate.py 4000 27000 | cut -c1-200)"
done
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
