cd $GRAFT_REPO_ROOT
export DEEPBINNER_INFLATE_NOCHECK=1
for i in 1 2; do
for name in base k2a1 k2a2 k2a4 k2a7; do
  lib=deepbinner_amd/csrc/_variants/$name.so
  [ "$name" = base ] && lib=deepbinner_amd/libdeepbinner_hip.so
  echo "$name: $(DEEPBINNER_HIP_LIB=$PWD/$lib timeout 200 python tools/inflate_rate.py 4000 27000 uniform | cut -c1-200)"
done
done
