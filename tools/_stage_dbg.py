import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
from deepbinner_amd import hip_backend
from deepbinner_amd.model_format import ModelWeights
w, _ = ModelWeights.load('/root/repo/deepbinner_amd/models/EXP-NBD103_read_starts.dbw')
m = hip_backend.HipModel(w, device=0)
g = np.load('/root/repo/tests/golden/stages_EXP-NBD103_read_starts.npz')
got = m.debug_stage(g['x'], 'E'); want = g['E']
print(got.shape, want.shape)
err = np.abs(got - want)
print('max err', err.max(), 'scale', np.abs(want).max())
# per channel block of 16, per position
e = err.reshape(err.shape[0], 32, 192)
for blk in range(12):
    b = e[:, :, blk*16:(blk+1)*16]
    print('channels %3d..%3d max err %.3e  worst positions %s' % (blk*16, blk*16+15, b.max(), np.argsort(b.max(axis=(0,2)))[-4:]))
b = e[0, :, 0:16]
print(np.round(b.max(axis=1), 4))
