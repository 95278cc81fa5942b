"""
deepbinner_amd - Deepbinner's classify path on an AMD MI355X (gfx950).

The modules carry the reference's names (``classify``, ``load_fast5s``, ``trim_signal``,
``realtime``, ``bin``, ``dtw_semi_global``, ``deepbinner`` for the command line) and its call
surface; the arithmetic is in three shared libraries behind C ABIs (``include/``):
``libdeepbinner_hip.so`` (the network, hand-written HIP), ``libdeepbinner_fast5.so`` (the fast5
loader, host C++) and ``libdeepbinner_dtw.so`` (semi-global DTW, HIP).  Nothing is imported
eagerly: ``import deepbinner_amd`` works on a machine without a GPU, the backends complain when
they are first used.
"""
