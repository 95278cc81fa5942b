"""A CPU-box guard on the code object that ships: the forward kernel's
correctness rests on hand-counted waits and on spacing the compiler does not see, its speed on
fitting 256 VGPRs without a spill - a compiler bump or an edit that breaks either is found HERE, not on
a GPU.  Reads the gfx950 code object out of the built libdeepbinner_hip.so (tools/code_object.py):
kernel metadata (llvm-readelf --notes) and the instruction stream (llvm-objdump -d)."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))
import code_object                                           # noqa: E402

LIB = os.path.join(REPO, 'deepbinner_amd', 'libdeepbinner_hip.so')
GOLD = os.path.join(REPO, 'tests', 'golden', 'code_object.json')
CSRC = os.path.join(REPO, 'deepbinner_amd', 'csrc')

pytestmark = pytest.mark.skipif(not os.path.exists(code_object.LLVM + '/llvm-objdump'),
                                reason='no ROCm LLVM tools here')


@pytest.fixture(scope='module')
def shipped():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return code_object.summary(LIB)


def layout_constants():
    """kLdsFloats and the MFMA count per window, from dbh_layout.h itself (compiled for the host)"""
    src = ('#include <cstdio>\n#include "dbh_layout.h"\nint main() { std::printf("%d %d\\n", dbh::kLdsFloats, '
           'dbh::forward_mfmas(13)); }\n')
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, 'layout')
        subprocess.run(['g++', '-std=c++17', '-I', CSRC, '-x', 'c++', '-', '-o', exe], input=src.encode(), check=True)
        lds_floats, mfmas = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    return int(lds_floats), int(mfmas)


def test_forward_kernel_registers_and_memory(shipped):
    md = shipped['metadata']
    assert int(md['vgpr_count']) <= 256 and int(md['agpr_count']) == 0
    assert int(md['vgpr_spill_count']) == 0, 'the forward kernel spills vector registers'
    assert int(md['private_segment_fixed_size']) == 0, 'the forward kernel uses scratch memory'
    lds_floats, _ = layout_constants()
    assert int(md['group_segment_fixed_size']) == 4 * lds_floats <= 160 * 1024
    c = shipped['census']
    assert c.get('scratch', 0) == 0
    # one workgroup of 8 waves per CU, 2 waves per SIMD: 256 registers each is what there is


def test_forward_kernel_matrix_instructions(shipped):
    c = shipped['census']
    mfma_ops = [k for k in c if k.startswith('mfma:')]
    assert mfma_ops == ['mfma:v_mfma_f32_16x16x4_f32'], mfma_ops          # exact fp32: nothing else
    gold = json.load(open(GOLD))
    assert c['mfma'] == gold['mfma_static'], 'static MFMA count changed: python tools/code_object.py --bless'
    assert c.get('lds_dma', 0) == gold['lds_dma'] and c.get('s_barrier', 0) == gold['s_barrier']
    _, per_window = layout_constants()
    assert per_window == 9156          # (dbh_layout.h: forward_mfmas; bench.py counts executed FLOP with it)


def test_hand_kept_hazards(shipped):
    # no reader of an MFMA result closer than the 10 wait states hipcc itself keeps
    assert shipped['mfma_read_hazards'] == [], shipped['mfma_read_hazards'][:5]
    # the literal s_waitcnt lgkmcnt(n) sites: the hand-counted LDS pipelines
    gold = json.load(open(GOLD))
    assert shipped['lgkm_waits'] == gold['lgkm_waits'], (
        'the LDS wait counts of the forward kernel changed - if that was meant: '
        'python tools/code_object.py --bless, and read the diff')


def test_guard_notices_a_removed_wait_and_a_spill():
    """The checks above on doctored instruction streams: they must fire."""
    base = ['v_mfma_f32_16x16x4_f32 v[20:23], v117, v125, v[20:23]', 'v_add_u32_e32 v7, 0x1a920, v3', 's_nop 7',
            's_nop 0', 'v_pk_add_f32 v[8:9], v[20:21], v[30:31] clamp']
    assert code_object.mfma_read_hazards(base) == []
    assert code_object.mfma_read_hazards(base[:2] + base[3:]) != []         # the s_nop 7 taken out
    assert code_object.mfma_read_hazards(['v_mfma_f32_16x16x4_f32 v[20:23], v1, v2, 0',
                                          'ds_write_b64 v5, v[22:23] offset:16']) != []
    assert code_object.mfma_read_hazards(['v_mfma_f32_16x16x4_f32 v[20:23], v1, v2, 0',
                                          'v_mfma_f32_16x16x4_f32 v[20:23], v3, v4, v[20:23]']) == []   # accumulate
    assert code_object.lgkm_wait_histogram(['s_waitcnt lgkmcnt(3)', 's_waitcnt vmcnt(0) lgkmcnt(0)',
                                            's_waitcnt lgkmcnt(3)']) == {'3': 2}
    assert code_object.census(['scratch_store_dword off, v2, off', 'v_mfma_f32_32x32x2_f32 v[0:15], v1, v2, 0'])[
        'scratch'] == 1
