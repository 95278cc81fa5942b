#!/usr/bin/env python3
"""The gfx950 code object inside libdeepbinner_hip.so: kernel metadata (registers, spills, scratch,
LDS) and a census of the forward kernel's instruction stream.  Used by tests/test_code_object.py
(the CPU-box guard on what ships) and by hand:  python tools/code_object.py [lib.so]"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
BUNDLE = 'hipv4-amdgcn-amd-amdhsa--gfx950'


def extract(lib, workdir):
    """-> path of the gfx950 ELF code object pulled out of lib's .hip_fatbin section"""
    fat = os.path.join(workdir, 'fatbin')
    co = os.path.join(workdir, 'gfx950.co')
    subprocess.run([f'{LLVM}/llvm-objcopy', f'--dump-section=.hip_fatbin={fat}', lib, os.path.join(workdir, 'copy.so')],
                   check=True)
    subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}',
                    f'--targets={BUNDLE}', f'--output={co}'], check=True)
    return co


def kernel_metadata(co):
    """-> {kernel symbol: {field: value}} from the AMDGPU metadata note"""
    text = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', co], check=True, capture_output=True,
                          text=True).stdout
    kernels, cur = {}, None
    in_kernels = False
    for line in text.split('\n'):
        if line.strip().startswith('amdhsa.kernels:'):
            in_kernels = True
            continue
        if not in_kernels:
            continue
        if line.startswith('amdhsa.') and not line.startswith('amdhsa.kernels'):
            break
        m = re.match(r'^\s+(- )?\.(\w+):\s+(.*)$', line)
        if not m:
            continue
        dash, key, val = m.groups()
        if dash and re.match(r'^  - ', line):         # a new kernel entry (list item at depth 1)
            cur = {}
            kernels[len(kernels)] = cur
        if cur is not None and re.match(r'^    \.|^  - \.', line):
            cur[key] = val.strip().strip("'")
    return {v['name']: v for v in kernels.values() if 'name' in v}


def disassemble(co, symbol):
    """-> the instruction lines (mnemonic + operands) of one kernel"""
    text = subprocess.run([f'{LLVM}/llvm-objdump', '-d', '--no-show-raw-insn', f'--disassemble-symbols={symbol}', co],
                          check=True, capture_output=True, text=True).stdout
    out = []
    for line in text.split('\n'):
        line = line.split('//')[0].strip()
        if not line or line.endswith(':') or line.startswith(('Disassembly', '/')) or 'file format' in line:
            continue
        out.append(line)
    return out


def census(insts):
    c = collections.Counter()
    for s in insts:
        op = s.split()[0]
        if op.startswith('v_mfma'):
            c['mfma'] += 1
            c['mfma:' + op] += 1
        elif op in ('v_writelane_b32', 'v_readlane_b32'):
            c[op] += 1
        elif op.startswith('v_pk_'):
            c['valu_pk'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('scratch_'):
            c['scratch'] += 1
        elif op.startswith('global_load_lds'):
            c['lds_dma'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_')):
            c['vmem'] += 1
            if op.startswith('flat_'):
                c['flat'] += 1
        elif op == 's_barrier':
            c['s_barrier'] += 1
        elif op == 's_waitcnt':
            c['s_waitcnt'] += 1
    c['instructions'] = len(insts)
    return dict(c)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), '..', 'deepbinner_amd',
                                                             'libdeepbinner_hip.so')
    with tempfile.TemporaryDirectory() as d:
        co = extract(lib, d)
        md = kernel_metadata(co)
        out = {}
        for name, m in md.items():
            keep = {k: m.get(k) for k in ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count',
                                          'sgpr_spill_count', 'private_segment_fixed_size',
                                          'group_segment_fixed_size', 'max_flat_workgroup_size')}
            out[name] = keep
        fwd = next(n for n in md if n.startswith('_ZN3dbh18dbh_forward_kernel'))
        out['forward_census'] = census(disassemble(co, fwd))
        print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
