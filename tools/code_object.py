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


# ---------------------------------------------------------------------------------------------
# The hazards the forward kernel keeps BY HAND.  hipcc pads the distance between an MFMA and the first
# instruction that reads its result - for instructions it sees: not for inline asm (the clamped
# packed adds / fmas of the Winograd output transforms, the ds_write_b64 of the edge rows), and its
# own LDS wait counts know nothing of the inline-asm ds_read pipelines.  So the shipped instruction
# stream is checked itself:
#   * a v_mfma_f32_16x16x4_f32 followed by a VALU / LDS / VMEM instruction that READS one of its four
#     result registers needs 10 wait states in between - what hipcc itself pads such a pair to (LLVM's
#     GCNHazardRecognizer, "XDL write VGPR -> VALU / VMEM / LDS read"; the hand-written blocks
#     start with s_nop 7, s_nop 2 = 11); every instruction in between counts one, s_nop N counts N + 1.  (An MFMA reading the result as its accumulator is not
#     subject to this: back-to-back accumulation is what the pipe is for.)
#   * the histogram of the literal `s_waitcnt lgkmcnt(n)` forms - the hand-counted ones are almost
#     all of the n > 0 - is compared with the committed one (tests/golden/code_object.json).
# ---------------------------------------------------------------------------------------------
_REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def _vregs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def mfma_read_hazards(insts, need=10):
    """-> [(index, instruction, producing mfma, wait states seen)] of readers that come too early"""
    bad = []
    pending = []                     # (result registers, wait states since, text)
    for i, s in enumerate(insts):
        op = s.split()[0]
        ops = s[len(op):]
        parts = [p.strip() for p in ops.split(',')]
        if op.startswith('v_mfma'):
            reads = _vregs(','.join(parts[1:3]))          # A and B operands (the accumulator is exempt)
            writes = _vregs(parts[0])
        elif op.startswith(('v_', 'ds_', 'global_', 'buffer_', 'flat_', 'scratch_')):
            if op.startswith('v_') and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
                reads, writes = _vregs(','.join(parts[1:])), _vregs(parts[0])
            elif op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
                reads, writes = _vregs(ops), set()
            elif ('_load' in op or op.startswith('ds_read')) and not op.startswith('global_load_lds'):
                # a load: the first operand is where the data will land (no read; and the register
                # allocator's reuse of an accumulator for it is the compiler's own business)
                reads, writes = _vregs(','.join(parts[1:])), set()
            else:                                         # stores, atomics, LDS-DMA: everything named is read
                reads, writes = _vregs(ops), set()
        else:
            reads, writes = set(), set()
        for regs, since, text in pending:
            if reads & regs and since < need:
                bad.append((i, s, text, since))
        step = 1
        if op == 's_nop':
            step = int(parts[0]) + 1 if parts and parts[0].isdigit() else 1
        pending = [(r, w + step, t) for (r, w, t) in pending if w + step < need and not (writes and writes >= r)]
        if op.startswith('v_mfma'):
            pending.append((writes, 0, s))
    return bad


def lgkm_wait_histogram(insts):
    h = collections.Counter()
    for s in insts:
        m = re.match(r'^s_waitcnt lgkmcnt\((\d+)\)$', s.strip())
        if m:
            h[int(m.group(1))] += 1
    return {str(k): h[k] for k in sorted(h)}


def summary(lib):
    """everything tests/test_code_object.py looks at, as one dict"""
    with tempfile.TemporaryDirectory() as d:
        co = extract(lib, d)
        md = kernel_metadata(co)
        fwd = next(n for n in md if n.startswith('_ZN3dbh18dbh_forward_kernel'))
        insts = disassemble(co, fwd)
        c = census(insts)
        return {
            'metadata': {k: md[fwd].get(k) for k in ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count',
                                                     'sgpr_spill_count', 'private_segment_fixed_size',
                                                     'group_segment_fixed_size')},
            'census': c,
            'lgkm_waits': lgkm_wait_histogram(insts),
            'mfma_read_hazards': [list(b) for b in mfma_read_hazards(insts)][:20],
        }


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--bless':
        # record what the built library carries (tests/golden/code_object.json): run after every
        # deliberate change of the forward kernel, and read the diff
        lib = os.path.join(os.path.dirname(__file__), '..', 'deepbinner_amd', 'libdeepbinner_hip.so')
        s = summary(lib)
        keep = {'mfma_static': s['census']['mfma'], 'lds_dma': s['census'].get('lds_dma', 0),
                's_barrier': s['census'].get('s_barrier', 0), 'lgkm_waits': s['lgkm_waits'],
                'vgpr_count': int(s['metadata']['vgpr_count']), 'sgpr_spill_count': int(s['metadata']['sgpr_spill_count'])}
        path = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'code_object.json')
        json.dump(keep, open(path, 'w'), indent=1, sort_keys=True)
        print(json.dumps(keep, indent=1, sort_keys=True))
        return
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
