"""Build-time check for gemm_nt_v3: the inline-asm operand loads write registers that hipcc
believes are ready at once; nothing may read (copy / spill) them before the counted wait that
covers them.  Usage: python tools/check_v3_isa.py [path/to/gemm_nt_v3.s]"""
import re
import subprocess
import sys
import tempfile
import os

def regs_of(text):
    regs = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', text):
        regs.add(int(m.group(1)))
    return regs

def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        here = os.path.dirname(os.path.abspath(__file__))
        src = os.path.join(here, '..', 'dino_amd', 'csrc', 'gemm_nt_v3.hip')
        path = os.path.join(tempfile.gettempdir(), 'gemm_nt_v3_check.s')
        subprocess.check_call(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-munsafe-fp-atomics',
                               '-S', '--cuda-device-only', src, '-o', path])
    lines = open(path).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_ZN12_GLOBAL__N_115gemm_nt_bf16_v3\w+:', l)]
    ok = True
    for st in starts:
        name = lines[st].split(':')[0]
        end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
        body = lines[st:end]
        dests, last = set(), None
        for k, l in enumerate(body):
            m = re.search(r'buffer_load_dword(?:x4)? (v\[\d+:\d+\]|v\d+), v\d+, s\[', l)
            if m and 'lds' not in l and body[k - 1].strip().startswith(';;#ASMSTART'):
                dests |= regs_of(m.group(1))
                last = k
        if last is None:
            print(name, 'no asm loads found')
            ok = False
            continue
        waits = [k for k, l in enumerate(body) if k > last and re.search(r's_waitcnt vmcnt\((0|4|8|16)\)', l)
                 and body[k - 1].strip().startswith(';;#ASMSTART')]
        w_end = waits[0]
        bad = []
        for k in range(last + 1, w_end):
            l = body[k].split(';')[0].strip()
            if not l or l.startswith('.') or l.endswith(':'):
                continue
            if regs_of(l) & dests:
                bad.append((k, l))
        scratch = [l for l in body if 'scratch_' in l or 'buffer_store_dword v' in l and 'offset' in l and 's[0:3]' in l]
        print(f'{name}: {len(dests)} load registers, {w_end - last} instructions to the covering wait, '
              f'{len(bad)} touch them, scratch ops {len(scratch)}')
        for b in bad[:10]:
            print('    ', b)
        ok = ok and not bad and not scratch
    sys.exit(0 if ok else 1)

main()
