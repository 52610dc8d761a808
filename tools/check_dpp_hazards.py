#!/usr/bin/env python3
"""Static check of the one software-visible hazard our inline assembly can create on gfx950:

    "VALU writes a VGPR; a DPP operand of one of the next TWO wait states reads it"   (2 wait states required)

hipcc's hazard recognizer inserts the s_nop itself between instructions it knows, but the body of an `asm` statement is
opaque to it (it neither sees a v_fmac_f64_dpp inside one as a DPP reader nor as a VALU writer).  This script compiles
the library's device code to assembly with the flags of the Makefile and walks every kernel: for each instruction with a
DPP control (row_newbcast, row_shr, quad_perm, ...) the registers of its DPP source (src0) must not be written by the
preceding VALU instructions within 2 wait states (an instruction = 1 wait state, `s_nop N` = N + 1).
Exit status 1 and a listing if a violation is found.  Usage: check_dpp_hazards.py [file.s]  (no argument: compile)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'gp_mpc_amd', 'csrc')
DPP = re.compile(r'\b(row_newbcast|row_shr|row_shl|row_ror|row_bcast|row_mirror|row_half_mirror|quad_perm|wave_shr|wave_shl|wave_ror|wave_rol|row_share|row_xmask)\b')
REG = re.compile(r'-?\|?v(\d+)\b|-?\|?v\[(\d+):(\d+)\]')


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(line):
    """-> (mnemonic, [operand strings]) or None for labels / directives / comments."""
    line = line.split(';')[0].strip()
    if not line or line.endswith(':') or line.startswith('.') or line.startswith('//'):
        return None
    parts = line.split(None, 1)
    mn = parts[0]
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    return mn, ops


def check(text):
    bad = []
    kernel = '?'
    hist = []                     # (wait states this entry provides, set of VGPRs it writes as a VALU op, text)
    for ln, raw in enumerate(text.splitlines(), 1):
        s = raw.strip()
        m = re.match(r'^([A-Za-z_][\w.$]*):', s)
        if m and not s.startswith('.L'):
            kernel = m.group(1)
            hist = []
            continue
        p = parse(raw)
        if p is None:
            continue
        mn, ops = p
        if mn == 's_nop':
            hist.append((int(ops[0], 0) + 1, set(), s))
            continue
        is_valu = mn.startswith('v_')
        if DPP.search(raw.split(';')[0]):
            # src0 = first source operand = operand 1 (operand 0 is vdst); modifiers follow the last operand after spaces
            src0 = ops[1].split()[0] if len(ops) > 1 else ''
            need = regs(src0)
            ws = 0
            for w, wr, txt in reversed(hist):
                if ws >= 2:
                    break
                if wr & need:
                    bad.append((kernel, ln, s, txt, ws))
                    break
                ws += w
        written = set()
        if is_valu and ops and not mn.startswith(('v_cmp', 'v_cmpx', 'v_readlane', 'v_readfirstlane')):
            written = regs(ops[0].split()[0])
        hist.append((1, written, s))
        if len(hist) > 8:
            hist = hist[-8:]
    return bad


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-value', '--cuda-device-only', '-S',
               os.path.join(CSRC, 'gpmpc_api.hip'), '-o', '-']
        text = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
    n_dpp = sum(1 for l in text.splitlines() if DPP.search(l.split(';')[0]) and parse(l))
    bad = check(text)
    print('%d DPP instructions checked, %d hazard violations' % (n_dpp, len(bad)))
    for k, ln, s, txt, ws in bad[:40]:
        print('  %s line %d: "%s" reads a register written %d wait state(s) earlier by "%s"' % (k, ln, s, ws, txt))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
