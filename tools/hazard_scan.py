#!/usr/bin/env python3
"""gfx9 hazard the compiler cannot pad inside inline asm: a VMEM instruction must not read an SGPR within 5 wait states of a
VALU write to it (v_readfirstlane / v_readlane / v_cmp with an SGPR destination ...).  hipcc inserts the s_nops for its own
instructions; the LDS-DMA and prefetch loads in this library are inline asm whose scalar base may come straight out of a
v_readfirstlane.  Usage: hazard_scan.py file.s [...] -- prints {(opcode, wait states): count} of violations per file
(assembly from `hipcc -S --cuda-device-only`); exit code 1 if there are any."""
import re
import sys

NEED = 5


def sregs(tok):
    tok = tok.strip(",")
    m = re.match(r"s\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    return scan_text(open(path).read())


def scan_text(text):
    ins = []
    for raw in text.splitlines():
        l = raw.split(";")[0].strip()
        if not l or l.startswith((".", "//")) or l.endswith(":"):
            continue
        ins.append(l)
    last, bad = {}, {}
    for i, l in enumerate(ins):
        parts = l.split()
        op, toks = parts[0], parts[1:]
        if op.startswith("v_") and toks and sregs(toks[0]):          # a VALU instruction whose destination is scalar
            for r in sregs(toks[0]):
                last[r] = i
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            for t in toks:
                for r in sregs(t):
                    if r in last:
                        ws = 0
                        for k in range(last[r] + 1, i):
                            o = ins[k].split()
                            ws += int(o[1]) + 1 if o[0] == "s_nop" else 1
                        if ws < NEED:
                            bad[(op, ws)] = bad.get((op, ws), 0) + 1
        elif op.startswith("s_") and toks:
            for r in sregs(toks[0]):
                last.pop(r, None)                                      # rewritten by the scalar unit: no hazard left
    return bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        bad = scan(p)
        print(p, bad if bad else "clean")
        rc |= bool(bad)
    sys.exit(rc)
