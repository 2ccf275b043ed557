#!/usr/bin/env python3
"""Build-time audit of the hand-issued MFMAs (conv_wino4.hip, conv_wino.hip): hipcc pads no hazards for an `asm`
statement, so a compiler-generated VALU instruction that writes a VGPR right in front of an inline-asm MFMA reading
it as A / B operand (e.g. a v_accvgpr_read bringing a parked value back) makes the MFMA read a stale register.
Compiles the file to ISA and reports every inline-asm MFMA whose A / B source was written by one of the two
instructions in front of the asm statement, unless the statement opens with wait states of its own (s_nop).
Usage: python tools/audit_asm_hazards.py [file.hip ...]     exit code 1 when a hazard is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "asvspoof2021_air_amd", "csrc")


def compile_to_asm(src, out):
    sys.path.insert(0, ROOT)
    from asvspoof2021_air_amd import build as b
    cmd = [b.HIPCC] + b.CFLAGS + b.EXTRA.get(os.path.basename(src), []) + ["--cuda-device-only", "-S", src, "-o", out, "-w"]
    subprocess.run(cmd, check=True, capture_output=True)


def regs(tok):
    """v12 -> {12}; v[4:7] -> {4..7}; anything else -> {}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def audit(asm_path):
    lines = open(asm_path).read().splitlines()
    kernel, hazards, n_mfma = "", [], 0
    prev = []  # last instructions outside asm statements: (text, set of VGPRs written)
    i = 0
    while i < len(lines):
        ln = lines[i].strip()
        if re.match(r"^[_A-Za-z0-9.$]+:", ln) and not ln.startswith(".L"):
            kernel = ln.split(":")[0]
            prev = []
        if ln.startswith(";;#ASMSTART"):
            body = []
            i += 1
            while i < len(lines) and not lines[i].strip().startswith(";;#ASMEND"):
                body.append(lines[i].strip())
                i += 1
            padded = bool(body) and body[0].startswith("s_nop")
            for b in body:
                if b.startswith("v_mfma"):
                    n_mfma += 1
                    ops = [t.strip() for t in b.split(None, 1)[1].split(",")]
                    src = regs(ops[1]) | regs(ops[2])
                    if not padded:
                        for text, written in prev[-2:]:
                            if written & src:
                                hazards.append((kernel, text, b))
            prev = []  # whatever the statement did, two instructions later nothing is pending
        elif ln and not ln.startswith((";", ".", "//")) and not ln.endswith(":"):
            parts = ln.split(None, 1)
            op = parts[0]
            written = set()
            if op.startswith("v_") and len(parts) > 1 and not op.startswith(("v_cmp", "v_accvgpr_write")):
                written = regs(parts[1].split(",")[0].strip())
            prev.append((ln, written))
            prev = prev[-2:]
        i += 1
    return n_mfma, hazards


def main(argv):
    files = argv or [os.path.join(CSRC, "conv_wino4.hip"), os.path.join(CSRC, "conv_wino.hip")]
    bad = 0
    for f in files:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            compile_to_asm(f, out)
            n, hz = audit(out)
        print("%s: %d inline-asm MFMAs, %d hazards" % (os.path.basename(f), n, len(hz)))
        for k, w, m in hz[:20]:
            print("   %s: `%s` feeds `%s`" % (k[:60], w, m))
        bad += len(hz)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
