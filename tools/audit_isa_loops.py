"""Per kernel of an ISA listing (hipcc -S --cuda-device-only): the basic blocks inside loops that hold MFMAs, with their
counts of v_accvgpr moves (accumulators / spilled values wandering between the register halves) and scratch accesses.
Found the split-bf16 weight gradient's 452 moves per 144 MFMAs (profiles/r05_conv_bf3.md).
Usage: python tools/audit_isa_loops.py file.s [min_moves_per_mfma]"""
import sys


def main():
    L = open(sys.argv[1]).read().split("\n")
    thresh = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
    starts = [k for k, l in enumerate(L) if l.startswith("_Z") and l.rstrip().endswith(")") is False and ": ;" in l]
    for i in starts:
        name = L[i].split(":")[0]
        e = next((j for j in range(i, len(L)) if "s_endpgm" in L[j]), len(L))
        blocks, cur = [], ["<entry>", []]
        for l in L[i:e]:
            if l.startswith(".LBB"):
                blocks.append(cur)
                cur = [l.split(":")[0] + (" LOOP" if "Loop" in l else ""), []]
            else:
                cur[1].append(l)
        blocks.append(cur)
        tot_m = tot_a = tot_s = 0
        for lab, ls in blocks:
            if "LOOP" not in lab:
                continue
            tot_m += sum("v_mfma" in x for x in ls)
            tot_a += sum("v_accvgpr" in x for x in ls)
            tot_s += sum("scratch_" in x for x in ls)
        if tot_m and (tot_a >= thresh * tot_m or tot_s):
            print("%-110s loop blocks: mfma %4d  v_accvgpr %4d  scratch %3d" % (name[:110], tot_m, tot_a, tot_s))


if __name__ == "__main__":
    main()
