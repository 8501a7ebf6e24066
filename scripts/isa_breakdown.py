"""Per-opcode breakdown of one kernel of the gfx950 assembly hipcc emits for ea_kernels.hip.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only csrc/ea_kernels.hip -o k.s
  python scripts/isa_breakdown.py k.s 'ea_eval_fused_kernelIfLi1ELi0ELi256ELb0' [more substrings ...]

Counts are static (every instruction of the kernel's text once); the fused kernel is straight-line code apart from
the uniform fix-up / general-quaternion branches, which are listed per basic block so they can be told apart.
Classes follow the issue costs measured in profiles/r01_issue_rates_2.txt: "cheap" = VALU with at most two VGPR
reads and no SGPR operand, "full" = three VGPR reads, an SGPR operand, DPP/permlane, conversions, all fp64."""
import collections
import re
import sys


def kernels(path):
    cur, out = None, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
        if cur is not None:
            out[cur].append(line.rstrip("\n"))
    return out


def classify(op, args):
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        nv = len(re.findall(r"\bv\d+|\bv\[\d+:\d+\]", args.split(",", 1)[1] if "," in args else ""))
        has_s = bool(re.search(r"\bs\d+|\bs\[\d+:\d+\]|vcc|exec", args.split(",", 1)[1] if "," in args else ""))
        if "_f64" in op or "dpp" in op or "permlane" in op or op.startswith(("v_cvt", "v_floor", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_readlane", "v_readfirstlane", "v_cmp", "v_cndmask", "v_mad_", "v_lshl", "v_ashr", "v_add_co", "v_addc", "v_mul_lo", "v_mul_hi", "v_med3", "v_min", "v_max")):
            return "valu_full"
        if nv >= 3 or has_s:
            return "valu_full"
        return "valu_cheap"
    return "other"


def main():
    ks = kernels(sys.argv[1])
    for want in sys.argv[2:]:
        for name, lines in ks.items():
            if want not in name:
                continue
            ops, cls, blocks = collections.Counter(), collections.Counter(), []
            blk = ["entry", collections.Counter()]
            for l in lines:
                m = re.match(r"^(\.LBB\w+):", l)
                if m:
                    blocks.append(blk)
                    blk = [m.group(1), collections.Counter()]
                    continue
                m = re.match(r"^\t(\w+)\s*(.*?)(\s*;.*)?$", l)
                if not m or m.group(1).startswith(".") or not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_)", m.group(1)):
                    continue
                op, args = m.group(1), m.group(2)
                c = classify(op, args)
                ops[op] += 1
                cls[c] += 1
                blk[1][c] += 1
            blocks.append(blk)
            print("==", name)
            print("  by class:", dict(cls), " VALU total:", cls["valu_full"] + cls["valu_cheap"])
            print("  by block:", "; ".join(f"{b[0]}:{sum(b[1].values())}(valu {b[1]['valu_full'] + b[1]['valu_cheap']})" for b in blocks if sum(b[1].values())))
            for op, n in ops.most_common():
                print(f"  {n:5d}  {op}")


if __name__ == "__main__":
    main()
