#!/usr/bin/env python3
"""Basic-block view of a kernel's gfx950 ISA (no GPU needed: hipcc cross-compiles).

    tools/isa_blocks.py orb_slam2_annotate_amd/csrc/k_desc.hip k_orient_descILi64ELi4ELi2ELb0      # blocks
    tools/isa_blocks.py orb_slam2_annotate_amd/csrc/k_desc.hip k_orient_descILi64ELi4ELi2ELb0 --mem  # + loads / stores / vmcnt waits
    tools/isa_blocks.py orb_slam2_annotate_amd/csrc/k_blur.hip --list                               # mangled kernel names

Per basic block: VALU / SALU / total instructions and the branches that end it; the footer gives the static totals, the
registers, the LDS bytes and the occupancy the compiler reports.  This is how the two changes of round 4's last session were
found: a run-time experiment switch that had cut the sampling of k_orient_desc into 32 blocks of 5-7
instructions, and `s_waitcnt vmcnt(0)` between loads the source issues back to back (--mem).  Dynamic counts are the block
counts times the trips you read off the loop structure; the hardware's own count is SQ_INSTS_VALU (tools/collect_valu.py).
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fdenormal-fp-math=ieee", "--cuda-device-only", "-S"]  # = csrc/Makefile's CXXFLAGS, device pass only


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = {a for a in sys.argv[1:] if a.startswith("--")}
    extra = [a[2:] for a in opts if a.startswith("---D")]  # (---DNAME=V passes -DNAME=V)
    if not args:
        raise SystemExit(__doc__)
    src = Path(args[0]).resolve()
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", str(out), str(src)], capture_output=True, text=True,
                           cwd=src.parent)
        if r.returncode != 0 or not out.exists():
            raise SystemExit(r.stderr[-3000:])
        lines = out.read_text().split("\n")
    kernels = [l[:-1].split(":")[0] for l in lines if re.match(r"^_Z\w+:", l)]
    if "--list" in opts or len(args) < 2:
        print("\n".join(kernels))
        return
    cand = [k for k in kernels if args[1] in k]
    if not cand:
        raise SystemExit(f"no kernel name contains {args[1]!r}; --list shows them")
    name = cand[0]
    i = next(k for k, l in enumerate(lines) if l.startswith(name + ":"))
    j = i
    while not lines[j].startswith(".Lfunc_end"):
        j += 1
    blocks, cur = [], ["entry", 0, 0, 0, []]
    mem = []
    for l in lines[i + 1:j]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), 0, 0, 0, []]
            continue
        t = l.split()
        if not t or t[0].startswith(";") or t[0].startswith("."):
            continue
        op = t[0]
        cur[3] += 1
        if op.startswith("v_"):
            cur[1] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            cur[4].append(op[2:] + " " + t[1])
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
            cur[2] += 1
        if "vmcnt" in l or op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            mem.append((cur[0], l.strip()[:100]))
    blocks.append(cur)
    print(name)
    for b in blocks:
        print("  %-12s valu %4d  salu %4d  all %4d  %s" % (b[0], b[1], b[2], b[3], ", ".join(b[4])))
    print("static: valu %d  salu %d  branches %d  blocks %d" % (sum(b[1] for b in blocks), sum(b[2] for b in blocks),
                                                              sum(len(b[4]) for b in blocks), len(blocks)))
    for l in lines[j:j + 80]:
        if any(k in l for k in ("NumVgprs:", "NumSgprs:", "ScratchSize", "Occupancy", "LDSByteSize")):
            print(" ", l.strip("; \t"))
    if "--mem" in opts:
        print("global / flat memory instructions and vmcnt waits, in program order:")
        for b, l in mem:
            print("  %-12s %s" % (b, l))


if __name__ == "__main__":
    main()
