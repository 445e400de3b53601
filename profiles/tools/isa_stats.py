#!/usr/bin/env python3
"""Static facts about the gfx950 code object bundled in libmld_hip.so, per kernel: registers, spills, scratch, LDS (from
the code object's metadata notes) and the instruction mix the verdicts quote (lane moves of spilled scalars, f64 division
sequences, scratch accesses, LDS permutes) from the disassembly.

    python3 profiles/tools/isa_stats.py [library.so] > profiles/r5_isa.txt

Runs anywhere (no GPU): llvm-objdump --offloading extracts the bundle, llvm-readelf --notes prints the metadata,
llvm-objdump -d the code."""
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
ROOT = Path(__file__).resolve().parents[2]
COUNTED = (("v_readlane", r"\bv_readlane_b32\b"), ("v_writelane", r"\bv_writelane_b32\b"),
           ("v_div_*", r"\bv_div_(scale|fmas|fixup)_f(32|64)\b"), ("scratch_*", r"\bscratch_(load|store)_"),
           ("ds_bpermute", r"\bds_bpermute_b32\b"), ("f64 VALU", r"\bv_[a-z0-9_]+_f64\b"),
           ("global/flat mem", r"\b(global|flat)_(load|store|atomic)_"), ("ds_*", r"\bds_(read|write|load|store|add|max|min|or)"),
           ("s_waitcnt", r"\bs_waitcnt\b"))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "mono_lidar_depth_amd" / "lib" / "libmld_hip.so"
    with tempfile.TemporaryDirectory() as td:
        work = Path(td) / lib.name
        shutil.copy(lib, work)
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(work)], capture_output=True, text=True, check=True)
        co = [p for p in Path(td).iterdir() if "amdgcn" in p.name]
        if not co:
            sys.exit("no amdgcn code object in " + str(lib))
        co = co[0]
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
        dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(co)], capture_output=True,
                             text=True).stdout
    # ---- metadata: one block per kernel ("  - .agpr_count: ..." opens it; its keys are sorted, .name sits in the middle)
    kernels = {}
    block = None
    blocks = []
    for line in notes.splitlines():
        if re.match(r"^  - \.", line):
            block = {}
            blocks.append(block)
            line = "    " + line[4:]
        m = re.match(r"^    \.(\w+):\s*(.*)$", line)
        if m and block is not None:
            block[m.group(1)] = m.group(2).strip().strip("'")
    for b in blocks:
        if b.get("name", "").startswith("_Z"):
            kernels[b["name"]] = {k: int(v) for k, v in b.items() if re.fullmatch(r"-?\d+", v or "")}
    # ---- disassembly: instructions per function
    funcs = {}
    name = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            name = m.group(1)
            funcs[name] = []
            continue
        if name and line.startswith("\t"):
            funcs[name].append(line.strip())
    dm = demangle(sorted(set(kernels) | set(funcs)))
    print(f"# ISA statistics of {lib.name} (gfx950 code object; profiles/tools/isa_stats.py)")
    print()
    hdr = ["kernel", "instr", "VGPR", "AGPR", "SGPR", "VGPR spills", "SGPR spills", "scratch B", "LDS B (static)"] + \
          [c[0] for c in COUNTED]
    print("| " + " | ".join(hdr) + " |")
    print("|" + "---|" * len(hdr))
    for sym in sorted(kernels, key=lambda s: -len(funcs.get(s, []))):
        md, ins = kernels[sym], funcs.get(sym, [])
        if not ins:
            continue
        short = re.sub(r"\(.*", "", dm.get(sym, sym)).replace("void ", "").replace("mld::", "")
        row = [f"`{short}`", str(len(ins)), str(md.get("vgpr_count", "")), str(md.get("agpr_count", "")),
               str(md.get("sgpr_count", "")), str(md.get("vgpr_spill_count", "")), str(md.get("sgpr_spill_count", "")),
               str(md.get("private_segment_fixed_size", "")), str(md.get("group_segment_fixed_size", ""))]
        text = "\n".join(ins)
        row += [str(len(re.findall(rx, text))) for _, rx in COUNTED]
        print("| " + " | ".join(row) + " |")
    total = sum(len(v) for k, v in funcs.items() if k in kernels)
    print()
    print(f"{len([k for k in kernels if funcs.get(k)])} kernels, {total} instructions, "
          f"code object text of the kernels ~{total * 6 // 1024} KiB (6 B per instruction on average)")


if __name__ == "__main__":
    main()
