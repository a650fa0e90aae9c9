#!/usr/bin/env python
"""VGPR / SGPR / spill / LDS figures of the gfx950 kernels bundled in an object file (from the code object's metadata notes).

    python tools/kernel_resources.py cryptonets_amd/lib/obj/cn_l_rr_f64.o [name filter]
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def resources(obj):
    """{demangled kernel name: dict(vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, lds, scratch)}"""
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), tmp)
        subprocess.check_call([LLVM + "/llvm-objdump", "--offloading", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in glob.glob(tmp + ".*") if "amdgcn" in f]
        if not co:
            raise RuntimeError("no gfx950 code object in %s" % obj)
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co[0]], text=True)
    out = {}
    for blk in notes.split("  - .agpr_count:")[1:]:
        f = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % key, blk) else 0
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = dict(agpr=int(blk.split()[0]), vgpr=f("vgpr_count"), sgpr=f("sgpr_count"), vgpr_spill=f("vgpr_spill_count"),
                         sgpr_spill=f("sgpr_spill_count"), lds=f("group_segment_fixed_size"), scratch=f("private_segment_fixed_size"))
    names = list(out)
    dem = subprocess.check_output(["c++filt"] + names, text=True).splitlines()
    return {d.split("(")[0]: out[n] for d, n in zip(dem, names)}



def flat_instructions(obj):
    """{mangled kernel name: count} of flat_load / flat_store / flat_atomic instructions in the gfx950 code of `obj`.  A flat access is what a
    pointer of unknown address space compiles to (an address read from a table, a select between a table entry and a kernel argument); it
    is counted on lgkmcnt as well as vmcnt, so waits for scalar loads and LDS traffic drain it too."""
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), tmp)
        subprocess.check_call([LLVM + "/llvm-objdump", "--offloading", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in glob.glob(tmp + ".*") if "amdgcn" in f]
        txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", co[0]], text=True)
    out, name = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
        elif name and re.match(r"^\s+flat_(load|store|atomic)", line):
            out[name] = out.get(name, 0) + 1
    return out


if __name__ == "__main__":
    res = resources(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-64s %5s %5s %5s %7s %7s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "v-spill", "s-spill", "scratch"))
    for k in sorted(res):
        if flt in k:
            r = res[k]
            print("%-64s %5d %5d %5d %7d %7d %8d" % (k[:64], r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch"]))
