#!/usr/bin/env python
"""FP64 instruction counts of the fused key-switch kernel, read from the BUILT code object (not from constants in a script).

`llvm-objdump --offloading` extracts the gfx950 code object from the key-switch translation unit's object file, `llvm-objdump -d`
disassembles it; the kernel's loops are recovered from its backward branches (limb loop > digit loop; the loop over the two inverse
transforms) and every `v_*_f64` instruction is weighted with the trip counts of the loops that contain it.  `fp64_per_thread(k, digits
per limb)` is what bench.py prices against the FP64 issue rate (one wave-instruction per ~4.4 cycles per SIMD,
profiles/r01_ubench_mulmod.txt) to get the key switch's arithmetic floor.

    python tools/ks_isa_counts.py            # prints the structure found in k_keyswitch_rr<13, ArF64T<0>, 1, true, false>
"""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
KERNEL = "_Z14k_keyswitch_rrILi13E6ArF64TILi0EELi1ELb1ELb0EE"          # k_keyswitch_rr<13, ArF64T<0>, 1, true, false>


def disassemble(obj, kernel=KERNEL):
    """[(address, mnemonic, operands)] of `kernel` in the device code object bundled in `obj`"""
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, os.path.basename(obj))
        os.symlink(obj, tmp)
        subprocess.check_call([OBJDUMP, "--offloading", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in glob.glob(tmp + ".*") if "amdgcn" in f]
        if not co:
            raise RuntimeError("no gfx950 code object in %s" % obj)
        txt = subprocess.check_output([OBJDUMP, "-d", co[0]], text=True)
    ins, on = [], False
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            on = m.group(1).startswith(kernel)
            continue
        if not on:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if not ins:
        raise RuntimeError("kernel %s not found" % kernel)
    return ins


def loops(ins):
    """{header address: [addresses of its back edges]}: backward branches, minus the compiler's out-of-line trampolines (an s_branch that
    directly follows another s_branch is the landing pad of a forward conditional branch jumping back into the body, not a loop)"""
    out = {}
    for k, (addr, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"(-?\d+)", args)
            if not m or (op == "s_branch" and k and ins[k - 1][1] == "s_branch"):
                continue
            nxt = ins[k + 1][0] if k + 1 < len(ins) else addr + 4
            off = int(m.group(1))
            off = off - 65536 if off >= 32768 else off         # simm16, in dwords from the next instruction
            tgt = nxt + 4 * off
            if tgt <= addr:
                out.setdefault(tgt, []).append(addr)
    return out


def structure(obj=None, kernel=KERNEL):
    obj = obj or os.path.join(ROOT, "cryptonets_amd", "lib", "obj", "cn_l_ks_f64l.o")
    ins = disassemble(obj, kernel)
    f64 = [a for a, op, _ in ins if op.startswith("v_") and "_f64" in op]
    other = [a for a, op, _ in ins if op.startswith("v_") and "_f64" not in op and not op.startswith("v_readfirstlane")]   # the rest of the VALU stream
    within = lambda lo, hi: sum(lo <= a <= hi for a in f64)
    owithin = lambda lo, hi: sum(lo <= a <= hi for a in other)
    # the kernel has two loop nests with FP64 work: (limb loop > digit loop), which the compiler rotates onto ONE header with two back
    # edges (digit: the nearer one, limb: the farther one), and behind it the loop over the two inverse transforms
    big = sorted((h, sorted(e)) for h, e in loops(ins).items() if within(h, max(e)) > 100)
    if len(big) == 3 and len(big[0][1]) == 1 and len(big[1][1]) == 2 and big[0][0] < big[1][0] and max(big[1][1]) < big[0][1][0]:
        # since the source words of a limb are requested a limb ahead (round 5) the limb loop has a header of its own: limb loop > digit loop with two back
        # edges (the nearer one: next digit; the farther one: the end of the recentring block that runs every `accmax` terms - counted once per limb, as before)
        (h0, (limb_end,)), (h1, (digit_end, _settle_end)), (h2, e2) = big
    elif len(big) == 2 and len(big[0][1]) == 2:
        (h1, (digit_end, limb_end)), (h2, e2) = big
        h0 = h1
    else:
        raise RuntimeError("unrecognised loop structure in the key-switch kernel: %s" % [(hex(h), [hex(x) for x in e]) for h, e in big])
    tail_end = max(e2)
    limb_only = lambda f: f(h0, limb_end) - f(h1, digit_end)
    return dict(kernel="k_keyswitch_rr<13, ArF64T<0>, 1, true, false>", instructions=len(ins), fp64_total=len(f64),
                fp64_digit_loop=within(h1, digit_end), fp64_limb_loop_only=limb_only(within),
                fp64_tail_loop=within(h2, tail_end), fp64_once=len(f64) - within(h0, limb_end) - within(h2, tail_end),
                valu_digit_loop=owithin(h1, digit_end), valu_limb_loop_only=limb_only(owithin),
                valu_tail_loop=owithin(h2, tail_end), valu_once=len(other) - owithin(h0, limb_end) - owithin(h2, tail_end))


def fp64_per_thread(k, digits_per_limb, obj=None, kernel=KERNEL):
    """FP64 instructions one thread executes for one (ciphertext, output limb): k source limbs x their digits, 2 inverse transforms"""
    s = structure(obj, kernel)
    return sum(d * s["fp64_digit_loop"] + s["fp64_limb_loop_only"] for d in digits_per_limb) + 2 * s["fp64_tail_loop"] + s["fp64_once"], s


def valu_per_thread(digits_per_limb, s):
    """non-FP64 VALU instructions of the same thread (digit extraction, selects, addressing; every one at least a full-rate issue slot)"""
    return sum(d * s["valu_digit_loop"] + s["valu_limb_loop_only"] for d in digits_per_limb) + 2 * s["valu_tail_loop"] + s["valu_once"]


SQUARE_KERNEL = "_Z14k_square_fusedILi13E6ArF64TILi0EELb1EE"          # k_square_fused<13, ArF64T<0>, true>: fused squaring, operand parked in LDS


def _branches(ins):
    """[(index, address, mnemonic, target address)] of every branch of the kernel"""
    out = []
    for k, (addr, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"(-?\d+)", args)
            if not m:
                continue
            off = int(m.group(1))
            off = off - 65536 if off >= 32768 else off
            nxt = ins[k + 1][0] if k + 1 < len(ins) else addr + 4
            out.append((k, addr, op, nxt + 4 * off))
    return out


def square_structure(obj=None, kernel=SQUARE_KERNEL):
    """FP64 / other VALU instructions ONE THREAD of k_square_fused executes for one (ciphertext, limb) block, from the built code object.
    The kernel is one rolled loop over the three steps (cn_k_rr.hip.h); the compiler rotates it so that the body starts with the inverse
    transform + store (all three steps), followed by the exit test, then EITHER the forward transform with the step-0 / step-1 tensor
    variants (each variant skipped by one forward branch in the other step) OR the step-2 block (A1^2 out of LDS).  Weights: inverse part
    x 3, forward part x 2, each skippable variant block x 1, step-2 block x 1."""
    obj = obj or os.path.join(ROOT, "cryptonets_amd", "lib", "obj", "cn_l_rr_f64l.o")
    ins = disassemble(obj, kernel)
    addrs = [a for a, _, _ in ins]
    is64 = lambda op: op.startswith("v_") and "_f64" in op
    isv = lambda op: op.startswith("v_") and "_f64" not in op and not op.startswith("v_readfirstlane")
    br = _branches(ins)
    back = [(a, t) for _, a, _, t in br if t <= a and a - t > 4096]
    if not back:
        raise RuntimeError("no step loop found in %s" % kernel)
    head = min(t for _, t in back)
    last = max(a for a, _ in back)
    end = addrs[-1]
    body = [(k, a, op, t) for k, a, op, t in br if head <= a <= last]
    exit_br = [(a, t) for _, a, op, t in body if op.startswith("s_cbranch") and t > last]
    if len(exit_br) != 1:
        raise RuntimeError("unrecognised exit structure in %s: %s" % (kernel, [(hex(a), hex(t)) for a, t in exit_br]))
    exit_at = exit_br[0][0]
    # the first forward conditional branch behind the exit test selects the step-2 block
    sel = [(a, t) for _, a, op, t in body if op.startswith("s_cbranch") and a > exit_at and exit_at < t <= last]
    if not sel:
        raise RuntimeError("no step selector found in %s" % kernel)
    sel_at, step2_at = sel[0]
    variants = [(a, t) for a, t in sel[1:] if a < step2_at and t <= step2_at + 8]
    if len(variants) != 2:
        raise RuntimeError("expected the step-0 / step-1 tensor variants in %s, found %s" % (kernel, [(hex(a), hex(t)) for a, t in variants]))

    def count(pred, lo, hi):                 # instructions with lo <= address < hi
        return sum(1 for a, op, _ in ins if lo <= a < hi and pred(op))
    res = {}
    for name, pred in (("fp64", is64), ("valu", isv)):
        inv = count(pred, head, exit_at)
        var = [count(pred, a + 4, t) for a, t in variants]
        fwd = count(pred, sel_at, step2_at) - sum(var)
        st2 = count(pred, step2_at, last + 4)
        once = count(pred, addrs[0], head) + count(pred, last + 4, end + 4)
        res[name] = dict(inverse_part=inv, forward_part=fwd, tensor_variants=var, step2_block=st2, once=once,
                         per_thread=3 * inv + 2 * fwd + sum(var) + st2 + once)
    if not 4500 <= res["fp64"]["per_thread"] <= 7500:
        raise RuntimeError("implausible FP64 count %d for k_square_fused (2 forward + 3 inverse 8192-point transforms are ~5600)" % res["fp64"]["per_thread"])
    res["kernel"] = "k_square_fused<13, ArF64T<0>, true>"
    res["instructions"] = len(ins)
    return res


if __name__ == "__main__":
    s = structure(sys.argv[1] if len(sys.argv) > 1 else None)
    print(json.dumps(s, indent=1))
    print(json.dumps(square_structure(), indent=1))
