#!/usr/bin/env python3
"""The asm-issued prefetch of the partition kernels (csrc/tc_msd.hpp) against the ISA the compiler made of it.

The loads are invisible to the compiler's wait counts -- that is their point -- so nothing guards their destination
registers between the load and the explicit `s_waitcnt vmcnt(0)` that lands them: the register allocator may read, move
or reuse them there, and then stale register contents become keys (it happened to another kernel of this library, which
prefetches by plain loads since).  check() cross-compiles the library's device code (no GPU) and
fails if any instruction between the last prefetch load and the landing wait touches a destination register.

  python scripts/check_asm_prefetch.py [-DMACRO ...]     # a variant build's definitions

`make -C text-compression_amd variant TAG=x DEFS="-DMSD_PROFILE"` runs it for every variant .so it builds, and
tests/test_asm_prefetch_registers.py for the default build and the diagnostic variants scripts/README.md documents."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "text-compression_amd")
KERNELS = ["msd_partition_kernelILb0ELb0EE", "msd_partition_kernelILb1ELb0EE",
           "msd_partition_kernelILb0ELb1EE", "msd_partition_kernelILb1ELb1EE"]


def _vregs(text):
    """vector registers named in an operand string: v7, v[12:15]"""
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return regs


def check(defs=()):
    """ISA of the build with the macro definitions `defs` (e.g. ["-DMSD_PROFILE"]); returns the number of prefetch groups
    checked, raises AssertionError with the offending instruction otherwise"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = "_".join(d.strip("-").replace("=", "") for d in defs) or "default"
    isa = "/tmp/textcomp_isa_%s.s" % tag
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                          "-I" + os.path.join(PKG, "csrc"), "-S", "--cuda-device-only", "-o", isa] + list(defs) +
                         [os.path.join(PKG, "csrc", "textcomp.hip")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = open(isa).read().split("\n")
    checked = 0
    for frag in KERNELS:
        start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and frag in l and l.rstrip().endswith(":") or (l.startswith("_Z") and frag in l and ":" in l.split(";")[0]))
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        body = lines[start:end]
        # asm blocks: (first line, last line, text)
        blocks, i = [], 0
        while i < len(body):
            if "#ASMSTART" in body[i]:
                j = next(k for k in range(i, len(body)) if "#ASMEND" in body[k])
                blocks.append((i, j, "\n".join(body[i + 1:j])))
                i = j
            i += 1
        loads = [(a, b, t) for a, b, t in blocks if "global_load" in t]
        assert loads, "no asm prefetch found in " + frag
        # groups of consecutive prefetch blocks (one group per place the prefetch was inlined), each followed by its wait
        groups, cur = [], [loads[0]]
        for blk in loads[1:]:
            if blk[0] - cur[-1][1] < 40: cur.append(blk)
            else:
                groups.append(cur); cur = [blk]
        groups.append(cur)
        for grp in groups:
            dest = set()
            for _, _, t in grp:
                for ln in t.split("\n"):
                    if "global_load" in ln:
                        dest |= _vregs(ln.split(",")[0])          # first operand: the destination
            last = grp[-1][1]
            waits = [a for a, b, t in blocks if a > last and "s_waitcnt vmcnt(0)" in t]
            assert waits, "prefetch without a landing wait in " + frag
            land = waits[0]
            for k in range(last + 1, land):
                ln = body[k].split(";")[0]
                if not ln.strip() or ln.strip().startswith(".") or ln.strip().endswith(":"):
                    continue
                ops = ln.strip().split(None, 1)
                touched = _vregs(ops[1]) & dest if len(ops) > 1 else set()
                assert not touched, "%s: line %d `%s` touches prefetch destination v%s before it has landed" % (
                    frag, k, ln.strip(), sorted(touched))
            checked += 1
    assert checked >= len(KERNELS)
    return checked


if __name__ == "__main__":
    import sys
    print("asm prefetch check (%s): %d groups clean" % (" ".join(sys.argv[1:]) or "default build", check(sys.argv[1:])))
