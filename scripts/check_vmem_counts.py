#!/usr/bin/env python3
"""Static check of the explicit s_waitcnt vmcnt(n) waits of the two-stages-ahead product sweeps
(WaveProblem::hess_prod_tape_d2, csrc/pantr_kernel.hip.h) in the COMPILED one-wavefront kernel.

A wait `vmcnt(n)` at a stage start is correct if at least n vector-memory operations were issued after the
request for the rows the stage is about to read; the source guarantees that through program order (each stage
requests rows, then issues its stores, between two asm statements the compiler may not move memory operations
across).  This script looks at the machine code instead: for every explicit wait with n in {5, 6, 7} it counts
the vector-memory instructions (LDS-DMA transfers, global stores and loads) between the previous explicit wait
and this one in the instruction stream, and requires count >= n.  Runs without a GPU (llvm-objdump).

usage: check_vmem_counts.py [library]   exit status 0 = all waits covered"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pantr-cdc2023-experiments_amd", "lib", "libpantr_mi355x.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
KERNEL = "_ZN5pantr18pantr_solve_kernelINS_10ChainModelELi0EEEvNS_9SolveArgsE"


def code_object(path):
    data = open(path, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        t = data[off:off + tl].decode()
        off += tl
        if "gfx950" in t:
            return data[i + o:i + o + s]
    raise SystemExit("no gfx950 code object in " + path)


def main():
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(code_object(LIB))
        f.flush()
        asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
    m = re.search(r"^[0-9a-f]+ <%s>:\n(.*?)(?=^[0-9a-f]+ <)" % re.escape(KERNEL), asm, re.S | re.M)
    if not m:
        raise SystemExit("kernel not found: " + KERNEL)
    ins = [l.split("//")[0].strip() for l in m.group(1).splitlines() if l.startswith("\t")]
    explicit = {"s_waitcnt vmcnt(2)", "s_waitcnt vmcnt(5)", "s_waitcnt vmcnt(6)", "s_waitcnt vmcnt(7)"}
    vmem = re.compile(r"^(global_load_lds_|global_store_|global_load_|scratch_)")
    checked, bad, last, count = 0, 0, None, 0
    for i, l in enumerate(ins):
        if vmem.match(l):
            count += 1
        if l in explicit:
            n = int(l[-2])
            # (two explicit waits back to back are the two arms of one `if`: the second sees what the first saw)
            if last is not None and i - last <= 12:
                count = prev_count
            if n >= 5 and last is not None:
                checked += 1
                if count < n:
                    bad += 1
                    print(f"instruction {i}: {l} after only {count} vector-memory instructions since the previous explicit wait")
            prev_count, last, count = count, i, 0
    print(f"{LIB}: {checked} explicit waits of the two-stages-ahead sweeps checked, {bad} not covered")
    return 1 if bad or checked < 6 else 0


if __name__ == "__main__":
    sys.exit(main())
