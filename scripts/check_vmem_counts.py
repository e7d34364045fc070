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
# the one-wavefront PANTR kernel: generic passes, and three / six elements per lane at compile time
KERNELS = ["_ZN5pantr18pantr_solve_kernelINS_10ChainModelELi0ELi%dEEEvNS_9SolveArgsE" % nj for nj in (0, 3, 6)]
SPEC_KERNELS = ["_ZN5pantr23pantr_solve_spec_kernelINS_10ChainModelELi%dEEEvNS_9SolveArgsE" % nj for nj in (0, 3, 6)]


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
    rc = max(check(asm, k) for k in KERNELS)
    return max([rc] + [check_waits_behind_transfers(asm, k) for k in KERNELS + SPEC_KERNELS])


def check_waits_behind_transfers(asm, kernel):
    """Inside a stage loop of a product (an innermost loop with LDS-DMA transfers) no `s_waitcnt vmcnt(0)` may follow a
    transfer within a few instructions: it would wait for the rows just requested, stage after stage.  The compiler
    puts such a wait in front of an LDS read whose vector it cannot tell from the landing buffers (round 4: the
    stage input read behind the request cost 13 % per product in one build and nothing in another)."""
    m = re.search(r"^[0-9a-f]+ <%s>:\n(.*?)(?=^[0-9a-f]+ <)" % re.escape(kernel), asm, re.S | re.M)
    if not m:
        raise SystemExit("kernel not found: " + kernel)
    lines = [l for l in m.group(1).splitlines() if l.startswith("\t")]
    ins = [l.split("//")[0].strip() for l in lines]
    addr = [int(l.split("//")[-1].strip().split(":")[0], 16) if "//" in l else -1 for l in lines]
    index = {a: i for i, a in enumerate(addr)}
    inner = [False] * len(ins)
    for i, l in enumerate(ins):
        b = re.match(r"(s_cbranch_\w+|s_branch) (\d+)", l)
        if not b:
            continue
        off = int(b.group(2))
        off -= 65536 if off >= 32768 else 0
        j = index.get(addr[i] + 4 + 4 * off)
        if off < 0 and j is not None and i - j < 450 and any(x.startswith("global_load_lds_") for x in ins[j:i]):
            inner[j:i + 1] = [True] * (i + 1 - j)
    bad = 0
    for i, l in enumerate(ins):
        if inner[i] and l.startswith("global_load_lds_"):
            for x in ins[i + 1:i + 8]:
                if x.startswith("global_load_lds_"):
                    break
                if x == "s_waitcnt vmcnt(0)":
                    bad += 1
                    print(f"instruction {i}: {l} is waited for right away")
                    break
    print(f"{LIB} {kernel[9:-16]}: {sum(inner)} instructions in stage loops, {bad} waits right behind a transfer")
    return 1 if bad or not any(inner) else 0


def parse(asm, kernel):
    m = re.search(r"^[0-9a-f]+ <%s>:\n(.*?)(?=^[0-9a-f]+ <)" % re.escape(kernel), asm, re.S | re.M)
    if not m:
        raise SystemExit("kernel not found: " + kernel)
    lines = [l for l in m.group(1).splitlines() if l.startswith("\t")]
    ins = [l.split("//")[0].strip() for l in lines]
    addr = [int(l.split("//")[-1].strip().split(":")[0], 16) if "//" in l else -1 for l in lines]
    return ins, addr


def predecessors(ins, addr):
    """preds[i]: the instructions control can come from (fall-through and branches)."""
    index = {a: i for i, a in enumerate(addr)}
    preds = [[] for _ in ins]
    for i, l in enumerate(ins):
        b = re.match(r"(s_cbranch_\w+|s_branch) (\d+)", l)
        if b:
            off = int(b.group(2))
            off -= 65536 if off >= 32768 else 0
            j = index.get(addr[i] + 4 + 4 * off)
            if j is not None:
                preds[j].append(i)
        if i + 1 < len(ins) and not l.startswith(("s_branch", "s_endpgm")):
            preds[i + 1].append(i)
    return preds


def check(asm, kernel):
    """Every explicit wait vmcnt(n), n >= 5, inside the product sweeps: at least n vector-memory instructions stand
    between it and the explicit wait before it - walked backwards through the control-flow graph (the sweeps' last
    stages are peeled, so the instruction before in the file need not be the instruction before in time), taking at
    every join the predecessor with more of them: a request under `if (k + 2 < N)` counts as made, as it did when
    this check scanned the instructions in file order (which branch conditions exclude each other it cannot know)."""
    ins, addr = parse(asm, kernel)
    preds = predecessors(ins, addr)
    explicit = {"s_waitcnt vmcnt(2)", "s_waitcnt vmcnt(5)", "s_waitcnt vmcnt(6)", "s_waitcnt vmcnt(7)"}
    vmem = re.compile(r"^(global_load_lds_|global_store_|global_load_|scratch_)")
    checked = bad = 0
    for w, l in enumerate(ins):
        if l not in explicit or int(l[-2]) < 5:
            continue
        n = int(l[-2])
        # the compiler's own partial waits for the loads of an element-wise pass (vmcnt(6), vmcnt(5), ... as the values
        # are consumed) are not the sweeps': those stand among LDS-DMA transfers
        if not any(x.startswith("global_load_lds_") for x in ins[max(0, w - 400):w + 400]):
            continue
        # ... nor is one that continues a descending run vmcnt(n + 1), vmcnt(n), ...: the loads of a pass being consumed
        back = [x for x in ins[max(0, w - 6):w] if x.startswith("s_waitcnt vmcnt(") or vmem.match(x)]
        if back and re.fullmatch(r"s_waitcnt vmcnt\((\d+)\)", back[-1]) and int(back[-1][16:-1]) > n:
            continue
        best, seen, stack = 0, set(), [(p, 0) for p in preds[w]]
        while stack and best < n:
            i, c = stack.pop()
            while True:
                if vmem.match(ins[i]):
                    c += 1
                best = max(best, c)
                if c >= n or (i, c) in seen or ins[i] in explicit or ins[i] == "s_waitcnt vmcnt(0)" or not preds[i]:
                    break
                seen.add((i, c))
                for p in preds[i][1:]:
                    stack.append((p, c))
                i = preds[i][0]
        checked += 1
        if best < n:
            bad += 1
            print(f"instruction {w}: {l} with at most {best} vector-memory instructions since the explicit wait before it")
    print(f"{LIB} {kernel[30:-16]}: {checked} explicit waits of the two-stages-ahead sweeps checked, {bad} not covered")
    return 1 if bad or checked < 6 else 0


if __name__ == "__main__":
    sys.exit(main())
