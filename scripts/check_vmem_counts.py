#!/usr/bin/env python3
"""Static check of the explicit `s_waitcnt vmcnt(n)` waits of the two-stages-ahead product sweeps
(WaveProblem::hess_prod_tape_d2, csrc/pantr_kernel.hip.h) in the COMPILED kernels of a library.  The Makefile
runs it on every library it links (a failure fails the build); tests/test_host.py runs it again.

The sweeps' own waits are marked in the machine code: they are written `s_waitcnt vmcnt(n) expcnt(6)` (expcnt(6)
never waits in a compute kernel and the compiler never emits it), so they are told from the compiler's waits without
guessing.  A marked wait vmcnt(n) at a stage start is right if at least n vector-memory operations were issued after
the request for the rows the stage is about to read (they complete in order).  In the source that request precedes the
marked wait before this one, so the check is: on EVERY path through the control-flow graph from the nearest earlier
synchronisation point - a marked wait, a full `s_waitcnt vmcnt(0)`, or the kernel's entry - to the marked wait, at least n
vector-memory instructions (LDS-DMA transfers, global loads and stores, scratch accesses) are issued; a path from a
full drain (or the entry) WITHOUT ANY LDS-DMA REQUEST on it is accepted as well (the rows a wait is for are requested
by LDS-DMA: with no such request since the drain they were requested before it, and have landed).  The minimum
over the paths is taken (round 4 took, at a join, the predecessor with more operations: a path with fewer was never
looked at).  Second check, all PANTR kernels of the chain: no stage loop of a product holds a full wait right behind
an LDS-DMA request (the compiler puts one in front of an LDS read it cannot tell from the landing buffers).

usage: check_vmem_counts.py [library]      exit status 0 = every marked wait covered on every path
A library built with -DPANTR_VMCNT_CONSERVATIVE has only `vmcnt(0) expcnt(6)` waits: nothing to count, passes."""
import heapq, os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pantr-cdc2023-experiments_amd", "lib", "libpantr_mi355x.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MARKED = re.compile(r"^s_waitcnt vmcnt\((\d+)\) expcnt\(6\)$")
VMEM = re.compile(r"^(global_load_lds_|global_store_|global_load_|global_atomic_|scratch_|buffer_|flat_)")


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


def kernels(asm):
    """name -> (instructions, addresses) of every function of the code object."""
    out = {}
    for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", asm, re.S | re.M):
        lines = [l for l in m.group(2).splitlines() if l.startswith("\t")]
        ins = [l.split("//")[0].strip() for l in lines]
        addr = [int(l.split("//")[-1].strip().split(":")[0], 16) if "//" in l else -1 for l in lines]
        out[m.group(1)] = (ins, addr)
    return out


def predecessors(ins, addr):
    """preds[i]: (p, taken) - the instructions control can come from: p's fall-through (taken = False) or its branch."""
    index = {a: i for i, a in enumerate(addr)}
    preds = [[] for _ in ins]
    for i, l in enumerate(ins):
        b = re.match(r"(s_cbranch_\w+|s_branch) (\d+)", l)
        if b:
            off = int(b.group(2))
            off -= 65536 if off >= 32768 else 0
            j = index.get(addr[i] + 4 + 4 * off)
            if j is not None:
                preds[j].append((i, True))
        if i + 1 < len(ins) and not l.startswith(("s_branch", "s_endpgm", "s_setpc")):
            preds[i + 1].append((i, False))
    return preds


# Path sensitivity, as little as the compiler's lowering of `if (c) A else B` needs.  It emits either
#     s_cbranch_vccz skip_A; A; s_cbranch_vccnz skip_B; B            (two branches on ONE unchanged condition register)
# or, structurised through a flag,
#     s_mov_b64 F, -1; vcc = exec & C; s_cbranch_vccz L; A; s_mov_b64 F, 0; L: vcc = exec & ~F; s_cbranch_vccnz skip_B; B
# - in both "A skipped and B skipped" is a path of the graph that no execution takes.  The backward search therefore
# carries what a path has established about condition registers - vcc, scc, exec from the conditional branches it passed
# and the edge it took out of each; scalar register pairs through `s_and_b64 / s_andn2_b64 vcc, exec, F`, `s_mov_b64` -
# as facts of three kinds (Z: is zero, NZ: is not zero, NA: is not all ones) and drops a path on which a fact meets an
# assignment that contradicts it (`s_mov_b64 F, 0` under NZ; `s_mov_b64 F, -1` under Z or NA).  Every rule only uses
# that exec is never zero while a wavefront runs.  A fact about a register is forgotten at any other instruction that
# may write it; when in doubt an instruction counts as a writer (forgetting only ever keeps MORE paths: the check stays
# on the safe side).  A lowering this does not know shows up as a false alarm, never as a missed one.
COND = {"s_cbranch_vccz": ("vcc", "Z"), "s_cbranch_vccnz": ("vcc", "NZ"), "s_cbranch_scc0": ("scc", "Z"),
        "s_cbranch_scc1": ("scc", "NZ"), "s_cbranch_execz": ("exec", "Z"), "s_cbranch_execnz": ("exec", "NZ")}
OPPOSITE = {"Z": "NZ", "NZ": "Z"}
SCC_KEEPERS = ("s_mov", "s_movk", "s_nop", "s_waitcnt", "s_branch", "s_cbranch", "s_load", "s_barrier", "s_sleep", "s_setprio",
               "s_mul_i32", "s_mul_hi", "s_cselect", "s_memtime", "s_memrealtime", "s_getreg", "s_sendmsg", "s_endpgm", "s_buffer_load",
               "s_store", "s_dcache", "s_icache", "s_sext", "s_pack")


def sregs(tok):
    """The scalar registers a token names: 's[44:45]' -> {44, 45}, 's7' -> {7}, 'vcc' -> {'vcc'} (else empty)."""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    return {tok} if tok in ("vcc", "exec", "vcc_lo", "vcc_hi", "exec_lo", "exec_hi") else set()


def contradicts(fact, value):
    """A register holding `value` (0 or -1) against a fact about it."""
    return (value == 0 and fact == "NZ") or (value == -1 and fact in ("Z", "NA"))


def step_back(l, facts):
    """Facts (dict register-token -> kind) that hold AFTER instruction `l`, carried to BEFORE it; None: contradiction."""
    facts = dict(facts)
    parts = l.replace(",", " ").split()
    op, args = parts[0], parts[1:]
    dst = args[0] if args else ""
    if op in ("s_and_b64", "s_andn2_b64") and len(args) == 3 and "exec" in args[1:] and dst in facts:
        src = args[2] if args[1] == "exec" else args[1]
        if op == "s_andn2_b64" and args[1] != "exec":
            src = None  # F & ~exec: says nothing useful
        kind = facts.pop(dst)
        if src and sregs(src) and kind in ("Z", "NZ"):
            # dst = exec & src:  dst != 0 => src != 0;  dst == 0 => src not all ones.   dst = exec & ~src: the other way round
            derived = {"s_and_b64": {"NZ": "NZ", "Z": "NA"}, "s_andn2_b64": {"NZ": "NA", "Z": "NZ"}}[op][kind]
            old = facts.get(src)
            if old is None or old == derived or {old, derived} == {"NA", "Z"}:
                facts[src] = "Z" if "Z" in (old, derived) else derived
            elif {old, derived} == {"NZ", "NA"}:
                pass  # both hold; keep the one already there
            else:
                return None  # Z against NZ
    elif op == "s_mov_b64" and len(args) == 2 and dst in facts:
        kind = facts.pop(dst)
        if args[1] in ("0", "-1"):
            if contradicts(kind, int(args[1])):
                return None
        elif sregs(args[1]) and args[1] not in facts:
            facts[args[1]] = kind
    # whatever else the instruction writes is forgotten
    written = set()
    if op.startswith("s_") and not op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_cmp", "s_bitcmp", "s_barrier",
                                                  "s_sleep", "s_setprio", "s_endpgm", "s_store", "s_dcache", "s_icache", "s_sendmsg")):
        written |= sregs(dst)
        if "saveexec" in op:
            written |= {"exec"}
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or (op.startswith("v_cmp") and op.endswith("_e64")) \
            or op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_div_scale", "v_mad_u64", "v_mad_i64")):
        for a in args[:2]:
            written |= sregs(a)
    if (op.startswith("v_cmp") and op.endswith("_e32")) or "vcc" in args[:1]:
        written |= {"vcc"}
    if op.startswith("v_cmpx") or dst == "exec":
        written |= {"exec"}
    if op.startswith("s_") and not op.startswith(SCC_KEEPERS):
        written |= {"scc"}
    for reg in list(facts):
        r = sregs(reg) or {reg}
        if r & written or (reg == "vcc" and written & {"vcc_lo", "vcc_hi"}) or (reg == "exec" and written & {"exec_lo", "exec_hi"}):
            del facts[reg]
    return facts


def feasible(ins, preds, start, facts, budget=200):
    """A path arrived at a synchronisation point `start` still assuming `facts` (e.g. a flag register not all ones, where
    the flag is set before the synchronisation point).  Walk on backwards, not counting anything: False if EVERY way into
    `start` contradicts the facts within `budget` instructions, True as soon as one way survives (or runs out of facts,
    of budget, or of predecessors)."""
    stack, seen = [(start, tuple(sorted(facts.items())), 0)], set()
    while stack:
        i, know, depth = stack.pop()
        if (i, know) in seen:
            continue
        seen.add((i, know))
        if not know or depth >= budget or not preds[i]:
            return True
        for p, taken in preds[i]:
            f = dict(know)
            op = ins[p].split(" ", 1)[0]
            if op in COND:
                reg, kind_taken = COND[op]
                kind = kind_taken if taken else OPPOSITE[kind_taken]
                if f.get(reg, kind) == OPPOSITE[kind]:
                    continue
                f[reg] = kind
            else:
                f = step_back(ins[p], f)
                if f is None:
                    continue
            stack.append((p, tuple(sorted(f.items())), depth + 1))
    return False


def check_marked_waits(name, ins, addr):
    preds = predecessors(ins, addr)
    checked = bad = conservative = 0
    for w, l in enumerate(ins):
        m = MARKED.match(l)
        if not m:
            continue
        n = int(m.group(1))
        if n == 0:
            conservative += 1
            continue
        checked += 1
        # Shortest backward path (weight 1 per vector-memory instruction) to the nearest synchronisation point.  A search
        # state is "control is about to reach instruction i, and on the rest of the way to the wait these facts hold there".
        # (`dma`: the path holds an LDS-DMA request - part of the state: a path without one is judged differently below)
        heap, dist, worst, tick = [(0, 0, w, (), False)], {}, None, 0
        while heap:
            c, _, i, know, dma = heapq.heappop(heap)
            if dist.get((i, know, dma), 1 << 30) <= c:
                continue
            dist[(i, know, dma)] = c
            for p, taken in preds[i]:
                facts = dict(know)
                op = ins[p].split(" ", 1)[0]
                if op in COND:  # leaving p by this edge says something about its condition register, which p does not change
                    reg, kind_taken = COND[op]
                    kind = kind_taken if taken else OPPOSITE[kind_taken]
                    if facts.get(reg, kind) == OPPOSITE[kind]:
                        continue  # contradicts what a later branch on the same, unchanged register established
                    facts[reg] = kind
                else:
                    facts = step_back(ins[p], facts)
                    if facts is None:
                        continue
                cp = c
                if MARKED.match(ins[p]):
                    ok, kind = cp >= n, "the marked wait before it"
                elif ins[p] == "s_waitcnt vmcnt(0)" or ins[p].startswith("s_waitcnt vmcnt(0) "):
                    # everything older than a full wait has landed: the rows this wait is for can only be in flight if
                    # they were requested on the way from there - a way without any LDS-DMA request is safe
                    ok, kind = cp >= n or not dma, "a full wait"
                elif not preds[p]:
                    ok, kind = cp >= n or not dma, "the kernel's entry"
                else:
                    cp += 1 if VMEM.match(ins[p]) else 0
                    if cp < n:  # (a path that already has its n operations needs no further look)
                        tick += 1
                        heapq.heappush(heap, (cp, tick, p, tuple(sorted(facts.items())), dma or ins[p].startswith("global_load_lds_")))
                    continue
                if not ok and facts and not feasible(ins, preds, p, facts):
                    continue  # what the path assumes is contradicted by an assignment further up, whichever way it came
                if not ok and (worst is None or cp < worst[0]):
                    worst = (cp, kind, p)
        if worst:
            bad += 1
            print(f"{name}: instruction {w}: `{l}` is reached from {worst[1]} (instruction {worst[2]}) on a path with "
                  f"only {worst[0]} vector-memory instructions")
    return checked, bad, conservative


def check_waits_behind_transfers(name, ins, addr):
    """Inside a stage loop of a product (an innermost loop with LDS-DMA transfers) no full `s_waitcnt vmcnt(0)` may follow
    a transfer within a few instructions: it would wait for the rows just requested, stage after stage (round 4: the
    stage input read behind the request cost 13 % per product in one build and nothing in another)."""
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
                    print(f"{name}: instruction {i}: {l} is waited for right away")
                    break
    return sum(inner), bad


def main():
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(code_object(LIB))
        f.flush()
        asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
    ks = kernels(asm)
    solve = {k: v for k, v in ks.items() if re.search(r"pantr_solve(_spec)?_kernelINS_10ChainModel", k)}
    if not solve:
        raise SystemExit("no PANTR kernel of the hanging chain found in " + LIB)
    rc = total = total_conservative = 0
    for name, (ins, addr) in sorted(ks.items()):
        short = re.sub(r"^_ZN5pantr\d+", "", name)[:60]
        checked, bad, conservative = check_marked_waits(short, ins, addr)
        total += checked
        total_conservative += conservative
        if checked or conservative:
            print(f"{os.path.basename(LIB)} {short}: {checked} marked waits checked on every path, {bad} not covered"
                  + (f", {conservative} conservative (vmcnt(0))" if conservative else ""))
        rc |= 1 if bad else 0
    if total < 6 and not total_conservative:
        print("fewer than six marked waits found: the two-stages-ahead sweeps are not in this library (or lost their marks)")
        rc = 1
    for name, (ins, addr) in sorted(solve.items()):
        short = re.sub(r"^_ZN5pantr\d+", "", name)[:60]
        n_inner, bad = check_waits_behind_transfers(short, ins, addr)
        # the comparators' kernels (method != pantr) replay no tape: no stage loop with transfers expected there
        pantr = re.search(r"ChainModelELi0E|spec_kernel", name) is not None
        print(f"{os.path.basename(LIB)} {short}: {n_inner} instructions in stage loops, {bad} full waits right behind a transfer")
        rc |= 1 if bad or (pantr and not n_inner) else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
