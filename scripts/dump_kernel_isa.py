#!/usr/bin/env python3
"""Disassembly of one kernel of a library (development aid): dump_kernel_isa.py <substring of the mangled name> [library]"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pattern, argv = sys.argv[1], sys.argv
sys.argv = [argv[0]] + argv[2:]          # check_vmem_counts reads the library path from its own argv[1]
import check_vmem_counts as cv
import subprocess, tempfile
lib = cv.LIB
with tempfile.NamedTemporaryFile(suffix=".elf") as f:
    f.write(cv.code_object(lib)); f.flush()
    asm = subprocess.run([cv.OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
for name, (ins, addr) in cv.kernels(asm).items():
    if pattern in name:
        print("#", name, len(ins), "instructions")
        for i, l in enumerate(ins):
            print(i, l)
