#!/usr/bin/env python3
"""Static check for the hand-counted vector-memory waits (rows of quant_rows_wave and the offset pairs of the 256 x 256 kernels are loaded into
VGPRs by inline-asm `global_load` whose completion hipcc does not track; the kernels wait with hand-written `s_waitcnt vmcnt(n)`.  If the compiler ever
placed a copy, a spill or any other use of such a register between the load and its wait, the kernel would read stale data and bit-exactness would depend on
register allocation).

The check disassembles the gfx950 code objects of libasq_hip.so and walks every selected kernel in address order with the hardware's model of `vmcnt`:
vector-memory instructions (loads, LDS-DMA loads, stores) enter a FIFO in issue order, `s_waitcnt vmcnt(n)` retires all but the youngest n, loads return in
order.  A VGPR range that is the destination of a load still in the FIFO must not appear as an operand of ANY instruction (read or write) -- including as
the address of a later memory instruction.  Every path of a kernel's control-flow graph is walked (both sides of a conditional branch; a block is revisited only with a
FIFO state it has not seen), with vmcnt's in-order model -- a store completing out of order relative to loads is not modelled.

    python tools/check_asm_waits.py [--pattern REGEX ...] [path/to/libasq_hip.so]        exit status 1 on a violation
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
# (kernel name regex, regex of the inline-asm register loads in that kernel's disassembly): only THESE loads are treated as untracked -- hipcc's own loads are
# waited for by hipcc (and use EXEC-masked complementary writes to one register that a mask-blind walk would flag)
DEFAULT_PATTERNS = [(r"quant_rows_wave", r"^global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off nt$"),                   # asq_quant.hip load16_nt_async
                    (r"gemm_i8_p16|gemm_i8_p8<", r"^global_load_dwordx2 v\[\d+:\d+\], v\[\d+:\d+\], off$")]            # asq_gemm_p16.h / asq_gemm_p8.h: the tile's offset pairs

VM_LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load)_")
VM_STORE = re.compile(r"^(global_store|buffer_store|flat_store|scratch_store|global_atomic|buffer_atomic|flat_atomic)_")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
WAIT = re.compile(r"vmcnt\((\d+)\)")
CONTEXT = 0


def vgprs(text):
    out = []
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.append((int(m.group(1)), int(m.group(2))))
        else:
            out.append((int(m.group(3)), int(m.group(3))))
    return out


def overlap(a, b):
    return a[0] <= b[1] and b[0] <= a[1]


BRANCH_END = ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap")


def check_kernel(name, lines, asm_load=None, max_steps=400_000):
    """lines: [(addr, mnemonic, operand text)] of one kernel.  Walks every path of the kernel's control-flow graph (both sides of a conditional branch, the target
    of an unconditional one; a block is revisited only with a vmcnt FIFO it has not been entered with before; operations older than the oldest register load
    are dropped from the FIFO, they cannot change when a register load retires).  Returns (violations, register loads seen, deepest FIFO of register loads
    behind a counted wait, True if the walk was cut short)."""
    addr_of = [int(a, 16) for a, _, _ in lines]
    index_at = {a: i for i, a in enumerate(addr_of)}
    # decode once: kind 0 plain, 1 waitcnt vmcnt(n), 2 end, 3 branch, 4 cbranch, 5 register load, 6 other vector memory operation
    dec = []
    for i, (addr, mn, ops) in enumerate(lines):
        if mn == "s_waitcnt":
            m = WAIT.search(ops)
            dec.append((1, int(m.group(1)), None) if m else (0, None, ()))
        elif mn in BRANCH_END:
            dec.append((2, None, None))
        elif mn == "s_branch" or mn.startswith("s_cbranch"):
            simm = int(ops.split()[0])
            simm = simm - 65536 if simm >= 32768 else simm
            dec.append((3 if mn == "s_branch" else 4, index_at.get(addr_of[i] + 4 + 4 * simm), None))
        else:
            regs = tuple(vgprs(ops))
            is_load = bool(VM_LOAD.match(mn))
            is_lds_dma = is_load and ("_lds_" in mn or ops.rstrip().endswith(" lds") or " lds " in ops)
            if is_load and not is_lds_dma and (asm_load is None or asm_load.match(f"{mn} {ops}".strip())):
                dec.append((5, regs[0] if regs else None, regs))
            elif is_load or VM_STORE.match(mn):
                dec.append((6, None, regs))
            else:
                dec.append((0, None, regs))
    bad, seen_bad, loads_seen, deepest = [], set(), set(), 0
    visited = set()
    stack = [(0, ())]
    steps, cut = 0, False
    while stack and not cut:
        idx, fifo = stack.pop()
        if (idx, fifo) in visited:
            continue
        visited.add((idx, fifo))
        fifo = list(fifo)
        while idx < len(lines):
            steps += 1
            if steps > max_steps:
                cut = True
                break
            kind, arg, regs = dec[idx]
            if kind == 1:
                if arg > 0 and fifo:
                    deepest = max(deepest, sum(1 for d in fifo if d is not None))
                del fifo[:max(0, len(fifo) - arg)]
                while fifo and fifo[0] is None:
                    fifo.pop(0)
                idx += 1
                continue
            if kind == 2:
                break
            if kind == 3 or kind == 4:
                st = tuple(fifo)
                if arg is not None and (arg, st) not in visited:
                    stack.append((arg, st))
                if kind == 3 or (idx + 1, st) in visited:
                    break
                visited.add((idx + 1, st))
                idx += 1
                continue
            if fifo and regs:
                for r in regs:
                    for d in fifo:
                        if d is not None and overlap(r, d) and (idx, d) not in seen_bad:
                            seen_bad.add((idx, d))
                            addr, mn, ops = lines[idx]
                            msg = f"{name}: {addr}: `{mn} {ops}` touches v[{r[0]}:{r[1]}] while the load into v[{d[0]}:{d[1]}] is still in flight"
                            if CONTEXT:
                                msg += "\n" + "\n".join(f"        {a}: {m} {o}" for a, m, o in lines[max(0, idx - CONTEXT):idx + 2])
                            bad.append(msg)
            if kind == 5:
                fifo.append(arg)
                loads_seen.add(idx)
            elif kind == 6 and fifo:      # (with no register load pending, older operations are irrelevant)
                fifo.append(None)
            if len(fifo) > 63:      # vmcnt is a 6-bit counter: the 64th operation cannot issue before the oldest has completed
                del fifo[:len(fifo) - 63]
            idx += 1
    return bad, len(loads_seen), deepest, cut


def kernels_of(path):
    td = tempfile.mkdtemp()
    try:
        so = os.path.join(td, "lib.so")
        shutil.copy(path, so)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], check=True, capture_output=True, cwd=td)
        for co in sorted(p for p in os.listdir(td) if "gfx950" in p):
            txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(td, co)], check=True, capture_output=True, text=True).stdout
            name, cur = None, []
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    if name and cur:
                        yield name, cur
                    name, cur = m.group(1), []
                    continue
                m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if m and name:
                    cur.append((m.group(3), m.group(1), m.group(2)))
            if name and cur:
                yield name, cur
    finally:
        shutil.rmtree(td, ignore_errors=True)


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def run(path, patterns):
    pats = [(re.compile(k), re.compile(l) if l else None) for k, l in patterns]
    todo = list(kernels_of(path))
    dem = demangle([n for n, _ in todo])
    report = {"kernels": 0, "loads": 0, "deepest": 0, "violations": [], "cut": []}
    for (name, lines), d in zip(todo, dem):
        for k, l in pats:
            if not k.search(d):
                continue
            bad, nloads, deepest, cut = check_kernel(d[:140], lines, l)
            if cut:
                report["cut"].append(d[:140])
            report["kernels"] += 1
            report["loads"] += nloads
            report["deepest"] = max(report["deepest"], deepest)
            report["violations"] += bad
            break
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pattern", action="append", help="kernel name regex (every register load of a matching kernel is then treated as untracked)")
    ap.add_argument("--context", type=int, default=0, help="print this many preceding instructions with each violation")
    ap.add_argument("path", nargs="?", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "autosmoothquant_amd", "libasq_hip.so"))
    a = ap.parse_args()
    CONTEXT = a.context
    rep = run(a.path, [(p, None) for p in a.pattern] if a.pattern else DEFAULT_PATTERNS)
    print(f"{rep['kernels']} kernels, {rep['loads']} register loads walked, up to {rep['deepest']} register loads in flight behind a counted wait, {len(rep['violations'])} violations, {len(rep['cut'])} walks cut short")
    for c in rep["cut"]:
        print("  cut short: " + c)
    for v in rep["violations"][:40]:
        print("  " + v)
    sys.exit(1 if rep["violations"] else 0)
