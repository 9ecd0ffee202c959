#!/usr/bin/env python3
"""ISA budget of the decode kernel: static gfx950 instruction counts per stage of a round, from the compiled code object.

  python profiles/tools/isa_budget.py [--kernel brotlig_decode_kernel] [--flags "-DX=1 ..."] [--json out.json] [--blocks]

Builds brotlig_hip.hip for gfx950 with -g (device only), checks that the kernel has the same size as in the build without
-g, disassembles it, asks llvm-symbolizer for the inline stack of every instruction and books the instruction
  * to a STAGE: the line of decode_pages<> (or of the stage function it was inlined from) that it came from, and
  * to a LOOP: the innermost natural loop (back edge in the kernel's control flow) it sits in, named by the source line
    of the loop's back-edge branch.
Classes: VALU (v_*), SALU (s_* except waits/branches), BRANCH (s_cbranch*/s_branch), WAIT (s_waitcnt, s_nop, s_barrier),
LDS (ds_*), VMEM (global_/buffer_/flat_/scratch_), SMEM (s_load*, s_memtime ...).

No GPU needed.  The dynamic side (trip counts) comes from the phase profile and is applied by --trips FILE (see
profiles/r04_isa_stage_budget.md for the reconciliation with the SQ counters).
"""
import argparse
import bisect
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "brotli_g_sdk_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
OUT = os.path.join(ROOT, "build", "isa")


def sh(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, text=True, **kw).stdout


def build(flags, debug):
    os.makedirs(OUT, exist_ok=True)
    tag = "g" if debug else "n"
    obj = os.path.join(OUT, f"budget_{tag}.o")
    co = os.path.join(OUT, f"budget_{tag}.co")
    sys.path.insert(0, ROOT)
    from brotli_g_sdk_amd._build import HIP_FLAGS                      # the product's flags
    cmd = ["hipcc"] + HIP_FLAGS + ["--cuda-device-only", "-c", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    if debug:
        cmd.append("-g")
    cmd += flags + [os.path.join(CSRC, "brotlig_hip.hip"), "-o", obj]
    subprocess.run(cmd, check=True, cwd=CSRC)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={obj}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    return co


def kernel_symbol(co, name):
    for ln in sh([f"{LLVM}/llvm-objdump", "-t", co]).splitlines():
        m = re.match(r"([0-9a-f]+)\s+g\s+F\s+\.text\s+([0-9a-f]+)\s+\S*\s*(\S+)$", ln)
        if m and name in m.group(3) and "timed" not in m.group(3).replace(name, ""):
            return m.group(3), int(m.group(1), 16), int(m.group(2), 16)
    raise SystemExit(f"kernel {name} not found")


def classify(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("scratch_"):
        return "SCRATCH"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "VMEM"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "BRANCH"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")):
        return "WAIT"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache", "s_store", "s_atomic")):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


def disassemble(co, sym):
    text = sh([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"--disassemble-symbols={sym}", co])
    insts = []
    for ln in text.splitlines():
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):[^<]*(?:<\S+\+0x([0-9a-f]+)>)?", ln)
        if m:
            insts.append({"addr": int(m.group(3), 16), "op": m.group(1), "args": m.group(2), "label": int(m.group(4), 16) if m.group(4) else None})
    return insts


def symbolize(co, addrs):
    p = subprocess.run([f"{LLVM}/llvm-symbolizer", f"--obj={co}", "--inlines", "--functions=short", "--basenames"],
                       input="\n".join(hex(a) for a in addrs) + "\n", capture_output=True, text=True, check=True)
    stacks, cur = [], []
    lines = p.stdout.splitlines()
    i = 0
    while i < len(lines):
        if lines[i] == "":
            stacks.append(cur)
            cur = []
            i += 1
            continue
        fn = lines[i]
        loc = lines[i + 1] if i + 1 < len(lines) else "?:0:0"
        m = re.match(r"(.*):(\d+):(\d+)$", loc)
        cur.append((fn, m.group(1) if m else "?", int(m.group(2)) if m else 0))
        i += 2
    if cur:
        stacks.append(cur)
    return stacks


# ---- stages of decode_pages by marker comments in the source (robust against line shifts) -----------------------------
def stage_table(src_lines):
    def find(s, after=0):
        for i, l in enumerate(src_lines):
            if i + 1 > after and s in l:
                return i + 1
        raise KeyError(s)
    fn = find("__device__ inline void decode_pages(")
    marks = [
        ("page start (start_pages, tables)", find("// ---- page start.  A half without a page", fn)),
        ("1 commands", find("// -- 1. one command per lane", fn)),
        ("2 ring", find("// -- 2. distance ring", fn)),
        ("3 positions", find("// -- 3. output positions", fn)),
        ("3a group loop head", find("const uint32_t ngroups = live ?", fn)),
        ("3b flush+slide", find("// -- 3b. flush the finished bytes", fn)),
        ("3c pieces+far loads", find("// -- 3c. my pieces in this group", fn)),
        ("3d bitmaps+deps", find("// literals of the group: consumption indices", fn)),
        ("4 literals decode", find("// -- 4. literals of the group", fn)),
        ("4b literal runs", find("// -- 4b. literal runs", fn)),
        ("5a far stores", find("// -- 5a. far sources", fn)),
        ("5b copy levels", find("// -- 5b. LZ77 copies", fn)),
        ("6 round end", find("// carry ring bookkeeping", fn)),
        ("7 page end (flush, delta)", find("// ---- page end for the halves", fn)),
        ("end", find("clk.flush(a.prof, lane);", fn) + 2),
    ]
    return fn, marks


def stage_of(stack, marks, fn_line):
    # the frame of decode_pages<>: its line says where in the round the instruction belongs
    for fn, f, line in stack:
        if fn.startswith("decode_pages") and f == "brotlig_round.h":
            if line < marks[0][1]:
                return "0 wave setup"
            for k in range(len(marks) - 1):
                if marks[k][1] <= line < marks[k + 1][1]:
                    return marks[k][0]
            return "0 wave setup"
    return "0 wave setup"


def round_loop_scratch(co, kernel="brotlig_decode_kernel"):
    """Scratch accesses inside the round loops of a code object: [(loop start, loop end, instructions, [(address, opcode), ...]), ...].
    A round loop is a natural loop (back edge) that contains the s_setprio around the copy levels -- only a round has one -- and no
    global atomic (the page loop around it takes pages from the work counter): both instantiations' round and group loops, whatever their size.  tests/test_kernel_isa.py asserts on this."""
    sym, start, _ = kernel_symbol(co, kernel)
    insts = disassemble(co, sym)
    loops = set()
    for i in insts:
        if i["op"].startswith(("s_cbranch", "s_branch")) and i["label"] is not None and start + i["label"] <= i["addr"]:
            loops.add((start + i["label"], i["addr"]))
    prio = [i["addr"] for i in insts if i["op"] == "s_setprio"]
    out = []
    for lo, hi in sorted(loops):
        if not any(lo <= p <= hi for p in prio):
            continue
        body = [i for i in insts if lo <= i["addr"] <= hi]
        if any(i["op"].startswith(("global_atomic", "flat_atomic")) for i in body):
            continue        # the page loop around the rounds (it takes pages from the work counter); a few reloads per PAGE are tolerated
        sc = [(hex(i["addr"] - start), i["op"]) for i in body if i["op"].startswith("scratch_")]
        out.append((hex(lo - start), hex(hi - start), len(body), sc))
    return out


def loops_scratch(co, kernel, min_insts=400):
    """Scratch accesses inside ANY natural loop of at least `min_insts` instructions of a kernel that holds no global atomic (a page loop
    takes pages from a counter; a few reloads per PAGE are tolerated): [(loop start, loop end, instructions, [(address, opcode), ...])].  For
    kernels whose round loops carry no s_setprio mark (the two-wavefront kernel)."""
    sym, start, _ = kernel_symbol(co, kernel)
    insts = disassemble(co, sym)
    loops = set()
    for i in insts:
        if i["op"].startswith(("s_cbranch", "s_branch")) and i["label"] is not None and start + i["label"] <= i["addr"]:
            loops.add((start + i["label"], i["addr"]))
    out = []
    for lo, hi in sorted(loops):
        body = [i for i in insts if lo <= i["addr"] <= hi]
        if len(body) < min_insts or any(i["op"].startswith(("global_atomic", "flat_atomic")) for i in body):
            continue
        out.append((hex(lo - start), hex(hi - start), len(body), [(hex(i["addr"] - start), i["op"]) for i in body if i["op"].startswith("scratch_")]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="brotlig_decode_kernel")
    ap.add_argument("--flags", default="")
    ap.add_argument("--json", default=None)
    ap.add_argument("--blocks", action="store_true", help="also list the loops with their static counts")
    ap.add_argument("--lines", default=None, help="stage name (prefix): list its instructions by innermost source line")
    a = ap.parse_args()
    flags = a.flags.split()
    co_g, co_n = build(flags, True), build(flags, False)
    sym, start, size = kernel_symbol(co_g, a.kernel)
    _, _, size_n = kernel_symbol(co_n, a.kernel)
    insts = disassemble(co_g, sym)
    same = size == size_n
    stacks = symbolize(co_g, [i["addr"] for i in insts])
    assert len(stacks) == len(insts), (len(stacks), len(insts))
    src = open(os.path.join(CSRC, "brotlig_round.h")).read().split("\n")     # (decode_pages<> lives there since the header was split per stage)
    fn_line, marks = stage_table(src)

    # control flow: back edges -> natural loops as address intervals [target, branch]
    addr_index = {i["addr"]: k for k, i in enumerate(insts)}
    loops = []
    label_re = re.compile(r"<[^>]*\+0x([0-9a-f]+)>|^(-?\d+)$")
    for k, i in enumerate(insts):
        if i["op"].startswith(("s_cbranch", "s_branch")):
            tgt = start + i["label"] if i["label"] is not None else None
            i["target"] = tgt
            if tgt is not None and tgt <= i["addr"]:
                loops.append((tgt, i["addr"], k))
    loops.sort(key=lambda t: (t[0], -t[1]))

    def loop_of(addr):
        best = None
        for lo, hi, k in loops:
            if lo <= addr <= hi and (best is None or hi - lo < best[1] - best[0]):
                best = (lo, hi, k)
        return best

    per_stage = collections.defaultdict(collections.Counter)
    per_loop = collections.defaultdict(collections.Counter)
    loop_name = {}
    line_detail = collections.defaultdict(collections.Counter)
    for i, st in zip(insts, stacks):
        cls = classify(i["op"])
        stage = stage_of(st, marks, fn_line)
        i["stage"], i["cls"] = stage, cls
        per_stage[stage][cls] += 1
        lp = loop_of(i["addr"])
        if lp is not None:
            key = (lp[0], lp[1])
            if key not in loop_name:
                bst = stacks[lp[2]]
                loop_name[key] = " <- ".join(f"{fn.split('<')[0]}:{ln}" for fn, f, ln in bst[:3])
            per_loop[key][cls] += 1
            per_loop[key]["stage:" + stage] += 1
        if a.lines and stage.startswith(a.lines):
            fn, f, ln = st[0]
            line_detail[(fn.split("<")[0], ln)][cls] += 1

    # Scratch accesses inside the round loops, from the build WITHOUT -g (the one that ships): -g moves the register allocation, and
    # a spill inside a round (one scratch round trip per round: 4-5 % of the kernel when it happened in round 4) does not show in the
    # -g listing.  Round loops = natural loops of 1 500 .. 4 000 instructions (one per instantiation of the page loop).
    spills_in_rounds = round_loop_scratch(co_n, a.kernel)
    classes = ["VALU", "SALU", "BRANCH", "WAIT", "LDS", "VMEM", "SCRATCH", "SMEM", "OTHER"]
    print(f"kernel {sym}: {len(insts)} instructions, {size} bytes (-g) / {size_n} bytes (no -g){'' if same else '  ** SIZES DIFFER: -g changed the code **'}")
    print(f"{'stage':36s} " + " ".join(f"{c:>7s}" for c in classes) + "   total")
    order = ["0 wave setup"] + [m[0] for m in marks[:-1]]
    tot = collections.Counter()
    for s in order:
        c = per_stage.get(s, collections.Counter())
        print(f"{s:36s} " + " ".join(f"{c[k]:7d}" for k in classes) + f"  {sum(c.values()):6d}")
        tot.update(c)
    print(f"{'TOTAL':36s} " + " ".join(f"{tot[k]:7d}" for k in classes) + f"  {sum(tot.values()):6d}")
    print("\nscratch accesses inside the round loops (the loops around the copy levels' s_setprio) of the build without -g (must be none):")
    seen = set()
    for lo, hi, n, sc in spills_in_rounds:
        if not any(abs(int(lo, 16) - int(l2, 16)) < 64 for l2 in seen) or sc:
            print(f"  loop {lo}-{hi} ({n} instructions): {len(sc)} {sc if sc else ''}")
        seen.add(lo)
    if a.blocks:
        print("\nloops (innermost attribution; address range, back-edge source):")
        for key in sorted(per_loop):
            c = per_loop[key]
            stages = sorted(((v, k[6:]) for k, v in c.items() if k.startswith("stage:")), reverse=True)
            print(f"  {key[0]:#x}-{key[1]:#x} " + " ".join(f"{k}={c[k]}" for k in classes if c[k]) + f"  [{stages[0][1]}]  {loop_name[key]}")
    if a.lines:
        print(f"\ninstructions of stage '{a.lines}*' by innermost source line:")
        for (fn, ln), c in sorted(line_detail.items(), key=lambda kv: -sum(kv[1].values()))[:60]:
            print(f"  {fn:32s}:{ln:5d} " + " ".join(f"{k}={c[k]}" for k in classes if c[k]) + "   | " + (src[ln - 1].strip()[:90] if fn and 0 < ln <= len(src) else ""))
    if a.json:
        json.dump({"kernel": sym, "bytes": size, "same_code_as_no_g": same, "stages": {s: dict(per_stage[s]) for s in per_stage},
                   "loops": [{"range": [k[0], k[1]], "name": loop_name[k], **{c: per_loop[k][c] for c in classes},
                              "stages": {kk[6:]: v for kk, v in per_loop[k].items() if kk.startswith("stage:")}} for k in sorted(per_loop)]},
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
