#!/usr/bin/env python3
"""Static instruction census of the hot loops of the propagation kernels (gfx950 ISA of one options build).

    python tools/isa_census.py [preset] [extra hipcc flags ...] > profiles/r05/isa_census.md

Compiles artis_engine.hip to assembly with line tables (-gline-tables-only: the code is the shipped build's), finds every
natural loop of a kernel (a backward branch to an earlier label), and counts the instructions of the loop's body by class --
VALU, SALU, vector memory, LDS, scalar memory, branches, waits -- and, separately, what the register allocator added:
v_readlane / v_writelane (SGPRs spilled to VGPR lanes) and scratch_load / scratch_store (VGPR spills). A loop is named by the
source lines most of its instructions come from. The counts are STATIC (every path of the body once), so they bound the
instructions of one wave-round from above; rare paths (the f64 fall-backs of the filters) are inside them.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from artis_amd import build as B  # noqa: E402

KERNELS = {  # substring of the mangled name -> label
    "9k_thermalILi1024ELi1ELb0E": "k_thermal<1024, 1, false>",
    "6k_rpktILb1ELi768ELb0E": "k_rpkt<true, 768>",
}
# the hot loops, found in the source by their tags: (kernel substring, name, reference lines, file, tag, largest body in assembly lines)
HOT = [
    ("9k_thermalILi1024ELi1ELb0E", "transition loop", "macroatom.cc:385-577", "artis_engine.hip", "[census: transition loop]", 520),
    ("6k_rpktILb1ELi768ELb0E", "opacity sum", "rpkt.cc:721-830", "physics.h", "[census: opacity sum]", 1100),
    ("6k_rpktILb1ELi768ELb0E", "line walk", "rpkt.cc:106-207", "physics.h", "[census: line walk]", 1500),
]


MIN_BODY = 120  # assembly lines (the transition loop's body is ~330 since round 6)


def tagged_range(path, tag):
    """source lines [first, last] of the loop statement that carries the tag (brace matching from its line)"""
    src = open(path).read().splitlines()
    first = next(i for i, ln in enumerate(src) if tag in ln)
    depth = 0
    for i in range(first, len(src)):
        code = src[i].split("//")[0]
        depth += code.count("{") - code.count("}")
        if depth == 0 and i > first:
            return first + 1, i + 1
    raise RuntimeError(tag)


_FUNC_CACHE = {}


def enclosing_function(path: str, line: int) -> str:
    """name of the function whose definition precedes `line` in a source file (a definition = a line at column 0 that opens one:
    AHD / __device__ / template-less `type name(`); '?' outside the tree"""
    if not os.path.exists(path):
        return "(library)"
    if path not in _FUNC_CACHE:
        defs = []
        for i, ln in enumerate(open(path, errors="replace").read().splitlines(), 1):
            m = re.match(r"^(?:AHD|ANOINLINE|__device__|__global__|static|inline|template\s*<[^>]*>\s*AHD)\b[^;]*?\b([A-Za-z_]\w*)\s*\(", ln)
            if m and not ln.rstrip().endswith(";"):
                defs.append((i, m.group(1)))
        _FUNC_CACHE[path] = defs
    name = "?"
    for i, n in _FUNC_CACHE[path]:
        if i > line:
            break
        name = n
    return name


def classify(op: str) -> str:
    if op.startswith(("v_readlane", "v_writelane")):
        return "sgpr_spill_lane_ops"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "classic"
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    pflags = [] if preset == "classic" else [f"-DARTIS_PRESET_{preset.upper()}"]
    tmp = tempfile.mkdtemp(prefix="isa_census_")
    cmd = ["/opt/rocm/bin/hipcc", *B.FLAGS, *pflags, *extra, "-gline-tables-only", "-save-temps=obj", "-o", os.path.join(tmp, "x.so"),
           os.path.join(B.CSRC, "artis_engine.hip")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, cwd=tmp)
    asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    lines = open(os.path.join(tmp, asm)).read().splitlines()
    files = {}
    for ln in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    print(f"# ISA census of the hot loops ({preset} build, hipcc {' '.join(B.FLAGS[:1] + extra)}; static counts per loop body)\n")
    for key, label in KERNELS.items():
        start = next((i for i, ln in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", ln)), None)
        if start is None:
            continue
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = lines[start:end]
        labels = {}
        insts = []  # (index in body, opcode, file, line)
        cur = (None, 0)
        for i, ln in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                labels[m.group(1)] = i
                continue
            m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
            if m:
                cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
                continue
            m = re.match(r"^\t([a-z][a-z0-9_]+)", ln)
            if m and not ln.startswith("\t."):
                insts.append((i, m.group(1), cur[0], cur[1], ln))
        loops = []
        for i, op, f, l, ln in insts:
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = ln.split()[-1]
                if tgt in labels and labels[tgt] < i:
                    loops.append((labels[tgt], i))
        tot = collections.Counter(classify(op) for _, op, *_ in insts)
        print(f"## {label}\n")
        print(f"whole kernel: {len(insts)} instructions; " + ", ".join(f"{k} {v}" for k, v in sorted(tot.items())) + "\n")
        print("| loop | ISA loop body: instr | VALU | SALU | VMEM | LDS | SMEM | branch | wait | v_readlane / v_writelane | scratch_ |")
        print("|---|---|---|---|---|---|---|---|---|---|---|")
        uniq = sorted(set(loops), key=lambda t: t[1] - t[0])
        sub = []
        for kkey, name, ref, fname, tag, max_body in HOT:
            if kkey != key:
                continue
            lo_l, hi_l = tagged_range(os.path.join(B.CSRC, fname), tag)
            # every inlined copy of the loop: the natural loops that hold instructions of the loop statement's own line (its
            # condition / increment), smallest first, none inside another that was taken
            head = [i for i, op, f, l, _ in insts if f == fname and l == lo_l]
            taken = []
            # a loop has several backward branches (to its header and to the flow blocks before it): of the natural loops of
            # MIN_BODY..max_body assembly lines that hold instructions of the loop statement's line, the largest, none inside another
            for lo, hi in sorted(uniq, key=lambda t: t[0] - t[1]):
                if MIN_BODY <= hi - lo <= max_body and any(lo <= i <= hi for i in head) and not any(tlo <= lo and hi <= thi for tlo, thi in taken):
                    taken.append((lo, hi))
            if not taken:
                print(f"| {name} ({ref}) | not found | | | | | | | | | |")
            for n, (lo, hi) in enumerate(sorted(taken)):
                inside = [(op, f, l) for i, op, f, l, _ in insts if lo <= i <= hi]
                c = collections.Counter(classify(op) for op, _, _ in inside)
                print(f"| {name}, copy {n + 1} of {len(taken)} ({ref}; {fname}:{lo_l}-{hi_l}) | {len(inside)} | {c['valu']} | {c['salu']} | {c['vmem']} | "
                      f"{c['lds']} | {c['smem']} | {c['branch']} | {c['wait']} | {c['sgpr_spill_lane_ops']} | {c['scratch']} |")
                if name == "transition loop" or os.environ.get("CENSUS_BLOCKS"):
                    # the body by SUB-BLOCK: its instructions grouped by the source function their line belongs to (the generator, the
                    # action-filter and direction-filter counts, the search, the target decode, the loop's own tests and counters ...)
                    blocks = collections.defaultdict(collections.Counter)
                    for op, f, l in inside:
                        src = os.path.join(B.CSRC, f) if f in ("physics.h", "tables.h", "artis_engine.hip", "model_build.h") else f
                        blocks[enclosing_function(src, l) if l else "(no line)"][classify(op)] += 1
                    sub.append((name, n + 1, blocks))
                if os.environ.get("CENSUS_LINES"):  # the body's instructions by source line (to stderr: not part of the table)
                    by = collections.Counter((f, l) for _, f, l in inside)
                    for (f, l), cnt in sorted(by.items(), key=lambda t: (t[0][0], t[0][1])):
                        print(f"    {name} copy {n + 1}: {f}:{l} {cnt}", file=sys.stderr)
        print()
        for lname, copy, blocks in sub:
            print(f"### {label}: {lname}, copy {copy}, by sub-block (source function of each instruction's line; static counts)\n")
            print("| sub-block | instr | VALU | SALU | VMEM | LDS | branch | wait |")
            print("|---|---|---|---|---|---|---|---|")
            for fn, c in sorted(blocks.items(), key=lambda t: -sum(t[1].values())):
                print(f"| `{fn}` | {sum(c.values())} | {c['valu']} | {c['salu']} | {c['vmem']} | {c['lds']} | {c['branch']} | {c['wait']} |")
            print()


if __name__ == "__main__":
    main()
