"""Instruction mix of one kernel in a hipcc -S listing: python tools/isa_mix.py lib.s <kernel name substring> [--loops]"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if pat in l and l.rstrip().split(";")[0].strip().endswith(":") and not l.startswith("\t"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start + 1:end]


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return op


def mix(seg):
    ops, groups = collections.Counter(), collections.Counter()
    for line in seg:
        line = line.strip()
        if not line or line[0] in ".;" or line.split(";")[0].strip().endswith(":"):
            continue
        op = line.split()[0]
        ops[op] += 1
        groups[classify(op)] += 1
    return ops, groups


ops, groups = mix(body)
print("kernel total", sum(ops.values()), dict(groups))
if "--loops" in sys.argv:
    # basic blocks: label -> instruction count, to see where the bulk sits
    cur, blocks = "entry", collections.OrderedDict()
    blocks[cur] = []
    for line in body:
        t = line.split(";")[0].strip()
        if t.endswith(":") and not line.startswith("\t"):
            cur = t[:-1]
            blocks[cur] = []
        else:
            blocks[cur].append(line)
    for name, seg in blocks.items():
        o, g = mix(seg)
        if sum(o.values()) >= 40:
            print(f"{name:14s} {sum(o.values()):5d}", dict(g))
print(ops.most_common(40))
