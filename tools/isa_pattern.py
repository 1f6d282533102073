"""Instruction-class pattern of the MFMA loops of a kernel in hipcc's -S output (M = MFMA, v = VALU, L = vector load,
W = s_waitcnt, s = other scalar, n = s_nop, D = LDS).  python tools/isa_pattern.py file.s kernel_substring [min_mfma]"""
import re, sys

def classify(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("global_load", "buffer_load", "flat_load")): return "L"
    if op.startswith(("global_store", "buffer_store")): return "S"
    if op.startswith("ds_"): return "D"
    if op.startswith("s_waitcnt"): return "W"
    if op.startswith("s_nop"): return "n"
    if op.startswith("v_"): return "v"
    if op.startswith("s_"): return "s"
    return "?"

def compress(seq):
    out, i = [], 0
    while i < len(seq):
        j = i
        while j < len(seq) and seq[j] == seq[i]: j += 1
        out.append(seq[i] + (str(j - i) if j - i > 1 else ""))
        i = j
    return " ".join(out)

def main():
    path, kern = sys.argv[1], sys.argv[2]
    min_m = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kern, l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur, name = [], [], "entry"
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((name, cur)); name, cur = m.group(1), []
            continue
        if not t or t.startswith((";", ".", "//")): continue
        cur.append(t.split()[0] + (" " + t.split(None, 1)[1] if t.startswith("s_cbranch") or t.startswith("s_branch") else ""))
    blocks.append((name, cur))
    for name, ins in blocks:
        nm = sum(1 for x in ins if x.startswith("v_mfma"))
        loop = any(x.startswith("s_cbranch") and x.endswith(name) for x in ins)
        if nm >= min_m:
            cl = [classify(x) for x in ins]
            cnt = {c: cl.count(c) for c in "MvLWsnD"}
            print(f"{name} loop={loop} n={len(ins)} {cnt}")
            print("   " + compress("".join(cl)))

main()
