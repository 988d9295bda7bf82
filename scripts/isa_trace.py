"""Instruction-class trace of a kernel in a gfx950 .s file (hipcc --save-temps): one letter per instruction, run-length encoded, per basic block.
M mfma, v VALU, t transcendental (sin/cos/exp/rcp/...), d ds_read, D ds_write/atomic, g global/buffer load, G store/atomic, s SALU, w s_waitcnt, n s_nop, b branch, p s_setprio, z s_sleep, ? other.
usage: python3 scripts/isa_trace.py file.s '<mangled kernel name substring>' [--raw]"""
import re, sys
def cls(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("v_sin", "v_cos", "v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): return "t"
    if op.startswith("v_"): return "v"
    if op.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle")): return "d"
    if op.startswith("ds_"): return "D"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")): return "g"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "G"
    if op == "s_waitcnt": return "w"
    if op == "s_nop": return "n"
    if op == "s_setprio": return "p"
    if op == "s_sleep": return "z"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm")): return "b"
    if op.startswith("s_"): return "s"
    return "?"
def rle(s):
    out = []; i = 0
    while i < len(s):
        j = i
        while j < len(s) and s[j] == s[i]: j += 1
        out.append(s[i] + (str(j - i) if j - i > 1 else "")); i = j
    return " ".join(out)
def main():
    path, name = sys.argv[1], sys.argv[2]
    raw = "--raw" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and name in l)
    block, label, counts = "", "entry", {}
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end") or l.strip().startswith("s_endpgm"):
            break
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            if block: print(f"{label:12s} [{len(block):4d}] {block if raw else rle(block)}")
            block, label = "", m.group(1); continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")): continue
        op = t.split()[0]
        c = cls(op); block += c; counts[c] = counts.get(c, 0) + 1
    if block: print(f"{label:12s} [{len(block):4d}] {block if raw else rle(block)}")
    print("totals", counts)
main()
