#!/usr/bin/env python3
"""Static hazard audit of the gfx950 code that ships in libtvr.so (or of a `--save-temps` .s file).

Why it exists: a round-1 build of the shade kernel returned wrong 16-lane groups on some boxes.  The mechanism could
not be reproduced in round 2 (the committed "corrupting" commit renders 10 full frames bit-reproducibly on three boxes), so the kernels
are held to rules that can be CHECKED on the instructions that ship, instead of to a belief about the mechanism:

  R1  vmcnt accounting: every VGPR an instruction reads (or overwrites) that is the destination of an outstanding VMEM load is covered by
      an `s_waitcnt vmcnt(N)` that retires that load — recomputed here independently of the compiler, along the fall-through path AND
      along the path on which every forward exec-masked branch (`s_cbranch_execz`) is taken, i.e. its loads are never issued
      (MI355X_MICROARCH.md "Compiler hazard": an un-waited VMEM op in an exec-masked sibling branch can move a wait).
  R2  phase rule: inside a tile loop no VMEM load is issued between the first and the last MFMA of the loop body ("matrix phase"), unless the
      kernel is listed as a single-MFMA-wave-per-SIMD kernel.  Reported per kernel as `vmem_in_matrix_phase`.
  R3  WAR adjacency: a VMEM load whose destination overlaps an A/B/C operand of an MFMA issued fewer than WAR_WINDOW instructions earlier.
      (What the round-1 ISA showed: `global_load_dwordx4 v[110:113]` directly behind `v_mfma ... v[110:113]`.)  Reported as `war_adjacent`.
  R4  a VALU result never feeds the DIRECTLY following MFMA: gfx950 needs one instruction (any: an `s_nop 0`, a satisfied `s_waitcnt`) between
      a VALU write and an MFMA read of the register (scripts/hwprobe/mfma_raw2.hip), hipcc pads for that, and the fp16 residuals come out of
      inline asm (`v_fma_mix_f32`, csrc/tvr_mfma.h) — reported as `valu_to_mfma_adjacent` (must be 0).  `valu_to_mfma_lt2` also counts
      the padded cases (one instruction in between; informational).

usage: isa_check.py <libtvr.so | file.s> [--kernel SUBSTR] [--json]
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
WAR_WINDOW = 16

_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(tok):
    out = set()
    for m in _REG.finditer(tok):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def disassemble(path):
    """-> {kernel name: [instruction strings]} from a .s file or from the gfx950 code objects bundled in a shared library."""
    if path.endswith(".s"):
        return split_kernels(open(path).read().split("\n"), asm=True)
    tmp = tempfile.mkdtemp(prefix="isa_check_")
    try:
        so = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, so)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        kernels = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
            kernels.update(resolve_branches(split_kernels(txt.split("\n"), asm=False)))
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def split_kernels(lines, asm):
    kernels, cur, name = {}, None, None
    for ln in lines:
        s = ln.strip()
        if asm:
            m = re.match(r"^(_Z\w+):", s)
            if m and not s.startswith(".L"):
                name, cur = m.group(1), []
                kernels[name] = cur
                continue
            if cur is None or not s or s.startswith((";", ".", "//")) and not s.startswith(".LBB"):
                continue
            if s.startswith("s_endpgm"):
                cur.append(s)
                cur = None
                continue
            cur.append(s.split(";")[0].strip() if not s.startswith(".LBB") else s.split(";")[0].strip())
        else:
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", s)
            if m:
                if m.group(1).startswith("_Z"):
                    name, cur = m.group(1), []
                    kernels[name] = cur
                elif cur is not None:
                    cur.append("LABEL " + m.group(1))
                continue
            if cur is None or not s:
                continue
            m = re.search(r"//\s*([0-9A-Fa-f]+):", s)
            s = s.split("//")[0].strip()
            if s:
                cur.append(s + (" @" + m.group(1) if m else ""))
    return {k: v for k, v in kernels.items() if v}


def resolve_branches(kernels):
    """objdump prints branch targets as signed dword offsets: turn them into `.LA<addr>` labels placed in front of the target instruction."""
    out = {}
    for name, ins in kernels.items():
        addr = []
        for x in ins:
            m = re.search(r" @([0-9A-Fa-f]+)$", x)
            addr.append(int(m.group(1), 16) if m else None)
        targets = {}
        clean = []
        for x, a in zip(ins, addr):
            y = re.sub(r" @[0-9A-Fa-f]+$", "", x)
            parts = y.split()
            if parts and (parts[0] == "s_branch" or parts[0].startswith("s_cbranch")) and a is not None and re.fullmatch(r"-?\d+", parts[-1]):
                off = int(parts[-1])
                if off >= 32768:
                    off -= 65536
                t = a + 4 + 4 * off
                targets[t] = f".LA{t:x}"
                y = f"{parts[0]} .LA{t:x}"
            clean.append((a, y))
        res = []
        for a, y in clean:
            if a in targets:
                res.append(targets[a] + ":")
            res.append(y)
        out[name] = res
    return out


def parse(ins):
    """-> (mnemonic, dst regs, src regs, raw) with the AMDGPU convention 'first operand(s) written'."""
    parts = ins.split(None, 1)
    mn = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    dst, src = set(), set()
    if not ops:
        return mn, dst, src
    if mn.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        if "lds" in ins.split():
            src = set().union(*[regs(o) for o in ops])                  # LDS-DMA: no VGPR destination
        else:
            dst = regs(ops[0])
            src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
    elif mn.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic", "ds_write", "ds_add", "ds_max", "ds_min")):
        src = set().union(*[regs(o) for o in ops])
        if "glc" in ins or "sc0" in ins.split() and "atomic" in mn:
            pass
    elif mn.startswith("v_cmp") or mn.startswith("v_cmpx") or mn.startswith("s_") or mn.startswith("v_readlane") or mn.startswith("v_readfirstlane"):
        src = set().union(*[regs(o) for o in ops])
    elif mn.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_subrev_co", "v_div_scale")):
        dst = regs(ops[0])
        src = set().union(*[regs(o) for o in ops[2:]]) if len(ops) > 2 else set()
    elif mn.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")):
        dst = regs(ops[0]) | regs(ops[1])
        src = set(dst)
    else:
        dst = regs(ops[0])
        src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        if mn.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_movrel")) or "dpp" in ins or "sdwa" in ins:
            src |= dst                                                 # accumulate / partial write reads the destination
    return mn, dst, src


def is_vmem_load(mn):
    return mn.startswith(("global_load", "buffer_load", "flat_load", "scratch_load"))


def is_vmem(mn):
    return mn.startswith(("global_", "buffer_", "flat_", "scratch_")) and not mn.startswith("buffer_wbl2") and not mn.startswith("buffer_inv")


def is_valu(mn):
    return mn.startswith("v_") and not mn.startswith("v_mfma") and not mn.startswith("v_smfmac")


def check_kernel(name, ins_list, skip_execz):
    """One linear pass.  skip_execz: treat every forward `s_cbranch_execz L` as taken (drop the instructions up to L)."""
    res = {"raw_violations": [], "war_adjacent": [], "valu_to_mfma_lt2": [], "valu_to_mfma_adjacent": [], "n_mfma": 0, "n_vmem_load": 0}
    outstanding = []                     # FIFO of (index, dst regs) of VMEM ops in issue order (stores have empty dst but still count)
    recent_mfma = []                     # (index, operand regs)
    recent_valu = []                     # (index, dst regs)
    labels = {}
    for i, ins in enumerate(ins_list):
        if ins.startswith("LABEL ") or re.match(r"^\.L\w+:", ins):
            labels[ins.replace("LABEL ", "").rstrip(":")] = i
    i, n = 0, len(ins_list)
    while i < n:
        ins = ins_list[i]
        if ins.startswith("LABEL ") or ins.endswith(":"):
            i += 1
            continue
        mn, dst, src = parse(ins)
        if mn == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                keep = int(m.group(1))
                outstanding = outstanding[len(outstanding) - keep:] if keep < len(outstanding) else outstanding
                if keep == 0:
                    outstanding = []
            i += 1
            continue
        if mn == "s_branch":                       # unconditional: continue at the target when it lies ahead (an if/else join) ...
            j = labels.get(ins.split()[-1])
            prev = ins_list[i - 1].split() if i > 0 else [""]
            if prev[0].startswith("s_cbranch"):    # ... the second arm of a two-way branch (`s_cbranch L1; s_branch EXIT`): the walk follows the FIRST arm when it lies ahead
                j1 = labels.get(prev[-1])          # (what follows the pair in the text is some other block, reached by a jump)
                if j1 is not None and j1 > i:
                    j = j1
            if j is not None and j > i:
                i = j
                continue
        if mn in ("s_cbranch_execz",) and skip_execz:
            tgt = ins.split()[-1]
            j = labels.get(tgt)
            if j is not None and j > i:
                i = j
                continue
        if mn in ("s_branch", "s_cbranch_vccnz", "s_cbranch_vccz", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_execnz") :
            tgt = ins.split()[-1]
            j = labels.get(tgt)
            if j is not None and j <= i:          # loop back edge: the compiler's state at the header is the merge; restart accounting there
                pass
        pend = set().union(*[d for _, d in outstanding]) if outstanding else set()
        hit = (src | dst) & pend
        if hit:
            res["raw_violations"].append((i, ins, sorted(hit)[:4]))
        if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
            res["n_mfma"] += 1
            for (k, d) in recent_valu:
                if i - k <= 2 and (d & src):
                    res["valu_to_mfma_lt2"].append((i, ins_list[k], ins))
                if i - k == 1 and (d & src):
                    res["valu_to_mfma_adjacent"].append((i, ins_list[k], ins))
            recent_mfma.append((i, src | dst))
            recent_mfma = [(k, r) for k, r in recent_mfma if i - k < WAR_WINDOW]
        if is_vmem(mn):
            if is_vmem_load(mn):
                res["n_vmem_load"] += 1
                for (k, r) in recent_mfma:
                    if i - k < WAR_WINDOW and (dst & r):
                        res["war_adjacent"].append((i, ins_list[k], ins))
                        break
            outstanding.append((i, dst))
        if is_valu(mn):
            recent_valu.append((i, dst))
            recent_valu = recent_valu[-4:]
        i += 1
    return res


def phase_report(ins_list):
    """Innermost-loop view: for every loop body (label .. backward branch to it) that contains MFMAs, count VMEM loads issued between
    its first and last MFMA."""
    labels, out = {}, []
    for i, ins in enumerate(ins_list):
        if ins.startswith("LABEL ") or re.match(r"^\.L\w+:", ins):
            labels[ins.replace("LABEL ", "").rstrip(":")] = i
    for i, ins in enumerate(ins_list):
        if ins.split()[0] in ("s_branch", "s_cbranch_vccnz", "s_cbranch_vccz", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_execnz"):
            j = labels.get(ins.split()[-1])
            if j is not None and j < i:
                body = ins_list[j:i]
                mf = [k for k, x in enumerate(body) if x.startswith("v_mfma")]
                if len(mf) >= 8:
                    vm = [k for k, x in enumerate(body) if is_vmem_load(x.split()[0]) and mf[0] < k < mf[-1]]
                    out.append({"loop_at": j, "len": i - j, "mfma": len(mf), "vmem_loads_between_first_and_last_mfma": len(vm)})
    return out


def audit(path, kernel_filter=None):
    ks = disassemble(path)
    report = {}
    for name, ins in ks.items():
        if kernel_filter and kernel_filter not in name:
            continue
        a, b = check_kernel(name, ins, False), check_kernel(name, ins, True)
        mf = [i for i, x in enumerate(ins) if x.startswith(("v_mfma", "v_smfmac"))]
        between = sum(1 for i, x in enumerate(ins) if mf and mf[0] < i < mf[-1] and is_vmem_load(x.split()[0]))
        report[name] = {
            "instructions": len(ins), "mfma": a["n_mfma"], "vmem_loads": a["n_vmem_load"], "vmem_loads_between_first_and_last_mfma": between,
            "raw_violations_fallthrough": len(a["raw_violations"]), "raw_violations_execz_taken": len(b["raw_violations"]),
            "war_adjacent": len(a["war_adjacent"]), "valu_to_mfma_lt2": len(a["valu_to_mfma_lt2"]), "valu_to_mfma_adjacent": len(a["valu_to_mfma_adjacent"]),
            "loops": phase_report(ins),
            "examples": {"raw": [x[1] for x in (a["raw_violations"] + b["raw_violations"])[:3]], "war": [f"{x[1]}  ->  {x[2]}" for x in a["war_adjacent"][:3]],
                         "valu_to_mfma": [f"{x[1]}  ->  {x[2]}" for x in a["valu_to_mfma_lt2"][:3]]},
        }
    return report


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    kf = None
    if "--kernel" in sys.argv:
        kf = sys.argv[sys.argv.index("--kernel") + 1]
        args = [a for a in args if a != kf]
    rep = audit(args[0], kf)
    if "--json" in sys.argv:
        print(json.dumps(rep, indent=1))
    else:
        for k, v in rep.items():
            if v["mfma"] == 0 and not v["raw_violations_fallthrough"] and not v["raw_violations_execz_taken"]:
                continue
            print(f"{k[:70]:70s} ins {v['instructions']:6d} mfma {v['mfma']:4d} vmem_ld {v['vmem_loads']:4d}  RAW {v['raw_violations_fallthrough']}/{v['raw_violations_execz_taken']}"
                  f"  loads-in-matrix-phase {v['vmem_loads_between_first_and_last_mfma']:3d}  WAR-adjacent {v['war_adjacent']:3d}  valu->mfma<2 {v['valu_to_mfma_lt2']:3d} (adjacent {v['valu_to_mfma_adjacent']})  loops {[(l['mfma'], l['vmem_loads_between_first_and_last_mfma']) for l in v['loops']]}")
            for kind, ex in v["examples"].items():
                for e in ex[:2]:
                    print(f"      {kind}: {e}")
