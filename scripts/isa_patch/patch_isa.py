import re, sys
src, dst, mode = sys.argv[1:4]
lines = open(src).read().split("\n")
out = []
NOPS = ["\ts_nop 15", "\ts_nop 15"]          # 2 x 16 wait states
i = 0
in_kernel = False
pending_after_mfma = False
for ln in lines:
    s = ln.strip()
    if re.match(r"^_Z12shade_kernel.*:", s):
        in_kernel = True
    if s.startswith(".end_amdhsa_kernel") or s.startswith("s_endpgm"):
        in_kernel = in_kernel and not s.startswith("s_endpgm")
    if in_kernel:
        if mode == "none":
            pass
        elif mode == "before_load":            # P1: separate every MFMA from the next VMEM load by 32 wait states
            if s.startswith("v_mfma"):
                pending_after_mfma = True
            elif pending_after_mfma and (s.startswith("global_load") or s.startswith("buffer_load") or s.startswith("flat_load")):
                out.extend(NOPS)
                pending_after_mfma = False
            elif s.startswith("s_cbranch") or s.endswith(":"):
                pending_after_mfma = False
        elif mode == "before_mfma":            # P2 (control): the same padding in front of the first MFMA of each group
            if s.startswith("v_mfma") and not pending_after_mfma:
                out.extend(NOPS)
                pending_after_mfma = True
            elif not s.startswith("v_mfma") and not s.startswith(";") and s:
                pending_after_mfma = False
        elif mode == "nop_each_mfma":          # P3: 4 wait states in front of EVERY MFMA (VALU-write -> MFMA-read margin)
            if s.startswith("v_mfma"):
                out.append("\ts_nop 3")
        elif mode == "drain_before_load":      # P4: before the first VMEM load after an MFMA, wait until the MFMA has certainly finished (8 passes) — s_nop 15 x1
            if s.startswith("v_mfma"):
                pending_after_mfma = True
            elif pending_after_mfma and s.startswith("global_load"):
                out.append("\ts_nop 15")
                pending_after_mfma = False
    out.append(ln)
open(dst, "w").write("\n".join(out))
