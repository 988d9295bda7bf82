"""CPU: the gfx950 instructions that SHIP in libtvr.so obey the rules DESIGN.md §4.2 derives for the shade kernel (scripts/isa_check.py):
an independent recount of every `s_waitcnt vmcnt` (no register is read while a load into it is outstanding — along the fall-through
path and with every exec-masked block skipped), no global load inside a tile's matrix phase, no load into a register that an MFMA issued
just before reads, and no register spills (a spill reload is a VMEM load the phase rule does not allow either)."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "scripts"))
LIB = os.path.join(ROOT, "jittor-myc-nerfs_amd", "lib", "libtvr.so")
GEN = "Lb1ELi3EEv8SceneDev"     # shade_kernel<SRC, DST, REF, RC, GEN = true, AR = 3>: the lockstep layer 1 of scenes with more than two encoding frequencies fetches W1's next
                                 # k-step from global memory between its MFMAs — by design outside the phase rule (csrc/tvr_shade.hip), and not the measured path
BWD_GEN = "mlp_train_backward_kernelILb0ELb1EE"        # its backward (round 6), csrc/tvr_mlp_train.hip


@pytest.fixture(scope="module")
def report():
    import __graft_entry__ as g
    import isa_check
    if not os.path.exists(LIB):
        g.build()
    rep = {k: v for k, v in isa_check.audit(LIB, "shade_kernel").items() if not k.endswith("E" + GEN + "9ShadeArgs")}
    rep.update(isa_check.audit(LIB, "shade16_kernel"))            # round 5: the render path's kernel on 16x16x32 tiles (it happens to satisfy the phase rule too)
    rep.update(isa_check.audit(LIB, "mlp_train_backward_kernel"))
    rep.update(isa_check.audit(LIB, "basis_backward_kernel"))
    return rep


def test_every_shade_kernel_variant_is_audited(report):
    # {queue, xyz->features, features->rgb} x {fp16 range check on, off} + the training forward (h -> rgb + activations), each x {TensorVMSplit, REFTensoRF},
    # + the render and mlp_render kernels in the two reduced-product arithmetics (2 kernels x 2 models x 2 range-check states x 2 modes)
    # + the backward kernels (mlp_train_backward x 2 models, basis_backward with 2 / 3 k-steps) + shade16_kernel x 2 range-check states x {TensorVMSplit, REFTensoRF (round 6)}
    # + mlp_train_backward_kernel<REF = false, GEN = true> (round 6: the fused training step of scenes with more than two encoding frequencies)
    assert len(report) == 39, sorted(report)

    assert all(v["mfma"] >= 27 for v in report.values())


def test_vmcnt_accounting_of_mfma_kernels(report):
    for name, v in report.items():
        assert v["raw_violations_fallthrough"] == 0 and v["raw_violations_execz_taken"] == 0, (name, v["examples"]["raw"])


def test_phase_rule_of_mfma_kernels(report):
    # (shade16_kernel<RC, REF = true>, round 6, requests the heads' ten fragments from global memory behind its phase boundary: hipcc issues all ten in front of the first MFMA,
    #  so the shipped code satisfies the rule although the kernel does not rely on it)
    for name, v in report.items():
        if BWD_GEN in name:
            # by design outside the phase rule, like the forward's lockstep layer 1: W1^T (208 KB) does not fit the LDS, every wave streams its fragments from L2 half a
            # slot ahead of the MFMAs that read them, into the registers of the half slot before, whose MFMAs have completed (each half slot ends with a read of its
            # accumulator — the discipline of tvr_gemm.hip's chunks).  The vmcnt accounting and the VALU -> MFMA rule hold for it like for every other kernel.
            assert v["vmem_loads_between_first_and_last_mfma"] > 0
            continue
        assert v["vmem_loads_between_first_and_last_mfma"] == 0, f"{name}: a global (or spill) load sits between the MFMAs of a tile"
        assert v["war_adjacent"] == 0, (name, v["examples"]["war"])


def test_no_valu_result_feeds_the_next_instruction_if_that_is_an_mfma():
    """gfx950 needs one instruction between a VALU write and an MFMA read of the register (scripts/hwprobe/mfma_raw2.hip); the fp16 residuals
    of every hi/lo split pass through inline asm (csrc/tvr_mfma.h).  Every MFMA kernel of the library, not only the shade kernels."""
    import isa_check
    rep = isa_check.audit(LIB, "")
    with_mfma = {k: v for k, v in rep.items() if v["mfma"] > 0}
    assert len(with_mfma) >= 14
    for name, v in with_mfma.items():
        assert v["valu_to_mfma_adjacent"] == 0, (name, v["examples"]["valu_to_mfma"])
    # and the checker sees the pattern when it is there
    bad = ["v_fma_mixhi_f16 v23, v27, -1.0, v28 op_sel:[1,0,0] op_sel_hi:[1,0,0]", "v_mfma_f32_32x32x16_f16 v[124:139], v[40:43], v[20:23], v[124:139]", "s_endpgm"]
    assert len(isa_check.check_kernel("k", bad, False)["valu_to_mfma_adjacent"]) == 1
    ok = [bad[0], "s_nop 0", bad[1], bad[2]]
    assert len(isa_check.check_kernel("k", ok, False)["valu_to_mfma_adjacent"]) == 0


def test_checker_flags_the_round1_pattern():
    """The checker must see what the round-1 build looked like: a load into an MFMA's A operand directly behind it, and a missing wait."""
    import isa_check
    bad = ["s_waitcnt vmcnt(0)", "v_mfma_f32_32x32x16_f16 v[6:21], v[110:113], v[22:25], 0", "global_load_dwordx4 v[110:113], v[104:105], off offset:272",
           "v_mfma_f32_32x32x16_f16 v[6:21], v[110:113], v[22:25], v[6:21]", "s_endpgm"]
    r = isa_check.check_kernel("k", bad, False)
    assert len(r["war_adjacent"]) == 1 and len(r["raw_violations"]) == 1      # the second MFMA reads v[110:113] with the load still outstanding
    good = bad[:3] + ["s_waitcnt vmcnt(0)"] + bad[3:]
    assert len(isa_check.check_kernel("k", good, False)["raw_violations"]) == 0
    # a wait that only covers the load if an exec-masked block in between ISSUED its own load
    masked = ["global_load_dwordx4 v[0:3], v[8:9], off", "s_cbranch_execz .LBB0_2", "global_load_dword v4, v[8:9], off", ".LBB0_2:",
              "s_waitcnt vmcnt(1)", "v_add_f32_e32 v5, v0, v0", "s_endpgm"]
    assert len(isa_check.check_kernel("k", masked, False)["raw_violations"]) == 0
    assert len(isa_check.check_kernel("k", masked, True)["raw_violations"]) == 1
