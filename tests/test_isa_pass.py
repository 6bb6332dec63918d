"""The ISA pass of the env-kernel build (high_speed_quadrupedal_locomotion_by_irrl_amd/isa_pass.py) on hand-written snippets:
it must add exactly the wait states the gfx950 ISA manual asks for around inline-assembly v_fmac_f32_dpp instructions, and
leave compiler-generated code alone.  Also: the product build really contains the hand-placed instructions, every one of them
hazard-free after the pass (re-running the pass on its own output adds nothing)."""
import os

from high_speed_quadrupedal_locomotion_by_irrl_amd import build, isa_pass

DPP = " quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
ASM = lambda body: "\t;;#ASMSTART\n" + body + "\t;;#ASMEND\n"


def _run(text):
    out, stats = isa_pass.run(text.splitlines(keepends=True))
    return "".join(out), stats


def _pad(n):
    return "".join("\tv_mul_f32_e32 v%d, v%d, v%d\n" % (200 + i, 201 + i, 202 + i) for i in range(n))


def test_reader_hazard_two_wait_states():
    # VALU writes v5, our DPP instruction reads v5 as its DPP operand right behind it: 2 wait states
    text = _pad(3) + "\tv_add_f32_e32 v5, v1, v2\n" + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + _pad(3)
    out, st = _run(text)
    assert "s_nop 1" in out and out.index("s_nop 1") < out.index(";;#ASMSTART") and st["wait_states_added"] == 2
    # one unrelated instruction in between: 1 wait state is still missing
    text = _pad(3) + "\tv_add_f32_e32 v5, v1, v2\n\tv_mul_f32_e32 v30, v31, v32\n" + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + _pad(3)
    out, st = _run(text)
    assert "s_nop 0" in out and st["wait_states_added"] == 1
    # two instructions in between, or a write to the NON-dpp operand / the accumulator: nothing to do
    for body in (_pad(3) + "\tv_add_f32_e32 v5, v1, v2\n" + _pad(2) + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP),
                 _pad(3) + "\tv_add_f32_e32 v7, v1, v2\n" + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP),
                 _pad(3) + "\tv_add_f32_e32 v9, v1, v2\n" + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP)):
        out, st = _run(body + _pad(3))
        assert st["wait_states_added"] == 0, out
    # an s_nop the compiler already placed counts
    text = _pad(3) + "\tv_add_f32_e32 v5, v1, v2\n\ts_nop 1\n" + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + _pad(3)
    assert _run(text)[1]["wait_states_added"] == 0


def test_writer_hazard_and_compiler_pairs():
    # our instruction writes v9, a compiler-generated DPP instruction reads it as DPP operand: the compiler could not know
    text = _pad(3) + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + "\tv_add_f32_dpp v11, v9, v12" + DPP + _pad(3)
    out, st = _run(text)
    assert st["nops_writer"] == 1 and st["wait_states_added"] == 2
    # compiler writer + compiler DPP reader: the compiler's own business (it already placed what is needed)
    text = _pad(3) + "\tv_add_f32_e32 v9, v1, v2\n\tv_add_f32_dpp v11, v9, v12" + DPP + _pad(3)
    assert _run(text)[1]["wait_states_added"] == 0


def test_transcendental_and_exec_and_block_boundaries():
    text = _pad(3) + "\tv_rcp_f32_e32 v7, v3\n" + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + _pad(3)          # trans result used at once
    assert _run(text)[1]["nops_trans"] == 1
    text = _pad(3) + "\tv_cmpx_lt_f32_e32 v1, v2\n" + _pad(2) + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + _pad(3)   # VALU wrote EXEC: 5
    out, st = _run(text)
    assert st["wait_states_added"] == 3                                                                     # 5 minus the two in between
    text = ".LBB0_7:\n" + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + _pad(3)                                  # unknown predecessors
    assert _run(text)[1]["wait_states_added"] == 2
    text = _pad(3) + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + "\ts_cbranch_vccnz .LBB0_9\n"                 # unknown successors
    out, st = _run(text)
    assert st["wait_states_added"] == 2 and out.index("s_nop 1") > out.index(";;#ASMEND")


def test_verifier_fails_closed():
    """the independent re-scan of the pass's output (isa_pass.verify, run by every build): an unresolved hazard stops the build"""
    import pytest
    bad = _pad(3) + "\tv_add_f32_e32 v5, v1, v2\n" + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + _pad(3)
    with pytest.raises(isa_pass.HazardError, match="DPP operand"):
        isa_pass.verify(bad.splitlines(keepends=True))
    assert isa_pass.verify(_run(bad)[0].splitlines(keepends=True)) == 1                      # the pass's own output is clean
    # v_swap_b32 writes BOTH of its operands
    swap = _pad(3) + "\tv_swap_b32 v3, v5\n" + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + _pad(3)
    with pytest.raises(isa_pass.HazardError):
        isa_pass.verify(swap.splitlines(keepends=True))
    out, st = _run(swap)
    assert st["wait_states_added"] == 2 and isa_pass.verify(out.splitlines(keepends=True)) == 1
    # a VALU write of EXEC near the end of a block: the successor block's DPP instruction cannot see it -> padded at the block end
    ex = _pad(3) + "\tv_cmpx_lt_f32_e32 v1, v2\n\ts_cbranch_execz .LBB0_3\n.LBB0_2:\n" + _pad(2) + ASM("\tv_fmac_f32_dpp v9, v5, v7" + DPP) + _pad(3)
    with pytest.raises(isa_pass.HazardError, match="EXEC"):
        isa_pass.verify(ex.splitlines(keepends=True))
    out, st = _run(ex)
    assert "s_nop 4" in out and out.index("s_nop 4") < out.index("s_cbranch_execz")
    isa_pass.verify(out.splitlines(keepends=True))
    # compiler-generated pairs are verified too (not fixed: that would be a compiler bug, and the build must stop)
    cc = _pad(3) + "\tv_add_f32_e32 v9, v1, v2\n\tv_add_f32_dpp v11, v9, v12" + DPP + _pad(3)
    with pytest.raises(isa_pass.HazardError):
        isa_pass.verify(cc.splitlines(keepends=True))
    # two instructions in one inline-assembly block: rejected by the pass and by the verifier
    two = _pad(3) + ASM("\tv_add_f32_e32 v5, v1, v2\n\tv_fmac_f32_dpp v9, v5, v7" + DPP) + _pad(3)
    with pytest.raises(isa_pass.HazardError, match="more than one instruction"):
        _run(two)


def test_verifier_on_the_real_env_kernel_assembly_of_both_lane_layouts():
    """the assembly the product library was built from (csrc/_obj/env_kernels_l{16,4,4w2}.s): every DPP instruction of every kernel
    -- hand-placed and compiler-generated -- is hazard-free; the same assembly without its wait states is not"""
    import pytest
    build.build()
    if not all(os.path.exists(os.path.join(build.CSRC, "_obj", "env_kernels_l%s.s" % l)) for l in (16, 4, "4w2")):
        build.build(force=True)
    for lanes, min_dpp in ((16, 2000), (4, 1000), ("4w2", 1000)):
        fixed = open(os.path.join(build.CSRC, "_obj", "env_kernels_l%s.s" % lanes)).readlines()
        assert isa_pass.verify(fixed) >= min_dpp
    # ... and the verifier does see a hazard in that assembly once the wait states are taken out again (the compiler's raw output itself may or
    # may not need any of the pass's: that depends on where its scheduler happens to put the hand-placed instructions)
    fixed = open(os.path.join(build.CSRC, "_obj", "env_kernels_l16.s")).readlines()
    with pytest.raises(isa_pass.HazardError):
        isa_pass.verify([l for l in fixed if not l.strip().startswith("s_nop")])


def test_product_build_contains_the_hand_placed_instructions_and_is_hazard_free():
    build.build()
    s = os.path.join(build.CSRC, "_obj", "env_kernels_l16.s")
    if not os.path.exists(s):       # library reused from a previous build without its intermediates: rebuild the unit's assembly
        build.build(force=True)
    text = open(s).read()
    assert text.count("v_fmac_f32_dpp") >= 100                       # the 16-lane step kernel's exchanges ride on the FMAs
    out, st = isa_pass.run(text.splitlines(keepends=True))
    assert st["asm_dpp"] == text.count("v_fmac_f32_dpp") and st["wait_states_added"] == 0   # idempotent: nothing left to fix


def test_kernarg_offsets_the_kernels_assume_are_the_ones_the_compiler_laid_out():
    """IRRL_BIND_ARGS / IRRL_BIND_POLICY_ARGS (csrc/env_kernels.hip) name EnvParams, EnvState and PolicyStepArgs by their byte offsets in the
    kernarg segment: arguments in declaration order, each at its natural alignment.  The code object's own metadata says where the compiler
    put them -- checked for every kernel of both lane layouts in the assembly the product library was built from."""
    import yaml
    build.build()
    if not all(os.path.exists(os.path.join(build.CSRC, "_obj", "env_kernels_l%s.s" % l)) for l in (16, 4, "4w2")):
        build.build(force=True)
    seen_policy = 0
    for lanes in (16, 4, "4w2"):
        lines = open(os.path.join(build.CSRC, "_obj", "env_kernels_l%s.s" % lanes)).read().split("\n")
        a = next(i for i, l in enumerate(lines) if l.strip() == ".amdgpu_metadata")
        b = next(i for i, l in enumerate(lines) if l.strip() == ".end_amdgpu_metadata")
        meta = yaml.safe_load("\n".join(lines[a + 1:b]))
        kernels = meta["amdhsa.kernels"]
        assert len(kernels) >= 9
        for k in kernels:
            args = [a for a in k[".args"] if not a[".value_kind"].startswith("hidden_")]
            if not (len(args) >= 2 and args[0][".value_kind"] == "by_value" and args[1][".value_kind"] == "by_value"):
                continue
            p, s = args[0], args[1]
            assert p[".offset"] == 0 and p[".size"] % 4 == 0 and p[".size"] >= 92 * 4, k[".name"]       # EnvParams first
            assert s[".size"] == 26 * 8 and s[".offset"] == (p[".size"] + 7) // 8 * 8, k[".name"]      # EnvState: 26 pointers, 8-aligned behind it
            nptr = {"irrl_rollout_persistent_actor_wave_kernel_l16": 4, "irrl_rollout_persistent_mlp_kernel_l16": 4, "irrl_rollout_persistent_kernel_l16": 4,
                    "irrl_rollout_persistent_actor_kernel_l16": 4, "irrl_step_policy_kernel_l16": 5}.get(k[".name"])
            if nptr:
                # the kernels that read PolicyStepArgs from the kernarg segment: `nptr` pointers behind EnvState, then the struct
                assert [a[".value_kind"] for a in args[2:3 + nptr]] == ["global_buffer"] * nptr + ["by_value"], k[".name"]
                pa = args[2 + nptr]
                assert pa[".size"] > 200 and pa[".offset"] == (s[".offset"] + s[".size"] + nptr * 8 + 7) // 8 * 8, k[".name"]
                seen_policy += 1
    assert seen_policy == 5
