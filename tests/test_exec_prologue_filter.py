"""The assembly filter of the build (grl_amd/_exec_prologue.py): vector copies the register
allocator put in front of a join block's exec restore move behind it; the copies of a `then` block, which must run
under the narrow mask, stay where they are."""
from grl_amd import _exec_prologue as ep


def _fix(text):
    out, fixed, skipped = ep.fix(text.strip("\n").split("\n"))
    return "\n".join(out), fixed, skipped


def test_copies_in_front_of_the_restore_move_behind_it():
    src = """
	s_and_saveexec_b64 s[26:27], s[54:55]
; %bb.494:                              ;   in Loop: Header=BB0_30 Depth=2
	v_cndmask_b32_e64 v103, v84, v85, s[50:51]
; %bb.495:                              ;   in Loop: Header=BB0_30 Depth=2
	v_mov_b32_e32 v198, v239
	s_mov_b64 s[76:77], s[60:61]
	v_accvgpr_write_b32 a21, v135
	s_or_b64 exec, exec, s[26:27]
	s_and_b64 s[6:7], s[30:31], s[28:29]
"""
    out, fixed, skipped = _fix(src)
    assert (fixed, skipped) == (1, 0)
    lines = [l.strip() for l in out.split("\n")]
    i = lines.index("s_or_b64 exec, exec, s[26:27]")
    assert lines[i - 1] == "s_mov_b64 s[76:77], s[60:61]"
    assert lines[i + 1:i + 3] == ["v_mov_b32_e32 v198, v239", "v_accvgpr_write_b32 a21, v135"]
    assert sorted(lines) == sorted(l.strip() for l in src.strip("\n").split("\n"))        # nothing added or lost


def test_the_copy_of_a_then_block_is_left_alone():
    src = """
	s_andn2_saveexec_b64 s[22:23], s[22:23]
; %bb.509:                              ;   in Loop: Header=BB0_500 Depth=3
	v_mov_b32_e32 v105, v104
; %bb.510:                              ;   in Loop: Header=BB0_500 Depth=3
	s_or_b64 exec, exec, s[22:23]
.LBB0_511:                              ;   in Loop: Header=BB0_500 Depth=3
	s_andn2_saveexec_b64 s[20:21], s[20:21]
"""
    out, fixed, skipped = _fix(src)
    assert (fixed, skipped) == (0, 0)
    assert out == src.strip("\n")


def test_a_scalar_that_reads_a_moved_vector_write_blocks_the_rewrite():
    src = """
.LBB0_7:
	v_mov_b32_e32 v3, v9
	v_readlane_b32 s4, v3, 2
	s_or_b64 exec, exec, s[8:9]
"""
    out, fixed, skipped = _fix(src)
    assert (fixed, skipped) == (0, 1)
    assert out == src.strip("\n")
