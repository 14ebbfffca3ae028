"""The hand-counted vector-memory waits, checked on the built library's disassembly (tools/check_asm_waits.py): the inline-asm register loads of
quant_rows_wave (row vectors, up to 24 in flight per wave) and of the 256 x 256 kernels (the tile's offset pairs) are invisible to hipcc's own s_waitcnt
insertion, so nothing but the authors' counted waits keeps an instruction from touching such a register early.  The walk covers every path of each kernel's
control-flow graph with the hardware's in-order vmcnt model.  No GPU needed; skipped when the library has not been built."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "autosmoothquant_amd", "libasq_hip.so")


def _tool():
    spec = importlib.util.spec_from_file_location("check_asm_waits", os.path.join(ROOT, "tools", "check_asm_waits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_checker_sees_a_planted_violation_and_respects_paths():
    t = _tool()
    ok = [("00", "global_load_dwordx4", "v[4:7], v[0:1], off nt"), ("08", "global_load_dwordx4", "v[8:11], v[0:1], off nt"),
          ("10", "s_waitcnt", "vmcnt(1)"), ("14", "v_add_f32_e32", "v12, v4, v5"), ("18", "s_waitcnt", "vmcnt(0)"),
          ("1c", "v_add_f32_e32", "v12, v8, v9"), ("20", "s_endpgm", "")]
    bad, n, deepest, cut = t.check_kernel("k", ok)
    assert not bad and n == 2 and deepest == 2 and not cut
    early = list(ok)
    early[3] = ("14", "v_add_f32_e32", "v12, v8, v5")      # reads the second load's register behind vmcnt(1)
    bad, _, _, _ = t.check_kernel("k", early)
    assert len(bad) == 1 and "v[8:11]" in bad[0]
    # an if / else writing the same register on two paths is not a violation: s_cbranch 2 skips the load and its branch, the else side moves a constant
    paths = [("00", "s_cbranch_vccnz", "3"), ("04", "global_load_dword", "v5, v[0:1], off"), ("0c", "s_branch", "1"), ("10", "v_mov_b32_e32", "v5, 0"),
             ("14", "s_waitcnt", "vmcnt(0)"), ("18", "v_add_f32_e32", "v6, v5, v5"), ("1c", "s_endpgm", "")]
    bad, _, _, _ = t.check_kernel("k", paths)
    assert not bad
    # ... while the same use without the wait is one, on the path that ran the load
    bad, _, _, _ = t.check_kernel("k", paths[:4] + [("14", "s_nop", "0")] + paths[5:])
    assert len(bad) == 1


@pytest.mark.skipif(not os.path.exists(LIB), reason="libasq_hip.so not built")
def test_no_instruction_touches_an_asm_loaded_register_before_its_counted_wait():
    t = _tool()
    rep = t.run(LIB, t.DEFAULT_PATTERNS)
    assert rep["kernels"] >= 100 and rep["loads"] >= 1000, rep      # the kernels and their asm loads were found (quant_rows_wave alone: 150 instances)
    assert rep["deepest"] >= 16, rep                                    # ... and the counted waits really run with loads in flight
    assert not rep["cut"], rep["cut"][:3]
    assert not rep["violations"], rep["violations"][:5]
