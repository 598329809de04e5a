"""CPU: the MFMA dependency lint over the SHIPPED library (tools/isa_hazards.py).

Every kernel of the built idee_amd/libidee_hip.so is disassembled (llvm-objdump on the gfx950 code objects of its .hip_fatbin)
and every instruction in the dependency window of an MFMA is checked against the wait states the HARDWARE needs, as measured
on an MI355X by tools/micro/mfma_hazard.hip (profiles/r05_mfma_hazard.txt).  This is the guard against the
class of failure met in round 4 — an MFMA chain through short-lived temporaries whose dead registers were rewritten by `asm`
VALU statements the compiler does not pad, and a partially overlapping SrcC that LLVM pads one wait state short — for ANY
future schedule, flag or instantiation: it runs on the binary that ships, not on a source pattern."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def report():
    import isa_hazards
    from idee_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} is not built (python idee_amd/csrc/build.py)")
    rep, nfuncs = isa_hazards.check_library(_lib.LIB_PATH)
    assert nfuncs > 300, nfuncs            # the whole library was disassembled, not one code object
    return rep


def test_no_instruction_inside_an_mfma_dependency_window(report):
    assert not report.unknown_mfma, f"MFMA instructions the hazard table does not cover: {sorted(report.unknown_mfma)}"
    assert not report.violations, "\n".join(report.violations[:20])


def test_every_rule_was_exercised(report):
    """The lint saw the patterns it is there for (a parser that silently skipped MFMAs would pass the test above)."""
    for key in ("raw_read[xdl->v]", "waw_valu[xdl->v]", "mfma_srcc_same[xdl->xdl]", "mfma_srcab[xdl->xdl]", "valu_to_mfma[v->xdl]",
                "mfma_srcc_same[smfma->smfma]", "war_srcab[xdl->v]"):
        assert report.counts[key] > 50, (key, report.counts[key])
    assert report.counts["MFMAs with C != D (register SrcC)"] > 1000


def test_probe_table_matches_committed_measurement():
    """The HW table of the lint is the committed probe output, line by line."""
    import isa_hazards as H
    txt = open(os.path.join(ROOT, "profiles", "r05_mfma_hazard.txt")).read()
    got, sect = {}, None
    for line in txt.splitlines():
        if line.startswith("=="):
            sect = "mix" if "mixed" in line else ("s" if "16x16x4_f32" in line else "x")
        elif "min safe wait states" in line:
            got.setdefault(sect, {})[line.split()[0]] = int(line.rsplit("min safe wait states", 1)[1].split()[0])
    x, s_, m = got["x"], got["s"], got["mix"]
    assert H.HW["raw_read"] == {("xdl", "v"): x["raw_valu"], ("smfma", "v"): s_["raw_valu"]}
    assert H.HW["waw_valu"] == {("xdl", "v"): x["waw_valu"], ("smfma", "v"): s_["waw_valu"]}
    assert H.HW["war_srcc"] == {("xdl", "v"): x["war_c"], ("smfma", "v"): s_["war_c"]}
    assert H.HW["war_srcab"] == {("xdl", "v"): max(x["war_a"], x["war_b"]), ("smfma", "v"): max(s_["war_a"], s_["war_b"])}
    assert H.HW["mfma_srcc_same"] == {("xdl", "xdl"): max(x["chain_cd"], x["chain_inpl"]), ("xdl", "smfma"): m["cd_xs"],
                                      ("smfma", "xdl"): m["cd_sx"], ("smfma", "smfma"): max(s_["chain_cd"], s_["chain_inpl"])}
    assert H.HW["mfma_srcc_part"] == {("xdl", "xdl"): x["chain_part"], ("xdl", "smfma"): m["part_xs"], ("smfma", "xdl"): m["part_sx"],
                                      ("smfma", "smfma"): s_["chain_part"]}
    assert H.HW["mfma_srcab"] == {("xdl", "xdl"): x["chain_ab"], ("xdl", "smfma"): m["ab_xs"], ("smfma", "xdl"): m["ab_sx"],
                                  ("smfma", "smfma"): s_["chain_ab"]}
    assert H.HW["mfma_waw"] == {("xdl", "xdl"): m["waw_xx"], ("xdl", "smfma"): m["waw_xs"], ("smfma", "xdl"): m["waw_sx"],
                                ("smfma", "smfma"): m["waw_ss"]}
    assert H.HW["valu_to_mfma"] == {("v", "xdl"): max(x["valu_to_ab"], x["valu_to_c"]), ("v", "smfma"): max(s_["valu_to_ab"], s_["valu_to_c"])}
    assert x["d_on_a"] == x["d_on_b"] == s_["d_on_a"] == s_["d_on_b"] == 0 and H.SAME_MFMA_DST_ON_SRCAB_OK
    assert got["x"]["pv_chain"] == 0


def test_isa_sched_summary_tokens():
    """tools/isa_sched.py: the token summary of an instruction stream (the scan DESIGN.md section 6 describes)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import isa_sched
    body = """
	global_load_dwordx4 v[0:3], v[8:9], off                    // 0000
	s_waitcnt vmcnt(0)                                         // 0004
	ds_read_b128 v[4:7], v10                                   // 0008
	s_waitcnt lgkmcnt(0)                                       // 000C
	v_mfma_f32_16x16x4_f32 v[12:15], v4, v0, v[12:15]          // 0010
	v_mfma_f32_16x16x4_f32 v[12:15], v5, v1, v[12:15]          // 0018
	v_add_u32_e32 v8, 4, v8                                    // 0020
	v_add_u32_e32 v9, 4, v9                                    // 0024
	scratch_load_dword v20, off, off                           // 0028
	s_cbranch_scc1 65530                                       // 002C
	global_load_lds_dwordx4 v[8:9], off                        // 0030
	s_barrier                                                  // 0034
""".split("\n")
    assert isa_sched.summarize(body) == ["GL", "W[vm(0)]", "DSR", "W[lg(0)]", "M2", "v2", "SCR", "BR", "DMA", "BAR"]
