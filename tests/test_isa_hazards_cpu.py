"""The emitted gfx950 ISA of csrc/rzcc.hip never touches the destination of a hand-issued (inline-asm) LDS read before the
`s_waitcnt lgkmcnt(0)` behind it (the scan kernel's `ds_read2_b64` fetch; gfx950 has no VGPR interlock for
LDS returns).  hipcc cross-compiles without a GPU; the assembly is cached under build_dev/isa/ (about 40 s when stale)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_isa_hazards as H  # noqa: E402

HAZARD = """
_Zkernel:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
.LBB0_1:
	;;#ASMSTART
	ds_read2_b64 v[48:51], v4 offset0:0 offset1:0x41
	;;#ASMEND
	v_fma_f64 v[10:11], v[12:13], v[14:15], v[10:11]
	s_cbranch_scc1 .LBB0_3
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mov_b64_e32 v[60:61], v[48:49]
	s_branch .LBB0_1
.LBB0_3:
	v_mov_b64_e32 v[84:85], v[50:51]
	s_endpgm
.Lfunc_end0:
"""


def test_checker_finds_a_read_of_a_pending_destination():
    (name, body), = list(H.kernels(HAZARD))
    found = H.check_kernel(name, body)
    # only the copy on the path WITHOUT a wait is a finding; the one behind the wait is clean
    assert len(found) == 1 and found[0][2] == [50, 51] and "v[84:85]" in found[0][1], found


def test_checker_is_clean_when_the_wait_is_on_every_path():
    fixed = HAZARD.replace(".LBB0_3:\n", ".LBB0_3:\n\ts_waitcnt lgkmcnt(0)\n")
    (name, body), = list(H.kernels(fixed))
    assert H.check_kernel(name, body) == []


def test_rzcc_isa_has_no_read_of_an_lds_destination_in_flight():
    path = H.emit("rzcc.hip")
    findings, seen, with_asm = H.check_file(path)
    assert seen >= 40 and with_asm >= 10, (seen, with_asm)  # every rzcc_scan_kernel / encoder instantiation was looked at
    assert findings == [], findings[:5]
