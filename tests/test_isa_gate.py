"""tools/ring_hazard_check.py (the in-flight register check of tools/seam_check_isa.sh) on synthetic gfx950 assembly: it must flag a
register of an asm-issued load that is copied, re-used or overwritten in front of the counted wait that covers it, follow loops into
their steady state, and pass the patterns the kernels rely on. No compiler, no GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "ring_hazard_check.py")


def run(tmp_path, body, name="_ZN12_GLOBAL__N_111ring_kernelEv"):
    path = tmp_path / "k.s"
    path.write_text("\t.text\n%s:\n%s\n.Lfunc_end0:\n" % (name, body))
    r = subprocess.run([sys.executable, TOOL, str(path), "ring_kernel"], capture_output=True, text=True)
    return r.returncode, r.stdout


CLEAN = """
\tglobal_load_dwordx4 v[10:13], v1, s[2:3] offset:1024
\tglobal_load_dwordx4 v[14:17], v1, s[2:3] offset:2048
\tglobal_load_lds_dwordx4 v[2:3], off
\tv_add_u32_e32 v1, 64, v1
\ts_waitcnt vmcnt(2)
\tv_mfma_f32_16x16x32_f16 a[0:3], v[10:13], v[20:23], a[0:3]
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\ts_waitcnt vmcnt(2)
\tv_mfma_f32_16x16x32_f16 a[0:3], v[14:17], v[20:23], a[0:3]
\ts_waitcnt vmcnt(0)
\tv_mov_b32_e32 v30, v10
\ts_endpgm
"""


def test_clean_ring_passes(tmp_path):
    rc, out = run(tmp_path, CLEAN)
    assert rc == 0 and "in-flight register touched: 0" in out, out


@pytest.mark.parametrize("bad", [
    "\tv_mov_b32_e32 v40, v12",                                        # the compiler copies a ring register with its load in flight
    "\tv_mfma_f32_16x16x32_f16 a[0:3], v[14:17], v[20:23], a[0:3]",   # consumed one wait too early
    "\tglobal_load_dwordx4 v[12:15], v1, s[2:3]",                     # slot overwritten by another load
    "\tglobal_store_dwordx4 v1, v[10:13], s[4:5]",                    # stored before it landed
])
def test_touching_a_pending_register_fails(tmp_path, bad):
    body = CLEAN.replace("\tv_add_u32_e32 v1, 64, v1\n", "\tv_add_u32_e32 v1, 64, v1\n" + bad + "\n", 1)
    rc, out = run(tmp_path, body)
    assert rc == 1 and "<- pending" in out, out


def test_loop_steady_state_is_replayed(tmp_path):
    # the load issued at the bottom of the body is only in flight at the top of the NEXT iteration
    body = """
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\ts_waitcnt vmcnt(0)
.LBB0_1:
\tv_mfma_f32_16x16x32_f16 a[0:3], v[10:13], v[20:23], a[0:3]
\ts_add_u32 s8, s8, 1
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\ts_cmp_lt_u32 s8, s9
\ts_cbranch_scc1 .LBB0_1
\ts_waitcnt vmcnt(0)
\ts_endpgm
"""
    rc, out = run(tmp_path, body)
    assert rc == 1 and "v_mfma" in out, out
    rc, out = run(tmp_path, body.replace(".LBB0_1:\n", ".LBB0_1:\n\ts_waitcnt vmcnt(0)\n"))
    assert rc == 0, out


def test_no_matching_kernel_is_an_error(tmp_path):
    rc, out = run(tmp_path, CLEAN, name="_ZN12_GLOBAL__N_15otherEv")
    assert rc == 1 and "nothing was checked" in out
