"""The assembly lint behind the library's build (tools/check_smem_inflight.py): it has to see a register that is touched
while an asm-issued scalar load may still be on its way to it, on any path, and nothing in code that waits first."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_smem_inflight", os.path.join(ROOT, "tools", "check_smem_inflight.py"))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)

HEAD = "_Z6kernelv:\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n"
LOAD = "\t;;#ASMSTART\n\ts_load_dwordx16 s[52:67], s[8:9], s0\n\t;;#ASMEND\n"
WAIT = "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n"


def reports(text):
    out = {}
    for name, body in lint.kernels(HEAD + text + TAIL):
        out.update(lint.check(name, body))
    return out


def test_a_wait_on_every_path_is_clean():
    text = (LOAD + "\ts_and_saveexec_b64 s[18:19], vcc\n\ts_cbranch_execz .LBB0_2\n"
            + WAIT + "\tv_mul_f32_e32 v1, s52, v0\n.LBB0_2:\n\ts_or_b64 exec, exec, s[18:19]\n" + WAIT
            + "\tv_readlane_b32 s52, v40, 17\n")
    assert reports(text) == {}


def test_a_register_reused_on_the_path_that_skips_the_wait_is_reported():
    # the restore of a spilled pair lands in s[52:53] on the path that never waited for the load into s[52:67]
    text = (LOAD + "\ts_and_saveexec_b64 s[18:19], vcc\n\ts_cbranch_execz .LBB0_2\n"
            + WAIT + "\tv_mul_f32_e32 v1, s52, v0\n.LBB0_2:\n\ts_or_b64 exec, exec, s[18:19]\n"
            + "\tv_readlane_b32 s52, v40, 17\n" + WAIT)
    found = reports(text)
    assert len(found) == 1 and list(found.values())[0][1] == [52]


def test_a_read_before_the_wait_is_reported():
    text = LOAD + "\tv_mul_f32_e32 v1, s60, v0\n" + WAIT
    assert [r[1] for r in reports(text).values()] == [[60]]


def test_a_partial_wait_does_not_count():
    text = LOAD + "\ts_waitcnt lgkmcnt(1)\n\ts_mov_b64 s[58:59], 0\n" + WAIT
    assert [r[1] for r in reports(text).values()] == [[58, 59]]


def test_a_loop_carries_the_load_round_to_its_head():
    text = (".LBB0_1:\n\ts_mov_b32 s53, 0\n" + LOAD + "\ts_cmp_lg_u32 s2, 0\n\ts_cbranch_scc1 .LBB0_1\n" + WAIT)
    assert [r[1] for r in reports(text).values()] == [[53]]


def test_the_library_s_own_assembly_passes():
    path = os.path.join(ROOT, "raytracer_amd", "csrc", "pb_render.device.s")
    if not os.path.exists(path):
        pytest.skip("the library has not been built here (make -C raytracer_amd/csrc writes the assembly it checks)")
    n = 0
    for name, body in lint.kernels(open(path).read()):
        n += 1
        assert lint.check(name, body) == {}, name
    assert n > 40
