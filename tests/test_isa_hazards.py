"""Static check of the SHIPPED device code (the fat binaries inside dl4vc_amd/csrc/*.o) for the one hazard gfx950 leaves to software
and ROCm 7.2's hazard recognizer can miss: an MFMA's destination read or overwritten by a vector / LDS / vector-memory instruction
too few wait states later, over EVERY control-flow path (tools/isa_hazard_check.py; HISTORY.md section 14.2).

Why it is a test: VERDICT r5 "weak" 12.  The bf16x3 segment kernel built under `-amdgpu-sched-strategy=iterative-ilp` gave outputs that
differed between identical sites.  The cause is not an LDS / LDS-DMA ordering the source leaves implicit (tools/isa_order_diff.py: no
memory instruction crossed a barrier, a wait or an asm block in any of the three builds) but a missing `s_nop` at the join behind the
`skip_last` branch of x3::gemm_x, which that schedule exposes and hipcc's default / max-ilp schedules do not.  The source now carries
the wait states itself; this test is the net under every other kernel and every future compiler or flag: it disassembles what the
library is linked from and fails on the first unprotected path.  No GPU needed (hipcc and llvm-objdump run here)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazard_check as H          # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(ROOT, "dl4vc_amd", "csrc")
OBJECTS = ["dan_kernels.o", "dan_kernels_bf16p.o", "dan_kernels_bf16x.o", "dan_train.o"]

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(LLVM, "llvm-objdump")), reason="ROCm LLVM tools not present")


def disassemble(obj, workdir):
    """host object -> .hip_fatbin section -> the gfx950 code object -> llvm-objdump -d text"""
    fat, co, out = (os.path.join(workdir, os.path.basename(obj) + e) for e in (".fatbin", ".co", ".s"))
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    with open(out, "w") as f:
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True, stdout=f)
    return out


@pytest.mark.parametrize("obj", OBJECTS)
def test_shipped_kernels_have_no_unprotected_mfma_result_read(obj, tmp_path):
    path = os.path.join(CSRC, obj)
    if not os.path.isfile(path):
        pytest.skip("%s not built (python -c 'import __graft_entry__ as g; g.build()')" % obj)
    listing = disassemble(path, str(tmp_path))
    kernels = H.parse(listing)
    n_mfma = sum(1 for items in kernels.values() for kind, t in items if kind == "ins" and t.startswith("v_mfma"))
    assert kernels and n_mfma > 0, "no MFMA found in %s: the disassembly was not understood" % obj
    bad = [b for name, items in kernels.items() for b in H.check_kernel(name, items)]
    msg = "\n".join("%s: [%d] %s -> [%d] %s: %d wait states on some path, %d needed" % b for b in bad[:10])
    assert not bad, "%d unprotected MFMA-result accesses in %s:\n%s" % (len(bad), obj, msg)
    # ... and the classic gfx9 wait-state rules (VALU-written SGPR read by vector memory / taken as a lane select, DPP after a VALU
    # write of its source or of EXEC, a wide store's data registers overwritten), over every path as well: the recognizer's hole is
    # not specific to MFMAs
    classic = [b for name, items in kernels.items() for b in H.check_kernel_classic(name, items)]
    msg = "\n".join("%s: [%d] %s -> [%d] %s: %d wait states on some path, %d needed (%s)" % b for b in classic[:10])
    assert not classic, "%d classic wait-state violations in %s:\n%s" % (len(classic), obj, msg)
    print("%s: %d kernels, %d MFMAs, every path protected" % (obj, len(kernels), n_mfma))


HIPCC = "/opt/rocm/bin/hipcc"

CALIBRATION_SRC = r"""
#include <hip/hip_runtime.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__global__ void k16x16x32(const bf8* a, float* out) {
    bf8 x = a[threadIdx.x], y = a[threadIdx.x + 64];
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
    out[threadIdx.x] = c[0] * 3.f;
}
__global__ void k32x32x16(const bf8* a, float* out) {
    bf8 x = a[threadIdx.x], y = a[threadIdx.x + 64];
    v16f c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
    out[threadIdx.x] = c[0] * 3.f;
}
__global__ void k16x16x4(const float* a, float* out) {
    float x = a[threadIdx.x], y = a[threadIdx.x + 64];
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0);
    out[threadIdx.x] = c[0] * 3.f;
}
"""


@pytest.mark.skipif(not os.path.isfile(HIPCC), reason="hipcc not present")
def test_wait_state_table_is_what_hipcc_itself_places_in_straight_line_code(tmp_path):
    """CALIBRATION of tools/isa_hazard_check.py::NEED: an MFMA followed at once by a VALU read of its result -- hipcc fills the gap
    with `s_nop N`; N + 1 is the figure the checker uses for that opcode (8 / 12 / 10 wait states for the three MFMAs the library
    issues).  A compiler whose figures differ changes this test first."""
    src = tmp_path / "cal.hip"
    src.write_text(CALIBRATION_SRC)
    out = tmp_path / "cal.s"
    subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out), str(src)], check=True,
                   stderr=subprocess.DEVNULL)
    kernels = H.parse(str(out))
    seen = {}
    for name, items in kernels.items():
        ins = [t for kind, t in items if kind == "ins"]
        i = next(k for k, t in enumerate(ins) if t.startswith("v_mfma"))
        op = ins[i].split()[0]
        gap = 0
        for t in ins[i + 1:]:
            if H.regs(ins[i][len(op):].split(",")[0]) & H.regs(t):
                break
            gap += int(t.split()[1], 0) + 1 if t.startswith("s_nop") else 1
        seen[op] = gap
        assert not H.check_kernel(name, items)
    assert seen == {"v_mfma_f32_16x16x32_bf16": 8, "v_mfma_f32_32x32x16_bf16": 12, "v_mfma_f32_16x16x4_f32": 10}, seen
    for op, need in seen.items():
        assert H.NEED[op] == need


def test_the_checker_sees_the_hazard_the_iterative_ilp_build_had():
    """The shape of the failure, as a listing: the last MFMA of a tile in front of a branch over a short block, its result read by the
    first instruction at the join.  The fall-through path has its wait states; the taken path has one.  (What ROCm 7.2 emitted for
    x3::segmentx_kernel under -amdgpu-sched-strategy=iterative-ilp, registers and all: HISTORY.md section 14.2.)"""
    listing = """
_Z6kernelv:
	v_mfma_f32_16x16x32_bf16 v[58:61], v[102:105], v[114:117], v[146:149]
	v_mfma_f32_16x16x32_bf16 v[54:57], v[98:101], v[114:117], v[142:145]
	s_cbranch_vccnz .LBB4_161
	ds_read_b128 v[114:117], v210 offset:49152
	ds_read_b128 v[118:121], v210 offset:49408
	s_waitcnt lgkmcnt(1)
	v_mfma_f32_16x16x32_bf16 v[46:49], v[102:105], v[114:117], v[46:49]
	v_mfma_f32_16x16x32_bf16 v[40:43], v[98:101], v[114:117], v[42:45]
	v_mfma_f32_16x16x32_bf16 v[40:43], v[106:109], v[114:117], v[40:43]
	v_mfma_f32_16x16x32_bf16 v[44:47], v[110:113], v[114:117], v[46:49]
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_16x16x32_bf16 v[46:49], v[102:105], v[118:121], v[44:47]
	v_mfma_f32_16x16x32_bf16 v[42:45], v[98:101], v[118:121], v[40:43]
.LBB4_161:
	%s
	v_cvt_pk_bf16_f32 v98, v54, v55
	v_cvt_pk_bf16_f32 v99, v56, v57
	s_endpgm
"""
    import tempfile
    for fix, expect in (("", True), ("s_nop 7", False), ("s_nop 5", True), ("s_nop 6", False)):
        with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
            f.write(listing % fix)
        try:
            kernels = H.parse(f.name)
            bad = [b for name, items in kernels.items() for b in H.check_kernel(name, items)]
        finally:
            os.remove(f.name)
        assert bool(bad) == expect, (fix, bad)
        if fix == "":
            assert any("v[54:57]" in b[2] and b[5] == 1 for b in bad), bad      # one wait state: the branch itself


def test_the_join_of_the_skip_last_branch_carries_its_own_wait_states():
    """The source-level protection stays in x3::gemm_x (a reviewer deleting the `s_nop 7` as dead weight must meet this test)."""
    src = open(os.path.join(CSRC, "dan_kernels_bf16x.hip")).read()
    body = src[src.index("__device__ __forceinline__ void gemm_x("):src.index("__device__ __forceinline__ void load_first(")]
    assert 'asm volatile("s_nop 7");' in body.split("sched_group_barrier(0x020, 4, 0);")[-1]


def _classic(body):
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write("_Z1kv:\n" + body + "\ts_endpgm\n")
    try:
        kernels = H.parse(f.name)
        return [(b[7], b[5], b[6]) for name, items in kernels.items() for b in H.check_kernel_classic(name, items)]
    finally:
        os.remove(f.name)


def test_classic_wait_state_rules_fire_where_they_should_and_only_there():
    """tools/isa_hazard_check.py::check_kernel_classic on hand-written listings: each rule with too few wait states, with exactly
    enough, and across a branch join (the hole's shape)."""
    sgpr = "VALU writes an SGPR, a vector-memory instruction reads it"
    assert _classic("\tv_readfirstlane_b32 s4, v0\n\tglobal_load_dword v1, v2, s[4:5]\n") == [(sgpr, 0, 5)]
    assert _classic("\tv_readfirstlane_b32 s4, v0\n\ts_nop 4\n\tglobal_load_dword v1, v2, s[4:5]\n") == []
    assert _classic("\tv_cmp_lt_i32_e64 s[8:9], s0, v182\n\ts_nop 1\n\tglobal_store_dwordx4 v39, v[150:153], s[8:9]\n") == [(sgpr, 2, 5)]
    assert _classic("\tv_readfirstlane_b32 s4, v0\n\tglobal_load_dword v1, v2, s[6:7]\n") == []                      # another SGPR
    dpp = "VALU writes a VGPR, a DPP instruction reads it"
    assert _classic("\tv_mov_b32_e32 v1, v2\n\tv_mov_b32_dpp v3, v1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") == [(dpp, 0, 2)]
    assert _classic("\tv_mov_b32_e32 v1, v2\n\ts_nop 1\n\tv_mov_b32_dpp v3, v1 row_shl:1 row_mask:0xf bank_mask:0xf\n") == []
    store = "a vector-memory store of more than 64 bits, a VALU instruction overwrites its data registers"
    assert _classic("\tglobal_store_dwordx4 v0, v[2:5], s[0:1]\n\tv_mov_b32_e32 v3, 0\n") == [(store, 0, 2)]
    assert _classic("\tglobal_store_dwordx4 v0, v[2:5], s[0:1]\n\ts_nop 1\n\tv_mov_b32_e32 v3, 0\n") == []
    assert _classic("\tglobal_store_dwordx2 v0, v[2:3], s[0:1]\n\tv_mov_b32_e32 v3, 0\n") == []                      # 64 bits: no hazard
    lane = "VALU writes an SGPR / VCC, v_readlane / v_writelane takes it as the lane select"
    assert _classic("\tv_add_co_u32_e32 v6, vcc, 0x4000, v30\n\tv_readlane_b32 s3, v5, vcc_lo\n") == [(lane, 0, 4)]
    # the hole's shape: the producer in front of a branch over a short block, the consumer first at the join
    tri = "\tv_readfirstlane_b32 s4, v0\n\ts_cbranch_scc1 .L1\n\tv_mov_b32_e32 v9, 0\n\tv_mov_b32_e32 v8, 0\n\tv_mov_b32_e32 v7, 0\n\tv_mov_b32_e32 v6, 0\n.L1:\n\t%sglobal_load_dword v1, v2, s[4:5]\n"
    assert _classic(tri % "") == [(sgpr, 1, 5)]                   # the fall-through path has 5, the taken one 1
    assert _classic(tri % "s_nop 3\n\t") == []
