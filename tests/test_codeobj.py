"""What the SHIPPED libsplat_hip.so's gfx950 code object says about the hot kernels (no GPU needed): register budgets that
decide residency, spills, scratch.  VERDICT r5 weak 9: round 5's near-selection compositor sat at 72 VGPRs with twenty scalar
registers spilled into a vector register that itself went to scratch -- one scratch_store at entry, six scratch_loads in front
of the walks -- and nothing guarded it.  Parsed with tools/codeobj.py (offload bundle + NT_AMDGPU_METADATA, no ROCm tool)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import codeobj  # noqa: E402

LIB = os.path.join(ROOT, "splat_amd", "libsplat_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def kernels():
    ks = {}
    for sym, md in codeobj.kernels(LIB).items():
        ks[codeobj.demangle(sym).split("(")[0].replace("void ", "")] = (sym, md)
    return ks


def test_every_kernel_of_the_path_is_in_the_code_object(kernels):
    for name in ("splat::cov3d_kernel", "splat::pack_scene_kernel", "splat::preprocess_kernel<true, false, false>",
                 "splat::preprocess_kernel<true, false, true>", "splat::scan_bucket_kernel<256>", "splat::select_near_kernel",
                 "splat::composite_exact_kernel<false, false, 2>", "splat::composite_exact_kernel<false, true, 2>",
                 "splat::composite_exact_kernel<true, false, 2>", "splat::composite_exact_kernel<false, false, 0>"):
        assert name in kernels, (name, sorted(kernels))


def test_k1_keeps_seven_workgroups_per_cu_and_no_scratch(kernels):
    # 256 threads: 72 VGPRs = seven waves per SIMD; one register more is a workgroup less per CU (LAB_NOTEBOOK.md: 76 VGPRs cost 4 %)
    for flav in ("<true, false, false>", "<true, true, false>", "<true, false, true>"):
        md = kernels["splat::preprocess_kernel" + flav][1]
        assert md[".vgpr_count"] <= 72, (flav, md[".vgpr_count"])
        assert md.get(".vgpr_spill_count", 0) == 0 and md.get(".sgpr_spill_count", 0) == 0, flav
        assert md[".private_segment_fixed_size"] == 0, flav
        assert md[".sgpr_count"] <= 96, flav          # 97+ scalar registers: six workgroups per CU (MI355X_MICROARCH.md, "Residency")
        assert md[".group_segment_fixed_size"] <= 11 * 1024, flav      # (fits beside seven compositor workgroups: 160 - 7 x 21.25 KB)


def test_compositor_register_and_lds_budgets(kernels):
    for flav in ("<false, false, 0>", "<false, false, 2>", "<false, true, 2>", "<false, false, 1>"):
        md = kernels["splat::composite_exact_kernel" + flav][1]
        assert md[".vgpr_count"] <= 72, (flav, md[".vgpr_count"])
        assert md[".sgpr_count"] <= 96, (flav, md[".sgpr_count"])
        assert md[".group_segment_fixed_size"] <= 22 * 1024 + 256, flav      # seven workgroups per CU: 7 x 22 KB of 160
    # the flavour without any out-of-line path (the sort launches ran): nothing spilled, nothing private
    md0 = kernels["splat::composite_exact_kernel<false, false, 0>"][1]
    assert md0.get(".vgpr_spill_count", 0) == 0 and md0.get(".sgpr_spill_count", 0) == 0 and md0[".private_segment_fixed_size"] == 0
    # the near-selection flavour (the default frame's): its private segment is the frame of repair_tile, a real function call
    # on a path few tiles take; the one vector register the metadata counts as spilled is saved around
    # THAT call (checked instruction by instruction below)
    md2 = kernels["splat::composite_exact_kernel<false, false, 2>"][1]
    assert md2.get(".vgpr_spill_count", 0) <= 1
    assert md2[".private_segment_fixed_size"] <= 256


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_the_hot_compositor_touches_scratch_only_around_the_repair_call(kernels, tmp_path):
    objs = codeobj.code_objects(LIB)
    blob = next(o for t, l in objs.items() if "gfx950" in t for o in l)
    co = tmp_path / "dev.co"
    co.write_bytes(blob)
    for flav in ("<false, false, 2>", "<false, true, 2>"):
        sym = kernels["splat::composite_exact_kernel" + flav][0]
        out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--disassemble-symbols=" + sym, str(co)],
                             capture_output=True, text=True, check=True).stdout
        ins = [ln.split("//")[0].strip() for ln in out.splitlines() if ln.startswith("\t")]
        assert len(ins) > 1000, "disassembly of %s came out empty" % sym
        calls = [k for k, i in enumerate(ins) if i.startswith("s_swappc_b64")]
        scratch = [k for k, i in enumerate(ins) if re.match(r"scratch_(load|store)", i)]
        assert len(calls) == 1, calls                    # repair_tile, and nothing else out of line
        assert scratch, "no save around the call any more: tighten this test (private_segment_fixed_size == 0?)"
        for k in scratch:
            assert abs(k - calls[0]) <= 8, "scratch access %d instructions away from the repair call: the hot path spills (%s)" % (k - calls[0], ins[k])
        # ... and no MFMA anywhere on this path, by design (BASELINE.json north_star: no dense contraction)
        assert not any(i.startswith("v_mfma") for i in ins)


def test_the_large_list_kernel_keeps_its_ballots_in_scalar_registers(kernels):
    # eight tiles' ballots at a time: sixteen spilled 42 scalar registers and was slower (profiles/r07_bin_large_v2_ab.txt)
    for flav in ("<false>", "<true>"):
        md = kernels["splat::bin_large_kernel" + flav][1]
        assert md.get(".sgpr_spill_count", 0) == 0 and md.get(".vgpr_spill_count", 0) == 0, flav
        assert md[".private_segment_fixed_size"] == 0 and md[".vgpr_count"] <= 64, flav
        assert md[".group_segment_fixed_size"] <= 9 * 1024, flav


def test_codeobj_tool_prints_a_table():
    if shutil.which("c++filt") is None:
        pytest.skip("c++filt missing")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "codeobj.py"), "composite_exact_kernel"], capture_output=True, text=True, check=True).stdout
    assert "composite_exact_kernel<false, false, 2>" in out
