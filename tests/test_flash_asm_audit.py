"""Host test (no GPU): the flash prefill kernel's gfx950 assembly keeps its accumulator in the AGPR file without spills.
The kernel pins 256 accumulator registers through asm operands and leaves hipcc a 256-VGPR allocation problem; a compiler
upgrade or an edit that tips it into spilling would still produce right answers, 3x slower -- this catches it at build time."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src,kernel,literal", [("mla_prefill_flash.hip", "mla_prefill_flash_pipe_kernel", True)])
def test_flash_kernel_assembly_has_no_spills_and_no_accumulator_traffic(tmp_path, src, kernel, literal):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_flash_asm

    csrc = os.path.join(ROOT, "chitu_amd", "csrc")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
           "-S", "--cuda-device-only", os.path.join(csrc, src), "-o", str(tmp_path / "k.s")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    seen, bad = (check_flash_asm.audit_literal if literal else check_flash_asm.audit)(str(tmp_path / "k.s"), kernel)
    assert seen and not bad, bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src,kernel", [("fp8_gemm_tiled.hip", "fp8_gemm_tiled_kernel"), ("bf16_gemm_tiled.hip", "bf16_gemm_tiled_kernel"),
                                        ("moe_tiled.hip", "moe_gemm_tiled_kernel"), ("gqa_prefill_flash.hip", "gqa_prefill_flash_kernel")])
def test_no_compiler_wait_inside_the_dma_fed_loops(tmp_path, src, kernel):
    """The loops fed by LDS-DMA hold no `s_waitcnt vmcnt` of the compiler's: it cannot see the DMA requests, its counter is
    in-order, so any such wait also drains the tiles in flight (check_flash_asm.audit_dma_loops).  (mla_decode.hip is not in
    the list: its tile loop keeps one compiler wait -- for the page-table fallback of splits above 32k tokens -- at the request
    point, right behind the kernel's own vmcnt(0) + barrier, where nothing is in flight; the compiler lays that loop out rotated
    and twice, which the audit's linear reading of the body cannot follow.)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_flash_asm

    csrc = os.path.join(ROOT, "chitu_amd", "csrc")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
           "-S", "--cuda-device-only", os.path.join(csrc, src), "-o", str(tmp_path / "k.s")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    seen, bad = check_flash_asm.audit_dma_loops(str(tmp_path / "k.s"), kernel)
    assert seen and not bad, bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_asm_loads_of_mla_decode_are_untouched_until_their_wait(tmp_path):
    """mla_decode.hip requests seqlens and four page-table entries through inline-asm loads and waits for them in a LATER asm
    statement (so that they head the kernel); the compiler treats their outputs as defined at once.  Nothing may touch those
    registers in between (check_flash_asm.audit_async_asm_loads reads the compiled ISA of both instantiations)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_flash_asm

    csrc = os.path.join(ROOT, "chitu_amd", "csrc")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
           "-S", "--cuda-device-only", os.path.join(csrc, "mla_decode.hip"), "-o", str(tmp_path / "k.s")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    seen, bad = check_flash_asm.audit_async_asm_loads(str(tmp_path / "k.s"), "mla_decode_kernel")
    assert len(seen) == 2 and not bad, bad
