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
