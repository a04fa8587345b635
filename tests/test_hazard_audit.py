"""CPU: the shipped code objects keep the required distance between every matrix instruction and the first
consumer of its destination registers (tools/hazard_audit.py disassembles libcsvae.so).  A build whose final
matrix instruction was followed too closely by the epilogue's first accumulator read produced run-to-run
differences in the last bits of two output columns a few launches in a hundred: far too rare for a
repeated-launch test to catch reliably, but visible in the instruction stream every time."""
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

LLVM_OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(LLVM_OBJDUMP), reason="needs the ROCm LLVM binutils")
def test_no_matrix_result_is_consumed_early(tmp_path):
    so = os.path.join(ROOT, "vae_conditional_sampling_amd", "libcsvae.so")
    if not os.path.exists(so):
        pytest.skip("libcsvae.so has not been built")
    out = tmp_path / "audit.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hazard_audit.py"), so, "--json", str(out)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    import json
    report = json.loads(out.read_text())
    names = " ".join(r["kernel"] for r in report)
    # every family of matrix kernels was seen (a silent parse failure would make the audit vacuous)
    for family in ("cs_mlp_split16_kernel", "cs_mlp_split_kernel", "cs_mlp_kernel", "cs_conv3x3_halo_split_kernel", "cs_conv_mfma_split_kernel",
                   "cs_conv3x3_halo_kernel", "cs_conv_mfma_kernel"):
        assert family in names, f"no kernel of {family} in the audit"
    assert sum(r["mfma"] for r in report) > 20000
    assert all(not r["early"] for r in report)
