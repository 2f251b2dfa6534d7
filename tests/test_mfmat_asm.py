"""The rollout of the transposed-sets kernel stores without lane masks: a lane without a row lands on a cell ANOTHER lane
owns, and the owner's store must follow in program order.  That is an order between lanes — to the compiler the two
addresses never alias — so it is pinned by a wavefront-scope fence in the hand-over (admm_mfmat.hip.h) and checked here on
the compiler's own assembly of BASELINE config 4's instantiation (with three steps per run of products and no fence the
compiler did swap two pairs: knots 25 and 37 of every instance came out wrong)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("form", [0, 2])
def test_handover_store_order_in_the_compiled_rollout(tmp_path, form):
    """form 0: the shipped one (plain stores behind a wavefront fence: this test is its guard); form 2: the fallback for a
    toolchain on which form 0 fails (one asm statement per hand-over, -DTMPC_MFMAT_HANDOVER=2) — known good here too"""
    from check_handover_order import decode, kernels
    csrc = os.path.join(ROOT, "tinympc-julia_amd", "csrc")
    src, out = str(tmp_path / "one.hip"), str(tmp_path / "one.s")
    with open(src, "w") as f:                                  # config 4's kernel alone (the whole family takes minutes)
        f.write('#include "mfmat_entry.hip.h"\nnamespace tmpc {\n'
                "template __global__ void admm_mfmat_kernel<6, 3, 50, 1, 0, 3, 0, 3, false>(const AdmmParams);\n"
                "template __global__ void admm_mfmat_kernel<6, 3, 50, 1, 0, 0, 0, 0, false>(const AdmmParams);\n}\n")
    # the Makefile's flags for the matrix-core instantiations
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form",
                    f"-DTMPC_MFMAT_HANDOVER={form}", f"-I{csrc}", "--cuda-device-only", "-S", src, "-o", out], check=True)
    ks = kernels(out)
    assert len(ks) == 2                                        # cones, plain box
    for lines in ks:
        missing, bad = decode(lines, 6, 3, 50)
        assert not missing, f"hand-over stores not found in the assembly: {missing[:8]}"
        assert not bad, f"hand-over stores out of order: {bad}"
