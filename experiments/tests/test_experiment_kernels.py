"""GPU, `make EXPERIMENTS=1` builds only: the opt-in kernels kept under experiments/csrc -- built to parity, measured slower than (or equal
to) the product kernels, outside the product library (DESIGN.md / LABNOTES.md say what each one taught):
  sed_conv_wir.hip      forward / data gradient with the weights resident in registers (SED_CONV_KERNEL=r)
  sed_conv_w4.hip       one wave per SIMD, all weights of a wave in registers, side work in the MFMA gaps (SED_CONV_KERNEL=4)
  sed_bwd_fused_cs.hip  fused weight + data gradient of blocks 2-3, the workgroups of a strip sliced by input channels (SED_BWD_FUSED_CS=1)
The cases call the product suite's own test functions (tests/test_gpu_kernels_ab.py, tests/test_gpu_kernels_oracle.py) with the
experiment kernels selected / the geometries only they cover.  Run: experiments/tools/test_experiments_build.sh (GPU box)."""
import importlib
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in (ROOT, os.path.join(ROOT, "tests")):
    if d not in sys.path:
        sys.path.insert(0, d)
import test_gpu_kernels_ab as AB            # noqa: E402
import test_gpu_kernels_oracle as KO        # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.experiments]
SHAPES = AB.SHAPES
test_forward_and_data_gradient_kernels = AB.test_forward_and_data_gradient_kernels.__wrapped__ if hasattr(AB.test_forward_and_data_gradient_kernels, "__wrapped__") else AB.test_forward_and_data_gradient_kernels


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    lib = importlib.import_module("soundeventdetection-pytorch_amd")._lib
    if not lib.lib().sed_build_flags() & 1:
        pytest.fail("experiments/tests needs a library built with `make EXPERIMENTS=1` (experiments/tools/test_experiments_build.sh)")
    return lib


WIR_SHAPES = [s for s in SHAPES if s[3] >= 64] + [(1, 3, 16, 128, 128), (5, 2, 8, 128, 128), (2, 1500, 16, 128, 128),
                                                   (3, 7, 32, 64, 64), (2, 31, 8, 64, 128), (2, 19, 8, 128, 64), (1, 64, 16, 64, 64)]


@pytest.mark.parametrize("B,H,W,Cin,Cout", WIR_SHAPES)
def test_resident_weight_kernel(L, monkeypatch, B, H, W, Cin, Cout):
    """csrc/sed_conv_wir.hip (weights in registers, row ring filled by LDS-DMA, images chained through one shared zero row)
    against the previous-generation LDS-weights kernel: ragged heights, images shorter than one step, steps that straddle
    two images, a single workgroup, every prologue / epilogue."""
    test_forward_and_data_gradient_kernels(L, monkeypatch, B, H, W, Cin, Cout, first="r")


W4_SHAPES = [s for s in WIR_SHAPES if (s[2], s[3], s[4]) in ((16, 128, 128), (8, 128, 128), (16, 64, 128), (16, 128, 64), (32, 64, 64))]


@pytest.mark.parametrize("B,H,W,Cin,Cout", W4_SHAPES)
def test_one_wave_per_simd_resident_weight_kernel(L, monkeypatch, B, H, W, Cin, Cout):
    """csrc/sed_conv_w4.hip (256-thread workgroups, all weights of a wave's 32 output channels in registers, side work in
    the MFMA gaps) against the previous-generation LDS-weights kernel, same cases as the two-waves-per-SIMD kernel."""
    test_forward_and_data_gradient_kernels(L, monkeypatch, B, H, W, Cin, Cout, first="4")




@pytest.mark.parametrize("W,Cin,Cout", KO.GEOM_C1_EXP)
@pytest.mark.parametrize("B,H,nwg", KO.FUSED_CASES)
def test_cin_sliced_fused_backward_conv1_vs_oracle(L, monkeypatch, B, H, nwg, W, Cin, Cout):
    """conv1 of blocks 2 / 3 (64 -> 128 at W = 16, 128 -> 128 at W = 8) through csrc/sed_bwd_fused_cs.hip, same oracle as block 1's."""
    KO.test_fused_backward_conv1_vs_oracle(L, monkeypatch, B, H, nwg, W, Cin, Cout)


@pytest.mark.parametrize("W,C,Cq,pool", KO.GEOM_C2_EXP)
@pytest.mark.parametrize("B,H,nwg", KO.FUSED_CASES)
def test_cin_sliced_fused_backward_conv2_vs_oracle(L, monkeypatch, B, H, nwg, W, C, Cq, pool):
    """conv2 of blocks 2 / 3 (128 -> 128 at W = 16 with pool 2 or 1, W = 8 with pool 1) through csrc/sed_bwd_fused_cs.hip."""
    KO.test_fused_backward_conv2_vs_oracle(L, monkeypatch, B, H, nwg, W, C, Cq, pool, "1")
