"""The shape of the lean step kernels' ISA (no GPU: hipcc -S of csrc/fishing_step.hip with the product's flags, scripts/isa_shape.py).

Two things this round's readings of the ISA found and fixed, held here so that they do not come back unseen:
* round 4 made the zig-zag walk a run-time field of the by-value argument struct and so strung FOUR dependent scalar-load round trips
  in front of every wave's first global load (the walk now rides in the preloaded argument word: DESIGN.md section 5);
* wherever control flow follows the noise generator (fishing-v4's derivation, Beverton-Holt / Myers / May), LLVM sank the whole
  Philox + Box-Muller block behind the wait for the tile's loads (pinned by an empty asm that reads the normals)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.fixture(scope="module")
def shapes():
    import isa_shape
    if isa_shape.hipcc_path() is None:
        pytest.skip("no hipcc")
    ks = isa_shape.listing()
    return {k: isa_shape.shape(v) for k, v in ks.items() if isa_shape.is_exact_lean(k)}


def test_every_hot_request_has_its_exact_kernel(shapes):
    for name in ("fishing::step_kernel_lean<float, 1, 12294, 4>", "fishing::step_kernel_lean<float, 0, 12294, 4>",
                 "fishing::step_kernel_lean<float, 2, 12294, 4>", "fishing::step_kernel_lean<float, 4, 8462, 4>",
                 "fishing::step_kernel_lean<float, 105, 8198, 4>", "fishing::step_kernel_lean<double, 105, 8198, 2>",
                 "fishing::step_kernel_lean<double, 1, 12294, 2>"):
        assert name in shapes, name
    assert len(shapes) >= 50


def test_no_chain_of_scalar_loads_in_front_of_the_tiles_loads(shapes):
    """At most three `s_waitcnt lgkmcnt` between a kernel's entry and its first global load, counted over every path (one of them
    sits in the branch only graph-replay launches at the zig-zag sizes take): the struct's argument batch, nothing else."""
    for name, s in shapes.items():
        assert s["first_global_load"] is not None and s["lgkm_waits_before_first_load"] <= 3, (name, s)


def test_the_noise_generator_runs_under_the_tiles_loads(shapes):
    """The Philox4x32-10 block's 32 x 32 -> 64-bit multiplies (19-20 of them) are issued BEFORE the first `s_waitcnt vmcnt`."""
    for name, s in shapes.items():
        assert s["wide_multiplies_before_first_vmcnt_wait"] >= 16, (name, s)
        assert s["first_vmcnt_wait"] is not None and s["first_vmcnt_wait"] > s["first_global_load"], (name, s)
