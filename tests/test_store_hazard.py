"""CPU: no kernel that can have a second wave on its SIMD contains the gfx950 store-data hazard (tools/check_store_hazard.py,
tools/store_war_hazard.hip)."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (cross-compiles gfx950 without a GPU)")
def test_no_unsafe_store_followed_by_a_vector_write_of_its_data():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_the_checker_fires_on_the_microbenchmark_that_demonstrates_the_hazard():
    """positive control (round-3 advisor): the scan must find the store / clobber pairs of tools/store_war_hazard.hip -- adjacent ones as
    adjacent, the ones with an instruction between as windowed -- or a silent regex mismatch would read as 'no hazard anywhere'"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.self_test()


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_training_forward_has_no_adjacent_store_overwrite_pair():
    """rounds 3-5 let mlp_fwd_kernel<true> keep 12 adjacent store / vector-write pairs because its register count forbids a second wave OF
    ITS OWN on the SIMD.  Round 6: a wave of another kernel (the library's second stream, another process) on the SIMD triggers the hazard
    just the same (tools/store_war_hazard_foreign.hip) -- a saved +0.0 became 0x7fffffff = NaN in a run where four processes shared a GPU.
    The pairs are gone (one s_nop 0 behind the colour layer's stores); no occupancy excuses an adjacent site any more."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory() as tmp:
        asm = mod.compile_to_asm(os.path.join(mod.CSRC, "niw_mlp_fwd.hip"), os.path.join(tmp, "fwd.s"))
        found = {k: v for k, v in mod.scan(asm).items() if "mlp_fwd_kernel" in k}
        text = open(asm).read()
    assert "mlp_fwd_kernelILb1E" in text, "the training forward is not in the assembly"
    for kernel, (adjacent, windowed, occupancy) in found.items():
        assert adjacent == 0, f"{kernel}: {adjacent} adjacent store / vector-write pairs"
