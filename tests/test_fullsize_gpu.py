"""Parity at BASELINE.json's full sizes: the native HIP training step — the variant bench.py measures:
compact GCL, bf16-planes GEMM operands, B-direct weights, row classes, active slots, message dropout p = 0.1 replayed
from the counter hash — against oracle/vae_cpu.py on the same synthetic batch, weights and eps, at

    configs[1]  LMD2  2-bar  B = 256  d = 256  L = 8   (the bench workload)
    configs[2]  LMD16 16-bar B = 64   d = 256  L = 8
    configs[4]  dense stress, one GPU's shard of the 8-GPU job reduced to B = 8 (d = 512, 16,256 edges per bar)

The oracle runs twice, in fp32 (the reference's arithmetic) and in fp64.  At these sizes the reference's own fp32
arithmetic sits 1e-4 .. 5e-4 from exact on the model outputs and 4e-2 .. 5e-2 on single gradient tensors (16
BatchNorm'd layers amplify summation-order rounding), so "within 1e-4 of the reference" is anchored on the fp64 result:

    outputs  (s_logits, c_logits, mu, log_var):  |HIP - fp64| <= 1e-4 rel (measured 2e-6 .. 8e-6), and
             |HIP - fp32 oracle| <= |fp32 oracle - fp64| + 1e-4  (the HIP path is no further from the reference than the
             reference is from exact arithmetic);
    losses   1e-6;
    gradients: worst tensor and whole-vector L2 error of HIP against fp64 at most 0.75x / 0.5x those of the fp32 oracle
             (measured 22x / 24x smaller at configs[1]; on the dense shard, measured 2.3-3x / 2.7x smaller, at most 1x)
             and below fixed caps.
Measured values: profiles/r02_fullsize_parity.json (tools/fullsize_parity.py)."""
import pytest

from util import FULLSIZE, REL_TOL, hip_vs_oracle_fullsize

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(FULLSIZE))
def test_native_step_matches_oracle_at_full_size(name):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import host_cores
    rep = hip_vs_oracle_fullsize(FULLSIZE[name], threads=host_cores())
    info = rep["info"]
    assert info["compact"] == 1 and info["planes"] == 1 and info["b_frag"] == 1 and info["n_slots"] < 15, info
    for k, e in rep["outputs"].items():
        assert e["hip_vs_o64"] < REL_TOL, (k, e)
        assert e["hip_vs_o32"] <= e["o32_vs_o64"] + REL_TOL, (k, e)
    for k, e in rep["losses"].items():
        assert e["hip_vs_o64"] < 1e-6, (k, e)
    g = rep["grad"]
    # (the dense shard's margin is 2.3-3x, and the fp32 oracle's own error moves with the host's thread count: there the
    #  HIP path only has to be no worse than the fp32 reference; the sparse configurations keep 20x of margin)
    kw, kl = (1.0, 1.0) if FULLSIZE[name]["dense"] else (0.75, 0.5)
    assert g["hip_vs_o64"]["worst_tensor_err"] <= kw * g["o32_vs_o64"]["worst_tensor_err"], g
    assert g["hip_vs_o64"]["rel_l2"] <= kl * g["o32_vs_o64"]["rel_l2"], g
    assert g["hip_vs_o64"]["worst_tensor_err"] < 3e-2 and g["hip_vs_o64"]["rel_l2"] < 3e-3, g
