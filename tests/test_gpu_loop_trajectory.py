"""-m gpu: all 50 sampler steps of BASELINE configs[1] at SD-1.5 shape against the committed fp32-oracle trajectory
(tests/golden/t1_sd15_loop_trajectory_g0.3.npz; generated on the CPU by tests/golden/make_loop_trajectory.py -- oracle/loops.py on the
seeded synthetic network with its output layer damped to 0.3, i.e. a contractive eps network like a trained one).  The HIP loop
runs on the oracle's inversion outputs; after every step the edited and the reconstruction-branch latents are compared.
Per-format limits (tests/helpers/gpu.py::lim): bfloat16 storage measured 3.4e-2 / 2.6e-2 at step 50 (profiles/r05_loop_divergence.txt B),
half storage a quarter of the bfloat16 limits."""
import os

import pytest
import torch

from helpers import gpu as G
from helpers import trajectory as TR

pytestmark = pytest.mark.gpu


def test_sd15_50_step_loop_follows_the_oracle_trajectory():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    assert os.path.exists(TR.fixture_path(0.3)), "fixture missing: python tests/golden/make_loop_trajectory.py 0.3"
    ts, e_edit, e_rec, rms = TR.hip_vs_oracle_trajectory(0.3)
    print(f"storage {G._lib.STORAGE}: SD-1.5-shape h_Edit_p2p_implicit vs oracle trajectory, edited rel L2 step 1 / 10 / 25 / 50: "
          f"{e_edit[0]:.3e} / {e_edit[9]:.3e} / {e_edit[24]:.3e} / {e_edit[-1]:.3e}; reconstruction {e_rec[0]:.3e} / {e_rec[9]:.3e} / "
          f"{e_rec[24]:.3e} / {e_rec[-1]:.3e}")
    assert all(v == v for v in e_edit + e_rec)                 # finite
    G.within(e_edit[0], 1.0e-2, what="edited latent after step 1")
    G.within(max(e_edit), 5.0e-2, what="edited latent, worst step of 50")
    G.within(max(e_rec), 4.0e-2, what="reconstruction branch, worst step of 50")
