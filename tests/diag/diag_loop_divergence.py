"""Diagnostic (not a test): HIP vs the fp32 oracle on BASELINE configs[1] at SD-1.5 shape, per sampler step, over all 50 steps
(K = 1, P2P Replace + Reweight + LocalBlend), in the storage format of this process (HEDIT_STORAGE).  One line per step: relative
L2 distance of the edited latent and of the reconstruction-branch latent after that step.

    python tests/diag/diag_loop_divergence.py [gain=0.3]
    HEDIT_STORAGE=f16 python tests/diag/diag_loop_divergence.py

Round 6: the ORACLE side is no longer run here.  Its trajectory (inversion outputs + [x_orig, x_edit] after every step) is generated
once in the build container by tests/golden/make_loop_trajectory.py and committed (tests/golden/t1_sd15_loop_trajectory_g<gain>.npz), so
this script costs seconds of GPU time instead of holding an MI355X for 15-45 minutes of host fp32 arithmetic (round 5 spent 146 of its
270 GPU-minutes that way; profiles/r05_loop_divergence.txt keeps those runs, including the bf16-storage emulation and the
full-gain chains).  gain = the damping of the synthetic network's output layer the fixture was made with (a random-init eps network
at full gain is not contractive: tests/helpers/models.py)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "h-edit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from helpers import trajectory as TR  # noqa: E402
from hedit import _lib  # noqa: E402

gain = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
t0 = time.time()
ts, e_edit, e_rec, rms = TR.hip_vs_oracle_trajectory(gain)
print(f"# storage {_lib.STORAGE}: HIP vs fp32 oracle trajectory, h_Edit_p2p_implicit, SD-1.5 shape, all {len(ts)} steps (t = {ts[0]} .. {ts[-1]}), K = 1, "
      f"output gain {gain}; {time.time() - t0:.0f} s including model set-up")
print("# step     t   edited rel.L2   reconstruction rel.L2   |x_edit| rms (oracle)")
for i, t in enumerate(ts):
    print(f"  {i + 1:4d}  {t:4d}     {e_edit[i]:.3e}        {e_rec[i]:.3e}            {rms[i]:.3f}")
print(f"# storage {_lib.STORAGE}: final edited {e_edit[-1]:.3e}, reconstruction {e_rec[-1]:.3e}; worst step edited {max(e_edit):.3e}, reconstruction {max(e_rec):.3e}")
