"""Golden vectors for the goal_buffer branch of __sample_goal (env_mujoco_util.py:208-212; kwarg init_buffer, :46), from the reference's OWN
Python (same stand-ins as make_glue_vectors.py; build container only).  For 48 seeds of the global numpy RNG the reference draws
random_idx = np.random.randint(0, len(buffer) - 1) and returns buffer[idx][1:4] / [4:7] as the reaching goal; the script records the buffer,
the index the reference consumed (recovered by replaying the seed), the goal it returned and the object / destination goals drawn after it.
Also recorded: the largest index over 4 000 draws (the last row is never used).  Writes tests/golden/glue_vectors_init_buffer.npz (data only)."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_glue_vectors import install_stubs, make_util  # noqa: E402


def main(out):
    warnings.simplefilter("ignore")
    install_stubs()
    rng = np.random.default_rng(20261005)
    nrows = 9
    buf = rng.uniform(-1, 1, (nrows, 26))
    u = make_util("reaching", rng)
    u.goal_buffer = buf
    u.interface.set_dest_xyz = lambda xy: None
    import io
    import contextlib
    idx, reach, obj, dest = [], [], [], []
    from env_script import env_mujoco_util as ref
    for k in range(48):
        seed = 9100 + k
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            rg, og, dg = u._JacoMujocoEnvUtil__sample_goal()
        np.random.seed(seed)
        i = np.random.randint(0, len(buf) - 1)
        ox, oy = ref.uniform(-0.1, 0.1), 0.65 + ref.uniform(-0.08, 0.02)
        assert og[0][0] == ox and og[0][1] == oy          # the draws after the index are the object goal's, in this order
        idx.append(i); reach.append(rg[0]); obj.append(og[0]); dest.append(dg[0])
    np.random.seed(1)
    top = max(np.random.randint(0, len(buf) - 1) for _ in range(4000))
    G = {"buffer": buf, "idx": np.array(idx, np.int32), "reach_goal": np.array(reach, np.float64), "obj_goal": np.array(obj), "dest_goal": np.array(dest),
         "largest_index_in_4000_draws": np.array([top], np.int32)}
    np.savez_compressed(out, **G)
    print("wrote", out, {k: v.shape for k, v in G.items()}, "largest index", top, "of", nrows, "rows")


if __name__ == "__main__":
    main(os.path.join(os.path.dirname(os.path.abspath(__file__)), "glue_vectors_init_buffer.npz"))
