"""CPU tier: oracle/glue.py against golden vectors captured from the reference's own Python
(tests/golden/make_glue_vectors.py ran env_script/env_mujoco_util.py behind third-party stand-ins)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import glue  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "glue_vectors.npz"))
BASE = np.array([0.0, 0.0, 0.157])


def test_get_rotation():
    for r, inv, out in zip(G["rot_in"], G["rot_inv"], G["rot_out"]):
        assert np.allclose(glue.get_rotation(r[0], r[1], r[2], r[3:6], bool(inv)), out, atol=1e-14)


def test_touch_class_bit_exact():
    got = np.array([glue.touch_class(s) for s in G["touch_sens"]])
    assert np.array_equal(got, G["touch_class"])
    assert set(G["touch_class"].tolist()) == {0, 1, 2, 3}


def test_reward_picking():
    for k in range(len(G["s_ee"])):
        r = glue.reward_picking(G["s_ee"][k], G["s_eeq"][k], G["s_obj"][k], G["s_touch"][k])
        assert abs(r - G["s_reward"][k]) < 1e-12


@pytest.mark.parametrize("task,key", [("picking", "s_term_pick"), ("placing", "s_term_place")])
def test_terminal_flags_bit_exact(task, key):
    seen = set()
    for k in range(len(G["s_ee"])):
        done, bonus, wb, succ = glue.terminal(task, G["s_q2"][k], G["s_ee"][k], G["s_obj"][k], G["s_dest_goal"][k], int(G["s_touch"][k]),
                                              int(G["s_nsteps"][k]), BASE)
        e = G[key][k]
        assert (bool(e[0]), int(e[3])) == (done, succ)                 # done / success: exact
        assert abs(e[1] - bonus) < 1e-12 and abs(e[2] - wb) < 1e-12
        seen.add((done, succ, bonus < 0))
    assert len(seen) >= 3   # the fixture exercises continue / success / failure branches


def test_env_timeout():
    for n in G["timeout_steps"]:
        done, bonus, wb, succ = glue.env_terminal("picking", int(n) - 1, 1.2, [0, .4, .3], [0, .65, .2], [.4, .3, .35], 0, 5, BASE)
        assert done == (n >= G["timeout_expect_done_at"][0])
        if done:
            assert (bonus, wb, succ) == (-10.0, 0.0, 0)


def test_subgoal_and_observation():
    for k in range(len(G["s_ee"])):
        nz = G["s_noise"][k]
        pos, ori = glue.rulebased_subgoal("picking", G["s_ee"][k], G["s_obj_goal"][k], G["s_obj"][k][1], G["s_dest_goal"][k], nz[:6])
        assert np.allclose(pos, G["s_sub_pos"][k], atol=1e-12) and np.allclose(ori, G["s_sub_ori"][k], atol=1e-10)
        touch = glue.touch_class(nz[6:])
        obs = glue.observation("picking", touch, G["s_ee"][k], G["s_eeq"][k], G["s_grip"][k], G["s_obj"][k], G["s_dest_goal"][k],
                               G["s_obj_goal"][k], nz[:6])
        assert obs.dtype == np.float32 and obs.shape == (26,)
        assert np.allclose(obs, G["s_obs"][k], atol=2e-7)


def test_take_action():
    for k in range(len(G["s_ee"])):
        a, g0 = G["s_act"][k][:7], G["s_act"][k][7]
        target, grip, ramp = glue.take_action(G["s_ee"][k], G["s_eeq"][k], a, g0)
        assert np.allclose(target, G["s_target"][k], atol=1e-12)
        assert abs(grip - G["s_grip_after"][k]) < 1e-15 and np.allclose(ramp, G["s_ramp"][k], atol=1e-15)
        assert np.allclose(G["s_mocap_hand_pos"][k], target[:3], atol=1e-12)
        q = glue.quat_from_euler(*target[3:6]); e = G["s_mocap_hand_quat"][k]
        assert min(np.abs(q - e).max(), np.abs(q + e).max()) < 1e-12
        # the marker placement consumes the 6 draws of the first _get_rulebased_subgoal call of the step
        pos, ori = glue.rulebased_subgoal("picking", G["s_ee"][k], G["s_obj_goal"][k], G["s_obj"][k][1], G["s_dest_goal"][k], G["s_noise_act"][k])
        assert np.allclose(pos, G["s_mocap_sub_pos"][k], atol=1e-12)


# ---- task 'reaching' (tests/golden/make_glue_vectors_reaching.py ran the reference's own branches)
R = np.load(os.path.join(ROOT, "tests", "golden", "glue_vectors_reaching.npz"))


def test_reaching_reward_and_terminal():
    seen = set()
    for k in range(len(R["r_ee"])):
        r = glue.reward_reaching(R["r_ee"][k], R["r_eeq"][k], R["r_goal"][k], BASE)
        assert abs(r - R["r_reward"][k]) < 1e-12
        done, bonus, wb, succ = glue.terminal("reaching", R["r_q2"][k], R["r_ee"][k], R["r_obj"][k], R["r_dest_goal"][k], 0, int(R["r_nsteps"][k]), BASE,
                                              ee_quat=R["r_eeq"][k], reach_goal=R["r_goal"][k])
        e = R["r_term"][k]
        assert bool(e[0]) == done and abs(e[1] - bonus) < 1e-12 and abs(e[2] - wb) < 1e-12
        assert succ == int(done and bonus > 100)      # the success flag the reference's 3-tuple lacks
        seen.add((done, bonus > 0))
    assert seen == {(False, False), (True, True), (True, False)}


def test_reaching_goal_sampling_with_float16_cast():
    for d, g in zip(R["g_draws"], R["g_goal"]):
        got = glue.sample_reach_goal(d, BASE)
        assert np.array_equal(got[3:], g[3:]) and np.allclose(got[:3], g[:3], atol=1e-15)      # orientation: float16 values, exact
        assert np.all(got[3:] == got[3:].astype(np.float16).astype(np.float64))


def test_reaching_goal_observation_branch():
    for k in range(len(R["r_ee"])):
        touch = glue.touch_class(R["r_sens"][k])
        obs = glue.observation("reaching", touch, R["r_ee"][k], R["r_eeq"][k], R["r_grip"][k], R["r_obj"][k], R["r_dest_goal"][k], None, None,
                               reach_goal=R["r_goal"][k])
        assert np.allclose(obs, R["r_obs"][k], atol=2e-7)


# ---------------------------------------------------------------- grasping / pickAndplace branches (round 3)
@pytest.fixture(scope="module")
def GG():
    return np.load(os.path.join(ROOT, "tests", "golden", "glue_vectors_grasping.npz"))


def test_grasping_reward_and_terminal_match_reference(GG):
    base = np.array([0.0, 0.0, 0.157])
    for k in range(len(GG["g_reward"])):
        ee, q, obj, touch = GG["g_ee"][k], GG["g_eeq"][k], GG["g_obj"][k], int(GG["g_touch"][k])
        assert abs(glue.reward_grasping(ee, q, obj, touch) - GG["g_reward"][k]) < 1e-12
        d, b, wb, succ = glue.terminal("grasping", GG["g_q2"][k], ee, obj, GG["g_dest_goal"][k], touch, int(GG["g_nsteps"][k]), base)
        rd, rb, rwb = GG["g_term_grasp"][k]
        assert bool(rd) == d and abs(rb - b) < 1e-12 and abs(rwb - wb) < 1e-12
        assert succ == int(d and b > 100)                                   # the 4th value the reference's 3-tuple lacks


def test_pickandplace_terminal_with_picked_flag_matches_reference(GG):
    base = np.array([0.0, 0.0, 0.157])
    seen = set()
    for k in range(len(GG["g_reward"])):
        picked = [bool(GG["g_picked_in"][k])]
        d, b, wb, succ = glue.terminal("pickAndplace", GG["g_q2"][k], GG["g_ee"][k], GG["g_obj"][k], GG["g_dest_goal"][k], int(GG["g_touch"][k]),
                                       int(GG["g_nsteps"][k]), base, picked=picked)
        rd, rb, rwb = GG["g_term_pp"][k]
        assert bool(rd) == d and rb == b and abs(rwb - wb) < 1e-12 and picked[0] == bool(GG["g_picked_out"][k])
        assert succ == int(b == 180.0)
        seen.add(float(b))
    assert seen == {-20.0, -1.0, 0.0, 20.0, 180.0}                          # every branch of :585-600 is exercised


def test_grasping_prereach_quantities_match_reference(GG):
    for k in range(len(GG["g_reward"])):
        ori = glue.grasp_reach_ori(GG["g_ee"][k], GG["g_obj_goal"][k], GG["g_gamma"][k])
        assert np.array_equal(ori, GG["g_reach_ori"][k])                    # float16 values: exact
        dist, ang = glue.grasp_prereach_conditions(GG["g_ee"][k], GG["g_eeq"][k], GG["g_obj_goal"][k], GG["g_goal_ori"][k])
        assert abs(dist - GG["g_dist"][k]) < 1e-12 and abs(ang - GG["g_angdiff"][k]) < 1e-9
    assert list(GG["task_max_steps"]) == [500, 1200]
    assert glue.env_terminal("grasping", 498, 1.3, np.ones(3), np.ones(3), np.ones(3), 0, 0, np.zeros(3))[1] != -10.0
    assert glue.env_terminal("grasping", 499, 1.3, np.ones(3), np.ones(3), np.ones(3), 0, 0, np.zeros(3)) == (True, -10.0, 0.0, 0)
    assert glue.env_terminal("pickAndplace", 1198, 1.3, np.ones(3), np.ones(3), np.ones(3), 0, 0, np.zeros(3), picked=[False])[1] != -10.0
    assert glue.env_terminal("pickAndplace", 1199, 1.3, np.ones(3), np.ones(3), np.ones(3), 0, 0, np.zeros(3), picked=[False]) == (True, -10.0, 0.0, 0)


# ---------------------------------------------------------------- releasing / carrying / pushing branches (round 4)
@pytest.fixture(scope="module")
def GR():
    return np.load(os.path.join(ROOT, "tests", "golden", "glue_vectors_releasing.npz"))


def test_releasing_carrying_pushing_terminal_match_reference(GR):
    """glue.terminal against the reference's own _get_terminal_inspection (env_mujoco_util.py:549-566,583-584): done flag and bonus
    exact; the success flag is the value the reference's 3-tuples lack (a 4-tuple only comes back from the singular-pose exit)."""
    base = np.array([0.0, 0.0, 0.157])
    seen = set()
    for k in range(len(GR["r_q2"])):
        args = (GR["r_q2"][k], GR["r_ee"][k], GR["r_obj"][k], GR["r_dest_goal"][k], int(GR["r_touch"][k]), int(GR["r_nsteps"][k]), base)
        for task, key in (("releasing", "r_term_rel"), ("carrying", "r_term_carry"), ("pushing", "r_term_push")):
            d, b, wb, succ = glue.terminal(task, *args, obj_vel=GR["r_objvel"][k])
            rd, rb, rwb, rlen = GR[key][k]
            assert bool(rd) == d and abs(rb - b) < 1e-12 and abs(rwb - wb) < 1e-12, (task, k)
            assert succ == int(d and b > 100)
            assert rlen == (4 if rb == -1.0 else 3)
            if task == "releasing":
                seen.add(round(float(b) if b < 100 else 200.0))
            else:
                assert d and b in (0.0, -1.0)                                   # every episode ends in its first step
        assert list(GR["r_reward"][k]) == [0.0, 0.0, 0.0]                       # _get_reward: 0 for all three (:432-439)
    assert seen == {-20, -1, 0, 200}
    assert list(GR["act_width"]) == [7, 7, 6] and list(GR["task_max_steps"]) == [500, 500, 500]
    assert glue.env_terminal("releasing", 499, 1.3, np.ones(3), np.ones(3), np.ones(3), 0, 0, np.zeros(3), obj_vel=np.zeros(3)) == (True, -10.0, 0.0, 0)
    assert GR["init_pushing_raises"][0] == 1                                    # the reference cannot reset task 'pushing' (documented in env.py)


# ---- kwarg init_buffer (tests/golden/make_glue_vectors_init_buffer.py ran the reference's own goal_buffer branch)
def test_goal_buffer_branch_of_sample_goal():
    B = np.load(os.path.join(ROOT, "tests", "golden", "glue_vectors_init_buffer.npz"))
    buf, n = B["buffer"], len(B["buffer"])
    for i, want in zip(B["idx"], B["reach_goal"]):
        # any uniform draw that lands in bin i of the n - 1 equal bins stands for the reference's randint result i
        got, idx = glue.sample_reach_goal_from_buffer(buf, (i + 0.5) / (n - 1))
        assert idx == i and np.array_equal(got, want)                       # row[1:4], row[4:7], untouched (no float16 cast)
    assert int(B["largest_index_in_4000_draws"][0]) == n - 2               # numpy's randint(0, n - 1): the last row is never drawn
    assert glue.sample_reach_goal_from_buffer(buf, 0.999999)[1] == n - 2 and glue.sample_reach_goal_from_buffer(buf, 0.0)[1] == 0
    assert set(B["idx"].tolist()) <= set(range(n - 1))
