"""Oracle pinning: oracle.gae against the reference's known-answer GAE test
(nnx_ppo/algorithms/ppo_test.py:229-264) through the committed fixture."""
import hashlib

import numpy as np

from oracle import gae as og


def _load(golden_dir):
    z = np.load(golden_dir / "gae_seed23.npz")
    return {k: z[k] for k in z.files}


def test_fixture_is_the_reference_case(golden_dir):
    z = _load(golden_dir)
    assert z["rewards"].shape == (100, 512) and z["values"].shape == (101, 512)
    assert int(z["done"].sum()) == 515 and int(z["truncation"].sum()) == 246
    assert not np.any(z["truncation"] & ~z["done"])
    a32 = z["advantages"].astype(np.float32)
    # recorded in SURVEY.md §7 / §8c when the reference's own loop was run
    assert hashlib.sha256(a32.tobytes()).hexdigest()[:16] == "16b0582343e22e20"
    assert abs(z["advantages"].mean() - (-0.028385276)) < 1e-8
    assert abs(z["advantages"].std() - 1.823125960) < 1e-8


def test_known_answer_loop_regenerates_fixture(golden_dir):
    z = _load(golden_dir)
    c = og.make_seed23_case()
    for k in ("rewards", "values", "done", "truncation"):
        assert np.array_equal(c[k], z[k])
    adv = og.gae_known_answer(c["rewards"], c["values"], c["done"], c["truncation"],
                              c["gamma"], c["lambda_"])
    assert np.array_equal(adv, z["advantages"])


def test_scan_restatement_matches_known_answer(golden_dir):
    z = _load(golden_dir)
    got = og.gae(z["rewards"], z["values"][:-1], z["values"][-1], z["done"],
                 z["truncation"], float(z["lambda_"]), float(z["gamma"]))
    assert np.max(np.abs(got - z["advantages"])) < 1e-12
    got32 = og.gae(z["rewards"], z["values"][:-1], z["values"][-1], z["done"],
                   z["truncation"], float(z["lambda_"]), float(z["gamma"]),
                   dtype=np.float32)
    assert got32.dtype == np.float32
    # the reference test's own bound (ppo_test.py:264)
    assert np.max(np.abs(got32.astype(np.float64) - z["advantages"])) < 1e-6


def test_edge_cases():
    # T=1, all done, all truncated, no done
    r = np.array([[1.0, 2.0, 3.0]])
    v = np.array([[0.5, 0.5, 0.5]])
    lv = np.array([10.0, 10.0, 10.0])
    done = np.array([[False, True, True]])
    tr = np.array([[False, False, True]])
    a = og.gae(r, v, lv, done, tr, 0.95, 0.9)
    assert np.allclose(a, [[1.0 + 9.0 - 0.5, 2.0 - 0.5, 0.0]])
    # empty time axis
    e = og.gae(np.zeros((0, 4)), np.zeros((0, 4)), np.zeros(4),
               np.zeros((0, 4), bool), np.zeros((0, 4), bool), 0.95, 0.99)
    assert e.shape == (0, 4)
