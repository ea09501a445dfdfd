"""CPU: the host-side quaternion helpers of the product package against the reference goldens."""
import os

import numpy as np

from densefusion_amd.lib import transformations as tf

G = os.path.join(os.path.dirname(__file__), "golden")


def test_quaternion_helpers_match_reference():
    g = np.load(os.path.join(G, "quaternion.npz"))
    for q, M, qb in zip(g["q"], g["M"], g["q_back"]):
        np.testing.assert_allclose(tf.quaternion_matrix(q), M, rtol=0, atol=1e-15)
        np.testing.assert_allclose(tf.quaternion_from_matrix(M, True), qb, rtol=0, atol=1e-15)
    assert np.allclose(tf.quaternion_from_matrix(g["doc_R123"], True), g["doc_q123"])
    assert np.allclose(tf.quaternion_matrix([0, 1, 0, 0]), np.diag([1, -1, -1, 1]))
