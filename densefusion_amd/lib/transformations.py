"""The two pose-algebra functions of lib/transformations.py that sit on the path, for callers that
still want them on the host (numpy fp64): ``quaternion_matrix`` (:1254-1278) and
``quaternion_from_matrix`` with ``isprecise=True`` (:1320-1341,1361-1363).  The device-side
equivalents used by the refine loop live in csrc/pose.hip.
"""
from __future__ import annotations

import math

import numpy

_EPS = numpy.finfo(float).eps * 4.0


def quaternion_matrix(quaternion):
    q = numpy.array(quaternion, dtype=numpy.float64, copy=True)
    n = float(numpy.dot(q, q))
    if n < _EPS:
        return numpy.identity(4)
    q *= math.sqrt(2.0 / n)
    o = numpy.outer(q, q)
    M = numpy.identity(4)
    M[0, 0], M[0, 1], M[0, 2] = 1.0 - o[2, 2] - o[3, 3], o[1, 2] - o[3, 0], o[1, 3] + o[2, 0]
    M[1, 0], M[1, 1], M[1, 2] = o[1, 2] + o[3, 0], 1.0 - o[1, 1] - o[3, 3], o[2, 3] - o[1, 0]
    M[2, 0], M[2, 1], M[2, 2] = o[1, 3] - o[2, 0], o[2, 3] + o[1, 0], 1.0 - o[1, 1] - o[2, 2]
    return M


def quaternion_from_matrix(matrix, isprecise=True):
    if not isprecise:
        raise NotImplementedError("only the isprecise=True branch is on the DenseFusion path")
    M = numpy.asarray(matrix, dtype=numpy.float64)[:4, :4]
    t = numpy.trace(M)
    if t > M[3, 3]:
        q = numpy.array([t, M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
    else:
        i, j, k = 0, 1, 2
        if M[1, 1] > M[0, 0]:
            i, j, k = 1, 2, 0
        if M[2, 2] > M[i, i]:
            i, j, k = 2, 0, 1
        t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
        v = numpy.empty(4)
        v[i], v[j], v[k], v[3] = t, M[i, j] + M[j, i], M[k, i] + M[i, k], M[k, j] - M[j, k]
        q = v[[3, 0, 1, 2]]
    q = q * (0.5 / math.sqrt(t * M[3, 3]))
    return -q if q[0] < 0.0 else q
