"""CPU oracle for the multi-frame fusion step (reference sunflower/predictor/flower_model.py:18-26,146-213).
TEST INFRASTRUCTURE ONLY -- only tests/ may import this.

filterpy==1.4.5 (the reference's Kalman filter, environment.yml) is not installed here and not vendored in the
reference, and the reference has no tests for this step: PARITY UNPINNED against filterpy itself.  The restatement
below is independent of the product code in flope_amd/sunflower/predictor/flower_model.py: with F = H = I and
P, Q, R multiples of the identity, every one of the 7 state components is the same SCALAR Kalman recursion
    p <- p + q;   k = p / (p + r);   x <- x + k (z - x);   p <- (1 - k)^2 p + k^2 r
so a track is simulated with scalars only, and association is restated with explicit loops.
"""
import math


def scalar_track(z_seq, q=0.001, r=0.1, p0=1.0):
    """z_seq: list of 7-vectors; the first initialises the track.  Returns (x[7], p) after the remaining updates,
    with the quaternion part (components 3..6) re-normalised after every update."""
    x = [float(v) for v in z_seq[0]]
    p = p0
    for z in z_seq[1:]:
        p = p + q
        k = p / (p + r)
        x = [xi + k * (float(zi) - xi) for xi, zi in zip(x, z)]
        p = (1 - k) ** 2 * p + k ** 2 * r
        n = math.sqrt(sum(v * v for v in x[3:]))
        x = x[:3] + [v / n for v in x[3:]]
    return x, p


def associate(frames, th=0.05):
    """frames: list of lists of 7-vectors.  Returns per-track measurement lists, following flower_model.py:160-213:
    the first frame opens one track per measurement; later, each measurement goes to the nearest track ANCHOR (the
    track's first measurement) if closer than th, else opens a new track (invisible to the rest of that frame)."""
    tracks = []
    for fi, frame in enumerate(frames):
        if fi == 0:
            tracks = [[m] for m in frame]
            continue
        n_anchor = len(tracks)
        for m in frame:
            best, bd = -1, float("inf")
            for ti in range(n_anchor):
                a = tracks[ti][0]
                d = math.sqrt(sum((m[c] - a[c]) ** 2 for c in range(3)))
                if d < bd:
                    best, bd = ti, d
            if bd < th:
                tracks[best].append(m)
            else:
                tracks.append([m])
    return tracks
