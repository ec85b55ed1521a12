"""Obstacle sets as data rows (cx, cy, kind, a, b): kind 0 = circle (a = radius), 1 = rectangle
(a = width, b = height).  Values: the reference's default list (gym_aqua/envs/aqua.py:59-66), the
"difficult" list of its -v2 ids (gym_aqua/__init__.py:24-31) and the 4 circle + 4 rectangle set the
benchmark configurations use."""
import numpy as np

CIRCLE, RECT = 0.0, 1.0

DEFAULT5 = np.array([
    [15, 75, CIRCLE, 5, 0],
    [20, 35, CIRCLE, 10, 0],
    [65, 85, RECT, 5, 5],
    [85, 20, CIRCLE, 10, 0],
    [85, 75, CIRCLE, 5, 0],
], dtype=np.float64)

DIFFICULT6 = np.array([
    [15, 70, CIRCLE, 5, 0],
    [25, 40, RECT, 10, 10],
    [40, 80, RECT, 10, 10],
    [55, 20, CIRCLE, 10, 0],
    [60, 55, RECT, 20, 20],
    [85, 75, CIRCLE, 5, 0],
], dtype=np.float64)

BENCH8 = np.array([
    [15, 75, CIRCLE, 5, 0],
    [20, 35, CIRCLE, 10, 0],
    [85, 20, CIRCLE, 10, 0],
    [85, 75, CIRCLE, 5, 0],
    [65, 85, RECT, 5, 5],
    [25, 40, RECT, 10, 10],
    [40, 80, RECT, 10, 10],
    [60, 55, RECT, 20, 20],
], dtype=np.float64)

NONE = np.zeros((0, 5), dtype=np.float64)


def rows_from(obstacles):
    """Accepts what the reference's constructor accepts (aqua.py:13,56-68): False/None -> no obstacles,
    True -> the default five, or a list of (np.array([x, y]), 'c', radius) / (np.array([x, y]), 'r', (w, h));
    additionally an [K][5] array of rows.  Returns float64 [K][5]."""
    if obstacles is None or obstacles is False:
        return NONE.copy()
    if obstacles is True:
        return DEFAULT5.copy()
    if isinstance(obstacles, np.ndarray) and obstacles.ndim == 2 and obstacles.shape[1] == 5:
        return np.ascontiguousarray(obstacles, dtype=np.float64)
    rows = []
    for item in obstacles:
        pos, kind, dims = item
        if kind == "c":
            rows.append([float(pos[0]), float(pos[1]), CIRCLE, float(dims), 0.0])
        elif kind == "r":
            rows.append([float(pos[0]), float(pos[1]), RECT, float(dims[0]), float(dims[1])])
        else:
            # the reference raises a bare Exception for an unknown kind (aqua.py:260, in render)
            raise Exception("unknown obstacle type %r" % (kind,))
    return np.asarray(rows, dtype=np.float64).reshape(-1, 5)


def as_reference_list(rows):
    """[K][5] rows -> the tuple list the reference keeps in env.obstacles (aqua.py:59-66)."""
    out = []
    for cx, cy, kind, a, b in np.asarray(rows, dtype=np.float64).reshape(-1, 5):
        if kind == CIRCLE:
            out.append((np.array([cx, cy]), "c", a))
        else:
            out.append((np.array([cx, cy]), "r", (a, b)))
    return out
