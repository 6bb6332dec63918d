"""Independent numpy model of the BlackPanther robot used ONLY to cross-check the oracle's physics
against first principles (kinetic-energy Hessian, momentum, energy).  It keeps the URDF's 17 links
separate (toes are NOT merged into the shanks) and shares no code with oracle/irrl_oracle.c.
Numbers: SURVEY 8(a)-M (black_panther.urdf:18-165, mirrored for the other legs)."""
import numpy as np


def quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def quat_rot(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rot_axis(axis, ang):
    axis = np.asarray(axis, float)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def links(gc):
    """-> list of (mass, com_world, R_world, inertia_local) for the 17 URDF links."""
    p = np.asarray(gc[0:3], float)
    R = quat_rot(np.asarray(gc[3:7], float))
    out = [(3.72, p + R @ np.array([0, 0, -0.003]), R, np.diag([0.016269, 0.050813, 0.060989]))]
    for leg in range(4):
        sf = 1.0 if leg < 2 else -1.0
        sy = -1.0 if leg % 2 == 0 else 1.0
        q = gc[7 + 3 * leg: 10 + 3 * leg]
        pa = p + R @ np.array([sf * 0.212, sy * 0.051, 0.0])
        Ra = R @ rot_axis([1, 0, 0], q[0])
        out.append((0.54, pa + Ra @ np.array([sf * 0.058, sy * 0.00485, 0.0]), Ra, np.diag([0.000391, 0.000739, 0.000488])))
        pt = pa + Ra @ np.array([0, sy * 0.085, 0])
        Rt = Ra @ rot_axis([0, -1, 0], q[1])
        It = np.diag([0.001724, 0.001907, 0.000468])
        It[1, 2] = It[2, 1] = -sy * 0.000228
        out.append((0.636, pt + Rt @ np.array([0, -sy * 0.019, -0.01865]), Rt, It))
        ps = pt + Rt @ np.array([0, 0, -0.201])
        Rs = Rt @ rot_axis([0, -1, 0], q[2])
        out.append((0.064, ps + Rs @ np.array([0, 0, -0.0865]), Rs, np.diag([0.000716, 0.000721, 0.000012])))
        ptoe = ps + Rs @ np.array([0, 0, -0.19])
        out.append((0.05, ptoe, Rs, np.eye(3) * 0.000025))
    return out


def toe_positions(gc):
    ls = links(gc)
    return np.array([ls[4 + 4 * leg][1] for leg in range(4)])


def advance(gc, gv, eps):
    gc = np.array(gc, float)
    out = gc.copy()
    out[0:3] += eps * gv[0:3]
    w = np.asarray(gv[3:6], float)
    ang = np.linalg.norm(w) * eps
    if ang != 0:
        ax = w / np.linalg.norm(w)
        dq = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        out[3:7] = quat_mul(dq, gc[3:7])  # world-frame angular velocity -> left multiplication
    out[7:] += eps * np.asarray(gv[6:], float)
    return out


def kinetic_energy(gc, gv, eps=1e-6):
    lp, lm = links(advance(gc, gv, eps)), links(advance(gc, gv, -eps))
    l0 = links(gc)
    T = 0.0
    for (m, cp, Rp, I), (_, cm, Rm, _), (_, _, R0, _) in zip(lp, lm, l0):
        v = (cp - cm) / (2 * eps)
        W = ((Rp - Rm) / (2 * eps)) @ R0.T
        w = np.array([W[2, 1], W[0, 2], W[1, 0]])
        T += 0.5 * m * v @ v + 0.5 * w @ (R0 @ I @ R0.T) @ w
    return T


def mass_matrix(gc, rotor=True):
    n = 18
    E = np.eye(n)
    Ti = np.array([kinetic_energy(gc, E[i]) for i in range(n)])
    M = np.zeros((n, n))
    for i in range(n):
        M[i, i] = 2 * Ti[i]
        for j in range(i + 1, n):
            M[i, j] = M[j, i] = kinetic_energy(gc, E[i] + E[j]) - Ti[i] - Ti[j]
    if rotor:
        for leg in range(4):
            for k, r in enumerate([0.003708, 0.003708, 0.008966]):
                M[6 + 3 * leg + k, 6 + 3 * leg + k] += r
    return M


def com_world(gc):
    ls = links(gc)
    m = sum(l[0] for l in ls)
    return sum(l[0] * l[1] for l in ls) / m, m


def random_config(rng, z=0.45):
    gc = np.zeros(19)
    gc[0:2] = rng.uniform(-1, 1, 2)
    gc[2] = z
    q = rng.normal(size=4)
    q[0] += 3.0
    gc[3:7] = q / np.linalg.norm(q)
    nominal = np.array([0, -0.78, 1.57] * 4)
    gc[7:] = nominal + rng.uniform(-0.4, 0.4, 12)
    return gc
