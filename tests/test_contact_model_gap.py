"""The stated MODEL GAP of the contact solve.  RaiSim resolves contacts with the per-contact iteration of
Hwangbo, Lee & Hutter, "Per-Contact Iteration Method for Solving Contact Dynamics" (RA-L 2018): Gauss-Seidel
over the contacts, each single-contact problem solved EXACTLY under Signorini's condition, the Coulomb cone and the maximum
dissipation principle -- for a slipping contact a bisection along the curve {cone boundary} x {v_n = target} for the point of
minimal kinetic energy.  RaiSim itself is closed source and absent, so the build's own rule (sticking solve; otherwise slide
along the sticking impulse's tangential direction with the normal condition kept exact; `solve_contact` in oracle/irrl_oracle.c
and csrc/env_core.hpp) cannot be compared with RaiSim -- but it CAN be compared with an independent numpy implementation of the
published method on the contact problems the oracle itself poses (Delassus blocks, free velocities, normals and restitution
targets captured from the running oracle through its probe).  The test states what agrees exactly and bounds what does not."""
import numpy as np
import pytest

import oracle as O
from conftest import load_env_cfg

S = O.S


def _single_contact_hwangbo(G, c, n, mu, vstar):
    """exact single-contact solve of the published method: velocity after the impulse v+ = c + G lam (c already holds the other
    contacts' contributions), target normal speed vstar.  -> lam"""
    cn = c @ n - vstar
    if cn >= 0.0:                                    # opening contact
        return np.zeros(3)
    lam = np.linalg.solve(G, vstar * n - c)          # sticking: v+ = vstar n exactly
    ln = lam @ n
    lt = lam - ln * n
    if ln > 0.0 and np.linalg.norm(lt) <= mu * ln:
        return lam
    # slipping: lam(theta) = ln(theta) (n + mu d(theta)) on the cone boundary with (c + G lam) . n = vstar; among those the
    # minimiser of the post-impact kinetic energy  h = 1/2 lam^T G lam + lam^T c'  (c' = c - vstar n)  -- maximum dissipation
    t1 = np.cross(n, [1.0, 0.0, 0.0])
    if np.linalg.norm(t1) < 0.1:
        t1 = np.cross(n, [0.0, 1.0, 0.0])
    t1 /= np.linalg.norm(t1)
    t2 = np.cross(n, t1)
    cp = c - vstar * n

    def point(th):
        w = n + mu * (np.cos(th) * t1 + np.sin(th) * t2)
        den = n @ G @ w
        if den <= 1e-12:
            return None, np.inf
        lam = (-cn / den) * w
        return lam, 0.5 * lam @ G @ lam + lam @ cp

    th = np.linspace(0.0, 2.0 * np.pi, 3601)
    e = np.array([point(t)[1] for t in th])
    k = int(np.argmin(e))
    lo, hi = th[max(k - 1, 0)], th[min(k + 1, len(th) - 1)]
    for _ in range(60):                               # bisection on the sign of dh/dtheta inside the bracketing interval
        mid = 0.5 * (lo + hi)
        d = 1e-7
        if point(mid + d)[1] > point(mid - d)[1]:
            hi = mid
        else:
            lo = mid
    return point(0.5 * (lo + hi))[0]


def _hwangbo_per_contact_iteration(prob, mu, sweeps=200):
    act = np.nonzero(prob["active"])[0]
    lam = np.zeros((4, 3))
    for _ in range(sweeps):
        delta = 0.0
        for i in act:
            c = prob["cfree"][i].copy()
            for j in act:
                if j != i:
                    c += prob["G"][3 * i:3 * i + 3, 3 * j:3 * j + 3] @ lam[j]
            new = _single_contact_hwangbo(prob["G"][3 * i:3 * i + 3, 3 * i:3 * i + 3], c, prob["n"][i], mu, prob["vstar"][i])
            delta = max(delta, np.abs(new - lam[i]).max())
            lam[i] = new
        if delta < 1e-13:
            break
    return lam


def _oracle_problem(case, solver=2, precision="f64"):
    """one 0.25 ms substep of the oracle from a prepared state; returns the captured contact problem and friction"""
    dt = 0.00025
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=1, control_dt=dt, simulation_dt=dt, ContactIterations=200, ContactTolerance=0.0,
                       ContactSolver=solver)
    env = O.OracleVecEnv(cfg, precision)
    st = env.get_state()
    st[0, S["GC"]:S["GC"] + 19] = 0
    st[0, S["GC"] + 2] = case.get("z", 0.2890)                 # nominal stance: toes ~1 mm into the ground
    st[0, S["GC"] + 3] = 1.0
    q = np.array([0, -0.78, 1.57] * 4, float)
    for leg in case.get("lifted", ()):                          # fold a leg up: its toe leaves the ground
        q[3 * leg:3 * leg + 3] = [0.0, -1.6, 2.5]
    st[0, S["GC"] + 7:S["GC"] + 19] = q
    st[0, S["GV"]:S["GV"] + 18] = 0
    st[0, S["GV"]:S["GV"] + 3] = case.get("v", (0, 0, 0))
    st[0, S["GV"] + 3:S["GV"] + 6] = case.get("w", (0, 0, 0))
    st[0, S["LAMW"]:S["LAMW"] + 12] = 0
    st[0, S["INCONTACT"]:S["INCONTACT"] + 4] = 0
    st[0, S["TQL"]:S["TQL"] + 12] = 0
    env.set_state(st)
    env.contact_probe(0)
    env.step(np.zeros((1, 12), np.float32))
    return env.contact_problem(), float(st[0, S["MATERIAL"]]), float(st[0, S["MATERIAL"] + 1]), float(st[0, S["MATERIAL"] + 2])


CASES = {
    "4 feet at rest (sticking)": dict(),
    "4 feet drifting at 0.05 m/s (friction cannot stop it within one substep: sliding)": dict(v=(0.05, 0.02, 0)),
    "4 feet sliding forwards 1.5 m/s": dict(v=(1.5, 0, 0)),
    "4 feet sliding diagonally + yaw": dict(v=(-1.0, 1.0, 0), w=(0, 0, 2.0)),
    "2 diagonal feet (trot stance), sliding sideways": dict(v=(0, 1.2, 0), lifted=(1, 2)),
    "1 foot, sliding": dict(v=(0.8, -0.5, 0), lifted=(1, 2, 3)),
    "4 feet landing at 0.3 m/s: above the restitution threshold": dict(v=(0, 0, -0.3)),
    "4 feet landing at 5 mm/s: below the restitution threshold": dict(v=(0, 0, -0.005)),
    "landing 0.4 m/s while sliding 1 m/s": dict(v=(1.0, 0, -0.4)),
}


@pytest.mark.parametrize("solver", [1, 3])
@pytest.mark.parametrize("name", list(CASES))
def test_published_rule_in_the_oracle_equals_the_numpy_per_contact_iteration(name, solver):
    """ContactSolver 1 (Gauss-Seidel) and 3 (simultaneous sweeps) with the published per-contact rule (oracle solve_contact_md: closed
    form up to a scalar Newton root) against the brute-force numpy implementation of the same published method (scan + bisection on
    the polar angle of the conic), on the robot's own contact problems: 1 / 2 / 4 feet, sticking, sliding, landing -- identical
    impulses (1e-6 of the largest one)."""
    case = CASES[name]
    prob, mu, rest, thr = _oracle_problem(case, solver=solver)
    lam_h = _hwangbo_per_contact_iteration(prob, mu)
    scale = np.abs(lam_h).max()
    assert scale > 0
    assert np.abs(prob["lam"] - lam_h).max() <= 1e-6 * scale, (name, prob["lam"], lam_h)


def test_published_rule_single_contact_random_problems_f64_and_f32():
    """the single-contact solve alone, on randomised problems around the captured ones (friction 0.2 .. 1.0, perturbed velocities,
    tilted normals): f64 oracle vs numpy brute force to 2e-6, f32 build of the same C source to 5e-4 (f32 cancellation in c.n)"""
    rng = np.random.default_rng(5)
    prob, mu, _, _ = _oracle_problem(CASES["4 feet sliding diagonally + yaw"], solver=1)
    worst64 = worst32 = 0.0
    n_slide = 0
    for trial in range(300):
        i = int(rng.integers(4))
        G = prob["G"][3 * i:3 * i + 3, 3 * i:3 * i + 3] * rng.uniform(0.7, 1.4)
        n = np.array([rng.normal(0, 0.2), rng.normal(0, 0.2), 1.0])
        n /= np.linalg.norm(n)
        c = rng.standard_normal(3) * np.array([1.0, 1.0, 0.3]) - 0.05 * n
        m = float(rng.uniform(0.2, 1.0))
        vs = float(rng.choice([0.0, 0.05]))
        ref = _single_contact_hwangbo(G, c, n, m, vs)
        sc = np.abs(ref).max()
        if sc == 0.0:
            assert np.all(O.solve_contact_md(G, c, n, vs, m) == 0.0)
            continue
        ln = ref @ n
        n_slide += np.linalg.norm(ref - ln * n) > (1 - 1e-7) * m * ln
        worst64 = max(worst64, np.abs(O.solve_contact_md(G, c, n, vs, m) - ref).max() / sc)
        worst32 = max(worst32, np.abs(O.solve_contact_md(G, c, n, vs, m, "f32") - ref).max() / sc)
    assert n_slide > 100, n_slide
    assert worst64 < 2e-6 and worst32 < 5e-4, (worst64, worst32)


@pytest.mark.parametrize("name", list(CASES))
def test_simultaneous_sweeps_and_gauss_seidel_reach_the_same_impulses(name):
    """the order inside a sweep (ContactSolver bit 1) changes the path, not the fixed point: converged impulses of the simultaneous
    iteration equal the Gauss-Seidel ones, for the build's first rule (2 vs 0) and for the published one (3 vs 1)"""
    for a, b in ((2, 0), (3, 1)):
        pa, _, _, _ = _oracle_problem(CASES[name], solver=a)
        pb, _, _, _ = _oracle_problem(CASES[name], solver=b)
        scale = np.abs(pb["lam"]).max()
        assert scale > 0 and np.abs(pa["lam"] - pb["lam"]).max() <= 1e-7 * scale, (name, a, b)


@pytest.mark.parametrize("name", list(CASES))
def test_block_gs_rule_against_the_published_per_contact_iteration(name):
    """ContactSolver 0 / 2 -- the build's FIRST sliding rule, kept as an option -- against the published method: the stated gap."""
    case = CASES[name]
    prob, mu, rest, thr = _oracle_problem(case)
    act = prob["active"]
    assert act.sum() == 4 - len(case.get("lifted", ())), (name, act)
    lam_o = prob["lam"]
    lam_h = _hwangbo_per_contact_iteration(prob, mu)
    # restitution targets: e * |v_n| above the threshold, 0 below
    vz = -case.get("v", (0, 0, 0))[2]
    if vz > thr:
        assert np.allclose(prob["vstar"][act], rest * vz, rtol=0.05)
    else:
        assert np.all(prob["vstar"][act] == 0.0)
    n = prob["n"]
    ln_o, ln_h = (lam_o * n).sum(1), (lam_h * n).sum(1)
    lt_o, lt_h = lam_o - ln_o[:, None] * n, lam_h - ln_h[:, None] * n
    # both obey the cone and push
    for ln, lt in ((ln_o, lt_o), (ln_h, lt_h)):
        assert np.all(ln[act] >= 0) and np.all(np.linalg.norm(lt[act], axis=1) <= mu * ln[act] * (1 + 1e-6) + 1e-12)
    # a foot that unloads in the substep (tipping on two feet) may carry no impulse in one or both solutions: compared where both push
    assert np.array_equal(ln_o[act] > 1e-9, ln_h[act] > 1e-9), (name, ln_o, ln_h)
    act = act & (ln_o > 1e-9) & (ln_h > 1e-9)
    sliding = np.linalg.norm(lt_h[act], axis=1) > (1 - 1e-6) * mu * ln_h[act]
    scale = np.abs(lam_h).max()
    if not sliding.any():
        # every contact sticks: the two methods solve the same linear system -> identical impulses
        assert np.abs(lam_o - lam_h).max() < 1e-8 * scale, (name, np.abs(lam_o - lam_h).max() / scale)
        return
    # slipping contacts: same normal velocity condition, both on the cone boundary; the tangential DIRECTION is where the rules
    # differ (sticking impulse's direction vs maximum dissipation).  The gap, on the robot's own contact problems:
    v_o = prob["cfree"] + (prob["G"] @ lam_o.reshape(12)).reshape(4, 3)
    v_h = prob["cfree"] + (prob["G"] @ lam_h.reshape(12)).reshape(4, 3)
    assert np.allclose((v_o * n).sum(1)[act], prob["vstar"][act], atol=1e-7)      # the build's rule keeps the normal condition exact
    assert np.allclose((v_h * n).sum(1)[act], prob["vstar"][act], atol=1e-7)
    cosang = np.sum(lt_o[act] * lt_h[act], axis=1) / (np.linalg.norm(lt_o[act], axis=1) * np.linalg.norm(lt_h[act], axis=1))
    ang = np.degrees(np.arccos(np.clip(cosang, -1, 1)))
    rel_n = np.abs(ln_o[act] - ln_h[act]) / ln_h[act]
    e_o = 0.5 * lam_o.reshape(12) @ prob["G"] @ lam_o.reshape(12) + lam_o.reshape(12) @ prob["cfree"].reshape(12)
    e_h = 0.5 * lam_h.reshape(12) @ prob["G"] @ lam_h.reshape(12) + lam_h.reshape(12) @ prob["cfree"].reshape(12)
    print("%-55s friction direction differs by %.2f deg (max), normal impulse by %.2f %%, energy objective %.6g vs %.6g" %
          (name, ang.max(), 100 * rel_n.max(), e_o, e_h))
    # Classical Coulomb friction is anti-parallel to the slip velocity; neither rule is (the published one minimises the kinetic
    # energy on the curve {cone boundary, v_n = target}, which tilts the friction through the off-diagonal Delassus terms).
    def off_coulomb(lam):
        v = prob["cfree"] + (prob["G"] @ lam.reshape(12)).reshape(4, 3)
        vt = v - (v * n).sum(1)[:, None] * n
        lt = lam - (lam * n).sum(1)[:, None] * n
        out = []
        for i in np.nonzero(act)[0]:
            if np.linalg.norm(vt[i]) > 1e-9 and np.linalg.norm(lt[i]) > 1e-12:
                out.append(np.degrees(np.arccos(np.clip(-(lt[i] @ vt[i]) / np.linalg.norm(lt[i]) / np.linalg.norm(vt[i]), -1, 1))))
        return max(out) if out else 0.0
    print("%-55s friction vs -slip: build %.1f deg, published %.1f deg" % ("", off_coulomb(lam_o), off_coulomb(lam_h)))
    # THE STATED GAP on the robot's own contact problems (measured: up to 26 deg / 20 % with four sliding feet, 0.8 deg / 0.2 % with
    # one): friction directions within 30 degrees of each other, normal impulses within 25 %, the build's friction within 10 degrees
    # of the classical anti-slip direction, and the build's impulses remove at least 90 % of the energy the published optimum removes
    assert ang.max() < 30.0 and rel_n.max() < 0.25, (name, ang, rel_n)
    assert off_coulomb(lam_o) < 10.0, (name, off_coulomb(lam_o))
    assert e_h < 0 and e_o <= 0.9 * e_h, (name, e_o, e_h)
