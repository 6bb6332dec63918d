"""Sim-to-sim check (run in the build container only: it reads the reference checkpout under /root/reference): the policy
the reference authors trained in RaiSim (IRRL/script/pkl/bp5_155.pkl) drives THIS build's physics (f64 oracle, Manual
mode, the reference's own test config IRRL/script/config/bp5_test.yaml) with a fixed velocity command, exactly like
`run_bp_v5.py --test`.  If the robot trots at the commanded speed without falling, the build-defined rigid-body + contact
model is close enough to RaiSim's for a RaiSim-trained controller -- the only cross-check against the closed-source
simulator available here.    usage: python tools/sim2sim_reference_policy.py [--solver N] [--emu] [cmd_vx ...]
(--solver: ContactSolver, 0..3; default = the shipped default 3: published per-contact rule, simultaneous sweeps)"""
import os, sys
import numpy as np
import yaml

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF = "/root/reference/IRRL/script"
import oracle as O
from high_speed_quadrupedal_locomotion_by_irrl_amd.checkpoint import NumpyLstmActor, read_checkpoint
from high_speed_quadrupedal_locomotion_by_irrl_amd.helper import obs_normalisation


def run(cmd_vx, steps=2000, verbose=True, emulated_kernel=False, solver=3):
    cfg = yaml.safe_load(open(os.path.join(REF, "config", "bp5_test.yaml")))["environment"]
    cfg["num_envs"] = 1
    cfg.setdefault("ContactIterations", 6); cfg.setdefault("ContactTolerance", 1.0e-4)
    cfg["ContactSolver"] = solver
    if emulated_kernel:   # the f32 KERNEL SOURCE (csrc/env_core.hpp, 16-lane layout) on emulated lanes instead of the f64 oracle
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from host_emulation import emu as E
        env = E.EmuVecEnv16(cfg)
    else:
        env = O.OracleVecEnv(cfg)
    _, params = read_checkpoint(os.path.join(REF, "pkl", "bp5_155.pkl"))
    ctrl = NumpyLstmActor.from_parameter_list(params, n_layers=2)
    mean, std, _, _ = obs_normalisation(cfg)
    ob = env.reset() if hasattr(env, "reset") else env.observe()
    cmd = np.array([cmd_vx, 0.0, 0.0])
    S = O.S
    vx, z, falls, tilt = [], [], 0, []
    for t in range(steps):
        o = np.array(ob[0], np.float64)
        o[0:3] = (cmd - mean[0:3]) / std[0:3]
        a = ctrl.predict(o)
        ob, r, d, x = env.step(a[None, :].astype(np.float32))
        st = env.get_state()[0]
        vx.append(st[19]); z.append(st[2]); tilt.append(1 - 2 * (st[4] ** 2 + st[5] ** 2))
        if d[0]:
            falls += 1
            ctrl.reset()
    vx = np.array(vx)
    res = dict(solver=solver, engine="kernel source f32 (emulated lanes)" if emulated_kernel else "oracle f64", cmd=cmd_vx, mean_vx_last_half=float(vx[steps // 2:].mean()), falls=falls, mean_height=float(np.mean(z)), min_upright=float(np.min(tilt)),
               wildcat=bool(cfg.get("WILDCAT")))
    if verbose:
        print(res)
    return res


if __name__ == "__main__":
    args = [v for v in sys.argv[1:]]
    emu = "--emu" in args
    solver = 3
    if "--solver" in args:
        i = args.index("--solver")
        solver = int(args[i + 1])
        del args[i:i + 2]
    for c in [float(v) for v in args if v != "--emu"] or [0.5, 1.0, 2.0, 3.0]:
        run(c, emulated_kernel=emu, solver=solver)
