"""Do two builds of libirrl_env.so compute the same thing, bit for bit?  Each library walks the same pools (benchmark config + the training config
with resets, 4096 and 8192 envs = both lane layouts) through the same seeded actions in its own process; SHA-256 over every step's
ob | reward | done | extraInfo and over the final pool state are compared.
usage: python tools/compare_libs.py LIB_A LIB_B [--steps 300]
       python tools/compare_libs.py LIB_A --against profiles/r06_trajectory_hashes.json   (hashes recorded by an earlier run on the same GPU model;
                                                                                           --record FILE writes them)"""
import hashlib, json, os, subprocess, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def child(steps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from hip_env import HipVecEnv
    out = {}
    for cfg_name, n in (("bp5_imitation.yaml", 4096), ("default_cfg.yaml", 4096), ("default_cfg.yaml", 8192)):
        cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, cfg_name)))["environment"]
        cfg["num_envs"] = n
        env = HipVecEnv(cfg)
        env.reset()
        rng = np.random.default_rng(7)
        h = hashlib.sha256()
        resets = 0
        for _ in range(steps):
            a = np.clip(0.4 * rng.standard_normal((n, 12)), -1, 1).astype(np.float32)
            ob, rew, done, extra = env.step(a)
            resets += int(done.sum())
            for x in (ob, rew, done, extra):
                h.update(np.ascontiguousarray(x).tobytes())
        h.update(np.ascontiguousarray(env.get_state()).tobytes())
        out["%s/%d" % (cfg_name, n)] = {"sha256": h.hexdigest(), "resets": resets}
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]))
        sys.exit(0)
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 300
    def opt(name):
        return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else None
    against, record = opt("--against"), opt("--record")
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    res = []
    for lib in libs[:2]:
        env = dict(os.environ, IRRL_ENV_LIB=os.path.abspath(lib))
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(steps)], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(o.stdout[-2000:], o.stderr[-2000:])
            sys.exit(2)
        res.append(json.loads(line[0][7:]))
    if record:
        json.dump({"steps": steps, "pools": res[0]}, open(record, "w"), indent=1)
    if against:
        ref = json.load(open(against))
        assert ref["steps"] == steps, "recorded with --steps %d" % ref["steps"]
        res.append(ref["pools"])
    ok = len(res) == 2
    for k in res[0]:
        same = len(res) == 2 and res[0][k] == res[1][k]
        ok &= same
        print("%-28s %d steps, %d resets: %s  %s" % (k, steps, res[0][k]["resets"], ("IDENTICAL" if same else "DIFFERENT") if len(res) == 2 else "recorded", res[0][k]["sha256"]))
    sys.exit(0 if ok or (record and len(res) == 1) else 1)
