#!/usr/bin/env python3
"""Build-container only: the reference's own SIMULATOR LOGS -> tests/golden/raisim_body_logs.json.

`/root/reference/Exp_Raw_Data/body-center-<date>.bin` + `Param-<date>.txt` are recordings of the reference authors' RaiSim evaluation harness
(not in the repository) driving the `bp5_155` policy: 13 float32 per 500 Hz frame = base position (3), quaternion wxyz (4), world linear
velocity (3), world angular velocity (3); decoder = Data_Visualization_Code/Figure3.py:17-75 (`seg_len` frames per segment, each segment stored
feature-major), frame period 0.002 s = Figure3.py:193.  They are the only outputs of `world_->integrate()` the reference holds.
This script imports nothing but numpy / yaml / json, decodes every log and writes per log: the Param keys, the first frame, summary statistics
over the window the figure scripts use, the rise curve of the starts from rest and the dominant lines of the z / pitch spectra.  No reference
source text goes into the fixture -- numbers only.      python tools/gen_raisim_log_fixture.py
"""
import glob
import json
import os

import numpy as np
import yaml

SRC = "/root/reference/Exp_Raw_Data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "raisim_body_logs.json")
DT = 0.002


def decode(bin_file, cfg):
    """Figure3.py:27-45: total = NoE * FoE/skip * Num_Of_Env frames in segments of seg_len, every segment stored as [13, frames of the segment]."""
    seg = int(cfg["seg_len"])
    total = int(cfg["NoE"]) * int(int(cfg["FoE"]) / int(cfg["skip_frame"])) * int(cfg["Num_Of_Env"])
    raw = np.fromfile(bin_file, dtype=np.float32)
    if raw.size != 13 * total:
        raise SystemExit("%s: %d floats, Param says %d frames" % (bin_file, raw.size, total))
    data = np.empty((13, total))
    for head in range(0, total, seg):
        tail = min(head + seg, total)
        data[:, head:tail] = raw[head * 13:tail * 13].reshape(13, -1)
    return data.T


def body_frame(d):
    w, x, y, z = d[:, 3], d[:, 4], d[:, 5], d[:, 6]
    R = np.zeros((len(d), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (w * y + x * z)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (w * x + y * z); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    vb = np.einsum("nji,nj->ni", R, d[:, 7:10])
    wb = np.einsum("nji,nj->ni", R, d[:, 10:13])
    roll = np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    pitch = np.arcsin(np.clip(2 * (w * y - x * z), -1, 1))
    return vb, wb, roll, pitch


def summary(d, window):
    """Statistics of the frames in `window` (a slice) -- the same function of a [frames, 13] array the tests apply to this build's recording
    (tests/parity_lib.py body_log_statistics is its twin)."""
    vb, wb, roll, pitch = body_frame(d)
    h = window
    return {"vx_body_mean": float(vb[h, 0].mean()), "vx_body_std": float(vb[h, 0].std()), "vy_body_mean": float(vb[h, 1].mean()),
            "z_mean": float(d[h, 2].mean()), "z_std": float(d[h, 2].std()), "roll_std": float(roll[h].std()),
            "pitch_mean": float(pitch[h].mean()), "pitch_std": float(pitch[h].std()), "yaw_rate_mean": float(d[h, 12].mean()),
            "roll_rate_body_std": float(wb[h, 0].std()), "pitch_rate_body_std": float(wb[h, 1].std()), "vz_std": float(d[h, 9].std())}


def spectrum_lines(sig, k=3):
    s = sig - sig.mean()
    f = np.fft.rfftfreq(len(s), DT)
    P = np.abs(np.fft.rfft(s)) * 2 / len(s)
    idx = np.argsort(P)[-k:][::-1]
    return [[float(f[i]), float(P[i])] for i in idx]


def main():
    logs = []
    for b in sorted(glob.glob(os.path.join(SRC, "body-center-*.bin"))):
        date = os.path.basename(b)[len("body-center-"):-len(".bin")]
        cfg = yaml.safe_load(open(os.path.join(SRC, "Param-%s.txt" % date)))
        d = decode(b, cfg)
        n = len(d)
        from_rest = abs(d[0, 7]) < 0.5 and abs(d[0, 0]) < 0.5
        reverse = d[:, 7].mean() < 0
        if from_rest:
            family, window = "start_from_rest_20s", slice(n // 2, n)
        elif n == 1000:
            family, window = "steady_2s", slice(0, n)               # Figure4.py:357-361 averages ALL frames of these
        else:
            family, window = "steady_20s", slice(n // 2, n)
        # consistency of the decoding itself: |q| = 1, x(t) is the integral of the logged v_x
        qn = np.sqrt((d[:, 3:7] ** 2).sum(1))
        dx_int = float(np.sum(0.5 * (d[1:, 7] + d[:-1, 7])) * DT)
        rec = {"name": date, "params": {k: v for k, v in cfg.items()}, "frames": n, "family": family, "runs_in_minus_x": bool(reverse),
               "first_frame": [float(v) for v in d[0]], "window": [window.start, window.stop], "stats": summary(d, window),
               "check": {"quat_norm_min": float(qn.min()), "quat_norm_max": float(qn.max()), "x_travel": float(d[-1, 0] - d[0, 0]),
                         "x_travel_from_logged_vx": dx_int},
               "z_spectrum_hz_amp": spectrum_lines(d[window, 2]), "pitch_spectrum_hz_amp": spectrum_lines(body_frame(d)[3][window])}
        if from_rest:
            vb = body_frame(d)[0]
            ts = [0.1, 0.2, 0.3, 0.5, 0.75, 1.0, 1.25, 1.5, 2.0, 3.0, 4.0]
            rec["rise"] = {"t": ts, "vx_body": [float(vb[int(t / DT) - 25:int(t / DT) + 25, 0].mean()) for t in ts]}   # 0.1 s = half a stride
            final = rec["stats"]["vx_body_mean"]
            above = np.nonzero(np.convolve(vb[:, 0], np.ones(100) / 100, "same") >= 0.9 * final)[0]
            rec["time_to_90_percent_s"] = float(above[0] * DT)
        logs.append(rec)
    # the power log of one run (Figure5.py:98-126): per frame Num_sub_loop x 12 joint torques then Num_sub_loop x 12 joint rates, 4 kHz
    power = None
    pf = os.path.join(SRC, "power-2021-07-07-08-25-45.bin")
    if os.path.exists(pf):
        raw = np.fromfile(pf, dtype=np.float32)
        frames = 1000
        sub = raw.size // frames // 24
        data = raw.reshape(sub * 24, frames).T                      # one segment (seg_len 1000)
        tq = data[:, :12 * sub].reshape(-1, 12)
        qd = data[:, 12 * sub:].reshape(-1, 12)
        power = {"name": "2021-07-07-08-25-45", "substeps_per_frame": int(sub), "samples": int(tq.shape[0]),
                 "note": "joint side (the figure script divides the knee torque by 1.55 and multiplies the knee rate by 1.55 to get motor side)",
                 "torque_rms": [float(v) for v in np.sqrt((tq ** 2).mean(0))], "torque_absmax": [float(v) for v in np.abs(tq).max(0)],
                 "rate_rms": [float(v) for v in np.sqrt((qd ** 2).mean(0))], "rate_absmax": [float(v) for v in np.abs(qd).max(0)]}
    have = {l["name"] for l in logs}
    param_only = []
    for p in sorted(glob.glob(os.path.join(SRC, "Param-*.txt"))):
        date = os.path.basename(p)[len("Param-"):-len(".txt")]
        if date not in have:
            cfg = yaml.safe_load(open(p))
            param_only.append({"name": date, "NoE": cfg.get("NoE"), "FoE": cfg.get("FoE"), "delay": cfg.get("delay", 0)})
    out = {"source": "Exp_Raw_Data/body-center-*.bin + Param-*.txt of the reference repository (RaiSim evaluation harness of the authors, policy bp5_155)",
           "decoder": "Data_Visualization_Code/Figure3.py:17-75; frame period 0.002 s (Figure3.py:193)",
           "frame_layout": ["x", "y", "z", "qw", "qx", "qy", "qz", "vx", "vy", "vz", "wx", "wy", "wz"],
           "generated_by": "tools/gen_raisim_log_fixture.py", "frame_dt": DT, "logs": logs, "power": power,
           "param_files_without_a_recording": param_only}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, "(%d logs, %d bytes)" % (len(logs), os.path.getsize(OUT)))
    for l in logs:
        s = l["stats"]
        print("%s %-20s delay %s mu %-5s vx %+.3f +- %.3f  z %.4f +- %.4f  roll std %.4f  pitch %+.4f +- %.4f" % (
            l["name"], l["family"], l["params"].get("delay", 0), l["params"]["Mu_Min"], s["vx_body_mean"], s["vx_body_std"], s["z_mean"], s["z_std"],
            s["roll_std"], s["pitch_mean"], s["pitch_std"]))


if __name__ == "__main__":
    main()
