// irrl_config.hpp -- host-side configuration: a minimal parser for the flat `environment:` YAML
// mapping the reference hands to its C++ side as a STRING (run_bp_v5.py:205-207 ->
// VectorizedEnvironment.hpp:135 YAML::Load), and the derivation of the kernel parameters.
//
// The reference aborts the process when a key is missing (READ_YAML -> RSFATAL_IF,
// RaisimGymEnv.hpp:41-42; key list Environment.hpp:1594-1659).  Here a missing key is an error
// string returned through the C-ABI (irrl_last_error) -- same contract ("every key is mandatory"),
// recoverable failure mode.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "env_params.h"

namespace irrl_host {

struct Config {
  std::map<std::string, std::string> kv;

  static std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return (a == std::string::npos) ? std::string() : s.substr(a, b - a + 1);
  }
  // Accepts block style ("key: value" per line, any indentation, '#' comments) and the flow style
  // ruamel/PyYAML emit for small mappings ("{a: 1, b: 2}").
  bool parse(const std::string &text, std::string &err) {
    std::string t = trim(text);
    if (!t.empty() && t[0] == '{') {
      if (t[t.size() - 1] != '}') { err = "unterminated flow mapping"; return false; }
      t = t.substr(1, t.size() - 2);
      for (size_t i = 0; i < t.size(); i++) if (t[i] == ',') t[i] = '\n';
    }
    size_t pos = 0;
    while (pos <= t.size()) {
      size_t nl = t.find('\n', pos);
      std::string line = t.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
      pos = (nl == std::string::npos) ? t.size() + 1 : nl + 1;
      // strip comments (a '#' preceded by start-of-line or whitespace)
      for (size_t i = 0; i < line.size(); i++)
        if (line[i] == '#' && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) { line = line.substr(0, i); break; }
      line = trim(line);
      if (line.empty() || line == "---") continue;
      size_t c = line.find(':');
      if (c == std::string::npos) { err = "cannot parse line: '" + line + "'"; return false; }
      std::string k = trim(line.substr(0, c)), v = trim(line.substr(c + 1));
      if (v.size() >= 2 && ((v[0] == '"' && v[v.size() - 1] == '"') || (v[0] == '\'' && v[v.size() - 1] == '\''))) v = v.substr(1, v.size() - 2);
      if (k.size() >= 2 && (k[0] == '"' || k[0] == '\'')) k = k.substr(1, k.size() - 2);
      kv[k] = v;
    }
    return true;
  }
  bool has(const std::string &k) const { return kv.find(k) != kv.end(); }
  bool get_double(const std::string &k, double &out, std::string &err) const {
    auto it = kv.find(k);
    if (it == kv.end()) { err = "Node cfg[\"" + k + "\"] doesn't exist"; return false; }
    const std::string &v = it->second;
    if (v == ".inf" || v == ".Inf") { out = INFINITY; return true; }
    char *end = nullptr;
    out = std::strtod(v.c_str(), &end);
    if (end == v.c_str() || *end != 0) {
      bool b;
      if (to_bool(v, b)) { out = b ? 1.0 : 0.0; return true; }
      err = "cfg[\"" + k + "\"] = '" + v + "' is not a number"; return false;
    }
    return true;
  }
  static bool to_bool(const std::string &v, bool &b) {
    if (v == "True" || v == "true" || v == "TRUE" || v == "yes" || v == "Yes" || v == "on") { b = true; return true; }
    if (v == "False" || v == "false" || v == "FALSE" || v == "no" || v == "No" || v == "off") { b = false; return true; }
    return false;
  }
  bool get_bool(const std::string &k, bool &out, std::string &err) const {
    auto it = kv.find(k);
    if (it == kv.end()) { err = "Node cfg[\"" + k + "\"] doesn't exist"; return false; }
    if (to_bool(it->second, out)) return true;
    double d;
    if (!get_double(k, d, err)) return false;
    out = (d != 0.0);
    return true;
  }
};

// Every key the reference reads, in its own order (Environment.hpp:1598-1658), then the vectorizer's
// (VectorizedEnvironment.hpp:146-171).  "RefTraj"/"render" are optional there too (try/catch, if()).
static const char *const kMandatoryKeys[] = {
    "abad", "period", "lam", "stand_height", "up_height", "down_height", "gait_step", "Vx", "Vy", "Omega", "LeanFront",
    "LeanHind", "Terrain", "Manual", "Crutial", "Filter", "Camera", "StochasticDynamics", "HeightVariable",
    "TimeBasedContact", "ManualTraj", "MotorDynamics", "ObsFilter", "WILDCAT", "ForceDisturbance", "Convert2Torque",
    "terminalRewardCoeff", "EndEffectorRewardCoeff", "BodyPosRewardCoeff", "BodyAttitudeRewardCoeff", "JointRewardCoeff",
    "VelRewardCoeff", "TorqueCoeff", "ContactCoeff", "Stiffness", "Stiffness_Low", "AbadRatio", "Damping", "Freq",
    "max_time", "CubeNum", "FPS", "ActionNoise", "ObsNoise", "GaitType", "MotorMaxTorque", "MotorCriticalSpeed",
    "MotorMaxSpeed", "num_envs", "num_threads", "simulation_dt", "control_dt", "seedd"};

// scalars derived from the configuration; call again whenever control_dt / period / lam / max_time change
inline void derive_params(EnvParams &P) {
  P.inv_control_dt = 1.0f / P.control_dt;
  P.inv_period = 1.0f / P.period;
  P.inv_lam = 1.0f / P.lam;
  P.inv_one_minus_lam = 1.0f / (1.0f - P.lam);
  P.cmd_resample_p = 0.5f / (P.max_time / P.control_dt);
  P.two_pi_over_period = 2.0f * 3.1415926f / P.period;
}

// the frame cadences the reference evaluates with the LIVE control_dt_, in double (Environment.hpp:733 `frame_idx % int(5 * period_ /
// control_dt_)` for the meteorite, :747 `int(period_ / control_dt_ * 10)` for state_disturbance) -- and the flags that depend on
// them.  Called by build_params and again by the control-dt setter (VectorizedEnvironment::setControlTimeStep).
inline void derive_cadences(const Config &c, double control_dt, EnvParams &P) {
  double period = 0.0;
  bool crutial = false, force = false;
  std::string e;
  c.get_double("period", period, e); c.get_bool("Crutial", crutial, e); c.get_bool("ForceDisturbance", force, e);
  P.attack_every = control_dt > 0.0 ? (int32_t)(5.0 * period / control_dt) : 0;
  P.crutial = (crutial && P.attack_every > 0) ? 1 : 0;
  P.disturb_every = control_dt > 0.0 ? (int32_t)(period / control_dt * 10.0) : 0;
  P.state_disturbance = (force && P.manual && P.disturb_every > 0) ? 1 : 0;
}

inline bool build_params(const Config &c, EnvParams &P, std::string &err) {
  for (const char *k : kMandatoryKeys)
    if (!c.has(k)) { err = std::string("Node cfg[\"") + k + "\"] doesn't exist"; return false; }
  auto num = [&](const char *k) { double v = NAN; std::string e; if (!c.get_double(k, v, e) && err.empty()) err = e; return v; };
  auto flag = [&](const char *k) { bool v = false; std::string e; if (!c.get_bool(k, v, e) && err.empty()) err = e; return v; };
  std::memset(&P, 0, sizeof(P));
  P.n_envs = (int32_t)num("num_envs");
  const double sim_dt = num("simulation_dt"), control_dt = num("control_dt");
  P.sim_dt = (float)sim_dt;
  P.control_dt = (float)control_dt;
  P.loop_count = (int32_t)(control_dt / sim_dt + 1e-10);  // Environment.hpp:711
  P.seed = (uint32_t)(int32_t)num("seedd");
  P.env_id_offset = c.has("EnvIdOffset") ? (uint32_t)(int64_t)num("EnvIdOffset") : 0u;   // [ext] global id of env 0 (multi-GPU shards)
  P.max_time = (float)num("max_time");
  P.abad = (float)num("abad"); P.period = (float)num("period"); P.lam = (float)num("lam");
  P.stand_height = (float)num("stand_height"); P.up_height_max = (float)num("up_height");
  P.Vx = (float)num("Vx"); P.Vy = (float)num("Vy"); P.Omega = (float)num("Omega");
  P.lean_front = (float)num("LeanFront"); P.lean_hind = (float)num("LeanHind");
  P.manual = flag("Manual"); P.height_variable = flag("HeightVariable"); P.time_based_contact = flag("TimeBasedContact");
  P.wildcat = flag("WILDCAT"); P.stochastic = flag("StochasticDynamics"); P.obs_filter = flag("ObsFilter");
  P.c_term = (float)num("terminalRewardCoeff"); P.c_ee = (float)num("EndEffectorRewardCoeff");
  P.c_pos = (float)num("BodyPosRewardCoeff"); P.c_att = (float)num("BodyAttitudeRewardCoeff");
  P.c_joint = (float)num("JointRewardCoeff"); P.c_vel = (float)num("VelRewardCoeff");
  P.c_torque = (float)num("TorqueCoeff"); P.c_contact = (float)num("ContactCoeff");
  const double stiff = num("Stiffness"), damp = num("Damping"), ratio = num("AbadRatio");
  P.kp[0] = (float)(stiff * ratio); P.kp[1] = (float)stiff; P.kp[2] = (float)stiff;  // Environment.hpp:338-350
  P.kd[0] = (float)(damp * ratio); P.kd[1] = (float)damp; P.kd[2] = (float)damp;
  // Environment.hpp:396 / 423-427 run in the constructor, before setControlTimeStep (VectorizedEnvironment.hpp:
  // 150-152), so they see the base-class default control_dt_ = 0.01 (RaisimGymEnv.hpp:111).
  P.filter_para = flag("Filter") ? (float)(1.0 - num("Freq") * 0.01) : 0.0f;
  P.obs_filter_alpha = P.obs_filter ? (float)(2.0 * 3.14 * 0.01 * 20.0 / (2.0 * 3.14 * 0.01 * 20.0 + 1.0)) : 1.0f;
  P.action_noise = (float)num("ActionNoise"); P.obs_noise = (float)num("ObsNoise");
  P.tau_max = (float)num("MotorMaxTorque"); P.w_crit = (float)num("MotorCriticalSpeed"); P.w_max = (float)num("MotorMaxSpeed");
  const int gait = (int)num("GaitType");
  const float ph[3][4] = {{0.5f, 0.0f, 0.0f, 0.5f}, {0.5f, 0.5f, 0.0f, 0.0f}, {0.0f, 0.25f, 0.5f, 0.75f}};  // Environment.hpp:398-409
  for (int i = 0; i < 4; i++) P.phase[i] = (gait >= 0 && gait <= 2) ? ph[gait][i] : 0.0f;
  const double lh = 0.085, lt = 0.209, lc = 0.2175;  // Environment.hpp:1949-1952
  P.max_len = (float)std::sqrt(lh * lh + (lc + lt) * (lc + lt));
  // build-defined extension keys (optional)
  P.contact_iters = c.has("ContactIterations") ? (int32_t)num("ContactIterations") : 6;
  if (P.contact_iters <= 0) P.contact_iters = 6;
  P.contact_tol = c.has("ContactTolerance") ? (float)num("ContactTolerance") : 0.0f;
  {
    // bit 1: order inside a sweep (set = the toes of a robot update simultaneously, clear = Gauss-Seidel FR, FL, HR, HL);
    // bit 0: per-contact rule (set = the published maximum-dissipation rule of RaiSim's solver, clear = the build's first rule)
    const int solver = c.has("ContactSolver") ? (int)num("ContactSolver") : 3;
    if (solver < 0 || solver > 3) { err = "ContactSolver must be 0..3 (bit 1: simultaneous sweeps, bit 0: published per-contact rule)"; return false; }
    P.contact_jacobi = (solver & 2) ? 1 : 0;
    P.contact_rule = (solver & 1) ? 1 : 0;
    // how ContactTolerance ends the simultaneous sweeps: 1 (default) = before a sweep whose change is PREDICTED to be below the
    // tolerance, 0 = after a sweep whose own change was (rounds 1-3); Gauss-Seidel sweeps always use 0
    const int ex = c.has("ContactExit") ? (int)num("ContactExit") : 1;
    if (ex != 0 && ex != 1) { err = "ContactExit must be 0 (confirmed) or 1 (predicted)"; return false; }
    P.contact_exit = (ex == 1 && P.contact_jacobi) ? 1 : 0;
  }
  // the build-defined contact keys are a closed set: a misspelt or unsupported one (e.g. a relaxation factor only some other
  // implementation knows) must not be ignored silently -- two engines would then solve different iterations
  for (const auto &it : c.kv)
    if (it.first.compare(0, 7, "Contact") == 0 && it.first != "ContactCoeff" && it.first != "ContactIterations" &&
        it.first != "ContactTolerance" && it.first != "ContactSolver" && it.first != "ContactExit") { err = "unsupported build-defined key cfg[\"" + it.first + "\"]"; return false; }
  P.clamp_r = P.tau_max / (P.w_max - P.w_crit);
  P.clamp_inv_den = 1.0f / (-P.w_max + P.w_crit);
  P.shared_noise = c.has("SharedNoiseScalar") ? (int32_t)flag("SharedNoiseScalar") : 1;
  P.randomize_per_episode = c.has("RandomizePerEpisode") ? (int32_t)flag("RandomizePerEpisode") : 0;
  if (!err.empty()) return false;
  if (P.n_envs <= 0) { err = "num_envs must be positive"; return false; }
  if (P.loop_count <= 0) { err = "control_dt / simulation_dt must be >= 1"; return false; }
  const bool crutial = flag("Crutial"), terrain = flag("Terrain"), manual_traj = flag("ManualTraj"), force = flag("ForceDisturbance");
  {
    const double cubes = num("CubeNum");
    P.cube_num = (float)(cubes > 0.0 ? (double)(int)cubes : 1.0);
  }
  P.terrain = terrain ? 1 : 0;  // the table itself is attached by the owner of the pool (irrl_terrain.hpp)
  P.hf_nx = 5000; P.hf_ny = 500;  // Environment.hpp:259-260
  P.hf_x0 = -250.0f; P.hf_y0 = -10.0f; P.hf_max = 0.0f;
  P.hf_inv_dx = (float)((5000 - 1) / 500.0); P.hf_inv_dy = (float)((500 - 1) / 20.0);
  {
    // -x0 / dx split into whole cells and a fraction (env_core.hpp terrain_sample)
    // (in double with the exact 1 / dx: the f32 rounding of hf_inv_dx then only multiplies |x| <~ 15 m, not the 250 m of the offset)
    const double ox = -(double)P.hf_x0 * ((5000 - 1) / 500.0), oy = -(double)P.hf_y0 * ((500 - 1) / 20.0);
    P.hf_ix0 = (int32_t)std::floor(ox); P.hf_iy0 = (int32_t)std::floor(oy);
    P.hf_fx = (float)(ox - std::floor(ox)); P.hf_fy = (float)(oy - std::floor(oy));
    P.hf_xlo = (float)(-(double)P.hf_ix0); P.hf_xhi = (float)((double)P.hf_nx - 1.001 - (double)P.hf_ix0);
    P.hf_ylo = (float)(-(double)P.hf_iy0); P.hf_yhi = (float)((double)P.hf_ny - 1.001 - (double)P.hf_iy0);
  }
  P.height = nullptr;
  P.ref_traj = (!manual_traj && !P.manual) ? 1 : 0;   // table attached by the owner of the pool (RefTraj CSV or irrl_env_set_ref_host)
  P.ref_rows = 0; P.ref = nullptr;
  // ForceDisturbance: with Manual it is state_disturbance (ENV:912-940, built); without Manual it is force_attack, whose trigger
  // `random() < 2 dt / T` (ENV:751) compares a 31-bit integer with a number < 1 and never fires -> nothing to build
  (void)crutial; (void)force;
  derive_cadences(c, control_dt, P);
  derive_params(P);
  return true;
}

}  // namespace irrl_host
