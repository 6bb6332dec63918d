// flexible_robot_pybind.cpp -- the reference's OWN operator boundary, compiled: pybind11 module `_flexible_robot`, class
// `FlexibleGymEnv`, the ctor and the 29 method names of flex_gym/env/raisim_gym.cpp:14-46 -- over the C-ABI of include/irrl_env.h.
//
// Array contract as in the reference (Eigen::Ref<Matrix<float,-1,-1,RowMajor>> & co., RaisimGymEnv.hpp:46-49,
// VectorizedEnvironment.hpp:268-272): C-contiguous float32 / bool numpy arrays passed BY REFERENCE and filled in place; a wrong
// dtype or layout is a TypeError (every array argument is `noconvert`: pybind11 then refuses instead of silently copying).
// Shapes are checked ([N,35], [N,12], [N], [N,E] ...) and a mismatch is a ValueError (the reference would read / write out of
// bounds).  The engine's errors (irrl_last_error) surface as RuntimeError where the reference aborts the process (RSFATAL_IF).
//
// Host-only translation unit: g++ -shared against libirrl_env.so (no HIP headers); built by build.py into
// native/_flexible_robot<EXT_SUFFIX>.  numpy arrays travel through the library's pinned staging (`*_host` entry points).
// DEVICE BUFFERS (round 4): step / reset / observe / isTerminalState also take any object that exposes `__cuda_array_interface__`
// (torch tensors on the MI355X, CuPy-style arrays): same shapes, dtypes and in-place semantics, no copy -- the pointers go straight to
// the device entry points (irrl_env_step & co.), stream-ordered on torch's current stream when the object is a torch tensor (else on
// the interface's `stream` entry, else the null stream).  A caller that keeps the reference's compiled boundary gets the same
// zero-copy path as the ctypes class of flexible_robot.py.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "irrl_env.h"

namespace py = pybind11;
using FArr = py::array_t<float, py::array::c_style>;
using BArr = py::array_t<bool, py::array::c_style>;

namespace {

void check(int rc) {
  if (rc != 0) throw std::runtime_error(std::string("irrl_env: ") + irrl_last_error());
}
float *mat(FArr &a, py::ssize_t rows, py::ssize_t cols, const char *name) {
  if (a.ndim() != 2 || a.shape(0) != rows || a.shape(1) != cols)
    throw py::value_error(std::string(name) + " must have shape (" + std::to_string(rows) + ", " + std::to_string(cols) + ")");
  if (!a.writeable()) throw py::value_error(std::string(name) + " must be writeable");
  return a.mutable_data();
}
float *vec(FArr &a, py::ssize_t rows, const char *name) {
  if (!((a.ndim() == 1 && a.shape(0) == rows) || (a.ndim() == 2 && a.shape(0) == rows && a.shape(1) == 1)))
    throw py::value_error(std::string(name) + " must have shape (" + std::to_string(rows) + ",)");
  if (!a.writeable()) throw py::value_error(std::string(name) + " must be writeable");
  return a.mutable_data();
}
uint8_t *bvec(BArr &a, py::ssize_t rows, const char *name) {
  if (!((a.ndim() == 1 && a.shape(0) == rows) || (a.ndim() == 2 && a.shape(0) == rows && a.shape(1) == 1)))
    throw py::value_error(std::string(name) + " must have shape (" + std::to_string(rows) + ",)");
  if (!a.writeable()) throw py::value_error(std::string(name) + " must be writeable");
  return reinterpret_cast<uint8_t *>(a.mutable_data());   // numpy bool is one byte holding 0 / 1, like the engine's done flags
}

// ---- an argument of step / reset / observe / isTerminalState: a numpy array (host path) or a device buffer ----
bool is_device(const py::handle &o) { return py::hasattr(o, "__cuda_array_interface__"); }
// numpy with the reference's contract: no conversion, no copy (what `noconvert` array_t arguments do)
FArr host_f32(const py::object &o, const char *name) {
  if (!py::isinstance<py::array>(o)) throw py::type_error(std::string(name) + " must be a numpy float32 array or a device buffer (__cuda_array_interface__)");
  py::array a = py::reinterpret_borrow<py::array>(o);
  if (!a.dtype().is(py::dtype::of<float>()) || !(a.flags() & py::array::c_style))
    throw py::type_error(std::string(name) + " must be a C-contiguous float32 array (no implicit conversion: the array is filled in place)");
  return py::reinterpret_borrow<FArr>(o);
}
BArr host_bool(const py::object &o, const char *name) {
  if (!py::isinstance<py::array>(o)) throw py::type_error(std::string(name) + " must be a numpy bool array or a device buffer (__cuda_array_interface__)");
  py::array a = py::reinterpret_borrow<py::array>(o);
  if (!a.dtype().is(py::dtype::of<bool>()) || !(a.flags() & py::array::c_style))
    throw py::type_error(std::string(name) + " must be a C-contiguous bool array (no implicit conversion: the array is filled in place)");
  return py::reinterpret_borrow<BArr>(o);
}
// device buffer: pointer of a C-contiguous array of the given item kind ('f' 4 bytes, or 'b' / 'u' 1 byte) and shape [rows] / [rows, cols]
void *dev_ptr(const py::object &o, char kind, int itemsize, py::ssize_t rows, py::ssize_t cols, const char *name) {
  py::dict ifc = o.attr("__cuda_array_interface__").cast<py::dict>();
  const std::string typestr = ifc["typestr"].cast<std::string>();
  const bool kind_ok = typestr.size() >= 3 && (typestr[1] == kind || (kind == 'b' && typestr[1] == 'u')) && std::stoi(typestr.substr(2)) == itemsize &&
                       (itemsize == 1 || typestr[0] == '<' || typestr[0] == '=');
  if (!kind_ok) throw py::type_error(std::string(name) + ": device buffer of the wrong dtype (" + typestr + ")");
  py::tuple shape = ifc["shape"].cast<py::tuple>();
  std::vector<py::ssize_t> sh;
  for (auto d : shape) sh.push_back(d.cast<py::ssize_t>());
  const bool shape_ok = cols > 0 ? (sh.size() == 2 && sh[0] == rows && sh[1] == cols) : ((sh.size() == 1 && sh[0] == rows) || (sh.size() == 2 && sh[0] == rows && sh[1] == 1));
  if (!shape_ok) throw py::value_error(std::string(name) + ": device buffer of the wrong shape");
  if (ifc.contains("strides") && !ifc["strides"].is_none()) {
    py::tuple st = ifc["strides"].cast<py::tuple>();
    py::ssize_t expect = itemsize;
    for (int i = (int)sh.size() - 1; i >= 0; i--) {
      if (sh[i] > 1 && st[i].cast<py::ssize_t>() != expect) throw py::type_error(std::string(name) + ": device buffer must be C-contiguous");
      expect *= sh[i];
    }
  }
  py::tuple data = ifc["data"].cast<py::tuple>();
  return reinterpret_cast<void *>(data[0].cast<size_t>());
}
// the stream the caller's framework is issuing on: torch's current stream for torch tensors, else the interface's `stream`, else null
void *dev_stream(const py::object &o) {
  const std::string mod = py::str(py::type::of(o).attr("__module__"));
  if (mod.rfind("torch", 0) == 0) {
    py::object cs = py::module_::import("torch").attr("cuda").attr("current_stream")(o.attr("device"));
    return reinterpret_cast<void *>(cs.attr("cuda_stream").cast<size_t>());
  }
  py::dict ifc = o.attr("__cuda_array_interface__").cast<py::dict>();
  if (ifc.contains("stream") && !ifc["stream"].is_none()) {
    const long long v = ifc["stream"].cast<long long>();
    if (v > 2) return reinterpret_cast<void *>((size_t)v);     // 1 / 2: the legacy / per-thread default stream
  }
  return nullptr;
}

// VectorizedEnvironment<ENVIRONMENT> (VectorizedEnvironment.hpp:127-382) as the engine sees it: a handle
class VecEnv {
 public:
  VecEnv(const std::string &resource_dir, const std::string &cfg, int device) : h_(irrl_env_create(resource_dir.c_str(), cfg.c_str(), device)) {
    if (!h_) throw std::runtime_error(std::string("FlexibleGymEnv: ") + irrl_last_error());
    n_ = irrl_env_num_envs(h_);
  }
  ~VecEnv() { if (h_) irrl_env_destroy(h_); }
  VecEnv(const VecEnv &) = delete;
  VecEnv &operator=(const VecEnv &) = delete;

  void init() { check(irrl_env_init(h_)); }
  std::vector<std::string> extra_names() const {
    std::vector<std::string> out;
    for (int j = 0; j < irrl_env_extra_dim(h_); j++) out.emplace_back(irrl_env_extra_name(h_, j));
    return out;
  }
  void reset(py::object ob) {
    if (is_device(ob)) { check(irrl_env_set_stream(h_, dev_stream(ob))); check(irrl_env_reset(h_, (float *)dev_ptr(ob, 'f', 4, n_, 35, "ob"))); return; }
    FArr a = host_f32(ob, "ob");
    check(irrl_env_reset_host(h_, mat(a, n_, 35, "ob")));
  }
  void observe(py::object ob) {
    if (is_device(ob)) { check(irrl_env_set_stream(h_, dev_stream(ob))); check(irrl_env_observe(h_, (float *)dev_ptr(ob, 'f', 4, n_, 35, "ob"))); return; }
    FArr a = host_f32(ob, "ob");
    check(irrl_env_observe_host(h_, mat(a, n_, 35, "ob")));
  }
  void step(py::object action, py::object ob, py::object reward, py::object done, py::object extra) {
    const int E = irrl_env_extra_dim(h_);
    if (is_device(action)) {
      // all five on the device (a mixed call would need a copy somewhere: refused like a wrong dtype)
      if (!(is_device(ob) && is_device(reward) && is_device(done) && is_device(extra)))
        throw py::type_error("step: action is a device buffer, so ob / reward / done / extraInfo must be device buffers too");
      check(irrl_env_set_stream(h_, dev_stream(action)));
      check(irrl_env_step(h_, (const float *)dev_ptr(action, 'f', 4, n_, 12, "action"), (float *)dev_ptr(ob, 'f', 4, n_, 35, "ob"),
                          (float *)dev_ptr(reward, 'f', 4, n_, 0, "reward"), (uint8_t *)dev_ptr(done, 'b', 1, n_, 0, "done"),
                          (float *)dev_ptr(extra, 'f', 4, n_, E, "extraInfo")));
      return;
    }
    FArr a = host_f32(action, "action"), o = host_f32(ob, "ob"), r = host_f32(reward, "reward"), x = host_f32(extra, "extraInfo");
    BArr d = host_bool(done, "done");
    check(irrl_env_step_host(h_, mat(a, n_, 12, "action"), mat(o, n_, 35, "ob"), vec(r, n_, "reward"), bvec(d, n_, "done"), mat(x, n_, E, "extraInfo")));
  }
  void test_step(FArr &action, FArr &ob, FArr &reward, BArr &done, FArr &extra) {
    check(irrl_env_test_step_host(h_, mat(action, n_, 12, "action"), mat(ob, n_, 35, "ob"), vec(reward, n_, "reward"), bvec(done, n_, "done"),
                                  mat(extra, n_, irrl_env_extra_dim(h_), "extraInfo")));
  }
  void set_seed(int seed) { check(irrl_env_set_seed(h_, seed)); }
  void close() { check(irrl_env_close(h_)); }
  void is_terminal(py::object done) {
    if (is_device(done)) { check(irrl_env_set_stream(h_, dev_stream(done))); check(irrl_env_is_terminal(h_, (uint8_t *)dev_ptr(done, 'b', 1, n_, 0, "done"))); return; }
    BArr d = host_bool(done, "done");
    check(irrl_env_is_terminal_host(h_, bvec(d, n_, "done")));
  }
  void set_sim_dt(double dt) { check(irrl_env_set_simulation_dt(h_, dt)); }
  void set_control_dt(double dt) { check(irrl_env_set_control_dt(h_, dt)); }
  int ob_dim() const { return irrl_env_ob_dim(h_); }
  int action_dim() const { return irrl_env_action_dim(h_); }
  int extra_dim() const { return irrl_env_extra_dim(h_); }
  int num_envs() const { return n_; }
  void curriculum_update() { check(irrl_env_curriculum_update(h_)); }
  void origin_state(FArr &out) { check(irrl_env_origin_state_host(h_, mat(out, n_, 41, "out"))); }
  // VectorizedEnvironment.hpp:223-226: ReferenceState dispatches to OriginState (the reference's bug, reproduced): the caller's
  // [N,24] array receives the first 24 origin-state entries of every env
  void reference_state(FArr &out) {
    float *dst = mat(out, n_, 24, "out");
    std::vector<float> tmp((size_t)n_ * 41);
    check(irrl_env_origin_state_host(h_, tmp.data()));
    for (int e = 0; e < n_; e++)
      for (int k = 0; k < 24; k++) dst[(size_t)e * 24 + k] = tmp[(size_t)e * 41 + k];
  }
  void joint_effort(FArr &out) { check(irrl_env_joint_effort_host(h_, mat(out, n_, 12, "out"))); }
  void generalized_force(FArr &out) { check(irrl_env_generalized_force_host(h_, mat(out, n_, 18, "out"))); }
  void inverse_mass_matrix(FArr &out) { check(irrl_env_inverse_mass_matrix_host(h_, mat(out, n_, 324, "out"))); }
  void nonlinear(FArr &out) { check(irrl_env_nonlinear_host(h_, mat(out, n_, 18, "out"))); }
  void set_contact_coeff(FArr &in) { check(irrl_env_set_contact_coeff_host(h_, mat(in, n_, 3, "contact_coeff"))); }
  void sphere_info(FArr &out) { check(irrl_env_sphere_info_host(h_, mat(out, n_, 4, "out"))); }
  // build-defined extras shared with the Python class of the same name (tests, checkpoints of the env state)
  py::array_t<double> get_state() {
    py::array_t<double> out({(py::ssize_t)n_, (py::ssize_t)IRRL_STATE_DIM});
    check(irrl_env_get_state_host(h_, out.mutable_data()));
    return out;
  }
  void set_state(py::array_t<double, py::array::c_style> &st) {
    if (st.ndim() != 2 || st.shape(0) != n_ || st.shape(1) != IRRL_STATE_DIM) throw py::value_error("state must have shape (N, IRRL_STATE_DIM)");
    check(irrl_env_set_state_host(h_, st.data()));
  }
  int lanes_per_robot() const { return irrl_env_lanes_per_robot(h_); }
  int waves_per_simd() const { return irrl_env_waves_per_simd(h_); }
  size_t handle() const { return reinterpret_cast<size_t>(h_); }

 private:
  irrl_env *h_;
  int n_ = 0;
};

}  // namespace

PYBIND11_MODULE(_flexible_robot, m) {
  m.doc() = "MI355X engine behind the reference's pybind11 boundary (flex_gym/env/raisim_gym.cpp:14-46)";
  m.def("engine_version", []() { return std::string(irrl_version()); });
  auto nc = [](const char *n) { return py::arg(n).noconvert(); };
  py::class_<VecEnv>(m, "FlexibleGymEnv")
      .def(py::init<std::string, std::string, int>(), py::arg("resourceDir"), py::arg("cfg"), py::arg("device") = 0)   // raisim_gym.cpp:16
      .def("init", &VecEnv::init)                                                                                         // :17
      .def("getExtraInfoNames", &VecEnv::extra_names)                                                                    // :18
      .def("reset", &VecEnv::reset, py::arg("ob"))                                                                       // :19 (numpy or device buffer)
      .def("observe", &VecEnv::observe, py::arg("ob"))                                                                   // :20
      .def("step", &VecEnv::step, py::arg("action"), py::arg("ob"), py::arg("reward"), py::arg("done"), py::arg("extraInfo"))   // :21, :23
      .def("setSeed", &VecEnv::set_seed)                                                                                 // :22
      .def("testStep", &VecEnv::test_step, nc("action"), nc("ob"), nc("reward"), nc("done"), nc("extraInfo"))            // :24
      .def("close", &VecEnv::close)                                                                                      // :25
      .def("isTerminalState", &VecEnv::is_terminal, py::arg("done"))                                                     // :26
      .def("setSimulationTimeStep", &VecEnv::set_sim_dt)                                                                 // :27
      .def("setControlTimeStep", &VecEnv::set_control_dt)                                                                // :28
      .def("getObDim", &VecEnv::ob_dim)                                                                                  // :29
      .def("getActionDim", &VecEnv::action_dim)                                                                          // :30
      .def("getExtraInfoDim", &VecEnv::extra_dim)                                                                        // :31
      .def("getNumOfEnvs", &VecEnv::num_envs)                                                                            // :32
      .def("startRecordingVideo", [](VecEnv &, const std::string &) {})                                                  // :33 headless engine: no-ops
      .def("stopRecordingVideo", [](VecEnv &) {})                                                                        // :34
      .def("showWindow", [](VecEnv &) {})                                                                                // :35
      .def("hideWindow", [](VecEnv &) {})                                                                                // :36
      .def("curriculumUpdate", &VecEnv::curriculum_update)                                                               // :37
      .def("OriginState", &VecEnv::origin_state, nc("out"))                                                              // :38
      .def("GetOriginStateDim", [](VecEnv &) { return 41; })                                                             // :39
      .def("ReferenceState", &VecEnv::reference_state, nc("out"))                                                        // :40
      .def("GetJointEffort", &VecEnv::joint_effort, nc("out"))                                                           // :41
      .def("GetGeneralizedForce", &VecEnv::generalized_force, nc("out"))                                                 // :42
      .def("GetInverseMassMatrix", &VecEnv::inverse_mass_matrix, nc("out"))                                              // :43
      .def("GetNonlinear", &VecEnv::nonlinear, nc("out"))                                                                // :44
      .def("SetContactCoefficient", &VecEnv::set_contact_coeff, nc("contact_coeff"))                                     // :45
      .def("GetSphereInfo", &VecEnv::sphere_info, nc("out"))                                                             // :46
      .def("get_state", &VecEnv::get_state)
      .def("set_state", &VecEnv::set_state, nc("state"))
      .def_property_readonly("lanes_per_robot", &VecEnv::lanes_per_robot)
      .def_property_readonly("waves_per_simd", &VecEnv::waves_per_simd)
      .def_property_readonly("handle", &VecEnv::handle);
}
