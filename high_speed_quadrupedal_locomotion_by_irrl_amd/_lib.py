"""ctypes binding of libirrl_env.so (include/irrl_env.h).  Fails loudly: there is no fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IRRL_ENV_LIB") or os.path.join(_HERE, "libirrl_env.so")  # override: A/B builds of the same ABI
_lib = None

fp = C.POINTER(C.c_float)
dp = C.POINTER(C.c_double)
u8 = C.POINTER(C.c_uint8)
vp = C.c_void_p

# name -> (restype, argtypes); exactly the symbols declared in include/irrl_env.h
SIGNATURES = {
    "irrl_last_error": (C.c_char_p, []),
    "irrl_version": (C.c_char_p, []),
    "irrl_env_create": (vp, [C.c_char_p, C.c_char_p, C.c_int]),
    "irrl_env_destroy": (None, [vp]),
    "irrl_env_init": (C.c_int, [vp]),
    "irrl_env_set_stream": (C.c_int, [vp, vp]),
    "irrl_env_num_envs": (C.c_int, [vp]),
    "irrl_env_lanes_per_robot": (C.c_int, [vp]),
    "irrl_env_waves_per_simd": (C.c_int, [vp]),
    "irrl_env_ob_dim": (C.c_int, [vp]),
    "irrl_env_action_dim": (C.c_int, [vp]),
    "irrl_env_extra_dim": (C.c_int, [vp]),
    "irrl_env_extra_name": (C.c_char_p, [vp, C.c_int]),
    "irrl_env_step": (C.c_int, [vp, vp, vp, vp, vp, vp]),
    "irrl_env_step_host": (C.c_int, [vp, fp, fp, fp, u8, fp]),
    "irrl_env_step_rows": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "irrl_env_step_rows_persistent": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "irrl_env_step_rows_out": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "irrl_env_step_rows_persistent_out": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "irrl_env_test_step_host": (C.c_int, [vp, fp, fp, fp, u8, fp]),
    "irrl_env_reset": (C.c_int, [vp, vp]),
    "irrl_env_reset_host": (C.c_int, [vp, fp]),
    "irrl_env_observe": (C.c_int, [vp, vp]),
    "irrl_env_observe_host": (C.c_int, [vp, fp]),
    "irrl_env_is_terminal": (C.c_int, [vp, vp]),
    "irrl_env_is_terminal_host": (C.c_int, [vp, u8]),
    "irrl_env_set_seed": (C.c_int, [vp, C.c_int]),
    "irrl_env_set_simulation_dt": (C.c_int, [vp, C.c_double]),
    "irrl_env_set_control_dt": (C.c_int, [vp, C.c_double]),
    "irrl_env_close": (C.c_int, [vp]),
    "irrl_env_curriculum_update": (C.c_int, [vp]),
    "irrl_env_origin_state_host": (C.c_int, [vp, fp]),
    "irrl_env_reference_state_host": (C.c_int, [vp, fp]),
    "irrl_env_joint_effort_host": (C.c_int, [vp, fp]),
    "irrl_env_generalized_force_host": (C.c_int, [vp, fp]),
    "irrl_env_inverse_mass_matrix_host": (C.c_int, [vp, fp]),
    "irrl_env_nonlinear_host": (C.c_int, [vp, fp]),
    "irrl_env_set_contact_coeff_host": (C.c_int, [vp, fp]),
    "irrl_env_sphere_info_host": (C.c_int, [vp, fp]),
    "irrl_env_get_state_host": (C.c_int, [vp, dp]),
    "irrl_env_set_state_host": (C.c_int, [vp, dp]),
    "irrl_env_heightfield_host": (C.c_int, [vp, fp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "irrl_env_set_ref_host": (C.c_int, [vp, fp, C.c_int, C.c_int]),
    "irrl_env_cfg_value": (C.c_double, [vp, C.c_char_p]),
    "irrl_env_counters_host": (C.c_int, [vp, C.POINTER(C.c_ulonglong)]),
    "irrl_env_counters": (C.c_int, [vp, vp]),
    "irrl_env_snapshot": (C.c_int, [vp]),
    "irrl_env_restore": (C.c_int, [vp]),
    "irrl_bench_actions": (C.c_int, [C.c_uint, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_float, vp, vp]),
    "irrl_gae": (C.c_int, [C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]),
    "irrl_calib_copy_dword": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "irrl_ppo_loss": (C.c_int, [C.c_size_t, C.c_int] + [vp] * 8 + [C.c_float, C.c_float] + [vp] * 3 + [C.c_int, vp]),
    "irrl_ppo_heads_loss": (C.c_int, [C.c_size_t, C.c_int, C.c_int] + [vp] * 12 + [C.c_float, C.c_float] + [vp] * 5 + [C.c_int, vp]),
    "irrl_mlp_ppo_grads": (C.c_int, [C.c_int, C.c_size_t, vp, C.c_int, C.c_int, C.c_int] + [vp] * 13 + [C.c_float, C.c_float, vp, C.c_int, vp]),
    "irrl_mlp_ppo_grads_bf16": (C.c_int, [C.c_int, C.c_size_t, vp, C.c_int, C.c_int, C.c_int] + [vp] * 13 + [C.c_float, C.c_float, vp, C.c_int, vp]),
    "irrl_mlp_ppo_partial_len": (C.c_int, []),
    "irrl_adv_moments": (C.c_int, [C.c_size_t, vp, vp, vp, vp, C.c_int, vp, vp, vp]),
    "irrl_mlp_pack_records": (C.c_int, [C.c_size_t, vp, vp, vp, vp, vp, vp, vp]),
    "irrl_mlp_record_floats": (C.c_int, []),
    "irrl_mlp_ppo_grads_bf16_rec": (C.c_int, [C.c_int, C.c_size_t, vp, vp] + [vp] * 8 + [C.c_float, C.c_float, vp, C.c_int, vp]),
    "irrl_adv_moments_rec": (C.c_int, [C.c_size_t, vp, vp, vp, C.c_int, vp, vp, vp]),
    "irrl_sum_rows": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "irrl_lstm_seq_forward_bf16": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [vp] * 11),
    "irrl_lstm_seq_backward_bf16": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [vp] * 15),
    "irrl_clip_adam": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_longlong, vp, vp]),
    "irrl_sum_rows_scatter": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "irrl_random_permutation": (C.c_int, [C.c_longlong, C.c_uint, C.c_uint, vp, vp]),
    "irrl_lstm_seq_forward": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "irrl_lstm_seq_forward_x": (C.c_int, [C.c_int] * 4 + [vp] * 11),
    "irrl_lstm_seq_backward": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "irrl_lstm_seq_backward_x": (C.c_int, [C.c_int] * 4 + [vp] * 14),
    "irrl_mlp_policy_step": (C.c_int, [C.c_int] * 4 + [vp] * 9 + [C.c_int, C.c_uint, C.c_longlong, vp, C.c_int] + [vp] * 4 + [C.c_longlong] + [vp] * 8),
    "irrl_lstm_policy_step": (C.c_int, [C.c_int] * 4 + [vp] * 11 + [C.c_int, C.c_uint, C.c_longlong, vp, C.c_int] + [vp] * 4 + [C.c_longlong] + [vp] * 8),
    "irrl_mlp_rollout": (C.c_int, [vp] + [C.c_int] * 4 + [vp] * 9 + [C.c_int, C.c_uint, C.c_longlong, vp, C.c_int] + [vp] * 4 + [C.c_longlong] + [vp] * 8 + [C.c_int, vp]),
    "irrl_lstm_rollout_supports": (C.c_int, [vp, C.c_int, C.c_int]),
    "irrl_lstm_rollout": (C.c_int, [vp] + [C.c_int] * 4 + [vp] * 11 + [C.c_int, C.c_uint, C.c_longlong, vp, C.c_int] + [vp] * 4 + [C.c_longlong] + [vp] * 8 + [C.c_int, vp]),
}


def load():
    """Load the HIP library.  Raises if it has not been built -- build it with
    ``python -m high_speed_quadrupedal_locomotion_by_irrl_amd.build`` (or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libirrl_env.so is missing: build the gfx950 extension first "
                           "(python -m high_speed_quadrupedal_locomotion_by_irrl_amd.build). "
                           "There is no CPU or PyTorch fallback for the env kernels.")
    # PyTorch-ROCm is the device-memory / stream plumbing of this engine and ships its own HIP runtime
    # (libamdhip64).  Import it BEFORE dlopen-ing the kernels so that the process holds exactly one HIP runtime
    # (two copies in one process cannot both open the device: the second one sees no GPU).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def version():
    """"gfx950;irrl-env rN;irrl-src-hash:<hex>" of the loaded library (the hash covers every source under csrc/ + the flags)"""
    return load().irrl_version().decode()


def last_error():
    return load().irrl_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise RuntimeError("irrl_env: " + last_error())
