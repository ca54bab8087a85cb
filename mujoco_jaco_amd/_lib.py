"""ctypes binding of libjaco_env.so (C ABI: include/jaco_env.h).

There is deliberately no fallback: if the HIP extension has not been built
(``python -c "import __graft_entry__ as g; g.build()"``) importing this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("JACO_ENV_LIB", "libjaco_env.so"))
ASSETS = os.path.join(_HERE, "assets")


class JacoConfig(ctypes.Structure):
    _fields_ = [("model_blob", ctypes.c_void_p), ("model_blob_size", ctypes.c_size_t), ("num_envs", ctypes.c_int),
                ("device", ctypes.c_int), ("frame_skip", ctypes.c_int), ("task", ctypes.c_int), ("seed", ctypes.c_uint64)]


# every symbol include/jaco_env.h declares: name -> (restype, argtypes)
_vp, _ci, _cd, _cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_char_p
_ip = ctypes.POINTER(ctypes.c_int)
SYMBOLS = {
    "jaco_create": (_ci, [ctypes.POINTER(JacoConfig), ctypes.POINTER(_vp)]),
    "jaco_destroy": (_ci, [_vp]),
    "jaco_last_error": (_cp, [_vp]),
    "jaco_dims": (_ci, [_vp, _ip, _ip, _ip, _ip, _ip, _ip]),
    "jaco_num_envs": (_ci, [_vp]),
    "jaco_set_state": (_ci, [_vp, _vp, _vp, _vp, _vp]),
    "jaco_get_state": (_ci, [_vp, _vp, _vp, _vp, _vp]),
    "jaco_reset_state": (_ci, [_vp, _vp]),
    "jaco_physics_step": (_ci, [_vp, _vp, _ci, _vp]),
    "jaco_get_sensordata": (_ci, [_vp, _vp, _vp]),
    "jaco_get_flags": (_ci, [_vp, _vp, _vp]),
    "jaco_clear_flags": (_ci, [_vp, _vp]),
    "jaco_get_stats": (_ci, [_vp, _vp, _vp]),
    "jaco_set_option": (_ci, [_vp, _cp, _cd]),
    "jaco_reset": (_ci, [_vp, _vp, _vp, _vp]),
    "jaco_placing_hold": (_ci, [_vp, _vp, _ci, _vp]),
    "jaco_grasping_prereach": (_ci, [_vp, _vp, _ci, _vp, _vp]),
    "jaco_step": (_ci, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "jaco_forward": (_ci, [_vp, _vp, _vp]),
    "jaco_take_action": (_ci, [_vp, _vp, _vp]),
    "jaco_terminal_inspection": (_ci, [_vp, _vp, _vp, _vp]),
    "jaco_set_noise": (_ci, [_vp, _vp]),
    "jaco_set_subgoal": (_ci, [_vp, _vp]),
    "jaco_set_init_buffer": (_ci, [_vp, _vp, _ci, _ci, _vp]),
    "jaco_get_task_state": (_ci, [_vp, _vp, _vp]),
    "jaco_set_task_state": (_ci, [_vp, _vp, _vp]),
    "jaco_task_row_floats": (_ci, []),
    "jaco_get_markers": (_ci, [_vp, _vp, _vp]),
    "jaco_get_last_terminal": (_ci, [_vp, _vp, _vp]),
    "jaco_get_terminal_obs": (_ci, [_vp, _vp, _vp]),
    "jaco_set_markers": (_ci, [_vp, _vp, _vp]),
    "jaco_set_frame_skip": (_ci, [_vp, _ci]),
    "jaco_physics_step_debug": (_ci, [_vp, _vp, _ci, _ci, ctypes.POINTER(ctypes.c_float), _ci]),
    "jaco_debug_dump_floats": (_ci, []),
    "jaco_launch_count": (ctypes.c_longlong, [_vp]),
    "jaco_debug_queue_words": (_ci, [_vp, _ip, _ci]),
    "jaco_kernel_time_ms": (_ci, [_vp, ctypes.POINTER(_cd), _ip]),
    "jaco_enable_timing": (_ci, [_vp, _ci]),
    "jaco_step_time_ms": (_ci, [_vp, ctypes.POINTER(_cd)]),
    "jaco_stage_profile": (_ci, [_vp, ctypes.POINTER(ctypes.c_uint64), _ci]),
}

_libs = {}


def load(variant=""):
    """variant "": the default layout (11 fused bodies, 21 dofs in blocks 9 + 6 + 6); "_d12": the build for jaco2_torque.xml and jaco2_curtain_torque_sensor.xml
    (12 hinge dofs in one tree + at most one free object); "_d30": the build for jaco2_dual_torque.xml."""
    if variant not in _libs:
        # torch first: its wheel bundles its own libamdhip64, and the process must run on ONE HIP runtime -- loaded the other way round
        # (this library's /opt/rocm runtime, then torch's) the second runtime finds no device ("no usable HIP device" in jaco_create)
        import torch  # noqa: F401
        path = LIB_PATH if not variant else os.path.join(_HERE, "libjaco_env%s.so" % variant)
        if not os.path.exists(path):
            raise ImportError(
                "mujoco_jaco_amd: %s is missing. Build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback." % path)
        L = ctypes.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _libs[variant] = L
    return _libs[variant]


def variant_for(blob_bytes):
    """Which build of the library steps this model: the loader of each build rejects models outside its compiled layout."""
    from .modelc import blob as blobmod
    M = blobmod.loads(blob_bytes)
    nv, nb = int(M["nv"][0]), int(M["f_nbody"][0])
    if nv > 21 or nb > 13:
        return "_d30"   # jaco2_dual_torque.xml: two arms + two objects (30 dofs, 20 fused bodies); sim-interface (ctrl) level only
    # the 12-hinge arm (proximal + distal finger joints in the arm's tree), alone (jaco2_torque.xml) or with one free object
    # (jaco2_curtain_torque_sensor.xml: 13 fused bodies, 18 dofs); sim-interface level
    return "_d12" if (nv, nb) in ((12, 12), (18, 13)) else ""


def model_path(name):
    return os.path.join(ASSETS, name + ".jacomdl")
