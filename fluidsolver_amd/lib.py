"""ctypes binding of libfluidsolver_hip.so -- the C ABI declared in include/fluidsolver.h.

There is no CPU fallback: if the HIP library is missing or a call fails, this module
raises.  (The CPU oracle lives under oracle/ and is test infrastructure only.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FS_PROBE_LIB=1 (measurement scripts only): the -DFS_PROBE build with the timing probes compiled in
LIB_PATH = os.path.join(HERE, "libfluidsolver_hip_probe.so" if os.environ.get("FS_PROBE_LIB") == "1" else "libfluidsolver_hip.so")

FS_OK = 0
COMM_ID_BYTES = 128
LINES_STATIC_MAX = 262144
ERRORS = {-1: "FS_ERR_INVALID", -2: "FS_ERR_HIP", -3: "FS_ERR_NOMEM", -4: "FS_ERR_HALO", -5: "FS_ERR_STATE"}

FIELD_VELOCITY, FIELD_DENSITY, FIELD_PRESSURE, FIELD_DIVERGENCE, FIELD_VORTICITY, FIELD_DISPLAY = range(6)
FIELD_PLANES = {FIELD_VELOCITY: 2, FIELD_DENSITY: 3, FIELD_PRESSURE: 1, FIELD_DIVERGENCE: 1,
                FIELD_VORTICITY: 1, FIELD_DISPLAY: 4}
DISPLAY_NONE, DISPLAY_DENSITY, DISPLAY_VELOCITY, DISPLAY_SPEED, DISPLAY_VORTICITY, DISPLAY_PARTICLES_ONLY = -1, 0, 1, 2, 3, 4
OPT_USE_GRAPH, OPT_BLOCK_DEPTH, OPT_WITH_PARTICLES, OPT_TILE_VARIANT, OPT_FUSE = 0, 1, 2, 3, 4
OPT_BLOCK_DEPTH_PRESSURE, OPT_BLOCK_DEPTH_DIFFUSE, OPT_STAGGER, OPT_STAGGER_MODE, OPT_SLAB_OVERLAP = 5, 6, 7, 8, 9
OPT_TILE_VARIANT_DIFFUSE, OPT_SLAB_PROFILE, OPT_SPLIT_PLANES, OPT_SPLIT_ROWS, OPT_BAND_DELAY = 10, 11, 12, 13, 14
OPT_SLAB_DEFER, OPT_SEGMENT_GRAPH_AFTER, OPT_SLAB_SERIALIZE, OPT_TRACER_SORT, OPT_SPLAT_IN_GRAPH = 15, 16, 17, 18, 19
(STAGE_VORTICITY, STAGE_APPLY_VORTICITY, STAGE_DIFFUSE_VELOCITY, STAGE_DIVERGENCE, STAGE_CLEAR_PRESSURE,
 STAGE_PRESSURE, STAGE_SUBTRACT_GRADIENT, STAGE_ADVECT_VELOCITY, STAGE_DIFFUSE_DENSITY, STAGE_ADVECT_DENSITY,
 STAGE_FADE, STAGE_RENDER, STAGE_PRESSURE_CLEARED, STAGE_PROJECT) = range(14)


class FluidSolverError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "FS_ERR"), code, msg))
        self.code = code


class Grid(C.Structure):
    """fs_grid"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("pitch", C.c_int), ("y0", C.c_int),
                ("rows", C.c_int), ("yb", C.c_int), ("ye", C.c_int)]


class Params(C.Structure):
    """fs_params (FluidSolver.swift:139-202, :227)"""
    _fields_ = [("viscosity", C.c_float), ("diffusion", C.c_float), ("deltaTime", C.c_float),
                ("vorticityStrength", C.c_float), ("fadeSpeed", C.c_float), ("solverIterations", C.c_int),
                ("displayMode", C.c_int), ("brightness", C.c_float)]


class ParticleParams(C.Structure):
    """fs_particle_params (ParticleSystem.swift:63-111)"""
    _fields_ = [("momentum", C.c_float), ("fluidForce", C.c_float), ("fluidDecayRate", C.c_float),
                ("linkToViscosity", C.c_int), ("viscosity", C.c_float), ("windowW", C.c_float),
                ("windowH", C.c_float)]


_f, _i, _p, _sz, _l = C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_long
_G = C.POINTER(Grid)

# every symbol include/fluidsolver.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "fs_pitch_for_width": (_i, [_i]),
    "fs_config_set": (_i, [_i, _i]),
    "fs_config_get": (_i, [_i]),
    "fs_k_addForces": (_i, [_G, _p, _p, _p, _p, _p] + [_f] * 9 + [_p]),
    "fs_k_diffuse": (_i, [_G, _p, _p, _i, _sz, _f, _f, _p]),
    "fs_k_diffuse_n": (_i, [_G, _p, _p, _i, _sz, _f, _f, _i, C.POINTER(_i), _p]),
    "fs_k_advect": (_i, [_G, _p, _p, _p, _p, _i, _sz, _f, _p, _p]),
    "fs_k_advect_fade": (_i, [_G, _p, _p, _p, _p, _i, _sz, _f, _f, _p, _p]),
    "fs_k_advect_both": (_i, [_G, _p, _p, _p, _p, _p, _p, _sz, _f, _i, _f, _p, _p]),
    "fs_k_divergence": (_i, [_G, _p, _p, _p, _p]),
    "fs_k_pressureSolve": (_i, [_G, _p, _p, _p, _p]),
    "fs_k_pressureSolve_n": (_i, [_G, _p, _p, _p, _i, _i, C.POINTER(_i), _p]),
    "fs_k_subtractGradient": (_i, [_G, _p, _p, _p, _p]),
    "fs_k_project_n": (_i, [_G, _p, _p, _p, _p, _p, _i, C.POINTER(_i), _p]),
    "fs_k_project_n_cfl": (_i, [_G, _p, _p, _p, _p, _p, _i, _p, C.POINTER(_i), _p]),
    "fs_k_pressureSolve_n_gradient": (_i, [_G, _p, _p, _p, _i, _i, _p, _p, C.POINTER(_i), _p]),
    "fs_k_calculateVorticity": (_i, [_G, _p, _p, _p, _p]),
    "fs_k_applyVorticity": (_i, [_G, _p, _p, _p, _f, _f, _p]),
    "fs_k_vorticity_apply": (_i, [_G, _p, _p, _p, _p, _f, _f, _p]),
    "fs_k_fadeDensity": (_i, [_G, _p, _i, _sz, _f, _p]),
    "fs_k_clear": (_i, [_G, _p, _i, _sz, _p]),
    "fs_k_renderDensity": (_i, [_G, _p, _sz, _i, _p, _sz, _f, _p]),
    "fs_k_renderVelocity": (_i, [_G, _p, _p, _p, _sz, _f, _p]),
    "fs_k_renderSpeed": (_i, [_G, _p, _p, _p, _sz, _f, _p]),
    "fs_k_particles_update": (_i, [_G, _p, _p, C.POINTER(ParticleParams), _l] + [_p] * 7 + [_p]),
    "fs_k_max_abs": (_i, [_G, _p, _p, _p]),
    "fs_k_subtractGradient_cfl": (_i, [_G, _p, _p, _p, _p, _p]),
    "fs_create": (_i, [_i, _i, _i, _l, C.POINTER(_p)]),
    "fs_destroy": (_i, [_p]),
    "fs_reset": (_i, [_p]),
    "fs_default_params": (_i, [C.POINTER(Params)]),
    "fs_set_params": (_i, [_p, C.POINTER(Params)]),
    "fs_get_params": (_i, [_p, C.POINTER(Params)]),
    "fs_set_option": (_i, [_p, _i, _i]),
    "fs_get_option": (_i, [_p, _i, C.POINTER(_i)]),
    "fs_add_force": (_i, [_p, _f, _f, _f, _f, C.POINTER(_f), _f, _f]),
    "fs_step": (_i, [_p]),
    "fs_sync": (_i, [_p]),
    "fs_read_field": (_i, [_p, _i, _p, _sz]),
    "fs_write_field": (_i, [_p, _i, _p, _sz]),
    "fs_slab_host_times": (_i, [_p, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "fs_slab_graph_stats": (_i, [_p, C.POINTER(C.c_long)]),
    "fs_slab_host_breakdown": (_i, [_p, C.POINTER(C.c_double)]),
    "fs_graph_stats": (_i, [_p, C.POINTER(C.c_long)]),
    "fs_field_device_ptr": (_i, [_p, _i, _i, C.POINTER(_p), C.POINTER(_i)]),
    "fs_stream": (_p, [_p]),
    "fs_timer_begin": (_i, [_p]),
    "fs_timer_end": (_i, [_p, C.POINTER(_f)]),
    "fs_stage": (_i, [_p, _i, _i]),
    "fs_particles_set": (_i, [_p, _l] + [_p] * 7),
    "fs_particles_get": (_i, [_p, _l] + [_p] * 7),
    "fs_particles_set_aos": (_i, [_p, _l, _p, _sz]),
    "fs_particles_get_aos": (_i, [_p, _l, _p, _sz]),
    "fs_default_particle_params": (_i, [C.POINTER(ParticleParams), _f, _f]),
    "fs_particles_update": (_i, [_p, C.POINTER(ParticleParams)]),
    "fs_particles_count": (_l, [_p]),
    "fs_particles_spawn": (_i, [_p, _f, _f, _l, _f, C.c_uint]),
    "fs_particles_reserve": (_i, [_p, _l]),
    "fs_particle_lines": (_i, [_p, _f, _f, _f, _p, _p]),
    "fs_particle_lines_static": (_i, [_p, _f, _f, _f, _p, _p]),
    "fs_k_particle_lines": (_i, [_l, _p, _p, _p, _p, _p, _f, _f, _f, _p, _p, _p]),
    "fs_set_particle_params": (_i, [_p, C.POINTER(ParticleParams)]),
    "fs_k_tracer_count": (_i, [_G, C.POINTER(ParticleParams), _i, _i, _l, _p, _p, _p, _p]),
    "fs_k_tracer_scatter": (_i, [_G, C.POINTER(ParticleParams), _i, _i, _l, _p, _p, _p, _p, _p]),
    "fs_comm_unique_id": (_i, [_p]),
    "fs_comm_create_rccl": (_i, [_i, _i, _i, _p, C.POINTER(_p)]),
    "fs_comm_create_local": (_i, [_i, C.POINTER(_p)]),
    "fs_comm_create_local_async": (_i, [_i, C.POINTER(_p)]),
    "fs_comm_create_loopback": (_i, [_i, _i, C.POINTER(_p)]),
    "fs_comm_destroy": (_i, [_p]),
    "fs_comm_selftest": (_i, [_p, _i]),
    "fs_create_slab": (_i, [_i, _i, _i, _p, _i, _i, C.POINTER(_p)]),
    "fs_slab_info": (_i, [_p, C.POINTER(_i)]),
    "fs_slab_stats": (_i, [_p, C.POINTER(_l)]),
    "fs_group_step": (_i, [C.POINTER(_p), _i]),
    "fs_slab_waits": (_i, [_p, C.POINTER(C.c_double), C.POINTER(_l)]),
    "fs_slab_plan_dump": (_i, [_i, _i, _i, _i, _i, _i, C.POINTER(Params), _i, C.POINTER(_i), C.c_char_p, _sz]),
    "fs_last_error": (C.c_char_p, [_p]),
    "fs_version": (C.c_char_p, []),
    "fs_selftest_divide": (_i, [_f, C.POINTER(C.c_ulonglong)]),
    "fs_harness_fill_uniform": (_i, [_p, _l, _f, _f, C.c_uint]),
}

_lib = None


def load():
    """Load the HIP library (raises if it has not been built: python -m fluidsolver_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s not found: build it with `python -m fluidsolver_amd.build` (there is no CPU fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error(ctx=None):
    msg = load().fs_last_error(ctx)
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, ctx=None):
    if rc != FS_OK:
        raise FluidSolverError(rc, last_error(ctx) or last_error(None))
    return rc


def pitch_for_width(w):
    return load().fs_pitch_for_width(int(w))


def full_grid(W, H, pitch=None):
    return Grid(W, H, pitch or pitch_for_width(W), 0, H, 0, H)
