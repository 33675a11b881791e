"""ctypes binding of castro_amd/libcastro_hydro_amd.so (the C ABI in include/castro_hydro_amd.h).

This module only loads the shared library and declares signatures.  There is NO
CPU fallback: if the library is missing, or no HIP device is present when a
context is created, it raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Two builds of the same sources and the same ABI (castro_amd/csrc/Makefile):
#   exact     libcastro_hydro_amd.so            -ffp-contract=off, IEEE division / sqrt: bit-identical to the CPU restatement the tests check against
#   contract  libcastro_hydro_amd_contract.so   FMA contraction, reciprocal division, rsq-based sqrt: rtol 1e-10, faster
# CASTRO_AMD_NUMERICS picks the process default (exact); HipHydro(device, numerics=...) picks per context; both libraries can
# be loaded side by side.  CASTRO_AMD_LIB overrides the path for every mode (A/B builds).
NUMERICS_MODES = ("exact", "contract")
DEFAULT_NUMERICS = os.environ.get("CASTRO_AMD_NUMERICS", "exact")
_LIB_FILES = {"exact": "libcastro_hydro_amd.so", "contract": "libcastro_hydro_amd_contract.so"}


def lib_path(numerics=None):
    mode = numerics or DEFAULT_NUMERICS
    if mode not in NUMERICS_MODES:
        raise ValueError("CASTRO_AMD_NUMERICS / numerics must be one of %s, not %r" % (NUMERICS_MODES, mode))
    if os.environ.get("CASTRO_AMD_LIB"):             # A/B builds (tools/ab_variants.sh): one library whatever the mode asked for
        return os.environ["CASTRO_AMD_LIB"]
    return os.path.join(_HERE, _LIB_FILES[mode])


LIB_PATH = lib_path()

NUM_STATE, NGDNV, NUM_GROW = 8, 4, 4
URHO, UMX, UMY, UMZ, UEDEN, UEINT, UTEMP, UFS = range(8)

OK, ERR_ARG, ERR_UNSUPPORTED, ERR_NOMEM, ERR_HIP = 0, -1, -2, -3, -4
UPDATE_ADD, UPDATE_FROM_SBORDER, FLUX_ASSIGN, STAGE_A, STAGE_B = 0, 1, 2, 4, 8
STAGE_VALID, STAGE_REST, BC_FILL = 16, 32, 64
# CASTRO_AMD_DER_* ids, in the order the reference registers the fields (Castro_setup.cpp:756-960)
DERIVE_IDS = {"pressure": 0, "kineng": 1, "soundspeed": 2, "Gamma_1": 3, "MachNumber": 4, "magvort": 5, "divu": 6,
              "eint_E": 7, "eint_e": 8, "logden": 9, "X(X)": 10, "abar": 11, "x_velocity": 12, "y_velocity": 13,
              "z_velocity": 14, "magvel": 15, "radvel": 16, "magmom": 17, "StateErr_0": 18, "StateErr_1": 19, "StateErr_2": 20,
              "circvel": 21, "angular_momentum_x": 22, "angular_momentum_y": 23, "angular_momentum_z": 24}

# every symbol include/castro_hydro_amd.h declares (checked by tests/test_capi_symbols.py)
EXPORTED_SYMBOLS = (
    "castro_amd_default_params", "castro_amd_finalize_params",
    "castro_amd_ctx_create", "castro_amd_ctx_destroy", "castro_amd_ctx_reserve",
    "castro_amd_ctx_scratch_bytes", "castro_amd_ctx_status", "castro_amd_ctx_poison_scratch", "castro_amd_ctx_set_source_corrector",
    "castro_amd_ctu_hydro_fab", "castro_amd_ctu_hydro_clean_fab", "castro_amd_ctu_hydro_fab_ex", "castro_amd_step_control", "castro_amd_ctu_hydro_mf", "castro_amd_fab_ops_p",
    "castro_amd_derive_fab",
    "castro_amd_error_tag_fab", "castro_amd_cc_interp_fab", "castro_amd_lincomb_fab", "castro_amd_avgdown_fab", "castro_amd_fluxreg_crse_init_fab",
    "castro_amd_fluxreg_fine_add_fab", "castro_amd_reflux_fab",
    "castro_amd_old_rotation_source_fab", "castro_amd_new_rotation_source_fab",
    "castro_amd_old_gravity_source_fab", "castro_amd_new_gravity_source_fab", "castro_amd_saxpy_fab", "castro_amd_clean_state_fab", "castro_amd_clean_state_reduce_fab",
    "castro_amd_estdt_fab", "castro_amd_sources_mf", "castro_amd_clean_state_reduce_mf", "castro_amd_estdt_mf",
    "castro_amd_bc_fill_fab", "castro_amd_copy_fab", "castro_amd_pack_fab", "castro_amd_unpack_fab",
    "castro_amd_pack_regions_fab", "castro_amd_unpack_regions_fab", "castro_amd_fillpatch_shell_fab", "castro_amd_apply_source_fab", "castro_amd_fab_ops",
    "castro_amd_sedov_init_fab", "castro_amd_sod_init_fab", "castro_amd_version", "castro_amd_abi_version", "castro_amd_numerics",
    "castro_amd_comm_version", "castro_amd_comm_unique_id", "castro_amd_comm_create", "castro_amd_comm_adopt", "castro_amd_comm_rank",
    "castro_amd_comm_size", "castro_amd_comm_destroy", "castro_amd_halo_plan_create", "castro_amd_halo_plan_destroy",
    "castro_amd_halo_plan_bytes_sent", "castro_amd_fill_boundary", "castro_amd_fill_boundary_ex", "castro_amd_halo_plan_wait_packed",
    "castro_amd_halo_group_create", "castro_amd_halo_group_destroy", "castro_amd_halo_group_bytes_sent", "castro_amd_fill_boundary_group",
    "castro_amd_fill_boundary_group_ex", "castro_amd_halo_group_wait_packed",
    "castro_amd_allreduce_min",
    "castro_amd_ctx_profile", "castro_amd_ctx_profile_count", "castro_amd_ctx_profile_get",
    "castro_amd_ctx_profile_reset",
    "castro_amd_berger_rigoutsos",
    "castro_amd_cmpflx_points", "castro_amd_ppm_points", "castro_amd_flatten_points", "castro_amd_trans_points",
)


class Fab(C.Structure):
    """castro_amd_fab: FArrayBox descriptor (pointer, lo, hi, ncomp)."""
    _fields_ = [("p", C.c_void_p), ("lo", C.c_int * 3), ("hi", C.c_int * 3), ("ncomp", C.c_int)]


# castro_amd_step_control's ctl vector (include/castro_hydro_amd.h)
CTL_DT, CTL_TIME, CTL_NSTEP, CTL_STATUS, CTL_RHOMIN, CTL_EST, CTL_DTHYDRO, CTL_HIST, CTL_NHIST, CTL_SIZE = 0, 1, 2, 3, 4, 5, 6, 8, 56, 64


class HaloRegion(C.Structure):
    """castro_amd_halo_region"""
    _fields_ = [("peer", C.c_int), ("sbox_lo", C.c_int * 3), ("sbox_hi", C.c_int * 3), ("rbox_lo", C.c_int * 3),
                ("rbox_hi", C.c_int * 3), ("send_tag", C.c_int), ("recv_tag", C.c_int)]


class HaloMsg(C.Structure):
    """castro_amd_halo_msg: one send or one receive of a many-box exchange (castro_amd_halo_group_create)"""
    _fields_ = [("fab", C.c_int), ("peer", C.c_int), ("lo", C.c_int * 3), ("hi", C.c_int * 3), ("tag", C.c_int)]


class HydroOpts(C.Structure):
    """castro_amd_hydro_opts"""
    _fields_ = [("flags", C.c_int), ("clean_ntimes", C.c_int), ("d_out", C.c_void_p), ("sborder_clean_ntimes", C.c_int),
                ("d_dt", C.c_void_p)]


class HydroBox(C.Structure):
    """castro_amd_hydro_box: one box of a castro_amd_ctu_hydro_mf call"""
    _fields_ = [("bxlo", C.c_int * 3), ("bxhi", C.c_int * 3), ("vbxlo", C.c_int * 3), ("vbxhi", C.c_int * 3),
                ("Sborder", Fab), ("src", Fab), ("S_new", Fab), ("flux", Fab * 3), ("mass_flux", Fab * 3), ("qe", Fab * 3)]


class SourceBox(C.Structure):
    """castro_amd_source_box: one box of a castro_amd_sources_mf call"""
    _fields_ = [("lo", C.c_int * 3), ("hi", C.c_int * 3), ("S_old", Fab), ("S_new", Fab), ("source", Fab), ("mass_flux", Fab * 3)]


class StateBox(C.Structure):
    """castro_amd_state_box: one box of castro_amd_clean_state_reduce_mf / castro_amd_estdt_mf"""
    _fields_ = [("lo", C.c_int * 3), ("hi", C.c_int * 3), ("state", Fab)]


class Rotation(C.Structure):
    """castro_amd_rotation"""
    _fields_ = [("omega", C.c_double * 3), ("center", C.c_double * 3), ("include_centrifugal", C.c_int),
                ("include_coriolis", C.c_int), ("rot_source_type", C.c_int), ("implicit_rotation_update", C.c_int)]


def make_rotation(rotational_period, rot_axis=3, center=(0.5, 0.5, 0.5), include_centrifugal=1, include_coriolis=1,
                  rot_source_type=4, implicit_rotation_update=1):
    """castro.rotational_period / castro.rot_axis -> omega (Source/rotation/Rotation.H:10-22)"""
    import math
    R = Rotation()
    for d in range(3):
        R.omega[d] = 0.0
        R.center[d] = center[d]
    if rotational_period > 0.0:
        R.omega[rot_axis - 1] = 2.0 * math.pi / rotational_period
    R.include_centrifugal, R.include_coriolis = include_centrifugal, include_coriolis
    R.rot_source_type, R.implicit_rotation_update = rot_source_type, implicit_rotation_update
    return R


class Geom(C.Structure):
    _fields_ = [("dx", C.c_double * 3), ("problo", C.c_double * 3), ("probhi", C.c_double * 3),
                ("domlo", C.c_int * 3), ("domhi", C.c_int * 3),
                ("lo_bc", C.c_int * 3), ("hi_bc", C.c_int * 3), ("coord", C.c_int)]


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "ppm_type", "riemann_solver", "use_flattening", "hybrid_riemann", "first_order_hydro",
        "cg_maxiter", "cg_blend", "transverse_use_eos", "transverse_reset_density",
        "transverse_reset_rhoe", "ppm_temp_fix", "plm_iorder", "plm_limiter", "use_pslope")] + \
        [(n, C.c_double) for n in (
            "difmag", "small_dens", "small_temp", "small_pres", "small_ener", "cg_tol",
            "dual_energy_eta1", "dual_energy_eta2", "cfl", "init_shrink", "change_max",
            "eos_gamma", "small_x", "T_guess", "abar", "pslope_cutoff_density")] + \
        [("limit_fluxes_on_small_dens", C.c_int), ("limit_fluxes_on_large_vel", C.c_int), ("speed_limit", C.c_double),
         ("source_term_predictor", C.c_int)]


class FabOp(C.Structure):
    """castro_amd_fab_op (include/castro_hydro_amd.h)"""
    _fields_ = [("kind", C.c_int), ("dir", C.c_int), ("ncomp", C.c_int), ("lo", C.c_int * 3), ("hi", C.c_int * 3),
                ("side", C.c_int), ("a", C.c_double), ("b", C.c_double), ("dst", Fab), ("src", Fab), ("src2", Fab)]


OP_COPY, OP_LINCOMB, OP_FLUXREG_CRSE_INIT, OP_FLUXREG_FINE_ADD, OP_REFLUX, OP_CLEAN, OP_INTERP_CLEAN, OP_AVGDOWN, OP_INTERP = 0, 1, 2, 3, 4, 5, 6, 7, 8

_libs = {}
ABI_VERSION = 5          # CASTRO_AMD_ABI_VERSION of include/castro_hydro_amd.h this binding was written against


def numerics_of(L):
    """"exact" / "contract" as the loaded library reports it (castro_amd_numerics)"""
    if not hasattr(L, "castro_amd_numerics"):
        return "exact"                                     # an A/B build of a revision before the second mode existed
    return L.castro_amd_numerics().decode()


def load(numerics=None):
    """Load the shared library of a numerics mode (raises if it has not been built)."""
    mode = numerics or DEFAULT_NUMERICS
    if mode in _libs:
        return _libs[mode]
    path = lib_path(mode)
    if not os.path.exists(path):
        raise ImportError(
            "castro_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C castro_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    L = C.CDLL(path)
    ab_build = bool(os.environ.get("CASTRO_AMD_LIB"))      # an A/B build, possibly of an older revision (tools/ab_variants.sh)
    if hasattr(L, "castro_amd_abi_version") or not ab_build:
        L.castro_amd_abi_version.restype = C.c_int
        if L.castro_amd_abi_version() != ABI_VERSION:
            raise ImportError("castro_amd: %s has ABI version %d, this binding expects %d: rebuild it"
                              % (path, L.castro_amd_abi_version(), ABI_VERSION))
        L.castro_amd_numerics.restype = C.c_char_p
        if not ab_build and numerics_of(L) != mode:
            raise ImportError("castro_amd: %s is a %r build, expected %r" % (path, numerics_of(L), mode))
    I3 = C.POINTER(C.c_int)
    PF = C.POINTER(Fab)
    L.castro_amd_version.restype = C.c_char_p
    L.castro_amd_default_params.argtypes = [C.POINTER(Params)]
    L.castro_amd_finalize_params.argtypes = [C.POINTER(Params)]
    L.castro_amd_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.castro_amd_ctx_destroy.argtypes = [C.c_void_p]
    L.castro_amd_ctx_reserve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.castro_amd_ctx_scratch_bytes.restype = C.c_longlong
    L.castro_amd_ctx_scratch_bytes.argtypes = [C.c_void_p]
    L.castro_amd_ctx_status.argtypes = [C.c_void_p, C.c_void_p]
    L.castro_amd_ctx_poison_scratch.argtypes = [C.c_void_p, C.c_void_p]
    L.castro_amd_ctx_set_source_corrector.argtypes = [C.c_void_p, PF]
    L.castro_amd_ctu_hydro_fab.argtypes = [
        C.c_void_p, I3, I3, I3, I3, PF, PF, PF, PF, PF, PF, C.POINTER(Geom), C.POINTER(Params),
        C.c_double, C.c_double, C.c_int, C.c_void_p]
    L.castro_amd_ctu_hydro_clean_fab.argtypes = [
        C.c_void_p, I3, I3, I3, I3, PF, PF, PF, PF, PF, PF, C.POINTER(Geom), C.POINTER(Params),
        C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.castro_amd_ctu_hydro_fab_ex.argtypes = [
        C.c_void_p, I3, I3, I3, I3, PF, PF, PF, PF, PF, PF, C.POINTER(Geom), C.POINTER(Params),
        C.c_double, C.c_double, C.POINTER(HydroOpts), C.c_void_p]
    L.castro_amd_ctu_hydro_mf.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(HydroBox), C.c_int,
                                          C.POINTER(Geom), C.POINTER(Params), C.c_double, C.c_double, C.POINTER(HydroOpts),
                                          C.c_void_p]
    L.castro_amd_fab_ops_p.argtypes = [C.c_void_p, C.c_int, C.POINTER(FabOp), C.POINTER(Params), C.c_void_p]
    L.castro_amd_sources_mf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(SourceBox), C.POINTER(C.c_double), C.c_int,
                                        C.POINTER(Rotation), C.POINTER(Geom), C.POINTER(Params), C.c_double, C.c_int, C.c_void_p]
    L.castro_amd_clean_state_reduce_mf.argtypes = [C.c_void_p, C.c_int, C.POINTER(StateBox), C.POINTER(Geom), C.POINTER(Params),
                                                   C.c_int, C.c_void_p, C.c_void_p]
    L.castro_amd_estdt_mf.argtypes = [C.c_void_p, C.c_int, C.POINTER(StateBox), C.POINTER(Geom), C.POINTER(Params), C.c_void_p, C.c_void_p]
    L.castro_amd_step_control.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_double, C.c_double,
                                          C.c_double, C.c_int, C.c_void_p]
    L.castro_amd_clean_state_fab.argtypes = [C.c_void_p, PF, I3, I3, C.POINTER(Params), C.c_int, C.c_void_p]
    L.castro_amd_clean_state_reduce_fab.argtypes = [C.c_void_p, PF, I3, I3, C.POINTER(Geom), C.POINTER(Params),
                                                    C.c_int, C.c_void_p, C.c_void_p]
    L.castro_amd_estdt_fab.argtypes = [C.c_void_p, PF, I3, I3, C.POINTER(Geom), C.POINTER(Params),
                                       C.c_void_p, C.c_void_p]
    L.castro_amd_derive_fab.argtypes = [C.c_void_p, C.c_int, PF, PF, C.c_int, I3, I3, C.POINTER(Geom), C.POINTER(Params),
                                        C.POINTER(C.c_double * 3), C.c_void_p]
    D3 = C.POINTER(C.c_double * 3)
    L.castro_amd_old_gravity_source_fab.argtypes = [C.c_void_p, PF, PF, I3, I3, D3, C.c_int, C.c_double, C.c_void_p]
    L.castro_amd_new_gravity_source_fab.argtypes = [C.c_void_p, PF, PF, PF, PF, I3, I3, D3, C.c_int, C.c_double,
                                                    C.POINTER(Geom), C.c_void_p]
    L.castro_amd_old_rotation_source_fab.argtypes = [C.c_void_p, PF, PF, I3, I3, C.POINTER(Rotation), C.POINTER(Geom),
                                                     C.c_double, C.c_void_p]
    L.castro_amd_new_rotation_source_fab.argtypes = [C.c_void_p, PF, PF, PF, PF, I3, I3, C.POINTER(Rotation), C.POINTER(Geom),
                                                     C.c_double, C.c_void_p]
    L.castro_amd_saxpy_fab.argtypes = [C.c_void_p, PF, C.c_double, PF, C.c_int, I3, I3, C.c_void_p]
    L.castro_amd_error_tag_fab.argtypes = [C.c_void_p, PF, C.c_int, PF, I3, I3, C.c_int, C.c_double, C.c_void_p]
    L.castro_amd_fab_ops.argtypes = [C.c_void_p, C.c_int, C.POINTER(FabOp), C.c_void_p]
    L.castro_amd_apply_source_fab.argtypes = [C.c_void_p, PF, PF, C.c_double, PF, C.c_int, I3, I3, C.POINTER(Params), C.c_int, C.c_void_p]
    L.castro_amd_cc_interp_fab.argtypes = [C.c_void_p, PF, PF, I3, I3, C.c_int, C.c_void_p]
    L.castro_amd_fillpatch_shell_fab.argtypes = [C.c_void_p, PF, PF, I3, I3, C.c_int, C.POINTER(Params), C.c_int, C.c_void_p]
    L.castro_amd_lincomb_fab.argtypes = [C.c_void_p, PF, C.c_double, PF, C.c_double, PF, C.c_int, I3, I3, C.c_void_p]
    L.castro_amd_avgdown_fab.argtypes = [C.c_void_p, PF, PF, I3, I3, C.c_int, C.c_void_p]
    L.castro_amd_fluxreg_crse_init_fab.argtypes = [C.c_void_p, PF, PF, I3, I3, C.c_int, C.c_double, C.c_void_p]
    L.castro_amd_fluxreg_fine_add_fab.argtypes = [C.c_void_p, PF, PF, I3, I3, C.c_int, C.c_int, C.c_double, C.c_void_p]
    L.castro_amd_reflux_fab.argtypes = [C.c_void_p, PF, PF, I3, I3, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    L.castro_amd_bc_fill_fab.argtypes = [C.c_void_p, PF, C.POINTER(Geom), C.c_void_p]
    L.castro_amd_copy_fab.argtypes = [C.c_void_p, PF, PF, I3, I3, C.c_void_p]
    L.castro_amd_pack_fab.argtypes = [C.c_void_p, PF, I3, I3, C.c_void_p, C.c_void_p]
    L.castro_amd_unpack_fab.argtypes = [C.c_void_p, PF, I3, I3, C.c_void_p, C.c_void_p]
    for f in (L.castro_amd_pack_regions_fab, L.castro_amd_unpack_regions_fab):
        f.argtypes = [C.c_void_p, PF, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.c_void_p, C.c_void_p]
    L.castro_amd_sedov_init_fab.argtypes = [C.c_void_p, PF, I3, I3, C.POINTER(Geom), C.POINTER(Params),
                                            C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
    L.castro_amd_sod_init_fab.argtypes = [C.c_void_p, PF, I3, I3, C.POINTER(Geom), C.POINTER(Params)] + \
        [C.c_double] * 6 + [C.c_int, C.c_double, C.c_void_p]
    if hasattr(L, "castro_amd_comm_version"):               # absent from A/B builds of revisions before the C-level halo exchange
        L.castro_amd_comm_version.restype = C.c_char_p
        L.castro_amd_comm_unique_id.argtypes = [C.c_void_p]
        L.castro_amd_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.castro_amd_comm_adopt.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
        L.castro_amd_comm_rank.argtypes = [C.c_void_p]
        L.castro_amd_comm_size.argtypes = [C.c_void_p]
        L.castro_amd_comm_destroy.argtypes = [C.c_void_p]
        L.castro_amd_halo_plan_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.POINTER(HaloRegion), C.c_int]
        L.castro_amd_halo_plan_destroy.argtypes = [C.c_void_p]
        L.castro_amd_halo_plan_bytes_sent.argtypes = [C.c_void_p]
        L.castro_amd_halo_plan_bytes_sent.restype = C.c_longlong
        L.castro_amd_fill_boundary.argtypes = [C.c_void_p, C.c_void_p, PF, C.POINTER(Geom), C.c_void_p]
        L.castro_amd_fill_boundary_ex.argtypes = [C.c_void_p, C.c_void_p, PF, C.POINTER(Geom), C.c_int, C.c_void_p]
        L.castro_amd_halo_plan_wait_packed.argtypes = [C.c_void_p, C.c_void_p]
        L.castro_amd_halo_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.POINTER(HaloMsg), C.c_int,
                                                   C.POINTER(HaloMsg), C.c_int]
        L.castro_amd_halo_group_destroy.argtypes = [C.c_void_p]
        L.castro_amd_halo_group_bytes_sent.argtypes = [C.c_void_p]
        L.castro_amd_halo_group_bytes_sent.restype = C.c_longlong
        L.castro_amd_fill_boundary_group.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Fab), C.POINTER(Geom), C.c_void_p]
        L.castro_amd_fill_boundary_group_ex.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Fab), C.POINTER(Geom), C.c_int, C.c_void_p]
        L.castro_amd_halo_group_wait_packed.argtypes = [C.c_void_p, C.c_void_p]
        L.castro_amd_allreduce_min.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.castro_amd_ctx_profile.argtypes = [C.c_void_p, C.c_int]
    L.castro_amd_ctx_profile_count.argtypes = [C.c_void_p]
    L.castro_amd_ctx_profile_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int,
                                             C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    L.castro_amd_ctx_profile_reset.argtypes = [C.c_void_p]
    L.castro_amd_berger_rigoutsos.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_int]
    V = C.c_void_p
    L.castro_amd_cmpflx_points.argtypes = [C.c_longlong, C.c_int, V, V, V, V, V, V, C.POINTER(Params), V, V]
    L.castro_amd_ppm_points.argtypes = [C.c_longlong, V, V, V, V, C.c_double, V, V]
    L.castro_amd_flatten_points.argtypes = [C.c_longlong, V, V, V, V]
    L.castro_amd_trans_points.argtypes = [C.c_longlong, C.c_int, C.c_int, V, V, V, V, V, V, C.c_double, C.c_double,
                                          C.POINTER(Params), V, V]
    _libs[mode] = L
    return L


def i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def default_params(**overrides):
    """castro_amd_params with the reference defaults for the Sedov setup; keyword overrides are applied
    and the derived floors recomputed (Castro_setup.cpp:222-288)."""
    p = Params()
    load().castro_amd_default_params(C.byref(p))
    if overrides:
        for k, v in overrides.items():
            if not hasattr(p, k):
                raise AttributeError("castro_amd_params has no field %r" % k)
            setattr(p, k, v)
        if any(k in overrides for k in ("eos_gamma", "small_dens", "small_temp", "abar")):
            if "small_pres" not in overrides:
                p.small_pres = 1.e-100
            if "small_ener" not in overrides:
                p.small_ener = 1.e-100
            load().castro_amd_finalize_params(C.byref(p))
    return p


def make_geom(n_cell, prob_lo=(0., 0., 0.), prob_hi=(1., 1., 1.), lo_bc=(2, 2, 2), hi_bc=(2, 2, 2),
              domlo=(0, 0, 0)):
    g = Geom()
    for d in range(3):
        g.problo[d] = prob_lo[d]
        g.probhi[d] = prob_hi[d]
        g.dx[d] = (prob_hi[d] - prob_lo[d]) / n_cell[d]
        g.domlo[d] = domlo[d]
        g.domhi[d] = domlo[d] + n_cell[d] - 1
        g.lo_bc[d] = lo_bc[d]
        g.hi_bc[d] = hi_bc[d]
    g.coord = 0
    return g


_I3 = C.c_int * 3


def fab_desc(ptr, lo, hi, ncomp):
    return Fab(ptr, _I3(int(lo[0]), int(lo[1]), int(lo[2])), _I3(int(hi[0]), int(hi[1]), int(hi[2])), int(ncomp))


_FAB_MEMO = {}


def fab_of(tensor, lo, hi):
    """Descriptor for a contiguous torch tensor shaped (ncomp, nz, ny, nx) covering box [lo, hi]
    (C order of that shape == AMReX FAB layout: i fastest, component slowest).  A descriptor is a function of
    (pointer, shape, box) only, so it is memoised: a level of small AMR boxes builds thousands per coarse step."""
    if tensor is None:
        return fab_desc(None, lo, hi, 0)
    shp = tensor.shape
    key = (tensor.data_ptr(), shp, lo if type(lo) is tuple else tuple(lo), hi if type(hi) is tuple else tuple(hi))
    f = _FAB_MEMO.get(key)
    if f is not None and tensor.is_contiguous():
        return f
    if not (tensor.is_contiguous() and shp[-1] == hi[0] - lo[0] + 1 and shp[-2] == hi[1] - lo[1] + 1 and shp[-3] == hi[2] - lo[2] + 1):
        raise AssertionError("FAB tensors must be contiguous and shaped (ncomp, nz, ny, nx) of their box: %s for %s"
                             % (tuple(shp), (lo, hi)))
    f = fab_desc(key[0], lo, hi, shp[0] if len(shp) == 4 else 1)
    if len(_FAB_MEMO) > 4096:
        _FAB_MEMO.clear()
    _FAB_MEMO[key] = f
    return f


def check(rc, what):
    if rc != OK:
        names = {ERR_ARG: "bad argument", ERR_UNSUPPORTED: "unsupported option", ERR_NOMEM: "out of device memory",
                 ERR_HIP: "HIP runtime error / no device"}
        raise RuntimeError("castro_amd: %s failed: %s (%d)" % (what, names.get(rc, "error"), rc))
