"""ctypes binding of libc8.so: one declaration per entry point of include/c8.h."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libc8.so")

C8_ELEM_TRI3, C8_ELEM_TET4, C8_ELEM_HEX8 = 3, 4, 8
C8_OK, C8_LOCAL_SOLVE_FAILED, C8_ERR_ARG, C8_ERR_DEVICE, C8_ERR_UNSUPPORTED, C8_NOT_CONVERGED = 0, -1, -2, -3, -4, -5
C8_SCATTER_ATOMIC, C8_SCATTER_COLORED, C8_SCATTER_GATHER = 0, 1, 2
C8_KERNEL_AUTO, C8_KERNEL_SLOT, C8_KERNEL_WAVE, C8_KERNEL_WAVE_AD, C8_KERNEL_NODE = 0, 1, 2, 3, 4
C8_SCALE_NONE, C8_SCALE_LOG, C8_SCALE_BOUNDS = 0, 1, 2

dp = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)


class C8Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("c8 error %d: %s" % (code, msg))
        self.code = code


class MeshDesc(C.Structure):
    _fields_ = [("elem_type", C.c_int32), ("num_nodes", C.c_int32), ("num_elems", C.c_int32),
                ("num_elem_sets", C.c_int32), ("coords", dp), ("conn", i32p), ("elem_set", i32p),
                ("num_extra_pairs", C.c_int32), ("extra_pairs", i32p)]


class ModelDesc(C.Structure):
    _fields_ = [("global_type", C.c_char_p), ("local_type", C.c_char_p), ("stabilization_multiplier", C.c_double),
                ("local_max_iters", C.c_int32), ("local_abs_tol", C.c_double), ("local_rel_tol", C.c_double),
                ("num_params", C.c_int32), ("params", dp), ("thickness", C.c_double),
                ("ls_sufficient_decrease", C.c_double), ("ls_min_backtrack", C.c_double), ("ls_max_backtrack", C.c_double),
                ("ls_max_evals", C.c_int32)]


class Dbc(C.Structure):
    _fields_ = [("resid", C.c_int32), ("eq", C.c_int32), ("n", C.c_int32), ("nodes", C.c_void_p), ("values", C.c_void_p)]


class Tbc(C.Structure):
    _fields_ = [("resid", C.c_int32), ("n", C.c_int32), ("nodes_per_face", C.c_int32), ("faces", C.c_void_p),
                ("traction", C.c_void_p)]


class NewtonOpts(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("abs_tol", C.c_double), ("rel_tol", C.c_double), ("line_search", C.c_int32),
                ("sufficient_decrease", C.c_double), ("min_backtrack", C.c_double), ("max_backtrack", C.c_double),
                ("max_evals", C.c_int32)]


class CalibrationDesc(C.Structure):
    _fields_ = [("num_faces", C.c_int32), ("nodes_per_face", C.c_int32), ("faces", C.c_void_p),
                ("weights", C.c_double * 3), ("balance_factor", C.c_double), ("coord_idx", C.c_int32),
                ("coord_value", C.c_double), ("coord_tol", C.c_double), ("reaction_comp", C.c_int32),
                ("dt_over_total_time", C.c_double)]


class LbfgsOpts(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("grad_tol", C.c_double), ("step_tol", C.c_double),
                ("max_ls_evals", C.c_int32), ("memory", C.c_int32)]


class LbfgsResult(C.Structure):
    _fields_ = [("iters", C.c_int32), ("evals", C.c_int32), ("status", C.c_int32), ("f", C.c_double),
                ("projected_gradient_norm", C.c_double)]


class HaloDesc(C.Structure):
    _fields_ = [("num_owned", C.c_int32), ("num_touched", C.c_int32),
                ("send_ptr", i64p), ("send_nodes", i32p),
                ("recv_ptr", i64p), ("recv_nodes", i32p), ("recv_col_ptr", i64p), ("recv_cols", i32p),
                ("import_ptr", i64p), ("import_nodes", i32p), ("export_ptr", i64p), ("export_nodes", i32p),
                ("num_dims", C.c_int32), ("num_residuals", C.c_int32)]


HOST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), i64p, C.POINTER(C.c_double), i64p)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
C8_HALO_B, C8_HALO_A = 1, 2
C8_COMM_ID_BYTES = 128
ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int)
OBJECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


class State(C.Structure):
    _fields_ = [("x", C.c_void_p * 2), ("x_prev", C.c_void_p * 2), ("xi_prev", C.c_void_p), ("xi", C.c_void_p)]


class System(C.Structure):
    _fields_ = [("A", (C.c_void_p * 2) * 2), ("b", C.c_void_p * 2)]


LINEAR_SOLVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(System), C.POINTER(C.c_void_p))

# every symbol include/c8.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("c8_create", C.c_int, [C.POINTER(MeshDesc), C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    ("c8_destroy", None, [C.c_void_p]),
    ("c8_last_error", C.c_char_p, []),
    ("c8_build_info", C.c_char_p, []),
    ("c8_num_local_dofs", C.c_int, [C.c_void_p]),
    ("c8_num_dims", C.c_int, [C.c_void_p]),
    ("c8_num_residuals", C.c_int, [C.c_void_p]),
    ("c8_num_local_points", C.c_int, [C.c_void_p]),
    ("c8_num_colors", C.c_int, [C.c_void_p]),
    ("c8_graph_nnz", C.c_int64, [C.c_void_p, C.c_int, C.c_int]),
    ("c8_graph", C.c_int, [C.c_void_p, C.c_int, C.c_int, i64p, i32p]),
    ("c8_init_variables", C.c_int, [C.c_void_p, dp]),
    ("c8_set_params", C.c_int, [C.c_void_p, dp]),
    ("c8_set_active_params", C.c_int, [C.c_void_p, C.c_int, C.c_int, i32p]),
    ("c8_num_active_params", C.c_int, [C.c_void_p]),
    ("c8_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("c8_set_scatter_mode", C.c_int, [C.c_void_p, C.c_int]),
    ("c8_get_scatter_mode", C.c_int, [C.c_void_p]),
    ("c8_set_stage_chunk", C.c_int, [C.c_void_p, C.c_int]),
    ("c8_set_stage_overlap", C.c_int, [C.c_void_p, C.c_int]),
    ("c8_set_shape_cache", C.c_int, [C.c_void_p, C.c_int]),
    ("c8_set_assign_mode", C.c_int, [C.c_void_p, C.c_int]),
    ("c8_set_gather_early_nodes", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("c8_gather_finish", C.c_int, [C.c_void_p]),
    ("c8_set_kernel_variant", C.c_int, [C.c_void_p, C.c_int]),
    ("c8_set_async", C.c_int, [C.c_void_p, C.c_int]),
    ("c8_status", C.c_int, [C.c_void_p]),
    ("c8_lbfgs_minimize", C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), OBJECTIVE_FN, C.c_void_p, C.POINTER(LbfgsOpts), C.POINTER(LbfgsResult)]),
    ("c8_set_allreduce", C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int]),
    ("c8_set_qoi_avg_disp", C.c_int, [C.c_void_p]),
    ("c8_set_qoi_calibration", C.c_int, [C.c_void_p, C.POINTER(CalibrationDesc)]),
    ("c8_set_measured", C.c_int, [C.c_void_p, C.c_void_p, C.c_double]),
    ("c8_qoi_preprocess", C.c_int, [C.c_void_p, C.POINTER(State), C.POINTER(C.c_double)]),
    ("c8_assemble_forward_jacobian", C.c_int, [C.c_void_p, C.POINTER(State), C.POINTER(System)]),
    ("c8_assemble_forward_jacobian_subset", C.c_int, [C.c_void_p, C.POINTER(State), C.POINTER(System), C.c_void_p, C.c_int]),
    ("c8_assemble_residual", C.c_int, [C.c_void_p, C.POINTER(State), C.POINTER(System)]),
    ("c8_assemble_adjoint_jacobian", C.c_int, [C.c_void_p, C.POINTER(State), C.c_void_p, C.c_void_p, C.POINTER(System)]),
    ("c8_solve_adjoint_local", C.c_int, [C.c_void_p, C.POINTER(State), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("c8_param_gradient", C.c_int, [C.c_void_p, C.POINTER(State), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    ("c8_eval_qoi", C.c_int, [C.c_void_p, C.POINTER(State), C.c_void_p]),
    ("c8_apply_dirichlet", C.c_int, [C.c_void_p, C.c_int, C.POINTER(Dbc), C.POINTER(C.c_void_p), C.POINTER(System), C.c_int]),
    ("c8_apply_traction", C.c_int, [C.c_void_p, C.c_int, C.POINTER(Tbc), C.POINTER(System)]),
    ("c8_face_points", C.c_int, [C.c_int, C.c_int, dp, i32p, dp]),
    ("c8_apply_A", C.c_int, [C.c_void_p, C.POINTER(System), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("c8_primal_solve_step", C.c_int, [C.c_void_p, C.POINTER(State), C.POINTER(System), C.c_int, C.POINTER(Dbc), C.c_int,
                                       C.POINTER(Tbc), C.POINTER(NewtonOpts), C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    ("c8_adjoint_solve_step", C.c_int, [C.c_void_p, C.POINTER(State), C.POINTER(System), C.c_int, C.POINTER(Dbc), C.c_void_p,
                                        C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("c8_transform_params", C.c_int, [C.c_int, dp, i32p, dp, dp, C.c_int, dp]),
    ("c8_transform_gradient", C.c_int, [C.c_int, dp, dp, i32p, dp, dp, dp]),
    ("c8_brick_mesh", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, dp, i32p]),
    ("c8_brick_partition", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p]),
    ("c8_comm_rccl_id", C.c_int, [C.c_void_p]),
    ("c8_comm_create_rccl", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("c8_comm_create_host", C.c_int, [C.c_int, C.c_int, HOST_EXCHANGE_FN, HOST_ALLREDUCE_FN, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("c8_comm_destroy", None, [C.c_void_p]),
    ("c8_comm_rank", C.c_int, [C.c_void_p]),
    ("c8_comm_size", C.c_int, [C.c_void_p]),
    ("c8_comm_allreduce_sum", C.c_int, [C.c_void_p, dp, C.c_int]),
    ("c8_halo_build", C.c_int, [C.c_int32, i64p, i32p, C.POINTER(HaloDesc), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("c8_halo_attach", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("c8_halo_destroy", None, [C.c_void_p]),
    ("c8_halo_gather_start", C.c_int, [C.c_void_p, C.POINTER(System), C.c_int]),
    ("c8_halo_gather_finish", C.c_int, [C.c_void_p, C.POINTER(System)]),
    ("c8_halo_gather", C.c_int, [C.c_void_p, C.POINTER(System), C.c_int]),
    ("c8_halo_scatter_x", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    ("c8_halo_send_bytes", C.c_int64, [C.c_void_p, C.c_int]),
    ("c8_halo_table", C.c_int, [C.c_void_p, C.c_int, i64p, C.POINTER(i64p)]),
]

_lib = None


def load_library():
    """Load libc8.so.  Raises if the HIP extension has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("calibr8_amd/libc8.so is missing: run `python -m calibr8_amd.build` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        # This Python layer keeps its arrays in torch tensors, and the torch wheel carries a HIP runtime of its own: it has
        # to be the one libc8.so binds to.  Loaded the other way round (libc8.so first, torch afterwards) the process ends
        # up with two runtimes and c8_create finds no device.  (A C or C++ host links libc8.so against its own runtime.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc < 0:
        raise C8Error(rc, load_library().c8_last_error().decode())
    return rc
