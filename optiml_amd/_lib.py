"""ctypes binding of libbcqp_hip.so (C ABI declared in include/bcqp.h).

There is no CPU fallback: if the shared library is missing, or a call fails, this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libbcqp_hip.so')

OK = 0
ERR_HIP, ERR_RCCL, ERR_NOT_PD, ERR_NONFINITE, ERR_BADARG, ERR_NOMEM = -1, -2, -3, -4, -5, -6
F64, F32, STREAM = 0, 1, 2
STORAGE = {'f64': 0, 'f32': 1, 'stream': 2}
KERNEL_LINEAR, KERNEL_POLY, KERNEL_RBF, KERNEL_SIGMOID, KERNEL_LAPLACIAN = 0, 1, 2, 3, 4
PLAIN, SVC, SVR = 0, 1, 2
PG, FW, AS, IP = 0, 1, 2, 3
AS_CG = 5   # ActiveSet with conjugate-gradient restricted solves (BQ_AS_CG)
STATUS = {0: 'unknown', 1: 'optimal', 2: 'stopped'}
GET_X, GET_G, GET_LP, GET_LM, GET_D, GET_MASK_L, GET_MASK_U, GET_X_NOW, GET_G_NOW, GET_DUAL = range(10)
NO_RANK_ONE = 16
FULL_PANEL = 32
PLACE_PANEL = 64
DENSE_ROWS, DENSE_LOWER = 256, 512   # OR-ed into the storage of bq_problem_create_dense
SMO_ALPHAS, SMO_ERRORS, SMO_SCALARS, SMO_STATS = range(4)
RULE_SGD, RULE_ADAM, RULE_AMSGRAD, RULE_ADAMAX, RULE_ADAGRAD, RULE_ADADELTA, RULE_RMSPROP = range(7)
MOM = {'none': 0, 'polyak': 1, 'nesterov': 2}
PROF_MATVEC, PROF_GRAM, PROF_CHOL, PROF_EXCH, PROF_PCSHARD = range(5)
COUNT_INNER, COUNT_MINRES, COUNT_REFACTOR, COUNT_REUSED, COUNT_NO_PRODUCT = range(5)
STATE_X, STATE_G, STATE_MULT, STATE_MASKS = 1, 2, 4, 8
ABI_VERSION = 2


class IterStat(C.Structure):
    _fields_ = [('iter', C.c_int64), ('f', C.c_double), ('r1', C.c_double), ('r2', C.c_double), ('r3', C.c_double)]


class SolverState(C.Structure):   # bq_solver_snapshot
    _fields_ = [('iter', C.c_int64), ('kind', C.c_int), ('have', C.c_int), ('f', C.c_double), ('best_lb', C.c_double),
                ('x', C.POINTER(C.c_double)), ('g', C.POINTER(C.c_double)), ('lp', C.POINTER(C.c_double)),
                ('lm', C.POINTER(C.c_double)), ('mask_l', C.POINTER(C.c_double)), ('mask_u', C.POINTER(C.c_double))]


class AlParams(C.Structure):
    _fields_ = [('rule', C.c_int32), ('momentum_type', C.c_int32), ('step_size', C.c_double), ('momentum', C.c_double),
                ('beta1', C.c_double), ('beta2', C.c_double), ('decay', C.c_double), ('offset', C.c_double),
                ('rho', C.c_double), ('tol', C.c_double), ('epochs', C.c_int64)]


STAT_DTYPE = np.dtype([('iter', np.int64), ('f', np.float64), ('r1', np.float64), ('r2', np.float64),
                       ('r3', np.float64)])

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.c_int64, C.c_int)

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_i64 = C.c_int64

# name -> (restype, argtypes); every symbol include/bcqp.h declares
PROTOTYPES = {
    'bq_abi_version': (C.c_int, []),
    'bq_last_error': (C.c_char_p, []),
    'bq_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'bq_ctx_create': (C.c_int, [C.c_int, C.POINTER(_vp)]),
    'bq_comm_unique_id': (C.c_int, [_vp]),
    'bq_ctx_create_rccl': (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    'bq_comm_init_report': (C.c_int, [C.c_char_p, C.c_size_t]),
    'bq_ctx_create_exchange': (C.c_int, [C.c_int, C.c_int, C.c_int, EXCHANGE_FN, _vp, C.POINTER(_vp)]),
    'bq_ctx_create_share': (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    'bq_ctx_destroy': (C.c_int, [_vp]),
    'bq_ctx_info': (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    'bq_ctx_comm_info': (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'bq_ctx_set_sym_allreduce': (C.c_int, [_vp, C.c_int]),
    'bq_ctx_profile': (C.c_int, [_vp, C.c_int]),
    'bq_ctx_profile_read': (C.c_int, [_vp, C.c_int, _dp, C.POINTER(_i64), C.c_int]),
    'bq_ctx_probe_bandwidth': (C.c_int, [_vp, _i64, C.c_int, _dp, _dp]),
    'bq_ctx_probe_mfma_f64': (C.c_int, [_vp, C.c_double, _dp]),
    'bq_ctx_probe_exchange': (C.c_int, [_vp, C.c_int, _i64, C.c_int, _dp, _dp]),
    'bq_ctx_set_collective_timeout': (C.c_int, [_vp, C.c_double]),
    'bq_ctx_set_placement_budget': (C.c_int, [_vp, C.c_double, C.c_double, C.c_double]),
    'bq_ctx_release_held_memory': (C.c_int, [_vp, C.POINTER(_i64)]),
    'bq_ctx_probe_stall': (C.c_int, [_vp, C.c_double, C.c_int]),
    'bq_row_block': (C.c_int, [_i64, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    'bq_sym_row_block': (C.c_int, [_i64, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    'bq_problem_create_dense': (C.c_int, [_vp, _i64, _dp, _dp, C.c_int, C.POINTER(_vp)]),
    'bq_problem_create_kernel': (C.c_int, [_vp, C.c_int, _i64, _i64, _dp, _dp, C.c_int, C.c_double, C.c_double,
                                           C.c_int, C.c_double, _dp, C.c_int, C.POINTER(_vp)]),
    'bq_problem_destroy': (C.c_int, [_vp]),
    'bq_problem_dims': (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    'bq_problem_layout': (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_i64)]),
    'bq_problem_matvec': (C.c_int, [_vp, _dp, _dp]),
    'bq_problem_eval': (C.c_int, [_vp, _dp, _dp, _dp]),
    'bq_problem_x_star': (C.c_int, [_vp, _dp, C.POINTER(C.c_int), C.POINTER(_i64)]),
    'bq_problem_gram_matvec': (C.c_int, [_vp, _dp, _dp]),
    'bq_problem_panel_rows': (C.c_int, [_vp, _i64, _i64, _dp]),
    'bq_problem_time_matvec': (C.c_int, [_vp, C.c_int, _dp]),
    'bq_problem_placement': (C.c_int, [_vp, C.POINTER(C.c_int), _dp, C.c_int]),
    'bq_solver_create': (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, C.c_double, _i64, C.c_double, C.POINTER(_vp)]),
    'bq_solver_destroy': (C.c_int, [_vp]),
    'bq_solver_run': (C.c_int, [_vp, _i64, C.POINTER(IterStat), _i64, C.POINTER(_i64), C.POINTER(C.c_int)]),
    'bq_solver_state': (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(C.c_int), _dp]),
    'bq_solver_get': (C.c_int, [_vp, C.c_int, _dp]),
    'bq_solver_set_inner': (C.c_int, [_vp, C.c_double, _i64]),
    'bq_solver_inner_iters': (C.c_int, [_vp, C.POINTER(_i64)]),
    'bq_solver_counter': (C.c_int, [_vp, C.c_int, C.POINTER(_i64)]),
    'bq_solver_get_state': (C.c_int, [_vp, C.POINTER(SolverState)]),
    'bq_solver_set_state': (C.c_int, [_vp, C.POINTER(SolverState)]),
    'bq_al_solver_create': (C.c_int, [_vp, C.POINTER(AlParams), _dp, _dp, _dp, _dp, _dp, C.POINTER(_vp)]),
    'bq_al_solver_dual_size': (C.c_int, [_vp, C.POINTER(_i64)]),
    'bq_al_solver_set_schedules': (C.c_int, [_vp, _dp, _dp, _i64]),
    'bq_smo_create': (C.c_int, [_vp, C.c_int, _dp, C.c_double, C.c_double, C.c_double, C.POINTER(_vp)]),
    'bq_smo_run': (C.c_int, [_vp, _i64, C.POINTER(_i64), C.POINTER(C.c_int)]),
    'bq_smo_get': (C.c_int, [_vp, C.c_int, _dp]),
    'bq_smo_destroy': (C.c_int, [_vp]),
    'bq_decision_function': (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, C.c_int, _i64, _i64, _dp, _dp,
                                       C.c_double, _i64, _dp, _dp]),
    'bq_gram_matrix': (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, C.c_int, _i64, _i64, _dp, _i64, _dp, _dp]),
    'bq_cholesky_solve': (C.c_int, [_vp, _i64, _dp, _dp, _dp, _dp]),
}


class BcqpError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f'libbcqp_hip error {code}: {message}')
        self.code = code


_lib = None


def load():
    """Load the shared library (once) and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build the HIP library first (python -m optiml_amd.build, needs hipcc). '
            'optiml_amd has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale: also loud
        fn.restype = res
        fn.argtypes = args
    if lib.bq_abi_version() != ABI_VERSION:
        raise RuntimeError('libbcqp_hip.so ABI version mismatch: rebuild it')
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        msg = load().bq_last_error()
        raise BcqpError(rc, msg.decode('utf-8', 'replace') if msg else '?')


def as_f64(a, n=None, name='array'):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None and a.size != n:
        raise ValueError(f'{name} has {a.size} elements, expected {n}')
    return a


def ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)
