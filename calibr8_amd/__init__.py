"""calibr8_amd -- MI355X-native element assembly and adjoint sensitivities for CALIBR8.

The product is the C-ABI shared library `libc8.so` (include/c8.h): hand-written gfx950 HIP
kernels behind the reference's `eval_*` seam.  This package is the thin Python host side used
by the tests and the bench: ctypes bindings plus an `Assembler` whose arrays are torch tensors
in HBM.  There is no CPU execution path: creating an `Assembler` without a HIP device raises.
"""
from .lib import C8Error, load_library  # noqa: F401
from .assembly import Assembler, LinearSystem, brick_mesh, brick_partition  # noqa: F401
from .primal import PrimalDriver, adjoint_gradient, scipy_solver  # noqa: F401
from .inverse import FEMUProblem, InverseProblem, lbfgs_minimize  # noqa: F401
