"""asr-featext-opencl_amd -- MI355X (gfx950) MFCC front end.

Python side of the drop-in boundary: thin ctypes bindings over ``libmfcchip.so`` (the C ABI of
``include/mfx.h``).  The compiled host mirror of the reference's ``ParamBase``/``MfccBase`` classes
lives in ``host/`` (C++); this module offers the same interface to Python callers and tests.

There is no CPU fallback here: if the shared library is missing or no HIP device is present the
constructors raise.  (The directory name is not an importable identifier; load it with
``importlib`` as ``__graft_entry__.load_package()`` does, or put the parent directory on
``sys.path`` and use ``importlib.import_module("asr-featext-opencl_amd")``.)
"""
from . import mfcc  # noqa: F401
from . import sharding  # noqa: F401
from .mfcc import (  # noqa: F401
    DYN_ACC,
    DYN_DELTA,
    DYN_NONE,
    NORM_CMN,
    NORM_CVN,
    NORM_MINMAX,
    NORM_NONE,
    MfccHip,
    MfxConfig,
    MfxError,
    host_dct_matrix,
    host_dct_mfma_operands,
    host_frame_count,
    host_mel_lane_plan,
    host_mel_table,
    KERNEL_TABLE,
    library_path,
    load_library,
    plan_kernel,
    reference_window,
)
