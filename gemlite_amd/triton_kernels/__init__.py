"""Import-path shims: `gemlite.triton_kernels.*` as downstream code and the reference's own tests import it
(tests/test_gemlitelineartriton.py:5-7, tests/test_mxfp.py:5-7 of the reference: `from gemlite.triton_kernels.config import KERNEL`).

There are no Triton kernels here — the compute lives in libgemlite_hip.so — but the CONFIGURATION names of that package are part
of the drop-in surface: `config` (KERNEL, AUTOTUNE, set_autotune, set_kernel_caching) and `utils` (get_closest_m, M_MAPPING,
IS_HIP, ...).  `gemlite_amd.alias_as_gemlite()` makes `import gemlite...` resolve to this package.
"""
from . import config, utils  # noqa: F401
