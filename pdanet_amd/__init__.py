"""pdanet_amd -- MI355X (gfx950) native implementation of PDA-SSD's point-sampling /
grouping hot path behind the reference's ``pcdet.ops.pointnet2.pointnet2_batch`` operator API.

Layout
  csrc/                    hand-written HIP kernels + the C ABI (include/pda_pointnet2.h)
  libpda_pointnet2.so      built in-tree by ``pdanet_amd.build.build()`` / ``make -C csrc``
  _lib.py                  ctypes loader (fails loudly when the library is missing)
  pointnet2_batch_cuda.py  mirror of the reference's pybind extension module (same names)
  pointnet2_utils.py       mirror of the reference's autograd Functions + grouper modules
There is no CPU fallback anywhere in this package; the CPU oracle lives in ``oracle/`` and is
test infrastructure only.
"""

__version__ = "0.1.0"
