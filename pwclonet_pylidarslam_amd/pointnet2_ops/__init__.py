"""Drop-in counterpart of the reference's ``pointnet2_ops`` package
(``slam/models/Pointnet2_PyTorch/pointnet2_ops_lib/pointnet2_ops``) on MI355X."""
from . import _ext, pointnet2_modules, pointnet2_utils, pytorch_utils  # noqa: F401

__version__ = "3.0.0"  # P2/_version.py:1
