"""``pointnet2_ops`` -- the name the reference imports its extension package under.

The reference's modules do ``import pointnet2_ops.pointnet2_modules`` / ``from pointnet2_ops import
pointnet2_utils`` (``P2/__init__.py:1-3``, ``pointnet2_ops_lib/pointnet2_ops/__init__.py:1-3``) and
``pointnet2_utils`` binds ``pointnet2_ops._ext`` (``pointnet2_utils.py:7-9``).  This package is that name for
the MI355X implementation: every submodule is the module of ``pwclonet_pylidarslam_amd.pointnet2_ops`` itself
(no copies), so ``pointnet2_ops._ext.group_points`` etc. launch the HIP kernels of libpwclo_hip.so.
Installed by the repository's ``setup.py`` next to ``pwclonet_pylidarslam_amd``; also importable from a checkout.
"""
import sys

from pwclonet_pylidarslam_amd.pointnet2_ops import (_ext, pointnet2_modules, pointnet2_utils,  # noqa: F401
                                                    pytorch_utils)

for _name, _mod in (("_ext", _ext), ("pointnet2_modules", pointnet2_modules),
                    ("pointnet2_utils", pointnet2_utils), ("pytorch_utils", pytorch_utils)):
    sys.modules[__name__ + "." + _name] = _mod

__version__ = "3.0.0+gfx950"
