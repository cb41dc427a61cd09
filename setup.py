"""Build + install: ``pip install .`` (or ``python setup.py build_ext --inplace`` for a checkout).

Counterpart of the reference's ``pointnet2_ops_lib/setup.py`` (:19-39): the same importable names
(``pointnet2_ops``, ``pointnet2_ops._ext``) but no CUDAExtension / nvcc arch list -- the kernels are compiled by
hipcc for gfx950 into ONE C-ABI shared library (``pwclonet_pylidarslam_amd/lib/libpwclo_hip.so``,
``include/pwclo_ops.h``) and the ``_ext`` surface is a Python module over it.
"""
import os
import sys

from setuptools import Command, find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


def _build_library():
    sys.path.insert(0, HERE)
    from pwclonet_pylidarslam_amd import build as hip_build
    return hip_build.build()


class build_ext(Command):
    """hipcc --offload-arch=gfx950 over csrc/*.hip -> lib/libpwclo_hip.so (in place)."""
    description = "compile the HIP kernels into libpwclo_hip.so"
    user_options = [("inplace", "i", "accepted for familiarity; the library is always built in place")]

    def initialize_options(self):
        self.inplace = False

    def finalize_options(self):
        pass

    def run(self):
        print("built", _build_library())


class build_py_with_library(build_py):
    def run(self):
        _build_library()
        super().run()


setup(
    name="pointnet2_ops",
    version="3.0.0+gfx950",
    description="MI355X (gfx950) point-cloud operators behind the pointnet2_ops / PWCLO-Net interfaces",
    packages=find_packages(include=["pwclonet_pylidarslam_amd", "pwclonet_pylidarslam_amd.*", "pointnet2_ops"]),
    package_data={"pwclonet_pylidarslam_amd": ["lib/*.so", "csrc/*"]},
    install_requires=["torch>=1.4"],
    cmdclass={"build_ext": build_ext, "build_py": build_py_with_library},
    zip_safe=False,
)
