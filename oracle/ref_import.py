"""Import the reference's PWCLO-Net Python layers on CPU -- TEST INFRASTRUCTURE,
build container only.

`/root/reference` does not exist on the GPU box, so this module is used only by
``oracle/gen_golden.py`` (fixture generation) and by the container-only tests
that validate ``oracle.model`` against the imported reference.  Nothing under
``-m gpu``, ``smoke()`` or ``bench.py`` imports it.

What the import needs:
  * env ``PYLIDAR_SLAM_PWCLONET_ABS_PATH`` = reference root and ``RELIDAR_SLAM_ABS_PATH`` =
    a temp dir holding a ``pyLiDAR_SLAM`` symlink to the reference (``slam/common/pose.py:16-17``);
  * in-process stubs for packages absent from this image (``omegaconf``, ``typeguard``,
    ``pyquaternion``): type-hint / attribute-bag use only on this path;
  * ``pointnet2_ops._ext`` pre-populated with a CPU stand-in: the C oracle (``oracle.ops``),
    because the reference's own ext is CUDA-only (``P2/pointnet2_utils.py:7-31``).
No reference file is copied, modified or byte-compiled (``sys.dont_write_bytecode``).
"""
import os
import sys
import tempfile
import types

REF_ROOT = "/root/reference"
P2_LIB = os.path.join(REF_ROOT, "slam/models/Pointnet2_PyTorch/pointnet2_ops_lib")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "slam", "models", "PWCLONet"))


class _DictConfig(dict):
    """Attribute bag standing in for omegaconf.DictConfig (``config.x`` / ``config.get``)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _install_stubs():
    if "omegaconf" not in sys.modules:
        m = types.ModuleType("omegaconf")
        m.DictConfig = _DictConfig
        m.MISSING = "???"

        class OmegaConf:
            @staticmethod
            def create(d=None):
                return _DictConfig(d or {})

        m.OmegaConf = OmegaConf
        sys.modules["omegaconf"] = m
    if "typeguard" not in sys.modules:
        m = types.ModuleType("typeguard")
        m.check_type = lambda *a, **k: None
        sys.modules["typeguard"] = m
    if "pyquaternion" not in sys.modules:
        m = types.ModuleType("pyquaternion")

        class Quaternion:  # imported by slam/common/pose.py:19, unused on this path
            def __init__(self, *a, **k):
                raise RuntimeError("pyquaternion stub: not available in this image")

        m.Quaternion = Quaternion
        sys.modules["pyquaternion"] = m


def _install_hydra_stubs():
    """``slam/training/loss_modules.py:8-11`` needs ``hydra.conf.{dataclass,MISSING,field}`` and a
    ``ConfigStore`` to register its config dataclasses with; neither is used by the loss arithmetic."""
    import dataclasses
    if "hydra" in sys.modules:
        return
    h, hc = types.ModuleType("hydra"), types.ModuleType("hydra.conf")
    hcore, hcs = types.ModuleType("hydra.core"), types.ModuleType("hydra.core.config_store")
    hc.dataclass, hc.MISSING, hc.field = dataclasses.dataclass, "???", dataclasses.field

    class ConfigStore:
        _inst = None

        @classmethod
        def instance(cls):
            cls._inst = cls._inst or cls()
            return cls._inst

        def store(self, *a, **k):
            pass

    hcs.ConfigStore = ConfigStore
    sys.modules.update({"hydra": h, "hydra.conf": hc, "hydra.core": hcore, "hydra.core.config_store": hcs})


def load_loss():
    """The reference's ``_PWCLONetLossModule`` / ``ExponentialWeights`` (loss_modules.py:147-545)."""
    load()
    _install_hydra_stubs()
    import importlib
    return importlib.import_module("slam.training.loss_modules")


_loaded = None


def load():
    """Returns a namespace with the reference classes/functions of the hot path."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    _install_stubs()

    link_root = tempfile.mkdtemp(prefix="pwclo_ref_")
    os.symlink(REF_ROOT, os.path.join(link_root, "pyLiDAR_SLAM"))
    os.environ["PYLIDAR_SLAM_PWCLONET_ABS_PATH"] = REF_ROOT
    os.environ["RELIDAR_SLAM_ABS_PATH"] = link_root

    from oracle import ops as oracle_ops  # CPU stand-in for the CUDA-only extension
    ext = types.ModuleType("pointnet2_ops._ext")
    for name in ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
                 "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
                 "group_points_grad"):
        setattr(ext, name, getattr(oracle_ops, name))
    sys.modules["pointnet2_ops._ext"] = ext

    if P2_LIB not in sys.path:
        sys.path.insert(0, P2_LIB)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    import importlib
    ns = types.SimpleNamespace()
    ns.pwclo_net = importlib.import_module("slam.models.PWCLONet.pwclo_net")
    ns.costvolume = importlib.import_module("slam.models.PWCLONet.costvolume")
    ns.flowpredictor = importlib.import_module("slam.models.PWCLONet.flowpredictor")
    ns.pose_calculator = importlib.import_module("slam.models.PWCLONet.pose_calculator")
    ns.pose_warp_refinement = importlib.import_module("slam.models.PWCLONet.pose_warp_refinement")
    ns.PWCLO_utils = importlib.import_module("slam.models.PWCLONet.PWCLO_utils")
    p2 = "slam.models.Pointnet2_PyTorch.pointnet2_ops_lib.pointnet2_ops"
    ns.pointnet2_modules = importlib.import_module(p2 + ".pointnet2_modules")
    ns.pointnet2_utils = importlib.import_module(p2 + ".pointnet2_utils")
    ns.pytorch_utils = importlib.import_module(p2 + ".pytorch_utils")
    ns.DictConfig = _DictConfig
    _loaded = ns
    return ns


def make_reference_model(device="cpu"):
    """``PWCLONet(config)`` exactly as ``slam/training/prediction_modules.py`` builds it."""
    ns = load()
    cfg = _DictConfig(num_input_channels=3, sequence_len=2, device=device, scalar_last=False,
                      type="pwclonet")
    return ns.pwclo_net.PWCLONet(cfg)
