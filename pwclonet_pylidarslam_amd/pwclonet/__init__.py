"""PWCLO-Net layers (``slam/models/PWCLONet`` in the reference) on the HIP operator stack."""
from .costvolume import CostVolume  # noqa: F401
from .flowpredictor import FlowPredictor  # noqa: F401
from .pose_calculator import PoseCalculator  # noqa: F401
from .pose_warp_refinement import PoseWarpRefinement  # noqa: F401
from .pwclo_net import PWCLONet  # noqa: F401
