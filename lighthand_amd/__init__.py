"""lighthand_amd -- MI355X-native engine for LightHand's heatmap-regression hot path.

Public surface (mirrors the reference callables):
  modeling.simplebaseline.pose_resnet.get_pose_net, modeling.hrnet.pose_hrnet.get_hrnet,
  heatmap.JointsMSELoss / get_max_preds / generate_target, optim.Adam, runtime.TrainStep.
Compute happens only in liblighthand_hip.so (hand-written gfx950 kernels); importing the
compute modules without the built library raises ``LightHandError``.
"""
from ._lib import LightHandError  # noqa: F401

__version__ = "0.1.0"
