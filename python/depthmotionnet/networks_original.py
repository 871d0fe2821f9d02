"""`from depthmotionnet.networks_original import *` -- same entry point as the reference
(python/depthmotionnet/networks_original.py), served by the MI355X-native path."""
from demon_amd.networks_original import BootstrapNet, IterativeNet, RefinementNet  # noqa: F401

__all__ = ["BootstrapNet", "IterativeNet", "RefinementNet"]
