"""Same entry point as the reference's python/depthmotionnet/v2/networks.py."""
from demon_amd.networks_v2 import BootstrapNet, IterativeNet, RefinementNet  # noqa: F401

__all__ = ["BootstrapNet", "IterativeNet", "RefinementNet"]
