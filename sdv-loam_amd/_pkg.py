"""Package surface of sdv-loam_amd (imported through the ``sdv_loam_amd`` alias)."""
from . import synthetic  # noqa: F401
