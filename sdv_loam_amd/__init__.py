"""Import alias: the product package lives in the directory ``sdv-loam_amd/`` (the name the build
contract asks for), which is not a valid Python identifier.  This stub makes it importable as
``sdv_loam_amd`` by pointing the package search path at that directory."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "sdv-loam_amd"))
from ._pkg import *  # noqa: F401,F403,E402
