"""pyjac_amd: MI355X-native batched species-rate + analytical-Jacobian evaluation
behind pyJac's pywrap API.  See DESIGN.md."""
from ._lib import LAYOUT_AOS, LAYOUT_SOA, PyjacError  # noqa: F401
from .evaluator import Evaluator  # noqa: F401
from .mechanism import read_mech  # noqa: F401
from .tables import MechTables, build_tables  # noqa: F401

__version__ = '0.1.0'
