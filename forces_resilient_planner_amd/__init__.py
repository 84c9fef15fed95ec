"""MI355X-native batched NMPC solver: drop-in for the FORCESNLPsolver_{normal,final}_solve path of
ZJU-FAST-Lab/forces_resilient_planner (see DESIGN.md / INTEGRATION.md)."""
from . import layout  # noqa: F401
