"""emba_amd — MI355X (gfx950) implementation of EMBA's per-event warp / residual / Jacobian / normal-equation
hot path behind the reference's `EMBA::LEGM` interface.  See DESIGN.md and include/emba_hip.h."""
from .legm import LEGM, LinearTrajectory, EventPacket  # noqa: F401
from ._lib import EmbaError  # noqa: F401
