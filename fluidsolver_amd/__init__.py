"""fluidsolver_amd -- MI355X-native stable-fluids step (host-side bindings of libfluidsolver_hip.so)."""
from . import lib  # noqa: F401
from .lib import FluidSolverError, Grid, Params, ParticleParams, load  # noqa: F401
from .presets import PALETTES, PRESETS, PRESETS_BY_NAME, FluidPreset, palette_color  # noqa: F401
from .solver import FluidSolver, ParticleSystem  # noqa: F401
