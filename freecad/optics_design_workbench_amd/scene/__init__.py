"""scene bake: FCStd-lite loader, CSG of analytic primitives, flat tables"""
from .placement import Placement
from .fcstd import Document, DocumentObject, open_fcstd
from .geometry import UnsupportedGeometry
from .bake import (BakedScene, Limits, bakeScene, bakeLimits, lightSources, opticalObjects,
                   simulationSettings, activeSimulationSettings, globalPlacements, allPlacementsAndPaths,
                   tracingSequence)
