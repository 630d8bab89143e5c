"""host-side mirrors of the reference's document element proxies"""
from . import point_source
from .make import (makeBox, makeSphere, makeCylinder, makeCone, makeTorus, makeCommon, makeCut, makeFuse,
                   makeOpticalGroup, makeMirror, makeLens, makeAbsorber, makeVacuum, makeGrating,
                   makePointSource, makeSimulationSettings)
