"""simulation driver, results store, device tracer, multi-GPU helpers"""
from .results_store import (SimulationResults, RawFolder, rawFolders, latestRawFolder, resultsFolderPath,
                            updateResultEntry)
from .simulation_loop import runSimulation
