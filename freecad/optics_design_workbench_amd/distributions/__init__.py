'''
Classes for (random) point generation following given distributions.
'''
from .random_number_generator import VectorRandomVariable, ScalarRandomVariable, SamplerTables
from .points_by_density import calcDiffDensity, calcHistDensity, generatePointsWithGivenDensity1D
