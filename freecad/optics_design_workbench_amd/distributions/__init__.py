'''
Classes for (random) point generation following given distributions.
'''
from .random_number_generator import VectorRandomVariable, ScalarRandomVariable, SamplerTables
